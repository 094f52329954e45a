// Experiment: the pool attention's key tiles by LDS-DMA (pool_dma_kernel.hip.inc) against the library's register-staged kernel
// (attention_x3.hip: pool_attn_x3_kernel<HD, fp16, fp16 in>, head-major K | V): partials compared bit for bit, then both timed alone at the
// C2 shape.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 pool_dma_probe.hip -o pool_dma_probe
#include <hip/hip_runtime.h>
#include "../../gecco_amd/csrc/attention_x3.hip"
namespace {
#include "pool_dma_kernel.hip.inc"
}
#include <stdio.h>
#include <string.h>
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, N = argc > 2 ? atoi(argv[2]) : 2048, C = 384, H = 8, ns = argc > 3 ? atoi(argv[3]) : 1, HD = C / H;
    void* kv16;
    float *ind, *po[2], *pml[2];
    (void)hipMalloc(&kv16, (size_t)B * N * 2 * C * 2); (void)hipMalloc(&ind, (size_t)H * 64 * HD * 4);
    const size_t no = (size_t)B * H * ns * 64 * HD, nm = (size_t)B * H * ns * 64 * 2;
    for (int i = 0; i < 2; ++i) { (void)hipMalloc(&po[i], no * 4); (void)hipMalloc(&pml[i], nm * 4); }
    std::vector<unsigned short> a((size_t)B * N * 2 * C);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < a.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a[i] = (unsigned short)(0x3800 | ((s >> 20) & 0x83FF)); }
    (void)hipMemcpy(kv16, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> q((size_t)H * 64 * HD);
    for (size_t i = 0; i < q.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; q[i] = (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
    (void)hipMemcpy(ind, q.data(), q.size() * 4, hipMemcpyHostToDevice);
    auto run = [&](int dma) {
        return dma ? pool_dma_launch_t<48>((const float*)kv16, ind, po[1], pml[1], B, N, C, H, ns, 0)
                   : pool_attn_x3_partials_launch((const float*)kv16, ind, po[0], pml[0], B, N, C, H, ns, 0, 2, 1, 1);
    };
    run(0); run(1);
    (void)hipDeviceSynchronize();
    std::vector<float> o0(no), o1(no), m0(nm), m1(nm);
    (void)hipMemcpy(o0.data(), po[0], no * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(o1.data(), po[1], no * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(m0.data(), pml[0], nm * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(m1.data(), pml[1], nm * 4, hipMemcpyDeviceToHost);
    printf("partials bit-identical: O %s, (m, l) %s\n", memcmp(o0.data(), o1.data(), no * 4) ? "NO" : "yes", memcmp(m0.data(), m1.data(), nm * 4) ? "NO" : "yes");
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int dma = 0; dma < 2; ++dma) {
        run(dma);
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 8; ++i) run(dma);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 8;
        printf("%-22s B %d N %d splits %d: %.1f us (K | V %.0f MB: %.2f TB/s)\n", dma ? "LDS-DMA tiles" : "register-staged tiles", B, N, ns, ms * 1e3,
               (double)B * N * 2 * C * 2 / 1e6, (double)B * N * 2 * C * 2 / ms / 1e9);
    }
    return 0;
}
