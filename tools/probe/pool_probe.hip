// Diagnostic: the mixed mode's pool attention (attention_x3.hip: pool_attn_x3_kernel<48, F16, IO16>, head-major fp16 K | V) alone at the C2
// shape (B 64 x N 2048, d 384, 8 heads, 2 key splits), built with one ingredient removed each (tools/probe/build_pool.sh).
#include <hip/hip_runtime.h>
#include "../../gecco_amd/csrc/attention_x3.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    const int B = 64, N = 2048, C = 384, H = 8, ns = 2;
    void* kv16;
    float *ind, *po, *pml;
    (void)hipMalloc(&kv16, (size_t)B * N * 2 * C * 2); (void)hipMalloc(&ind, (size_t)H * 64 * (C / H) * 4);
    (void)hipMalloc(&po, (size_t)B * H * ns * 64 * (C / H) * 4); (void)hipMalloc(&pml, (size_t)B * H * ns * 64 * 2 * 4);
    std::vector<unsigned short> a((size_t)B * N * 2 * C);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < a.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a[i] = (unsigned short)(0x3800 | ((s >> 20) & 0x83FF)); }
    (void)hipMemcpy(kv16, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> q((size_t)H * 64 * (C / H));
    for (size_t i = 0; i < q.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; q[i] = (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
    (void)hipMemcpy(ind, q.data(), q.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int hm = 1; hm >= 0; --hm) {
        pool_attn_x3_partials_launch((const float*)kv16, ind, po, pml, B, N, C, H, ns, 0, 2, 1, hm);
        pool_attn_x3_partials_launch((const float*)kv16, ind, po, pml, B, N, C, H, ns, 0, 2, 1, hm);
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 8; ++i) pool_attn_x3_partials_launch((const float*)kv16, ind, po, pml, B, N, C, H, ns, 0, 2, 1, hm);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 8;
        printf("%-14s %s: %.1f us (201 MB of K | V: %.2f TB/s)\n", argv[0], hm ? "head-major" : "row-major ", ms * 1e3, 201.3e6 / ms / 1e9);
    }
    return 0;
}
