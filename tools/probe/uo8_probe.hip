// Diagnostic: the fused unpool attention + h8 out_proj kernel (unpool_outproj_h8.hip) alone at the C2 shape (B 64 x N 2048, d 384,
// 8 heads), per-block phase stamps (-DUO8_STAMPS).  Build: tools/probe/build_uo8.sh
#include <hip/hip_runtime.h>
#include "../../gecco_amd/csrc/unpool_outproj_h8.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    const int B = 64, N = 2048, C = 384, H = 8;
    float *x, *kvh, *bias, *stats;
    void *q16, *wimg, *kvimg;
    (void)hipMalloc(&x, (size_t)B * N * C * 4); (void)hipMalloc(&q16, (size_t)B * N * C * 2); (void)hipMalloc(&kvh, (size_t)B * 64 * 2 * C * 4);
    (void)hipMalloc(&wimg, (size_t)C * C * 4); (void)hipMalloc(&kvimg, unpool_outproj_h8_kv_bytes(B, C, H)); (void)hipMalloc(&bias, C * 4);
    (void)hipMalloc(&stats, (size_t)B * (N / 128) * 2 * C * 4);
    std::vector<float> h((size_t)B * N * C);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
    (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(kvh, h.data(), (size_t)B * 64 * 2 * C * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(bias, h.data(), C * 4, hipMemcpyHostToDevice);
    {
        std::vector<unsigned short> a((size_t)B * N * C);
        for (size_t i = 0; i < a.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a[i] = (unsigned short)(0x3800 | ((s >> 20) & 0x83FF)); }
        (void)hipMemcpy(q16, a.data(), a.size() * 2, hipMemcpyHostToDevice);
        for (size_t i = 0; i < (size_t)C * C * 2; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a[i] = (unsigned short)(0x2800 | ((s >> 20) & 0x83FF)); }
        (void)hipMemcpy(wimg, a.data(), (size_t)C * C * 4, hipMemcpyHostToDevice);
    }
    kvh_image_launch(kvh, kvimg, B, C, H, 0);
    UnpoolH8Args g{}; g.x = x; g.q16 = q16; g.kv_img = kvimg; g.w_img = wimg; g.bias = bias; g.stats = stats; g.B = B; g.rows = N; g.H = H;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    unpool_outproj_h8_launch(g, C, 0); unpool_outproj_h8_launch(g, C, 0);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 8; ++i) unpool_outproj_h8_launch(g, C, 0);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 8;
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 8; ++i) kvh_image_launch(kvh, kvimg, B, C, H, 0);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms2; (void)hipEventElapsedTime(&ms2, a, b); ms2 /= 8;
    printf("%-12s %.1f us (502 MB: %.2f TB/s)   k | v image %.1f us\n", argv[0], ms * 1e3, 502e6 / ms / 1e9, ms2 * 1e3);
#ifdef UO8_STAMPS
    (void)hipDeviceSynchronize();
    static unsigned long long hs[2048 * 4];
    (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_uo8_stamps), sizeof(hs));
    const int nb = B * (N / 128) < 2048 ? B * (N / 128) : 2048;
    double d[3] = {0, 0, 0};
    for (int i = 0; i < nb; ++i) for (int k = 0; k < 3; ++k) d[k] += (double)(hs[i * 4 + k + 1] - hs[i * 4 + k]);
    printf("   stamps (ticks, mean per block over %d blocks): attention (8 heads) %.0f  six column tiles %.0f  tail %.0f\n", nb, d[0] / nb, d[1] / nb, d[2] / nb);
    static unsigned long long he[2048 * 4];
    (void)hipMemcpyFromSymbol(he, HIP_SYMBOL(g_uo8_epi), sizeof(he));
    double e1 = 0, e2 = 0;
    for (int i = 0; i < nb; ++i) { e1 += (double)he[i * 4]; e2 += (double)he[i * 4 + 1]; }
    printf("   inside the six epilogues (wave 0): transposed phase %.0f  row phase %.0f\n", e1 / nb, e2 / nb);
#endif
    return 0;
}
