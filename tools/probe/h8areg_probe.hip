// Diagnostic: the h8 register-fed kernel (mlp.2 shape: B 64 x N 2048, K 768 -> 384, residual + statistics) alone, with per-block
// phase stamps (-DH8_STAMPS) and one ingredient removed per build.  Build: tools/probe/build_h8.sh
#include <hip/hip_runtime.h>
#include "../../gecco_amd/csrc/gemm_h8_areg.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    const int B = 64, N = 2048, K = argc > 1 ? atoi(argv[1]) : 768, Nout = 384;
    float *Aimg, *W, *bias, *img, *C, *stats;
    (void)hipMalloc(&Aimg, (size_t)B * N * K * 3); (void)hipMalloc(&W, (size_t)Nout * K * 4); (void)hipMalloc(&img, (size_t)Nout * K * 4);
    (void)hipMalloc(&C, (size_t)B * N * Nout * 4); (void)hipMalloc(&bias, Nout * 4); (void)hipMalloc(&stats, (size_t)B * (N / 128) * 2 * Nout * 4);
    std::vector<float> h((size_t)B * N * Nout);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
    (void)hipMemcpy(C, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < (size_t)Nout * K; ++i) h[i] *= 0.05f;
    (void)hipMemcpy(W, h.data(), (size_t)Nout * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(bias, h.data(), Nout * 4, hipMemcpyHostToDevice);
    // a plausible activation image: fp16 values of magnitude ~1 (bytes of random fp16 in [0.5, 2)), lo bytes small
    {
        std::vector<unsigned short> a((size_t)B * N * K * 3 / 2);
        for (size_t i = 0; i < a.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a[i] = (unsigned short)(0x3800 | ((s >> 20) & 0x83FF)); }
        (void)hipMemcpy(Aimg, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    }
    {   // timing only: a W stream of plausible fp16 / fp8 bit patterns (the image builder lives in gemm_h8_astat.hip)
        std::vector<unsigned short> a((size_t)Nout * K * 2);
        for (size_t i = 0; i < a.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a[i] = (unsigned short)(0x2800 | ((s >> 20) & 0x83FF)); }
        (void)hipMemcpy(img, a.data(), a.size() * 2, hipMemcpyHostToDevice);
    }
    GemmArgs g{}; g.A = Aimg; g.bias = bias; g.residual = C; g.C = C; g.stats = stats;
    g.B = B; g.rows = N; g.K = K; g.Nout = Nout; g.lda = K; g.ldw = K; g.ldc = Nout; g.ldr = Nout; g.a_img = 2; g.w_img = img; g.precision = 1;
    if (!gemm_h8_areg_supported(g)) { printf("unsupported\n"); return 1; }
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    gemm_h8_areg_launch(g, 0); gemm_h8_areg_launch(g, 0);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 8; ++i) gemm_h8_areg_launch(g, 0);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 8;
    printf("%-12s K=%d %.1f us  %.1f TF of 2MNK\n", argv[0], K, ms * 1e3, 2.0 * B * N * K * Nout / ms / 1e9);
#ifdef H8_STAMPS
    (void)hipDeviceSynchronize();
    static unsigned long long hs[4096 * 4];
    (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_h8a_stamps), sizeof(hs));
    const int nb = B * (N / 128) * 3 < 4096 ? B * (N / 128) * 3 : 4096;
    double d[3] = {0, 0, 0};
    for (int i = 0; i < nb; ++i) for (int k = 0; k < 3; ++k) d[k] += (double)(hs[i * 4 + k + 1] - hs[i * 4 + k]);
    printf("   stamps (ticks, mean per block over %d blocks): prologue (first stage + A landed) %.0f  K loop %.0f  epilogue %.0f\n", nb, d[0] / nb, d[1] / nb, d[2] / nb);
#endif
    return 0;
}
