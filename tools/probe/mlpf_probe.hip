// Diagnostic: the fused point MLP alone at the C2 shape (B = 64, N = 2048, C = 384), with phase stamps (-DMLPF_STAMPS).
#include <hip/hip_runtime.h>
#include "../../gecco_amd/csrc/mlp_fused_f16.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, N = 2048, C = 384, Wd = 768;
    float *x, *pa, *po, *img, *par, *stats;
    (void)hipMalloc(&x, (size_t)B * N * C * 4); (void)hipMalloc(&pa, B * C * 4); (void)hipMalloc(&po, B * C * 4);
    (void)hipMalloc(&img, (size_t)C * Wd * 4); (void)hipMalloc(&par, 4096 * 4); (void)hipMalloc(&stats, (size_t)B * (N / 128) * 2 * C * 4);
    std::vector<float> h((size_t)B * N * C);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 20011) / 10000.f - 1.f;
    (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> one(B * C, 1.f), zero(B * C, 0.f);
    (void)hipMemcpy(pa, one.data(), B * C * 4, hipMemcpyHostToDevice); (void)hipMemcpy(po, zero.data(), B * C * 4, hipMemcpyHostToDevice);
    std::vector<_Float16> w((size_t)C * Wd * 2);
    for (size_t i = 0; i < w.size(); ++i) w[i] = (_Float16)(((float)((i * 2246822519u) % 2001) / 1000.f - 1.f) * 0.03f);
    (void)hipMemcpy(img, w.data(), w.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> pv(4096, 0.01f); pv[2048] = 1.f;
    (void)hipMemcpy(par, pv.data(), pv.size() * 4, hipMemcpyHostToDevice);
    MlpArgs g{};
    g.x = x; g.pro_a = pa; g.pro_o = po; g.w_stream = img; g.b0 = par; g.b2 = par + 1024; g.alpha = par + 2048; g.act = 1;
    g.stats = stats; g.B = B; g.rows = N;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) mlp_fused_f16_launch(g, C, Wd, 0);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 8; ++i) mlp_fused_f16_launch(g, C, Wd, 0);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 8;
    printf("fused mlp B=%d: %.1f us per launch, %.0f TFLOP/s, %.2f TB/s of x in + x in + x out (err %d)\n", B, ms * 1e3,
           4.0 * B * N * C * Wd / ms / 1e9, 3.0 * B * N * C * 4 / ms / 1e9, (int)hipGetLastError());
#ifdef MLPF_STAMPS
    (void)hipDeviceSynchronize();
    static unsigned long long hs[2048 * 8];
    (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_mlpf_stamps), sizeof(hs));
    const char* names[] = {"params + A build", "chunks (GEMM a | b)", "epilogue"};
    const int nb = B * N / 128 < 2048 ? B * N / 128 : 2048;
    for (int k = 0; k < 3; ++k) {
        double d = 0;
        for (int i = 0; i < nb; ++i) d += (double)(hs[i * 8 + k + 1] - hs[i * 8 + k]);
        printf("   %-20s %8.0f cycles\n", names[k], d / nb);
    }
#endif
    return 0;
}
