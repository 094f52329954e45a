// Diagnostic: the A-stationary fp16 kernel alone at the C2 shapes, with per-block phase stamps (-DASTAT_STAMPS).
#include <hip/hip_runtime.h>
#include "../../gecco_amd/csrc/gemm_f16_dma.hip"
#include "../../gecco_amd/csrc/gemm_f16_astat.hip"
#include <stdio.h>
#include <vector>

int main() {
    const int B = 64, N = 2048, K = 384;
    struct Site { const char* name; int Nout, nsplit; bool act; } sites[] = {{"kv|q_proj", 1152, 768, false}, {"mlp.0+act", 768, 0, true}};
    float *A, *W, *pa, *po, *bias, *alpha, *img; _Float16 *C1, *C2;
    (void)hipMalloc(&A, (size_t)B * N * K * 4); (void)hipMalloc(&W, 1152 * K * 4); (void)hipMalloc(&img, 1152 * K * 4);
    (void)hipMalloc(&C1, (size_t)B * N * 768 * 2); (void)hipMalloc(&C2, (size_t)B * N * 384 * 2);
    (void)hipMalloc(&pa, B * K * 4); (void)hipMalloc(&po, B * K * 4); (void)hipMalloc(&bias, 1152 * 4); (void)hipMalloc(&alpha, 4);
    std::vector<float> h((size_t)B * N * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 20011) / 10000.f - 1.f;
    (void)hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(W, h.data(), 1152 * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(pa, h.data(), B * K * 4, hipMemcpyHostToDevice); (void)hipMemcpy(po, h.data(), B * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(bias, h.data(), 1152 * 4, hipMemcpyHostToDevice);
    float one = 1.f; (void)hipMemcpy(alpha, &one, 4, hipMemcpyHostToDevice);
    for (auto& s : sites) {
        split_f16_tiled_launch(W, img, s.Nout, K, K, 0);
        GemmArgs g{}; g.A = A; g.pro_a = pa; g.pro_o = po; g.bias = bias; g.alpha = alpha; g.act = s.act; g.C = (float*)C1;
        g.B = B; g.rows = N; g.K = K; g.Nout = s.Nout; g.lda = K; g.ldw = K; g.ldc = s.nsplit ? s.nsplit : s.Nout; g.ldr = g.ldc;
        g.precision = 2; g.w_img = img; g.c_f16 = 1;
        if (s.nsplit) { g.C2 = (float*)C2; g.bias2 = bias; g.n_split = s.nsplit; g.ldc2 = s.Nout - s.nsplit; }
        if (!gemm_f16_astat_supported(g)) { printf("unsupported\n"); return 1; }
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        gemm_f16_astat_launch(g, 0); gemm_f16_astat_launch(g, 0);
        (void)hipEventRecord(a, 0);
        for (int i = 0; i < 8; ++i) gemm_f16_astat_launch(g, 0);
        (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 8;
        printf("%-12s %.3f ms  %.1f TF  (HBM bytes %.0f MB -> %.2f TB/s)\n", s.name, ms, 2.0 * B * N * K * s.Nout / ms / 1e9,
               (B * (double)N * K * 4 + B * (double)N * s.Nout * 2) / 1e6, (B * (double)N * K * 4 + B * (double)N * s.Nout * 2) / ms / 1e9);
#ifdef ASTAT_STAMPS
        (void)hipDeviceSynchronize();
        static unsigned long long hs[4096 * 8];
        (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_astat_stamps), sizeof(hs));
        const int nb = B * N / 128;
        double d[2] = {0, 0};
        for (int i = 0; i < nb; ++i) for (int k = 0; k < 2; ++k) d[k] += (double)(hs[i * 8 + k + 1] - hs[i * 8 + k]);
        printf("   stamps (cycles, mean per block over %d blocks): A build %.0f  all column tiles %.0f\n", nb, d[0] / nb, d[1] / nb);
#endif
    }
    return 0;
}
