// Diagnostic (fp16 kernel): time the fused LDS-DMA GEMM with parts of its memory traffic aliased away
// (ldc = 0: every C row lands on the same bytes; ldr = 0 / lda = 0 likewise for the residual / A reads) to see
// which stream the kernel waits for.  Results are wrong by construction; timing only.
#include <hip/hip_runtime.h>
#include "../../gecco_amd/csrc/gemm_f16_dma.hip"
#include <stdio.h>
#include <vector>
#ifndef PREC0
#define PREC0 0
#endif

static float time_ms(const GemmArgs& g, int it = 8) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    gemm_f16_dma_launch(g, 0); gemm_f16_dma_launch(g, 0);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < it; ++i) gemm_f16_dma_launch(g, 0);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / it;
}

int main() {
    const int B = 64, N = 2048;
    struct Site { const char* name; int K, Nout; bool pro, res, stats, act; } sites[] = {
        {"kv|q_proj", 384, 1152, true, false, false, false}, {"out_proj+res+stats", 384, 384, false, true, true, false},
        {"mlp.0+act", 384, 768, true, false, false, true}, {"mlp.2+res+stats", 768, 384, false, true, true, false}};
    float *A, *W, *C, *R, *pa, *po, *bias, *alpha, *stats; float* img;
    (void)hipMalloc(&A, (size_t)B * N * 768 * 4); (void)hipMalloc(&W, 1152 * 768 * 4); (void)hipMalloc(&C, (size_t)B * N * 1152 * 4);
    (void)hipMalloc(&R, (size_t)B * N * 768 * 4); (void)hipMalloc(&pa, B * 768 * 4); (void)hipMalloc(&po, B * 768 * 4);
    (void)hipMalloc(&bias, 1152 * 4); (void)hipMalloc(&alpha, 4); (void)hipMalloc(&stats, (size_t)B * 16 * 2 * 768 * 4);
    (void)hipMalloc(&img, 1152 * 768 * 4);
    std::vector<float> h((size_t)B * N * 768);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    (void)hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(R, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W, h.data(), 1152 * 768 * 4, hipMemcpyHostToDevice); (void)hipMemcpy(pa, h.data(), B * 768 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(po, h.data(), B * 768 * 4, hipMemcpyHostToDevice); (void)hipMemcpy(bias, h.data(), 1152 * 4, hipMemcpyHostToDevice);
    float one = 1.f; (void)hipMemcpy(alpha, &one, 4, hipMemcpyHostToDevice);
    for (int prec = 2; prec < 3; ++prec)
        for (auto& s : sites) {
            GemmArgs g{}; g.A = A; g.W = W; g.bias = bias; g.pro_a = s.pro ? pa : nullptr; g.pro_o = s.pro ? po : nullptr;
            g.alpha = alpha; g.residual = s.res ? R : nullptr; g.C = C; g.stats = s.stats ? stats : nullptr;
            g.B = B; g.rows = N; g.K = s.K; g.Nout = s.Nout; g.lda = s.K; g.ldw = s.K; g.ldc = s.Nout; g.ldr = s.Nout; g.act = s.act;
            g.precision = prec; g.w_img = img;
#ifdef PROBE_A16
            g.a_f16 = 1; g.pro_a = nullptr; g.pro_o = nullptr; if (!s.res) g.c_f16 = 1;
#endif
            split_f16_tiled_launch(W, img, s.Nout, s.K, s.K, 0);
            const double fl = 2.0 * B * N * s.K * s.Nout;
            const float t0 = time_ms(g);
            GemmArgs g1 = g; g1.ldc = 0;            const float t1 = time_ms(g1);
            GemmArgs g2 = g1; g2.ldr = 0;           const float t2 = time_ms(g2);
            GemmArgs g3 = g2; g3.lda = 0;           const float t3 = time_ms(g3);
#ifdef GEMM_STAMPS
            {
                gemm_f16_dma_launch(g, 0); (void)hipDeviceSynchronize();
                static unsigned long long hs[16384 * 8];
                (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_stamps), sizeof(hs));
                const int nb = (B * (N / 128) * ((s.Nout + 127) / 128)) < 16384 ? (B * (N / 128) * ((s.Nout + 127) / 128)) : 16384;
                double d[4] = {0, 0, 0, 0};
                for (int i = 0; i < nb; ++i)
                    for (int k = 0; k < 4; ++k) d[k] += (double)(hs[i * 8 + k + 1] - hs[i * 8 + k]);
                unsigned long long t0 = ~0ull, t1 = 0;
                for (int i = 0; i < nb; ++i) { if (hs[i * 8] < t0) t0 = hs[i * 8]; if (hs[i * 8 + 4] > t1) t1 = hs[i * 8 + 4]; }
                printf("   stamps (100 MHz ticks, mean per block over %d blocks): park %.1f  first-stage %.1f  k-loop %.1f  epilogue %.1f | kernel span %.0f ticks\n",
                       nb, d[0] / nb, d[1] / nb, d[2] / nb, d[3] / nb, (double)(t1 - t0));
            }
#endif
            printf("%-6s %-20s full %.3f ms (%5.1f TF) | C aliased %.3f | +residual aliased %.3f | +A aliased %.3f (%5.1f TF)\n",
                   "fp16", s.name, t0, fl / t0 / 1e9, t1, t2, t3, fl / t3 / 1e9);
        }
    return 0;
}
