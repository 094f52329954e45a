// Diagnostic build of the fused GEMM with s_memtime stamps around prologue / main loop / epilogue
// (never shipped: the product library is built without GEMM_STAMPS).  Prints mean phase durations in
// shader cycles for the C2 call-site shapes.
#include <hip/hip_runtime.h>
#define GEMM_STAMPS
__device__ unsigned long long* g_stamps;
#include "../../gecco_amd/csrc/gemm_f32.hip"
#include <stdio.h>
#include <vector>
#include <algorithm>

int main() {
    const int B = 64, N = 2048;
    struct Site { const char* name; int K, Nout; bool pro, res, stats, act; } sites[] = {
        {"kv_proj", 384, 768, true, false, false, false}, {"q_proj", 384, 384, true, false, false, false},
        {"out_proj+res+stats", 384, 384, false, true, true, false}, {"mlp.0+act", 384, 768, true, false, false, true},
        {"mlp.2+res+stats", 768, 384, false, true, true, false}};
    float *A, *W, *C, *R, *pa, *po, *bias, *alpha, *stats;
    hipMalloc(&A, (size_t)B * N * 768 * 4); hipMalloc(&W, 768 * 768 * 4); hipMalloc(&C, (size_t)B * N * 768 * 4);
    hipMalloc(&R, (size_t)B * N * 768 * 4); hipMalloc(&pa, B * 768 * 4); hipMalloc(&po, B * 768 * 4);
    hipMalloc(&bias, 768 * 4); hipMalloc(&alpha, 4); hipMalloc(&stats, (size_t)B * 16 * 2 * 768 * 4);
    std::vector<float> h((size_t)B * N * 768);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(R, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), 768 * 768 * 4, hipMemcpyHostToDevice); hipMemcpy(pa, h.data(), B * 768 * 4, hipMemcpyHostToDevice);
    hipMemcpy(po, h.data(), B * 768 * 4, hipMemcpyHostToDevice); hipMemcpy(bias, h.data(), 768 * 4, hipMemcpyHostToDevice);
    float one = 1.f; hipMemcpy(alpha, &one, 4, hipMemcpyHostToDevice);
    unsigned long long* d_st; hipMalloc(&d_st, 8192 * 4 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st));
    for (auto& s : sites) {
        GemmArgs g{}; g.A = A; g.W = W; g.bias = bias; g.pro_a = s.pro ? pa : nullptr; g.pro_o = s.pro ? po : nullptr;
        g.alpha = alpha; g.residual = s.res ? R : nullptr; g.C = C; g.stats = s.stats ? stats : nullptr;
        g.B = B; g.rows = N; g.K = s.K; g.Nout = s.Nout; g.lda = s.K; g.ldw = s.K; g.ldc = s.Nout; g.ldr = s.Nout; g.act = s.act;
        for (int it = 0; it < 3; ++it) gemm_f32_launch(g, 0);
        hipDeviceSynchronize();
        const int nb = B * 16 * (s.Nout / 128);
        std::vector<unsigned long long> st(nb * 4);
        hipMemcpy(st.data(), d_st, nb * 32, hipMemcpyDeviceToHost);
        double p = 0, m = 0, e = 0; unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < nb; ++i) { p += st[4*i+1]-st[4*i]; m += st[4*i+2]-st[4*i+1]; e += st[4*i+3]-st[4*i+2]; }
        // s_memtime is per-XCD: use blocks of XCD 0 (blockIdx % 8 == 0) for the span
        for (int i = 0; i < nb; i += 8) { t0 = std::min(t0, st[4*i]); t1 = std::max(t1, st[4*i+3]); }
        const int nk = s.K / 32;
        printf("%-20s blocks %5d  prologue %7.0f  main %7.0f (%5.0f/K-step, ideal 2 blocks/CU: %d)  epilogue %7.0f  | XCD0 span %llu cycles, rounds %.1f\n",
               s.name, nb, p / nb, m / nb, m / nb / nk, 2 * 4096, e / nb, t1 - t0, nb / 512.0);
    }
    return 0;
}
