// Diagnostic: the 64-column-tile kv_proj | q_proj kernel alone at the C2 shape (B 64 x N 2048, K 384 -> 768 | 384, V two-term),
// with per-block phase stamps (-DH8_STAMPS) and one ingredient removed per build (-DH8_DIAG_*).  Build: tools/probe/build_h8.sh
#include <hip/hip_runtime.h>
#include "../../gecco_amd/csrc/gemm_h8_astat.hip"
#include <stdio.h>
#include <algorithm>
#include <vector>

int main(int argc, char** argv) {
    const int B = 64, N = 2048, K = 384, N1 = 768, N2 = 384, hd = 48;
    const int lo = argc > 1 ? atoi(argv[1]) : 1;
    const int zero = argc > 2 ? atoi(argv[2]) : 0;   // 1: all-zero x and weights (the clock a kernel reaches depends on operand toggling)
    float *A, *W, *pa, *po, *bias, *img; _Float16 *C1, *C2;
    (void)hipMalloc(&A, (size_t)B * N * K * 4); (void)hipMalloc(&W, (size_t)(N1 + N2) * K * 4); (void)hipMalloc(&img, (size_t)(N1 + N2) * K * 4);
    (void)hipMalloc(&C1, (size_t)B * N * N1 * 2); (void)hipMalloc(&C2, (size_t)B * N * N2 * 2);
    (void)hipMalloc(&pa, B * K * 4); (void)hipMalloc(&po, B * K * 4); (void)hipMalloc(&bias, N2 * 4);
    std::vector<float> h((size_t)B * N * K);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (float)((double)(s >> 11) / 9007199254740992.0 * 4.0 - 2.0); }
    if (zero) std::fill(h.begin(), h.end(), 0.f);
    (void)hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < (size_t)(N1 + N2) * K; ++i) h[i] *= 0.05f;
    (void)hipMemcpy(W, h.data(), (size_t)(N1 + N2) * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(pa, h.data(), B * K * 4, hipMemcpyHostToDevice); (void)hipMemcpy(po, h.data(), B * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(bias, h.data(), N2 * 4, hipMemcpyHostToDevice);
    SplitJobs jobs; jobs.n = 2;
    const int lb = lo ? N1 / 128 : 0, le = lo ? N1 / 64 : 0;
    jobs.job[0] = SplitJob{W, img, N1, K, K, 1 | (lb << 8) | (le << 20)};
    jobs.job[1] = SplitJob{W + (size_t)N1 * K, img + kvq_image_bytes(N1, K, (le - lb) * 64) / 4, N2, K, K, 1};
    h8_image_multi_launch(jobs, 0);
    GemmArgs g{}; g.A = A; g.pro_a = pa; g.pro_o = po; g.C = (float*)C1; g.C2 = (float*)C2; g.bias2 = bias; g.n_split = N1; g.ldc2 = N2;
    g.B = B; g.rows = N; g.K = K; g.Nout = N1 + N2; g.lda = K; g.ldw = K; g.ldc = N1; g.c_f16 = 1; g.w_img = img; g.hm_hd = hd; g.precision = 2;
    g.lo_begin = lb; g.lo_tiles = le;
    if (!gemm_kvq_astat_supported(g)) { printf("unsupported\n"); return 1; }
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    gemm_kvq_astat_launch(g, 0); gemm_kvq_astat_launch(g, 0);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 8; ++i) gemm_kvq_astat_launch(g, 0);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 8;
    printf("%-10s lo=%d %.1f us  %.1f TF of 2MNK\n", argv[0], lo, ms * 1e3, 2.0 * B * N * K * (N1 + N2) / ms / 1e9);
#ifdef H8_STAMPS
    (void)hipDeviceSynchronize();
    static unsigned long long hs[1024 * 4];
    (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_h8_stamps), sizeof(hs));
    const int nb = 1024;
    double d[2] = {0, 0};
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < nb; ++i) {
        for (int k = 0; k < 2; ++k) d[k] += (double)(hs[i * 4 + k + 1] - hs[i * 4 + k]);
        if (hs[i * 4] < t0) t0 = hs[i * 4];
        if (hs[i * 4 + 2] > t1) t1 = hs[i * 4 + 2];
    }
    printf("   stamps (s_memtime ticks, mean per block over %d blocks): A build %.0f  all column tiles %.0f; first start -> last end %llu\n",
           nb, d[0] / nb, d[1] / nb, t1 - t0);
#endif
    return 0;
}
