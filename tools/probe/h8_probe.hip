// Diagnostic: the h8 A-stationary mlp.0 kernel alone at the C2 shape (B 64 x N 2048, K 384 -> 768), with per-block phase
// stamps (-DH8_STAMPS) and one ingredient removed per build (-DH8_DIAG_*).  Build: tools/probe/build_h8.sh
#include <hip/hip_runtime.h>
#include "../../gecco_amd/csrc/gemm_h8_astat.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    const int B = 64, N = argc > 1 ? atoi(argv[1]) : 2048, K = 384, Nout = 768;
    float *A, *W, *pa, *po, *bias, *alpha, *img, *C;
    (void)hipMalloc(&A, (size_t)B * N * K * 4); (void)hipMalloc(&W, (size_t)Nout * K * 4); (void)hipMalloc(&img, (size_t)Nout * K * 4);
    (void)hipMalloc(&C, (size_t)B * N * Nout * 4);
    (void)hipMalloc(&pa, B * K * 4); (void)hipMalloc(&po, B * K * 4); (void)hipMalloc(&bias, Nout * 4); (void)hipMalloc(&alpha, 4);
    std::vector<float> h((size_t)B * N * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 20011) / 10000.f - 1.f;
    (void)hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < (size_t)Nout * K; ++i) h[i] *= 0.05f;
    (void)hipMemcpy(W, h.data(), (size_t)Nout * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(pa, h.data(), B * K * 4, hipMemcpyHostToDevice); (void)hipMemcpy(po, h.data(), B * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(bias, h.data(), Nout * 4, hipMemcpyHostToDevice);
    float one = 1.f; (void)hipMemcpy(alpha, &one, 4, hipMemcpyHostToDevice);
    SplitJobs jobs; jobs.n = 1; jobs.job[0] = SplitJob{W, img, Nout, K, K, 0};
    h8_image_multi_launch(jobs, 0);
    GemmArgs g{}; g.A = A; g.pro_a = pa; g.pro_o = po; g.bias = bias; g.alpha = alpha; g.act = 1; g.C = C;
    g.B = B; g.rows = N; g.K = K; g.Nout = Nout; g.lda = K; g.ldw = K; g.ldc = Nout; g.c_img = 1; g.w_img = img;
    if (!gemm_h8_astat_supported(g)) { printf("unsupported\n"); return 1; }
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    gemm_h8_astat_launch(g, 0); gemm_h8_astat_launch(g, 0);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 8; ++i) gemm_h8_astat_launch(g, 0);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 8;
    printf("%-10s %.1f us  %.1f TF of 2MNK\n", argv[0], ms * 1e3, 2.0 * B * N * K * Nout / ms / 1e9);
#ifdef H8_STAMPS
    (void)hipDeviceSynchronize();
    static unsigned long long hs[1024 * 4];
    (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_h8_stamps), sizeof(hs));
    const int nb = B * N / 256 < 1024 ? B * N / 256 : 1024;
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
