// Diagnostic: the fp16 weight-gradient kernel (gemm_tn_f16.hip) alone at the training shapes of C2 (48 x 2048 rows; 768 x 384,
// 384 x 768, 384 x 384 outputs), built with one ingredient removed each (tools/probe/build_tn.sh): where its time goes.
#include <hip/hip_runtime.h>
#include "../../gecco_amd/csrc/gemm_tn_f16.hip"
#include <stdio.h>
#include <vector>

int xcd_remap_dummy;

int main(int argc, char** argv) {
    const int Z = 48, R = 2048;
    float *dy, *x, *parts, *pa, *po;
    (void)hipMalloc(&dy, (size_t)Z * R * 768 * 4); (void)hipMalloc(&x, (size_t)Z * R * 768 * 4); (void)hipMalloc(&parts, (size_t)Z * 768 * 384 * 4);
    (void)hipMalloc(&pa, (size_t)Z * 768 * 4); (void)hipMalloc(&po, (size_t)Z * 768 * 4);
    std::vector<float> h((size_t)Z * R * 768);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
    (void)hipMemcpy(dy, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(pa, h.data(), (size_t)Z * 768 * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(po, h.data(), (size_t)Z * 768 * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int shapes[3][2] = {{768, 384}, {384, 768}, {384, 384}};
    for (int pro = 0; pro < 2; ++pro)
        for (int k = 0; k < 3; ++k) {
            TnArgs g{};
            g.A = dy; g.Bm = x; g.C = parts; g.Z = Z; g.R = R; g.N = shapes[k][0]; g.K = shapes[k][1]; g.lda = g.N; g.ldb = g.K;
            g.sA = (size_t)R * g.N; g.sB = (size_t)R * g.K; g.group = 1; g.f16 = 1;
            if (pro) { g.pro_a = pa; g.pro_o = po; }
            gemm_tn_f16_launch(g, 0); gemm_tn_f16_launch(g, 0);
            (void)hipEventRecord(a, 0);
            for (int i = 0; i < 8; ++i) gemm_tn_f16_launch(g, 0);
            (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 8;
            const double by = 4.0 * Z * R * (g.N + g.K);
            printf("%-12s N %d K %d pro %d: %.1f us (operands once %.0f MB: %.2f TB/s; %d blocks x %d steps)\n", argv[0], g.N, g.K, pro, ms * 1e3, by / 1e6,
                   by / ms / 1e9, gemm_tn_f16_tiles(g.N, g.K) * Z, R / 32);
        }
    return 0;
}
