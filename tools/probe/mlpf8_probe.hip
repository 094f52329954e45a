// Diagnostic: the fused h8 point MLP (mlp_fused_h8.hip) alone at the C2 shape (B 64 x N 2048, d 384 -> 768 -> 384), per-block phase
// stamps of one P1 wave and one P2 wave (-DMF8_STAMPS).  Build: tools/probe/build_mlpf8.sh
#include <hip/hip_runtime.h>
#include "../../gecco_amd/csrc/mlp_fused_h8.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    const int B = 64, N = 2048, C = 384, Wd = 768;
    float *x, *pa, *po, *W0, *W2, *b0, *b2, *alpha, *stats;
    void* img;
    (void)hipMalloc(&x, (size_t)B * N * C * 4); (void)hipMalloc(&pa, B * C * 4); (void)hipMalloc(&po, B * C * 4);
    (void)hipMalloc(&W0, (size_t)Wd * C * 4); (void)hipMalloc(&W2, (size_t)Wd * C * 4); (void)hipMalloc(&b0, Wd * 4); (void)hipMalloc(&b2, C * 4);
    (void)hipMalloc(&alpha, 4); (void)hipMalloc(&stats, (size_t)B * (N / 128) * 2 * C * 4); (void)hipMalloc(&img, mlp_fused_h8_image_bytes(C, Wd));
    std::vector<float> h((size_t)B * N * C);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
    (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(po, h.data(), B * C * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < (size_t)Wd * C; ++i) h[i] *= 0.05f;
    (void)hipMemcpy(W0, h.data(), (size_t)Wd * C * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W2, h.data() + 1000, (size_t)Wd * C * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(b0, h.data(), Wd * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(b2, h.data(), C * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < B * C; ++i) h[i] = 1.0f + h[i];
    (void)hipMemcpy(pa, h.data(), B * C * 4, hipMemcpyHostToDevice);
    const float a = 0.9f;
    (void)hipMemcpy(alpha, &a, 4, hipMemcpyHostToDevice);
    mlp_fused_h8_image_launch(W0, W2, img, C, Wd, 0);
    MlpH8Args g{}; g.x = x; g.pro_a = pa; g.pro_o = po; g.w_img = img; g.b0 = b0; g.b2 = b2; g.alpha = alpha; g.act = 1; g.stats = stats; g.B = B; g.rows = N;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    mlp_fused_h8_launch(g, C, Wd, 0); mlp_fused_h8_launch(g, C, Wd, 0);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 8; ++i) mlp_fused_h8_launch(g, C, Wd, 0);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 8;
    printf("%-14s %.1f us  %.1f TF of 2MNK (2 products)   err %d\n", argv[0], ms * 1e3, 4.0 * B * N * C * Wd / ms / 1e9, (int)hipGetLastError());
#ifdef MF8_STAMPS
    (void)hipDeviceSynchronize();
    static unsigned long long hs[2048 * 8];
    (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_mf8_stamps), sizeof(hs));
    const int nb = B * (N / 128) < 2048 ? B * (N / 128) : 2048;
    double d[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int i = 0; i < nb; ++i) for (int w = 0; w < 2; ++w) for (int k = 0; k < 3; ++k) d[w][k] += (double)(hs[i * 8 + w * 4 + k + 1] - hs[i * 8 + w * 4 + k]);
    printf("   stamps (ticks, mean per block over %d blocks): P1 wave: y build %.0f  schedule %.0f  tail %.0f | P2 wave: wait %.0f  schedule %.0f  epilogue %.0f\n",
           nb, d[0][0] / nb, d[0][1] / nb, d[0][2] / nb, d[1][0] / nb, d[1][1] / nb, d[1][2] / nb);
#endif
    return 0;
}
