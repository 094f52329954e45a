// Diagnostic: the one-launch inducer chain alone at the C2 shape (B = 64, C = 384), with phase stamps (-DCHAIN_STAMPS).
#include <hip/hip_runtime.h>
#include "../../gecco_amd/csrc/inducer_chain_f16.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, C = 384, Wd = 768, H = 8, ns = 2, HD = C / H;
    const int two = argc > 2 ? atoi(argv[2]) : 1;   // two-term weights (the mixed mode's chain)
    const size_t img_floats = ((size_t)C * C + (size_t)Wd * C + (size_t)C * Wd + (size_t)2 * C * C) / 2 * (two ? 2 : 1);
    float *part_o, *part_ml, *img, *par, *t, *h_out, *kvh;
    (void)hipMalloc(&part_o, (size_t)B * H * ns * 64 * HD * 4); (void)hipMalloc(&part_ml, (size_t)B * H * ns * 64 * 2 * 4);
    (void)hipMalloc(&img, img_floats * 4); (void)hipMalloc(&par, 16 * 768 * 4); (void)hipMalloc(&t, B * 4);
    (void)hipMalloc(&h_out, (size_t)B * 64 * C * 4); (void)hipMalloc(&kvh, (size_t)B * 64 * 2 * C * 4);
    std::vector<float> h((size_t)B * H * ns * 64 * HD);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 20011) / 10000.f - 1.f;
    (void)hipMemcpy(part_o, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> ml((size_t)B * H * ns * 64 * 2);
    for (size_t i = 0; i < ml.size(); ++i) ml[i] = (i & 1) ? 1.f + (float)(i % 7) : (float)(i % 5) - 2.f;
    (void)hipMemcpy(part_ml, ml.data(), ml.size() * 4, hipMemcpyHostToDevice);
    std::vector<_Float16> w(img_floats * 2);
    for (size_t i = 0; i < w.size(); ++i) w[i] = (_Float16)(((float)((i * 2246822519u) % 2001) / 1000.f - 1.f) * 0.05f);
    (void)hipMemcpy(img, w.data(), w.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> pv(16 * 768);
    for (size_t i = 0; i < pv.size(); ++i) pv[i] = 0.5f + (float)(i % 13) / 26.f;
    (void)hipMemcpy(par, pv.data(), pv.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(t, 0, B * 4);
    ChainArgs g{};
    g.part_o = part_o; g.part_ml = part_ml; g.nsplit = ns; g.H = H; g.w_stream = img;
    g.b0 = par; g.b2 = par + 768; g.bkv = par + 2 * 768; g.alpha = par + 3 * 768; g.act = 1;
    g.n1_scale_w = par + 4 * 768; g.n1_scale_b = par + 5 * 768; g.n1_bias_w = par + 6 * 768; g.n1_bias_b = par + 7 * 768;
    g.n2_scale_w = par + 8 * 768; g.n2_scale_b = par + 9 * 768; g.n2_bias_w = par + 10 * 768; g.n2_bias_b = par + 11 * 768;
    g.two_term = two; g.t = t; g.ctx_dim = 1; g.G = 32; g.eps = 1e-5f; g.h_out = h_out; g.kvh = kvh; g.B = B;
    const int cl = argc > 3 ? atoi(argv[3]) : 0;   // the cluster form: C / 128 blocks per sample
    unsigned* flags = nullptr;
    if (cl) {
        float *x1, *x3, *xu;
        (void)hipMalloc(&x1, (size_t)B * 64 * C * 4); (void)hipMalloc(&x3, (size_t)B * 64 * C * 4); (void)hipMalloc(&xu, (size_t)B * 64 * Wd * 2);
        (void)hipMalloc(&flags, (size_t)16 * B * 8 * 4); (void)hipMemset(flags, 0, (size_t)16 * B * 8 * 4);
        g.cluster = 1; g.x1 = x1; g.x3 = x3; g.xu = reinterpret_cast<unsigned*>(xu); g.flags = flags;
    }
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) { inducer_chain_f16_launch(g, C, Wd, 0); if (cl) g.flags += B * 8; }   // fresh counters per launch
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) { inducer_chain_f16_launch(g, C, Wd, 0); if (cl) g.flags += B * 8; }
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("inducer chain B=%d: %.1f us per launch (err %d)\n", B, ms * 100.f, (int)hipGetLastError());
#ifdef CHAIN_STAMPS
    (void)hipDeviceSynchronize();
    static unsigned long long hs[256 * 16];
    (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_chain_stamps), sizeof(hs));
    const char* names[] = {"params", "ring issue", "merge", "GEMM1", "norm_1", "y16 write", "GEMM2", "GEMM3", "norm_2", "h write", "GEMM4"};
    const int nb = (cl ? 3 * B : B) < 256 ? (cl ? 3 * B : B) : 256;
    if (!cl) {
        for (int k = 0; k < 10; ++k) {
            double d = 0;
            for (int i = 0; i < nb; ++i) d += (double)(hs[i * 16 + k + 1] - hs[i * 16 + k]);
            printf("   %-10s %8.0f ticks\n", names[k], d / nb);
        }
    } else {
        const int seq[] = {0, 1, 2, 3, 11, 4, 5, 14, 12, 6, 7, 13, 8, 9, 10};
        const char* cn[] = {"params", "ring issue", "merge", "GEMM1 own", "publish+wait 0", "load + norm_1", "y16 write", "GEMM2 own", "publish+wait 1",
                            "load u", "GEMM3 own", "publish+wait 2", "load + norm_2", "h write", "GEMM4 own"};
        for (int k = 0; k < 14; ++k) {
            double d = 0;
            for (int i = 0; i < nb; ++i) d += (double)(hs[i * 16 + seq[k + 1]] - hs[i * 16 + seq[k]]);
            printf("   %-24s %8.0f ticks\n", cn[k], d / nb);
        }
        double d = 0;
        for (int i = 0; i < nb; ++i) d += (double)(hs[i * 16 + 10] - hs[i * 16 + 0]);
        printf("   %-24s %8.0f ticks\n", "whole block", d / nb);
    }
#endif
    return 0;
}
