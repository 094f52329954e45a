// Diagnostic: the one-launch point MLP of the "w2" mode (gecco_amd/csrc/mlp_fused_w.hip) alone at the C2 shape (B 64 x N 2048, d 384 -> 768 ->
// 384): three row tiles against a float64 reference on the host (the pre-activations the kernel dumps, the output from the kernel's own
// rounded hidden layer, the output from the exact hidden layer, the column partials), then the launch time and per-block stamps
// (-DMFW_STAMPS).  Build: tools/probe/build_mlpfw.sh
#include <hip/hip_runtime.h>
#define MFW_DEV
#include "../../gecco_amd/csrc/mlp_fused_w.hip"
#include <math.h>
#include <string.h>
#include <stdio.h>
#include <vector>

static float h2f(float v) { return (float)(_Float16)v; }
// e2m3 with a block scale as the kernel forms it: 32 values -> scale byte, values rounded to fp6 (nearest even, saturating at 7.5)
static int scale_byte_h(double m) {
    if (!(m > 0)) return 127;
    const float f = (float)m * (16.0f / 15.0f);
    unsigned u; memcpy(&u, &f, 4);
    const int e = (int)(u >> 23) - 2;
    return e < 1 ? 1 : e;
}
static double q6(double v) {   // v already divided by the scale
    const double a = fabs(v);
    double q;
    if (a >= 7.5) q = 7.5;
    else if (a < 1.0) q = nearbyint(a * 8.0) / 8.0;
    else { int e; frexp(a, &e); const double step = ldexp(1.0, e - 1 - 3); q = nearbyint(a / step) * step; if (q > 7.5) q = 7.5; }
    return v < 0 ? -q : q;
}
static void quant32(const double* v, double* out) {
    double m = 0; for (int i = 0; i < 32; ++i) m = fmax(m, fabs(v[i]));
    const double sc = ldexp(1.0, scale_byte_h(m) - 127);
    for (int i = 0; i < 32; ++i) out[i] = q6(v[i] / sc) * sc;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, N = argc > 2 ? atoi(argv[2]) : 2048, C = 384, Wd = 768;
    const size_t rows = (size_t)B * N;
    float *x, *out, *pa, *po, *W0, *W2, *b0, *b2, *alpha, *stats, *dbg;
    void* img;
    (void)hipMalloc(&x, rows * C * 4); (void)hipMalloc(&out, rows * C * 4); (void)hipMalloc(&pa, B * C * 4); (void)hipMalloc(&po, B * C * 4);
    (void)hipMalloc(&W0, (size_t)Wd * C * 4); (void)hipMalloc(&W2, (size_t)Wd * C * 4); (void)hipMalloc(&b0, Wd * 4); (void)hipMalloc(&b2, C * 4);
    (void)hipMalloc(&alpha, 4); (void)hipMalloc(&stats, (size_t)B * (N / 128) * 2 * C * 4); (void)hipMalloc(&img, mlp_fused_w_image_bytes(C, Wd));
    (void)hipMalloc(&dbg, rows * Wd * 4);
    std::vector<float> hx(rows * C), hpa(B * C), hpo(B * C), hW0((size_t)Wd * C), hW2((size_t)Wd * C), hb0(Wd), hb2(C);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
    for (auto& v : hx) v = 1.5f * rnd();
    for (auto& v : hpa) v = 1.0f + 0.3f * rnd();
    for (auto& v : hpo) v = 0.2f * rnd();
    for (auto& v : hW0) v = 0.08f * rnd();
    for (auto& v : hW2) v = 0.05f * rnd();
    for (auto& v : hb0) v = 0.1f * rnd();
    for (auto& v : hb2) v = 0.1f * rnd();
    const int mode = argc > 3 ? atoi(argv[3]) : 0;   // 1: fp16-exact W1 (its second term vanishes), 2: fp16-exact y (pa 1, po 0), 4: fp16-exact W2
    if (mode & 1) for (auto& v : hW0) v = h2f(v);
    if (mode & 2) { for (auto& v : hx) v = h2f(v); for (auto& v : hpa) v = 1.f; for (auto& v : hpo) v = 0.f; }
    if (mode & 4) for (auto& v : hW2) v = h2f(v);
    printf("mode %d\n", mode);
    (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(pa, hpa.data(), hpa.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(po, hpo.data(), hpo.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W0, hW0.data(), hW0.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W2, hW2.data(), hW2.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(b0, hb0.data(), hb0.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(b2, hb2.data(), hb2.size() * 4, hipMemcpyHostToDevice);
    const float a = 0.9f;
    (void)hipMemcpy(alpha, &a, 4, hipMemcpyHostToDevice);
    // the kernel works on s u (s = sqrt(log2(e) / 2) / |alpha| folded into mlp.0's weights and bias): the host checks in that domain
    const float ws = 0.84932180028801907f / a;
    std::vector<float> hW0s(hW0.size()), hb0s(hb0.size());
    for (size_t i = 0; i < hW0.size(); ++i) hW0s[i] = hW0[i] * ws;
    for (size_t i = 0; i < hb0.size(); ++i) hb0s[i] = hb0[i] * ws;
    int rc = mlp_fused_w_image_launch(W0, b0, W2, b2, img, C, Wd, alpha, 1, 0);
    MlpWArgs g{};
    g.x = x; g.out = out; g.pro_a = pa; g.pro_o = po; g.w_img = img; g.alpha = alpha; g.act = 1; g.stats = stats; g.B = B; g.rows = N;
    g.dbg_u = dbg;
    rc |= mlp_fused_w_launch(g, C, Wd, 0);
    (void)hipDeviceSynchronize();
    printf("launch rc %d, err %d\n", rc, (int)hipGetLastError());

    {   // ---- the weight stream against the weights: the lo operands of phase-1 stage (t, half) = (3, 1), group gi = 1, both terms and blocks
        std::vector<unsigned> im(mlp_fused_w_image_bytes(C, Wd) / 4);
        (void)hipMemcpy(im.data(), img, im.size() * 4, hipMemcpyDeviceToHost);
        auto dec6 = [](unsigned v) { const int sg = (v >> 5) & 1, ex = (v >> 3) & 3, mn = v & 7; const double m = ex == 0 ? mn / 8.0 : (1.0 + mn / 8.0) * ldexp(1.0, ex - 1); return sg ? -m : m; };
        const int t = 3, half = 1, gi = 1, g2 = 3 * half + gi;
        const unsigned* st = im.data() + (size_t)(2 * t + half) * 44 * 256;
        for (int term = 0; term < 2; ++term) for (int j = 0; j < 2; ++j) {
            double worst = 0, big = 0;
            for (int l = 0; l < 64; ++l) {
                const int r = l & 31, h = l >> 5, c0 = 2 + 14 * gi;
                unsigned dw[6];
                for (int e = 0; e < 4; ++e) dw[e] = st[(c0 + (term ? 11 : 8) + j) * 256 + l * 4 + e];
                for (int e = 0; e < 2; ++e) dw[4 + e] = st[(c0 + (term ? 13 : 10)) * 256 + j * 128 + l * 2 + e];
                const int sb = (st[l * 4 + gi] >> (8 * (2 * term + j))) & 0xff;
                for (int i = 0; i < 32; ++i) {
                    const int bit = 6 * i;
                    unsigned long long w = (unsigned long long)dw[bit >> 5] | ((bit >> 5) + 1 < 6 ? (unsigned long long)dw[(bit >> 5) + 1] << 32 : 0ull);
                    const double v = dec6((unsigned)((w >> (bit & 31)) & 63)) * ldexp(1.0, sb - 127);
                    const float wf = hW0s[(size_t)(64 * t + 32 * j + r) * C + 64 * g2 + 16 * (i >> 3) + 8 * h + (i & 7)];
                    const double ref = term ? (double)wf : (double)wf - (double)h2f(wf);
                    worst = fmax(worst, fabs(v - ref)); big = fmax(big, fabs(ref));
                }
            }
            printf("stream check: phase 1 lo operand term %d block %d: max |err| %.3e of max %.3e\n", term, j, worst, big);
        }
    }
    // ---- reference on three row tiles
    const int T = N / 128, ntiles = B * T;
    const int check[3] = {0, ntiles / 2 + 5 < ntiles ? ntiles / 2 + 5 : 0, ntiles - 1};
    std::vector<float> ho(128 * C), hu(128 * Wd), hst(2 * C);
    double wu_j[2] = {0, 0}, wu_h[2] = {0, 0}, wu_r[4] = {0, 0, 0, 0};
    double worst_e = 0;
    double worst_u = 0, big_u = 0, worst_o1 = 0, worst_o2 = 0, big_o = 0, worst_s = 0, big_s = 0;
    for (int ci = 0; ci < 3; ++ci) {
        const int tile = check[ci], b = tile / T;
        const size_t r0 = (size_t)tile * 128;
        (void)hipMemcpy(ho.data(), out + r0 * C, ho.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hu.data(), dbg + r0 * Wd, hu.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hst.data(), stats + (size_t)tile * 2 * C, hst.size() * 4, hipMemcpyDeviceToHost);
        std::vector<double> cs(2 * C, 0.0);
        for (int m = 0; m < 128; ++m) {
            double y[384], uex[768], hk[768], hex[768];
            for (int k = 0; k < C; ++k) {
                float v = fmaf(hx[(r0 + m) * C + k], hpa[b * C + k], hpo[b * C + k]);
                v = fminf(fmaxf(v, -3584.f), 3584.f);
                y[k] = v;
            }
            // the kernel's operands: yh, the fp6 forms of yh and of 2^11 (y - yh) per (64-k group, lane half: k = 64 g + 16 s + 8 h + e)
            double yh[384], y6[384], yl6[384];
            for (int k = 0; k < C; ++k) yh[k] = h2f((float)y[k]);
            for (int gk = 0; gk < 6; ++gk) for (int hh = 0; hh < 2; ++hh) {
                double a[32], bq[32], lo[32], lq[32];
                for (int i = 0; i < 32; ++i) { const int k = 64 * gk + 16 * (i >> 3) + 8 * hh + (i & 7); a[i] = yh[k]; lo[i] = h2f((float)((y[k] - yh[k]) * 2048.0)); }
                quant32(a, bq); quant32(lo, lq);
                for (int i = 0; i < 32; ++i) { const int k = 64 * gk + 16 * (i >> 3) + 8 * hh + (i & 7); y6[k] = bq[i]; yl6[k] = lq[i] / 2048.0; }
            }
            for (int n = 0; n < Wd; ++n) {
                {   // emulated u
                    double e = hb0s[n];
                    for (int gk = 0; gk < 6; ++gk) for (int hh = 0; hh < 2; ++hh) {
                        double wl[32], wq[32], w6[32], wf[32];
                        for (int i = 0; i < 32; ++i) { const int k = 64 * gk + 16 * (i >> 3) + 8 * hh + (i & 7); const float w = hW0s[(size_t)n * C + k]; wl[i] = (double)w - (double)h2f(w); wf[i] = w; }
                        quant32(wl, wq); quant32(wf, w6);
                        for (int i = 0; i < 32; ++i) { const int k = 64 * gk + 16 * (i >> 3) + 8 * hh + (i & 7); e += yh[k] * (double)h2f(hW0s[(size_t)n * C + k]) + y6[k] * wq[i] + yl6[k] * w6[i]; }
                    }
                    worst_e = fmax(worst_e, fabs(e - (double)hu[m * Wd + n]));
                }
                double acc = hb0[n];
                for (int k = 0; k < C; ++k) acc += y[k] * (double)hW0[(size_t)n * C + k];
                uex[n] = acc;
                const double uk = hu[m * Wd + n];
                worst_u = fmax(worst_u, fabs(uk - acc * ws));
                wu_j[(n >> 5) & 1] = fmax(wu_j[(n >> 5) & 1], fabs(uk - acc * ws));
                wu_h[(n >> 2) & 1] = fmax(wu_h[(n >> 2) & 1], fabs(uk - acc * ws));
                wu_r[m >> 5] = fmax(wu_r[m >> 5], fabs(uk - acc * ws));
                big_u = fmax(big_u, fabs(acc * ws));
                hk[n] = h2f((float)((exp2(-uk * uk) - 0.7) / 0.28));   // uk = s u
                hex[n] = (exp(-acc * acc / (2.0 * a * a)) - 0.7) / 0.28;
            }
            for (int c = 0; c < C; ++c) {
                double o1 = (double)hb2[c] + hx[(r0 + m) * C + c], o2 = o1;
                for (int n = 0; n < Wd; ++n) {
                    o1 += hk[n] * (double)hW2[(size_t)c * Wd + n];
                    o2 += hex[n] * (double)hW2[(size_t)c * Wd + n];
                }
                const double got = ho[m * C + c];
                worst_o1 = fmax(worst_o1, fabs(got - o1));
                worst_o2 = fmax(worst_o2, fabs(got - o2));
                big_o = fmax(big_o, fabs(o2 - hx[(r0 + m) * C + c]));
                cs[c] += got;
                cs[C + c] += got * got;
            }
        }
        for (int i = 0; i < 2 * C; ++i) {
            worst_s = fmax(worst_s, fabs(cs[i] - hst[i]));
            big_s = fmax(big_s, fabs(cs[i]));
        }
    }
    printf("pre-activation u:            max |err| %.3e of max |u| %.3e  -> %.2e\n", worst_u, big_u, worst_u / big_u);
    printf("   u against the host emulation of the kernel's arithmetic: max |diff| %.3e\n", worst_e);
    printf("   u err by hidden block j: %.2e %.2e | by lane half h: %.2e %.2e | by wave: %.2e %.2e %.2e %.2e\n", wu_j[0], wu_j[1], wu_h[0], wu_h[1], wu_r[0], wu_r[1], wu_r[2], wu_r[3]);
    printf("output vs rounded-hidden ref: max |err| %.3e of max |mlp| %.3e -> %.2e\n", worst_o1, big_o, worst_o1 / big_o);
    printf("output vs exact ref:          max |err| %.3e                    -> %.2e\n", worst_o2, worst_o2 / big_o);
    printf("column partials:              max |err| %.3e of %.3e -> %.2e\n", worst_s, big_s, worst_s / big_s);

    // ---- time
    g.dbg_u = nullptr;
    g.out = x;   // in place, as the network runs it
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) mlp_fused_w_launch(g, C, Wd, 0);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) mlp_fused_w_launch(g, C, Wd, 0);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-14s B %d N %d: %.1f us  (%.1f TF of 2MNK, two products)   err %d\n", argv[0], B, N, ms * 1e3, 4.0 * rows * C * Wd / ms / 1e9, (int)hipGetLastError());
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) mlp_fused_w_image_launch(W0, b0, W2, b2, img, C, Wd, alpha, 1, 0);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("   weight stream build: %.1f us\n", ms * 100);
#ifdef MFW_STAMPS
    (void)hipDeviceSynchronize();
    static unsigned long long hs[1024 * 8];
    (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_mfw_stamps), sizeof(hs));
    const int nb = ntiles < 256 ? ntiles : 256;
    double d[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < nb; ++i) for (int k = 0; k < 5; ++k) d[k] += (double)(hs[i * 8 + k + 1] - hs[i * 8 + k]);
    printf("   stamps of the LAST tile (ticks, mean over %d blocks): start->tile %.0f | coefficients + y build %.0f | phase 1 %.0f | fp6 forms + phase 2 %.0f | partials %.0f\n",
           nb, d[0] / nb, d[1] / nb, d[2] / nb, d[3] / nb, d[4] / nb);
    double de = 0, da = 0;
    double dp = 0;
    for (int i = 0; i < nb; ++i) { de += (double)hs[i * 8 + 6]; da += (double)(hs[i * 8 + 7] & 0xffffffffull); dp += (double)(hs[i * 8 + 7] >> 32); }
    printf("   of the last tile: waits + barriers of the 48 stage entries %.0f, the 12 activation blocks %.0f, the 12 output-block epilogues %.0f\n", de / nb, da / nb, dp / nb);
#endif
    return 0;
}
