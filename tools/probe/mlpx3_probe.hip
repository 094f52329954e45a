// Correctness + timing probe of mlp_x3_fused.hip (standalone).
#include <hip/hip_runtime.h>
#include "kernels/mlp_x3_fused.hip"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

static double rnd() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }
static double gauss() { double u = 0; for (int i = 0; i < 12; ++i) u += (double)rand() / RAND_MAX; return u - 6.0; }

static void check(int C, int rows, int rps, int act) {
    const int WD = 2 * C, B = rows / rps;
    std::vector<float> hx((size_t)rows * C), hW0((size_t)WD * C), hW2((size_t)C * WD), hb0(WD), hb2(C), ha((size_t)B * C), ho((size_t)B * C);
    for (auto& v : hx) v = (float)gauss() * 1.5f;
    for (auto& v : hW0) v = (float)(rnd() / sqrt((double)C));
    for (auto& v : hW2) v = (float)(rnd() / sqrt((double)WD));
    for (auto& v : hb0) v = (float)rnd() * 0.1f;
    for (auto& v : hb2) v = (float)rnd() * 0.1f;
    for (auto& v : ha) v = 1.f + 0.2f * (float)rnd();
    for (auto& v : ho) v = 0.2f * (float)rnd();
    float *x, *W0, *W2, *b0, *b2, *a, *o, *alpha, *st; void* img;
    const int bm = mlp_x3_fused_row_tile(C);
    (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&W0, hW0.size() * 4); (void)hipMalloc(&W2, hW2.size() * 4); (void)hipMalloc(&b0, WD * 4);
    (void)hipMalloc(&b2, C * 4); (void)hipMalloc(&a, ha.size() * 4); (void)hipMalloc(&o, ho.size() * 4); (void)hipMalloc(&alpha, 4);
    (void)hipMalloc(&st, (size_t)(rows / bm) * 2 * C * 4); (void)hipMalloc(&img, mlp_x3_stream_bytes(C));
    (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(W0, hW0.data(), hW0.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(W2, hW2.data(), hW2.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(b0, hb0.data(), WD * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(b2, hb2.data(), C * 4, hipMemcpyHostToDevice); (void)hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(o, ho.data(), ho.size() * 4, hipMemcpyHostToDevice);
    const float al = 0.9f; (void)hipMemcpy(alpha, &al, 4, hipMemcpyHostToDevice);
    int rc1 = mlp_x3_stream_launch(W0, W2, img, C, 0);
    MlpX3Args g{}; g.x = x; g.pro_a = a; g.pro_o = o; g.w_stream = img; g.b0 = b0; g.b2 = b2; g.alpha = alpha; g.act = act; g.stats = st;
    g.rows_total = rows; g.rows_per_sample = rps;
    int rc = mlp_x3_fused_launch(g, C, 0);
    (void)hipDeviceSynchronize();
    hipError_t e = hipGetLastError();
    std::vector<float> got(hx.size()), hst((size_t)(rows / bm) * 2 * C);
    (void)hipMemcpy(got.data(), x, got.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hst.data(), st, hst.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0, sterr = 0, stref = 0;
    std::vector<double> cs((size_t)(rows / bm) * 2 * C, 0.0), y(C), h(WD);
    for (int m = 0; m < rows; ++m) {
        const int bb = m / rps;
        for (int k = 0; k < C; ++k) y[k] = (double)hx[(size_t)m * C + k] * ha[(size_t)bb * C + k] + ho[(size_t)bb * C + k];
        for (int u = 0; u < WD; ++u) {
            double s = hb0[u];
            for (int k = 0; k < C; ++k) s += y[k] * hW0[(size_t)u * C + k];
            if (act == 1) s = (exp(-s * s / (2.0 * al * al)) - 0.7) / 0.28;
            if (act == 3) s = s > 0 ? s : 0;
            h[u] = s;
        }
        for (int c = 0; c < C; ++c) {
            double s = hb2[c];
            for (int u = 0; u < WD; ++u) s += h[u] * hW2[(size_t)c * WD + u];
            const double ref = hx[(size_t)m * C + c] + s, gv = got[(size_t)m * C + c];
            maxerr = fmax(maxerr, fabs(gv - ref)); maxref = fmax(maxref, fabs(ref));
            cs[((size_t)(m / bm) * 2 + 0) * C + c] += gv; cs[((size_t)(m / bm) * 2 + 1) * C + c] += gv * gv;
        }
    }
    for (size_t i = 0; i < hst.size(); ++i) { sterr = fmax(sterr, fabs(hst[i] - cs[i])); stref = fmax(stref, fabs(cs[i])); }
    printf("check C %d rows %d act %d: rc %d/%d hip %d  max err %.3e (rel %.2e)%s   stats rel %.2e%s\n", C, rows, act, rc1, rc, (int)e, maxerr, maxerr / maxref,
           maxerr / maxref < 1e-4 ? "  OK" : "  FAIL", sterr / stref, sterr / stref < 1e-5 ? "  OK" : "  FAIL");
    (void)hipFree(x); (void)hipFree(W0); (void)hipFree(W2); (void)hipFree(b0); (void)hipFree(b2); (void)hipFree(a); (void)hipFree(o); (void)hipFree(alpha); (void)hipFree(st); (void)hipFree(img);
}

int main(int argc, char** argv) {
    srand(2);
    check(384, 512, 256, 1);
    check(384, 256, 128, 3);
    check(512, 256, 128, 1);
    check(256, 256, 256, 0);
    if (argc > 1) return 0;
    const int B = 64, N = 2048, C = 384, WD = 768;
    const size_t rows = (size_t)B * N;
    float *x, *W0, *W2, *b0, *b2, *a, *o, *alpha, *st; void* img;
    (void)hipMalloc(&x, rows * C * 4); (void)hipMalloc(&W0, WD * C * 4); (void)hipMalloc(&W2, C * WD * 4); (void)hipMalloc(&b0, WD * 4); (void)hipMalloc(&b2, C * 4);
    (void)hipMalloc(&a, B * C * 4); (void)hipMalloc(&o, B * C * 4); (void)hipMalloc(&alpha, 4); (void)hipMalloc(&st, (rows / 128) * 2 * C * 4); (void)hipMalloc(&img, mlp_x3_stream_bytes(C));
    std::vector<float> h(rows * C);
    for (auto& v : h) v = (float)gauss();
    (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < (size_t)WD * C; ++i) h[i] = (float)(rnd() / 40);      // small weights: x stays bounded over the repeats
    (void)hipMemcpy(W0, h.data(), WD * C * 4, hipMemcpyHostToDevice); (void)hipMemcpy(W2, h.data(), WD * C * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(b0, h.data(), WD * 4, hipMemcpyHostToDevice); (void)hipMemcpy(b2, h.data(), C * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < (size_t)B * C; ++i) h[i] = 0.5f;
    (void)hipMemcpy(a, h.data(), B * C * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < (size_t)B * C; ++i) h[i] = 0.01f;
    (void)hipMemcpy(o, h.data(), B * C * 4, hipMemcpyHostToDevice);
    const float one = 1.f; (void)hipMemcpy(alpha, &one, 4, hipMemcpyHostToDevice);
    mlp_x3_stream_launch(W0, W2, img, C, 0);
    MlpX3Args g{}; g.x = x; g.pro_a = a; g.pro_o = o; g.w_stream = img; g.b0 = b0; g.b2 = b2; g.alpha = alpha; g.act = 1; g.stats = st;
    g.rows_total = (int)rows; g.rows_per_sample = N;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        mlp_x3_fused_launch(g, C, 0);
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i) mlp_x3_fused_launch(g, C, 0);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        const double fl = 4.0 * rows * C * WD;
        printf("mlp_x3_fused C2: %.3f ms  %.1f TF (x3 executed %.1f TF)\n", ms, fl / ms / 1e9, 3 * fl / ms / 1e9);
#ifdef MX_STAMPS
        {
            static unsigned long long hs[1024 * 8];
            (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_mx_stamps), sizeof hs);
            double d[5] = {0, 0, 0, 0, 0};
            for (int b2 = 0; b2 < 1024; ++b2) { const unsigned long long* q = hs + b2 * 8; d[0] += q[1] - q[0]; d[1] += q[2] - q[1]; d[2] += q[3] - q[2]; d[3] += q[4] - q[0]; d[4] += q[5] - q[4]; }
            printf("   cycles: y build %.0f | first chunk %.0f | second chunk %.0f | start..end of main %.0f | epilogue %.0f\n", d[0] / 1024, d[1] / 1024, d[2] / 1024, d[3] / 1024, d[4] / 1024);
        }
#endif
    }
    return 0;
}
