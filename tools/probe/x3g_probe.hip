// Correctness + timing probe of gemm_x3_planes.hip (standalone: hipcc, no torch).
//   ./x3g_probe            correctness on small shapes vs a host fp64 reference, then timing at the C2 call sites
#include <hip/hip_runtime.h>
#include "kernels/gemm_x3_planes.hip"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

static double rnd() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }
static double gauss() { double u = 0; for (int i = 0; i < 12; ++i) u += (double)rand() / RAND_MAX; return u - 6.0; }

struct Dev { float *x, *W, *C, *R, *bias, *stats, *alpha; void *Y, *img, *Cp; };

static float time_ms(const X3Args& g, int it = 10) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    gemm_x3_planes_launch(g, 0); gemm_x3_planes_launch(g, 0);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < it; ++i) gemm_x3_planes_launch(g, 0);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / it;
}

static int check(int rows_total, int rps, int K, int Nout, bool residual, bool stats, bool planes, int act, int n_split) {
    std::vector<float> hx((size_t)rows_total * K), hW((size_t)Nout * K), hb(Nout), hR((size_t)rows_total * Nout);
    for (auto& v : hx) v = (float)gauss();
    for (auto& v : hW) v = (float)(rnd() / sqrt((double)K));
    for (auto& v : hb) v = (float)rnd() * 0.1f;
    for (auto& v : hR) v = (float)gauss();
    float *x, *W, *C, *R, *bias, *st, *alpha; void *Y, *img;
    (void)hipMalloc(&x, hx.size() * 4); (void)hipMalloc(&W, hW.size() * 4); (void)hipMalloc(&C, (size_t)rows_total * Nout * 4);
    (void)hipMalloc(&R, hR.size() * 4); (void)hipMalloc(&bias, Nout * 4); (void)hipMalloc(&st, (size_t)(rows_total / 64) * 2 * Nout * 4);
    (void)hipMalloc(&alpha, 4); (void)hipMalloc(&Y, hx.size() * 4); (void)hipMalloc(&img, planes_image_bytes(Nout, K));
    (void)hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(R, hR.data(), hR.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(bias, hb.data(), Nout * 4, hipMemcpyHostToDevice);
    float al = 0.9f; (void)hipMemcpy(alpha, &al, 4, hipMemcpyHostToDevice);
    (void)hipMemset(C, 0xFF, (size_t)rows_total * Nout * 4);
    affine_split_planes_launch(x, nullptr, nullptr, Y, rows_total, rps, K, 0);
    split_planes_image_launch(W, img, Nout, K, K, 0);
    X3Args g{}; g.Y = Y; g.Wimg = img; g.bias = bias; g.alpha = alpha; g.act = act; g.residual = residual ? R : nullptr; g.C = C;
    g.stats = stats ? st : nullptr; g.rows_total = rows_total; g.rows_per_sample = rps; g.K = K; g.Nout = Nout; g.ldc = Nout; g.ldr = Nout;
    g.c_planes = planes; g.skew = -1;
    float* C2 = nullptr;
    if (n_split) { (void)hipMalloc(&C2, (size_t)rows_total * (Nout - n_split) * 4); g.C2 = C2; g.bias2 = bias + n_split; g.n_split = n_split;
                   g.ldc = n_split; g.ldc2 = Nout - n_split; }
    int rc = gemm_x3_planes_launch(g, 0);
    (void)hipDeviceSynchronize();
    hipError_t e = hipGetLastError();
    std::vector<float> hC((size_t)rows_total * Nout), hC2;
    (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
    if (n_split) { hC2.resize((size_t)rows_total * (Nout - n_split)); (void)hipMemcpy(hC2.data(), C2, hC2.size() * 4, hipMemcpyDeviceToHost); }
    const int bm = gemm_x3_planes_row_tile(Nout, n_split);
    std::vector<float> hst((size_t)(rows_total / bm) * 2 * Nout);
    if (stats) (void)hipMemcpy(hst.data(), st, hst.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0, sterr = 0, stref = 0;
    std::vector<double> colsum((size_t)(rows_total / bm) * 2 * Nout, 0.0);
    const int mstep = stats ? 1 : 7;
    for (int m = 0; m < rows_total; m += mstep)
        for (int n = 0; n < Nout; n += (stats ? 1 : 3)) {
            double s = hb[n];
            for (int k = 0; k < K; ++k) s += (double)hx[(size_t)m * K + k] * hW[(size_t)n * K + k];
            if (act == 1) s = (exp(-s * s / (2.0 * al * al)) - 0.7) / 0.28;
            if (act == 3) s = s > 0 ? s : 0;
            if (residual) s += hR[(size_t)m * Nout + n];
            double got;
            if (planes) {
                const unsigned short* p = reinterpret_cast<const unsigned short*>(hC.data()) + (size_t)m * Nout * 2 + (size_t)(n >> 5) * 64 + (n & 31);
                unsigned int hi = (unsigned int)p[0] << 16, lo = (unsigned int)p[32] << 16;
                float fh, fl; __builtin_memcpy(&fh, &hi, 4); __builtin_memcpy(&fl, &lo, 4); got = (double)fh + fl;
            } else if (n_split && n >= n_split) got = hC2[(size_t)m * (Nout - n_split) + n - n_split];
            else got = hC[(size_t)m * (n_split ? n_split : Nout) + n];
            maxerr = fmax(maxerr, fabs(got - s)); maxref = fmax(maxref, fabs(s));
            if (stats) { colsum[((size_t)(m / bm) * 2 + 0) * Nout + n] += got; colsum[((size_t)(m / bm) * 2 + 1) * Nout + n] += got * got; }
        }
    if (stats) for (size_t i = 0; i < hst.size(); ++i) { sterr = fmax(sterr, fabs(hst[i] - colsum[i])); stref = fmax(stref, fabs(colsum[i])); }
    printf("check rows %d K %d Nout %d res %d stats %d planes %d act %d split %d: rc %d hip %d  max err %.3e (rel %.2e)%s\n", rows_total, K, Nout,
           residual, stats, planes, act, n_split, rc, (int)e, maxerr, maxerr / maxref, maxerr / maxref < 2e-4 ? "  OK" : "  FAIL");
    if (stats) printf("      stats err %.3e (rel %.2e)%s\n", sterr, sterr / stref, sterr / stref < 1e-4 ? "  OK" : "  FAIL");
    (void)hipFree(x); (void)hipFree(W); (void)hipFree(C); (void)hipFree(R); (void)hipFree(bias); (void)hipFree(st); (void)hipFree(alpha); (void)hipFree(Y); (void)hipFree(img);
    if (C2) (void)hipFree(C2);
    return 0;
}

int main(int argc, char** argv) {
    srand(1);
    check(512, 256, 384, 768, false, false, false, 0, 0);
    check(512, 256, 384, 384, true, true, false, 0, 0);
    check(512, 256, 768, 384, true, true, false, 0, 0);
    check(512, 256, 384, 768, false, false, true, 1, 0);
    check(512, 256, 384, 1152, false, false, false, 0, 768);
    check(512, 256, 512, 512, true, true, false, 0, 0);
    check(512, 256, 512, 1024, false, false, false, 3, 0);
    check(512, 256, 128, 256, false, false, true, 0, 0);
    if (argc > 1) return 0;
    // timing at the C2 call sites, random normal data
    const int B = 64, N = 2048;
    const size_t rows = (size_t)B * N;
    float *x, *W, *C, *R, *bias, *st, *alpha; void *Y, *img;
    (void)hipMalloc(&x, rows * 768 * 4); (void)hipMalloc(&W, 1152 * 768 * 4); (void)hipMalloc(&C, rows * 1152 * 4); (void)hipMalloc(&R, rows * 384 * 4);
    (void)hipMalloc(&bias, 1152 * 4); (void)hipMalloc(&st, (size_t)(rows / 64) * 2 * 1152 * 4); (void)hipMalloc(&alpha, 4);
    (void)hipMalloc(&Y, rows * 768 * 4); (void)hipMalloc(&img, 1152 * 768 * 4);
    std::vector<float> h(rows * 768);
    const bool zero = getenv("ZERO") != nullptr;
    for (auto& v : h) v = zero ? 0.f : (float)gauss();
    (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(R, h.data(), rows * 384 * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < 1152 * 768; ++i) h[i] = zero ? 0.f : (float)(rnd() / 20);
    (void)hipMemcpy(W, h.data(), 1152 * 768 * 4, hipMemcpyHostToDevice); (void)hipMemcpy(bias, h.data(), 1152 * 4, hipMemcpyHostToDevice);
    float one = 1.f; (void)hipMemcpy(alpha, &one, 4, hipMemcpyHostToDevice);
    struct Site { const char* name; int K, Nout, split; bool res, stats, planes; int act; } sites[] = {
        {"kv_proj|q_proj", 384, 1152, 768, false, false, false, 0}, {"out_proj+res+stats", 384, 384, 0, true, true, false, 0},
        {"mlp.0+act -> planes", 384, 768, 0, false, false, true, 1}, {"mlp.2+res+stats", 768, 384, 0, true, true, false, 0}};
    for (int rep = 0; rep < 2; ++rep)
    for (auto& s : sites) {
        affine_split_planes_launch(x, nullptr, nullptr, Y, rows, N, s.K, 0);
        split_planes_image_launch(W, img, s.Nout, s.K, s.K, 0);
        X3Args g{}; g.Y = Y; g.Wimg = img; g.bias = bias; g.alpha = alpha; g.act = s.act; g.residual = s.res ? R : nullptr; g.C = C;
        g.stats = s.stats ? st : nullptr; g.rows_total = (int)rows; g.rows_per_sample = N; g.K = s.K; g.Nout = s.Nout; g.ldc = s.Nout; g.ldr = s.Nout;
        g.c_planes = s.planes;
        if (s.split) { g.C2 = C + rows * 768; g.bias2 = bias + s.split; g.n_split = s.split; g.ldc = s.split; g.ldc2 = s.Nout - s.split; }
        const double fl = 2.0 * rows * s.K * s.Nout;
        printf("%-22s", s.name);
        for (int sk : {0}) { g.skew = sk * (s.K / 384); const float t = time_ms(g); printf("  skew %d: %.3f ms %5.1f TF |", g.skew, t, fl / t / 1e9); }
        printf("\n");
#ifdef X3_STAMPS
        {
            static unsigned long long hs[256 * 2 * 16 * 8];
            (void)hipDeviceSynchronize();
            (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_x3_stamps), sizeof hs);
            const int ntl = (int)((rows / 128) * (s.Nout / 384) / 256);
            for (int grp = 0; grp < 2; ++grp) {
                double d01 = 0, d12 = 0, d23 = 0, d30 = 0; int n = 0, n2 = 0;
                for (int b = 0; b < 256; ++b)
                    for (int t = 1; t < ntl && t < 16; ++t) {
                        const unsigned long long* q = hs + ((size_t)(b * 2 + grp) * 16 + t) * 8;
                        d01 += (double)(q[1] - q[0]); d12 += (double)(q[2] - q[1]); d23 += (double)(q[3] - q[2]); ++n;
                        if (t + 1 < ntl && t + 1 < 16) { d30 += (double)(q[8] - q[3]); ++n2; }
                    }
                printf("      group %d: main %.0f | epilogue: residual wait %.0f, compute+stores %.0f | gap to next main %.0f  (s_memtime ticks, 100 MHz => x21 cycles)\n", grp, d01 / n, d12 / n,
                       d23 / n, n2 ? d30 / n2 : 0.0);
            }
        }
#endif
    }
    {   // the cast pass
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a, 0);
        for (int i = 0; i < 10; ++i) affine_split_planes_launch(x, bias, bias, Y, rows, N, 384, 0);
        (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("affine_split_planes (B N 384): %.3f ms\n", ms / 10);
    }
    return 0;
}
