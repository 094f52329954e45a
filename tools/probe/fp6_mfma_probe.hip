// Semantics check of the fp6 (e2m3) path on gfx950: v_cvt_scalef32_pk32_fp6_f16 / 2xpk16_fp6_f32 (scale = divisor?), the packed layout as the
// f8f6f4 MFMA reads it (cbsz = blgp = 2), per-lane E8M0 block scales (byte select).  Exactly representable values -> the product is exact.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x6 __attribute__((ext_vector_type(6)));

template <int OPSEL>
__global__ void probe(const float* A, const float* B, const int* ea, const int* eb, float* C, float* dec) {
    // one wave.  lane (r, h): A operand = row r of A (32 x 64), its 32 values k = 32 h + i;  B operand = column r of B (64 x 32)
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f16x32 va;
    f32x16 vb0, vb1;
    for (int i = 0; i < 32; ++i) va[i] = (_Float16)A[r * 64 + 32 * h + i];
    for (int i = 0; i < 16; ++i) { vb0[i] = B[(32 * h + i) * 32 + r]; vb1[i] = B[(32 * h + 16 + i) * 32 + r]; }
    const int sa = ea[lane], sb = eb[lane];           // E8M0 bytes of this lane's blocks
    const float fa = ldexpf(1.f, sa - 127), fb = ldexpf(1.f, sb - 127);
    const u32x6 pa = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(va, fa);
#ifdef B_VIA_F16
    f16x32 vbh;
    for (int i = 0; i < 16; ++i) { vbh[i] = (_Float16)vb0[i]; vbh[16 + i] = (_Float16)vb1[i]; }
    const u32x6 pb = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vbh, fb);
#else
    const u32x6 pb = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(vb0, vb1, fb);
#endif
    i32x8 oa = {0, 0, 0, 0, 0, 0, 0, 0}, ob = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = 0; e < 6; ++e) { oa[e] = (int)pa[e]; ob[e] = (int)pb[e]; }
    // the scale bytes sit in byte 2 of a register whose other bytes are junk: opsel = 2 must pick them
    const int ra = OPSEL == 2 ? (0x11000022 | (sa << 16)) : sa, rb = OPSEL == 2 ? (0x33000044 | (sb << 16)) : sb;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(oa, ob, acc, 2, 2, OPSEL, ra, OPSEL, rb);
    // accumulator: lane (r, h) holds column r, rows (e & 3) + 8 (e >> 2) + 4 h
    for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
    const f32x32 d = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(pb, 1.0f);
    if (lane == 5) for (int i = 0; i < 32; ++i) dec[i] = d[i] * fb;
}

int main() {
    const float set[] = {0.f, 0.125f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f, 7.5f, -0.5f, -1.f, -2.f, -3.f, -6.f, 1.25f};
    float *dA, *dB, *dC, *dd; int *dea, *deb;
    (void)hipMalloc(&dA, 32 * 64 * 4); (void)hipMalloc(&dB, 64 * 32 * 4); (void)hipMalloc(&dC, 32 * 32 * 4); (void)hipMalloc(&dd, 128);
    (void)hipMalloc(&dea, 256); (void)hipMalloc(&deb, 256);
    const char* names[] = {"all scales 2^0", "one scale per row / column (both lane halves equal)", "one scale per lane (row, 32-k block)",
                           "per lane, k = 32 t + 16 h + j layout (lane half h: 16 values of each 32-k block), scale per (row, lane)"};
    for (int mode = 0; mode < 4; ++mode)
        for (int opsel = 0; opsel <= 2; opsel += 2) {
            std::vector<float> A(32 * 64), B(64 * 32), C(32 * 32);
            std::vector<int> ea(64), eb(64);
            unsigned s = 7;
            auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8); };
            for (int l = 0; l < 64; ++l) { ea[l] = 127 - 6 + (int)(rnd() % 10); eb[l] = 127 - 3 + (int)(rnd() % 7); }
            if (mode == 0) for (int l = 0; l < 64; ++l) ea[l] = eb[l] = 127;
            if (mode == 1) for (int l = 0; l < 32; ++l) { ea[l + 32] = ea[l]; eb[l + 32] = eb[l]; }
            // the kernel gives lane (r, h) the values k = 32 h + i; the scale used to BUILD them is that of the lane holding them
            for (int r = 0; r < 32; ++r)
                for (int k = 0; k < 64; ++k) {
                    A[r * 64 + k] = set[rnd() % 16] * ldexpf(1.f, ea[r + 32 * (k / 32)] - 127);
                    B[k * 32 + r] = set[rnd() % 16] * ldexpf(1.f, eb[r + 32 * (k / 32)] - 127);
                }
            (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(dea, ea.data(), 256, hipMemcpyHostToDevice); (void)hipMemcpy(deb, eb.data(), 256, hipMemcpyHostToDevice);
            if (mode == 3) continue;
            if (opsel == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, dA, dB, dea, deb, dC, dd);
            else hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, dA, dB, dea, deb, dC, dd);
            (void)hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0, big = 0;
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double ref = 0;
                    for (int k = 0; k < 64; ++k) ref += (double)A[i * 64 + k] * B[k * 32 + j];
                    worst = fmax(worst, fabs(ref - C[i * 32 + j]));
                    big = fmax(big, fabs(ref));
                }
            printf("%-60s opsel %d: max |diff| %.3e (max |ref| %.3e)\n", names[mode], opsel, worst, big);
            if (mode == 0 && opsel == 0) {
                std::vector<float> dec(32);
                (void)hipMemcpy(dec.data(), dd, 128, hipMemcpyDeviceToHost);
                printf("   B pack of lane 5 decoded (index: got / B[k = i][5]):");
                for (int i = 0; i < 32; ++i) printf(" %d:%g/%g", i, dec[i], B[i * 32 + 5]);
                printf("\n");
            }
        }
    return 0;
}
