// Operand layout and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands (gfx950), found by
// comparing the instruction with a host reference under candidate lane/byte -> k maps.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/fp8_mfma_layout.hip -o tools/probe/fp8_mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const unsigned char* A, const unsigned char* B, float* C, int scale_a, int scale_b) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = reinterpret_cast<const int*>(A + lane * 32)[i];
        b[i] = reinterpret_cast<const int*>(B + lane * 32)[i];
    }
    f32x16 c;
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int e = 0; e < 16; ++e) C[lane * 16 + e] = c[e];
}

// e4m3fn encode of small exactly representable values
static unsigned char enc(float v) {
    if (v == 0.f) return 0;
    unsigned char s = v < 0 ? 0x80 : 0;
    v = fabsf(v);
    int e;
    float m = frexpf(v, &e);   // v = m 2^e, m in [0.5, 1)
    int E = e - 1 + 7;         // 1.xxx form exponent + bias
    int man = (int)roundf((m * 2 - 1) * 8);
    if (E <= 0) { man = (int)roundf(v / ldexpf(1.f, -9)); E = 0; }
    return s | (unsigned char)(E << 3) | (unsigned char)(man & 7);
}

int main() {
    const int M = 32, K = 64;
    std::vector<float> Af(M * K), Bf(K * M);
    srand(5);
    for (auto& v : Af) v = (float)(rand() % 9 - 4);          // integers -4..4
    for (auto& v : Bf) v = (float)(rand() % 9 - 4) * 0.5f;   // halves
    std::vector<float> ref(M * M, 0.f);
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < M; ++j) {
            float s = 0;
            for (int kk = 0; kk < K; ++kk) s += Af[i * K + kk] * Bf[kk * M + j];
            ref[i * M + j] = s;
        }
    unsigned char *dA, *dB;
    float* dC;
    hipMalloc(&dA, 64 * 32); hipMalloc(&dB, 64 * 32); hipMalloc(&dC, 64 * 16 * 4);
    const char* names[] = {"k = 32 h + j", "k = 16 (j >> 4) * 2 + 16 h + (j & 15)  [16-blocks interleaved over h]", "k = 8 (2 (j >> 3) + h) + (j & 7)  [8-blocks interleaved]",
                           "k = 2 j + h"};
    for (int hyp = 0; hyp < 4; ++hyp) {
        std::vector<unsigned char> hA(64 * 32), hB(64 * 32);
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 32; ++j) {
                const int r = lane & 31, h = lane >> 5;
                int kk;
                if (hyp == 0) kk = 32 * h + j;
                else if (hyp == 1) kk = 32 * (j >> 4) + 16 * h + (j & 15);
                else if (hyp == 2) kk = 8 * (2 * (j >> 3) + h) + (j & 7);
                else kk = 2 * j + h;
                hA[lane * 32 + j] = enc(Af[r * K + kk]);
                hB[lane * 32 + j] = enc(Bf[kk * M + r]);
            }
        hipMemcpy(dA, hA.data(), hA.size(), hipMemcpyHostToDevice);
        hipMemcpy(dB, hB.data(), hB.size(), hipMemcpyHostToDevice);
        for (int sc = 0; sc < 3; ++sc) {
            // sc 0: scales 0 (compiler may pick the unscaled form); 1: both 127 (= 2^0); 2: scale_b = 124 (2^-3)
            const int sa = sc == 0 ? 0 : 127, sb = sc == 0 ? 0 : (sc == 1 ? 127 : 124);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, sa, sb);
            std::vector<float> C(64 * 16);
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            double err = 0, ratio = 0; int n = 0;
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 16; ++e) {
                    const int col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    const float want = ref[row * M + col];
                    err = fmax(err, fabs(C[lane * 16 + e] - want));
                    if (want != 0) { ratio += C[lane * 16 + e] / want; ++n; }
                }
            printf("hypothesis %d (%s), scales (%d, %d): max |C - ref| = %g, mean C/ref = %g\n", hyp, names[hyp], sa, sb, err, ratio / n);
        }
    }
    return 0;
}
