// Matrix-pipe rate of the scaled f8f6f4 MFMA by operand format on gfx950 (both operands fp8 / fp6 / fp4) beside the fp16 MFMA:
// is a 6-bit cross term really half the cycles — and half the TIME on this power-limited chip?  tools/probe/fp6_rate <iters>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int FMT>   // 0 fp8, 2 fp6 (e2m3), 4 fp4; -1: fp16 32x32x16 (4 per "step" = the same 64 k)
__global__ __launch_bounds__(256, 2) void rate_kernel(const int* __restrict__ src, float* __restrict__ out, int iters) {
    i32x8 a[2], b[2];
    for (int i = 0; i < 2; ++i)
        for (int e = 0; e < 8; ++e) {
            a[i][e] = src[(threadIdx.x * 16 + i * 8 + e) & 4095];
            b[i][e] = src[(threadIdx.x * 16 + i * 8 + e + 2048) & 4095];
        }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (FMT >= 0) {
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i & 1], b[i >> 1], acc[i], FMT, FMT, 0, 127, 0, 127);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f16x8 fa = __builtin_bit_cast(f16x8, __builtin_shufflevector(a[i & 1], a[i & 1], 0, 1, 2, 3));
                    const f16x8 fb = __builtin_bit_cast(f16x8, __builtin_shufflevector(b[i >> 1], b[i >> 1], 4, 5, 6, 7));
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[i], 0, 0, 0);
                }
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int FMT>
void run(const char* name, const int* src, float* out, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 512;
    hipLaunchKernelGGL(rate_kernel<FMT>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(rate_kernel<FMT>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    const double flops = 2.0 * 32 * 32 * 64 * 4.0 * iters * 4 * blocks;   // 4 accumulators x 4 waves
    printf("%-28s %8.3f ms  %8.1f TFLOP/s   (err %d)\n", name, ms, flops / ms / 1e9, (int)hipGetLastError());
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const int zero = argc > 2 ? atoi(argv[2]) : 0;
    int* src; float* out;
    (void)hipMalloc(&src, 4096 * 4); (void)hipMalloc(&out, 512 * 256 * 4);
    std::vector<int> h(4096);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = zero ? 0 : (int)(s & 0x3f3f3f3f) | 0x20202020; }   // finite in every format
    (void)hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<-1>("fp16 32x32x16 x4", src, out, iters);
    run<0>("f8f6f4 fp8 x fp8", src, out, iters);
    run<2>("f8f6f4 fp6 x fp6", src, out, iters);
    run<4>("f8f6f4 fp4 x fp4", src, out, iters);
    return 0;
}
