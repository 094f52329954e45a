// Diagnostic: what matrix rate does an LDS-fed fp16 MFMA loop reach per wave-tile shape and waves per SIMD?  No global
// traffic, no barriers: every wave reads its A / B fragments (ds_read_b128, conflict-free layout) from a block-resident
// LDS panel and issues v_mfma_f32_32x32x16_f16 on TM x TN blocks of 32 x 32 — the ceiling of any LDS-operand GEMM
// with that register blocking.  Prints TFLOP/s chip-wide and the LDS bytes per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int TM, int TN, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += WAVES * 64) smem[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    const int r = lane & 31, h = lane >> 5;
    // row r of a [32][32 k] fp16 block at r * 64 B, chunk (2 h + c) ^ ((r >> 2) & 3): the GEMM kernels' layout
    int off[2];
    for (int c = 0; c < 2; ++c) off[c] = r * 16 + (((2 * h + c) ^ ((r >> 2) & 3)) << 2);
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
        const float* base = smem + (it & 3) * 2048;   // K-step slot
        f16x8 a[TM][2], b[TN][2];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c) a[i][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(base + i * 512 + off[c]));
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) b[j][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(base + 8192 + j * 512 + off[c]));
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][c], b[j][c], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j) s += acc[i][j][0] + acc[i][j][15];
    if (s == 12345.678f) out[tid] = s;
}

template <int TM, int TN, int WAVES>
void run(const char* name) {
    float* out; (void)hipMalloc(&out, 4096);
    const int iters = 4096, blocks = 256 * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<TM, TN, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<TM, TN, WAVES>), dim3(blocks), dim3(WAVES * 64), 65536, 0, out, iters);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((k<TM, TN, WAVES>), dim3(blocks), dim3(WAVES * 64), 65536, 0, out, iters);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double fl = (double)blocks * WAVES * iters * TM * TN * 2 * 32768.0;
    printf("%-28s %7.1f TFLOP/s   %.2f KiB of LDS reads per MFMA   (%d waves/SIMD, err %d)\n", name, fl / ms / 1e9,
           (double)(TM + TN) / (TM * TN), WAVES / 4, (int)hipGetLastError());
}

int main() {
    run<1, 2, 4>("32 x 64, 1 wave/SIMD");
    run<1, 2, 8>("32 x 64, 2 waves/SIMD");
    run<1, 4, 4>("32 x 128, 1 wave/SIMD");
    run<1, 4, 8>("32 x 128, 2 waves/SIMD");
    run<2, 2, 4>("64 x 64, 1 wave/SIMD");
    run<2, 2, 8>("64 x 64, 2 waves/SIMD");
    run<2, 4, 4>("64 x 128, 1 wave/SIMD");
    run<2, 4, 8>("64 x 128, 2 waves/SIMD");
    run<4, 4, 4>("128 x 128, 1 wave/SIMD");
    return 0;
}
