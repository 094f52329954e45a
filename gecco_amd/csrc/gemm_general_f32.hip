// General strided-batched fp32-MFMA GEMM for the backward pass and the materialised (training) attention:
//
//   C[z](M x N) = scale * op(A[z])(M x K) * op(B[z])(K x N) (+ bias[n])
//
// Each operand is either "row-major in its own index" (X[row][k], k contiguous — what nn.Linear weights and
// activations are for the forward product) or "k-major" (X[k][row], row contiguous — what the SAME tensors are
// for the backward products: dX = dY * W reads W[n][k] with n as the reduction index, dW = dY^T * X reads both
// dY and X with the point index m as the reduction index).  Choosing the layout per operand means no tensor is
// ever transposed in memory for the backward pass.
//
// Batch index z = z1 * zdiv + z2 with independent (outer, inner) strides per operand: (b, head) for the attention
// products, b alone for per-sample weight-gradient partials (summed afterwards by reduce_batch_kernel in a fixed
// order: the gradients are run-to-run deterministic, no float atomics).
//
// Same MFMA core as gemm_f32.hip (v_mfma_f32_32x32x2_f32, k-permuted fragments); 128x128x16 tiles, 4 waves.
// Throughput here is secondary to generality (the training step is not the headline metric).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int GBM = 128, GBN = 128, GBK = 16, GNT = 256;
constexpr int G_LDP = GBK + 4;     // row-major operand tile: [128][16 + 4]
constexpr int G_LDK = GBM + 4;     // k-major operand tile:  [16][128 + 4]
constexpr int G_OPER = 128 * G_LDP > GBK * G_LDK ? 128 * G_LDP : GBK * G_LDK;  // floats per operand per stage

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(GNT) void gemm_general_kernel(GemmGeneralArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[2 * 2 * G_OPER];
    const int tilesM = (g.M + GBM - 1) / GBM, tilesN = (g.N + GBN - 1) / GBN;
    const int per = tilesM * tilesN;
    const int z = blockIdx.x / per, tloc = blockIdx.x % per;
    const int m0 = (tloc / tilesN) * GBM, n0 = (tloc % tilesN) * GBN;
    const int z1 = z / g.zdiv, z2 = z % g.zdiv;
    const float* __restrict__ A = g.A + (size_t)z1 * g.sA1 + (size_t)z2 * g.sA2;
    const float* __restrict__ Bp = g.B + (size_t)z1 * g.sB1 + (size_t)z2 * g.sB2;
    float* __restrict__ C = g.C + (size_t)z1 * g.sC1 + (size_t)z2 * g.sC2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    // staging: 2 float4 per operand per thread per K-step
    f32x4 ra[2], rb[2];
    auto load_op = [&](const float* __restrict__ X, int ld, int row0, int rows, bool km, int kt, f32x4* dst) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = i * GNT + tid;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (km) {  // tile [16 k][128 rows], 16-byte chunks along the row index
                const int kr = f >> 5, c4 = f & 31;
                const int k = kt * GBK + kr, row = row0 + c4 * 4;
                if (k < g.K && row < rows) v = *reinterpret_cast<const f32x4*>(X + (size_t)k * ld + row);
            } else {   // tile [128 rows][16 k], 16-byte chunks along k
                const int rr = f >> 2, c4 = f & 3;
                const int k = kt * GBK + c4 * 4, row = row0 + rr;
                if (k < g.K && row < rows) v = *reinterpret_cast<const f32x4*>(X + (size_t)row * ld + k);
            }
            dst[i] = v;
        }
    };
    auto store_op = [&](float* S, bool km, const f32x4* src) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = i * GNT + tid;
            if (km) *reinterpret_cast<f32x4*>(S + (f >> 5) * G_LDK + (f & 31) * 4) = src[i];
            else *reinterpret_cast<f32x4*>(S + (f >> 2) * G_LDP + (f & 3) * 4) = src[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (g.K + GBK - 1) / GBK;
    load_op(A, g.lda, m0, g.M, A_KM, 0, ra);
    load_op(Bp, g.ldb, n0, g.N, B_KM, 0, rb);
    store_op(smem, A_KM, ra);
    store_op(smem + G_OPER, B_KM, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int s = kt & 1;
        const float* As = smem + s * 2 * G_OPER;
        const float* Bs = As + G_OPER;
        if (kt + 1 < nk) {
            load_op(A, g.lda, m0, g.M, A_KM, kt + 1, ra);
            load_op(Bp, g.ldb, n0, g.N, B_KM, kt + 1, rb);
        }
#pragma unroll
        for (int kk = 0; kk < GBK / 8; ++kk) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (wm * 2 + i) * 32 + r;
                if (A_KM) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fa[i][e] = As[(kk * 8 + 4 * h + e) * G_LDK + row];
                } else {
                    fa[i] = *reinterpret_cast<const f32x4*>(As + row * G_LDP + kk * 8 + 4 * h);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = (wn * 2 + j) * 32 + r;
                if (B_KM) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) fb[j][e] = Bs[(kk * 8 + 4 * h + e) * G_LDK + row];
                } else {
                    fb[j] = *reinterpret_cast<const f32x4*>(Bs + row * G_LDP + kk * 8 + 4 * h);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i][e], fb[j][e], acc[i][j]);
        }
        if (kt + 1 < nk) {
            store_op(smem + (s ^ 1) * 2 * G_OPER, A_KM, ra);
            store_op(smem + (s ^ 1) * 2 * G_OPER + G_OPER, B_KM, rb);
        }
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + (wn * 2 + j) * 32 + r;
        const bool nok = n < g.N;
        const float bias = (g.bias && nok) ? g.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * 2 + i) * 32 + mfma_row(e, h);
                if (nok && m < g.M) C[(size_t)m * g.ldc + n] = acc[i][j][e] * g.scale + bias;
            }
    }
}

// out[i] = sum_z parts[z * stride + i]   (fixed order z = 0, 1, ..: deterministic parameter gradients)
// 16 bytes per thread, eight independent loads in flight per round; the additions stay strictly in z order.
__global__ __launch_bounds__(256) void reduce_batch_kernel(const float* __restrict__ parts, float* __restrict__ out,
                                                           size_t n, int Z, size_t stride, int accumulate) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n && !(stride & 3)) {
        f32x4 s = accumulate ? *reinterpret_cast<const f32x4*>(out + i) : f32x4{0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 8 <= Z; z += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(parts + (size_t)(z + u) * stride + i));
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < Z; ++z) s += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(parts + (size_t)z * stride + i));
        *reinterpret_cast<f32x4*>(out + i) = s;
    } else {
        for (size_t j = i; j < n && j < i + 4; ++j) {
            float s = accumulate ? out[j] : 0.f;
            for (int z = 0; z < Z; ++z) s += parts[(size_t)z * stride + j];
            out[j] = s;
        }
    }
}

// The same sum for FEW outputs and MANY partials (bias / activation-parameter gradients: n <= a few hundred, Z in the
// thousands): one thread per output would walk Z loads serially.  A block owns COLS outputs; its 256 / COLS z-lanes sum
// interleaved partials (z = lane, lane + ZL, ..) in z order, and a fixed-shape LDS tree combines the lanes: the result
// depends on (Z, COLS) only — deterministic — though not on the strict z order of the kernel above.
template <int COLS>
__global__ __launch_bounds__(256) void reduce_batch_wide_kernel(const float* __restrict__ parts, float* __restrict__ out,
                                                                size_t n, int Z, size_t stride, int accumulate) {
    constexpr int ZL = 256 / COLS;
    __shared__ float red[256];
    const int c = threadIdx.x % COLS, zl = threadIdx.x / COLS;
    const size_t i = (size_t)blockIdx.x * COLS + c;
    float s = 0.f;
    if (i < n) {
        int z = zl;
        for (; z + 7 * ZL < Z; z += 8 * ZL) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = parts[(size_t)(z + u * ZL) * stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < Z; z += ZL) s += parts[(size_t)z * stride + i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int o = ZL / 2; o > 0; o >>= 1) {
        if (zl < o) red[threadIdx.x] += red[threadIdx.x + o * COLS];
        __syncthreads();
    }
    if (zl == 0 && i < n) out[i] = accumulate ? out[i] + red[c] : red[c];
}

}  // namespace

int gemm_general_launch(const GemmGeneralArgs& g, hipStream_t st) {
    if ((g.lda & 3) || (g.ldb & 3)) return -2;
    if (g.a_kmajor ? (g.M & 3) : (g.K & 3)) return -2;   // the contiguous index is read in 16-byte pieces
    if (g.b_kmajor ? (g.N & 3) : (g.K & 3)) return -2;
    const int tilesM = (g.M + GBM - 1) / GBM, tilesN = (g.N + GBN - 1) / GBN;
    const dim3 grid((unsigned)((size_t)g.Z * tilesM * tilesN));
    if (g.a_kmajor && g.b_kmajor) hipLaunchKernelGGL((gemm_general_kernel<true, true>), grid, dim3(GNT), 0, st, g);
    else if (g.a_kmajor) hipLaunchKernelGGL((gemm_general_kernel<true, false>), grid, dim3(GNT), 0, st, g);
    else if (g.b_kmajor) hipLaunchKernelGGL((gemm_general_kernel<false, true>), grid, dim3(GNT), 0, st, g);
    else hipLaunchKernelGGL((gemm_general_kernel<false, false>), grid, dim3(GNT), 0, st, g);
    return (int)hipGetLastError();
}

int reduce_batch_launch(const float* parts, float* out, size_t n, int Z, size_t stride, int accumulate,
                        hipStream_t st) {
    if (Z >= 64 && n <= 16384) {   // few outputs, many partials: spread the partials over the lanes
        if (n >= 16)
            hipLaunchKernelGGL(reduce_batch_wide_kernel<16>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, parts, out, n, Z,
                               stride, accumulate);
        else
            hipLaunchKernelGGL(reduce_batch_wide_kernel<1>, dim3((unsigned)n), dim3(256), 0, st, parts, out, n, Z, stride,
                               accumulate);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, parts, out, n, Z,
                       stride, accumulate);
    return (int)hipGetLastError();
}
