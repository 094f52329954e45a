// Weight gradients of the training path in split-bf16 arithmetic, gfx950:
//
//     C[g, n, k] = sum over the samples z of group g, sum over rows m:  A[z, m, n] * B[z, m, k]
//
// with A = dY (Z, R, N) and B = X (Z, R, K) both row-major — dW = dY^T X of every nn.Linear (autograd of
// reference models/set_transformer.py / mlp.py under Lightning's loss.backward(), diffusion.py:213-222).  The
// contraction runs over the ROWS of both operands, so both MFMA fragments are transposed reads: a (32 rows x 128
// columns) fp32 tile of each operand is split into bf16 hi | lo planes (a = hi + lo, gemm_f32_dma.hip) stored in the
// 4-row x 32-column blocks of attention_x3.hip's value tile and read with ds_read_b64_tr_b16 (the hardware transpose);
// the same k permutation applies to both fragments, so products pair up.  Three v_mfma_f32_32x32x16_bf16 per product
// (a_lo b_hi + a_hi b_lo + a_hi b_hi), fp32 accumulation: ~2^-16 per product at 16x the fp32 matrix rate / 3.
// 128 x 128 output tile per block, 2 x 2 waves of 64 x 64, 32 rows per step, next step's tiles in registers while this
// one multiplies, two LDS stages (one barrier per step).  One partial per GROUP of samples, summed afterwards in a
// fixed order (reduce_batch_kernel): bit-reproducible gradients, 1/8 of the partial traffic of one per sample.
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

// 4 fp32 -> 4 bf16 hi (top 16 bits) and 4 bf16 lo = rne(x - hi), each packed in two dwords
__device__ __forceinline__ void tn_split4(const f32x4& x, u32x2& hi, u32x2& lo) {
    bf16x4 l;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const unsigned ua = __float_as_uint(x[2 * p]), uc = __float_as_uint(x[2 * p + 1]);
        hi[p] = __builtin_amdgcn_perm(uc, ua, 0x07060302u);
        l[2 * p] = (__bf16)(x[2 * p] - __uint_as_float(ua & 0xFFFF0000u));
        l[2 * p + 1] = (__bf16)(x[2 * p + 1] - __uint_as_float(uc & 0xFFFF0000u));
    }
    lo = __builtin_bit_cast(u32x2, l);
}

// element offset of (row, col) in a [32][128] plane: 4-row x 32-column blocks of 128 elements (256 B = all 64 banks); the
// row's 64-byte slot inside its block is rotated by the column block, so the four column blocks a store instruction touches
// for one row land on four different bank groups (unrotated they share one: SQ_LDS_BANK_CONFLICT was 33 % of the LDS-active
// cycles); a transposed read still covers whole blocks (its four rows take the four slots in another order)
__device__ __forceinline__ int tn_off(int row, int col) {
    return ((row >> 2) * 4 + (col >> 5)) * 128 + ((row + (col >> 5)) & 3) * 32 + (col & 31);
}

constexpr int TN_PLANE = 32 * 128;   // u16 per plane

// PRO: the AdaGN apply on the B operand (TnArgs::pro_a) — its own instantiation, so the plain weight gradients run the code
// they ran before it existed
template <bool PRO>
__global__ __launch_bounds__(256, 2) void gemm_tn_x3_kernel(TnArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16* lds = reinterpret_cast<u16*>(smem);   // [2 stages][A hi | A lo | B hi | B lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int tilesK = (g.K + 127) / 128;
    // the output tiles of one sample group read the same rows of dY / X: the XCD-aware virtual grid puts them on ONE XCD, so its
    // L2 serves the re-reads (dispatch order deals consecutive blocks to different XCDs, each with its own L2)
    const int ntile = (int)gridDim.x;
    const int vb = g.xcd ? xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y)) : (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int bx = vb % ntile, by = vb / ntile;
    const int n0 = (bx / tilesK) * 128, k0 = (bx % tilesK) * 128;
    // N, K need not fill the last tile (the conditioner's 96 / 192 / 48 / 672-wide layers): a thread's four columns are the
    // same in every step, so one predicate per operand zero-fills what lies beyond the matrix
    const bool aok = n0 + (tid & 31) * 4 < g.N, bok = k0 + (tid & 31) * 4 < g.K;
    const int z0 = by * g.group, z1 = min(g.Z, z0 + g.group);
    const int msteps = g.R / 32, nsteps = (z1 - z0) * msteps;

    // global tile loads: thread -> 4 x (row, 4 columns) of each operand
    f32x4 ra[4], rb[4];
    const unsigned va = aok ? (unsigned)(((tid >> 5) * g.lda + (tid & 31) * 4) * 4) : 0x7fffffffu;
    const unsigned vb_ = bok ? (unsigned)(((tid >> 5) * g.ldb + (tid & 31) * 4) * 4) : 0x7fffffffu;
    f32x4 pa4 = {1.f, 1.f, 1.f, 1.f}, po4 = {0.f, 0.f, 0.f, 0.f};   // AdaGN coefficients of the thread's four X columns, sample zc
    int zc = -1;
    auto load = [&](int s) {
        const int z = z0 + s / msteps, m0 = (s % msteps) * 32;
        if (PRO && z != zc && bok) {
            pa4 = *reinterpret_cast<const f32x4*>(g.pro_a + (size_t)z * g.K + k0 + (tid & 31) * 4);
            po4 = *reinterpret_cast<const f32x4*>(g.pro_o + (size_t)z * g.K + k0 + (tid & 31) * 4);
        }
        zc = z;
        // raw buffer loads (gemm_tn_f16.hip): one per-lane byte offset per operand — out of range, so zeros, beyond the matrix — and
        // scalar offsets for the slab and the row group, instead of eight 64-bit addresses
        const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + (size_t)z * g.sA), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Bm + (size_t)z * g.sB), 0, 0x7fffffff, 0x00020000);
        const unsigned sa0 = (unsigned)((m0 * g.lda + n0) * 4), sb0 = (unsigned)((m0 * g.ldb + k0) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, va, sa0 + (unsigned)(i * 8 * g.lda * 4), 0));
            rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, vb_, sb0 + (unsigned)(i * 8 * g.ldb * 4), 0));
        }
    };
    // bias gradient = column sums of dY: the blocks of the first K tile add up the rows they stage anyway
    const bool want_cs = g.colsum != nullptr && k0 == 0;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    auto store = [&](int stage) {
        u16* st = lds + stage * 4 * TN_PLANE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + i * 256, row = f >> 5, c4 = f & 31, o = tn_off(row, c4 * 4);
            u32x2 hi, lo;
            if (want_cs) cs += ra[i];
            tn_split4(ra[i], hi, lo);
            *reinterpret_cast<u32x2*>(st + o) = hi;
            *reinterpret_cast<u32x2*>(st + TN_PLANE + o) = lo;
            // the AdaGN apply, if any, here — where the loaded values are consumed a step after their loads were issued (applied
            // at the load it would make every step wait for its own global loads)
            tn_split4(PRO ? (bok ? rb[i] * pa4 + po4 : rb[i]) : rb[i], hi, lo);
            *reinterpret_cast<u32x2*>(st + 2 * TN_PLANE + o) = hi;
            *reinterpret_cast<u32x2*>(st + 3 * TN_PLANE + o) = lo;
        }
    };
    // transposed-read addressing (attention_x3.hip): lane 4q+p of a 16-lane group points at row q, columns 4p .. 4p+3
    const int tq = (lane & 15) >> 2, tp = lane & 3, tcol = 16 * ((lane >> 4) & 1) + 4 * tp;
    auto frag = [&](const u16* plane, int sg, int blk) -> bf16x8 {
        typedef __attribute__((address_space(3))) s16x4* lp;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(plane + tn_off(16 * sg + 4 * h + tq, blk * 32 + tcol)));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(plane + tn_off(16 * sg + 8 + 4 * h + tq, blk * 32 + tcol)));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (nsteps > 0) load(0);
    for (int s = 0; s < nsteps; ++s) {
        const int stage = s & 1;
        store(stage);
        __syncthreads();   // stage complete; every wave is past its reads of the other stage's previous contents
        if (s + 1 < nsteps) load(s + 1);
        const u16* st = lds + stage * 4 * TN_PLANE;
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = frag(st, sg, wn * 2 + i);
                al[i] = frag(st + TN_PLANE, sg, wn * 2 + i);
                bh[i] = frag(st + 2 * TN_PLANE, sg, wk * 2 + i);
                bl[i] = frag(st + 3 * TN_PLANE, sg, wk * 2 + i);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }
    if (want_cs) {   // (block-uniform) the 8 threads of a column group, summed in thread order
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(smem);
        red[tid] = cs;
        __syncthreads();
        if (tid < 32 && n0 + tid * 4 < g.N) {
            f32x4 s = red[tid];
#pragma unroll
            for (int j = 1; j < 8; ++j) s += red[tid + 32 * j];
            *reinterpret_cast<f32x4*>(g.colsum + (size_t)by * g.N + n0 + tid * 4) = s;
        }
    }
    float* Cb = g.C + (size_t)by * g.N * g.K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * 64 + i * 32 + mfma_row(e, h), k = k0 + wk * 64 + j * 32 + r;
                if (n < g.N && k < g.K) Cb[(size_t)n * g.K + k] = acc[i][j][e];
            }
}

}  // namespace

bool gemm_tn_x3_supported(const TnArgs& g) {
    // one sample's rows are addressed with 31-bit byte offsets (buffer loads)
    return g.Z > 0 && g.group > 0 && g.R >= 32 && g.R % 32 == 0 && g.N > 0 && g.K > 0 && !(g.N & 3) && !(g.K & 3) && !(g.lda & 3) &&
           !(g.ldb & 3) && (size_t)g.R * g.lda * 4 < 0x7fffffffu && (size_t)g.R * g.ldb * 4 < 0x7fffffffu;
}

int gemm_tn_x3_launch(const TnArgs& g, hipStream_t st) {
    if (!gemm_tn_x3_supported(g)) return -9;
    const int G = (g.Z + g.group - 1) / g.group;
    const size_t lds = (size_t)2 * 4 * TN_PLANE * 2;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr = true;
    }
    static int xcd = -1;
    if (xcd < 0) {
        const char* e = getenv("GECCO_TN_XCD");   // 0: plain dispatch order (A/B runs)
        xcd = (e && atoi(e) == 0) ? 0 : 1;
    }
    TnArgs ga = g;
    ga.xcd = xcd;
    const dim3 grid(((g.N + 127) / 128) * ((g.K + 127) / 128), G);
    if (g.pro_a) hipLaunchKernelGGL(gemm_tn_x3_kernel<true>, grid, dim3(256), lds, st, ga);
    else hipLaunchKernelGGL(gemm_tn_x3_kernel<false>, grid, dim3(256), lds, st, ga);
    return (int)hipGetLastError();
}
