// The point MLP of a layer as ONE launch, the hidden layer never leaves the registers (gfx950):
//
//     x += mlp.2( act( mlp.0( AdaGN(x) ) ) ),  + the GroupNorm column partials of the new x      (written for feature_dim 384, width 768;
//                                                                                                     a template over the width: 128, 256, 512 too — WCfg)
//
// Reference: models/set_transformer.py:164-166 (x = x + self.mlp(self.mlp_norm(x, t_embed))), models/mlp.py:5-39,
// models/activation.py:17-24, models/normalization.py:36-44.
//
// Arithmetic ("w2" mode).  y = AdaGN(x) and both weights carry two terms, the hidden layer one (its second term is what no
// register budget holds, see below): with y = yh + yl (yh = fp16(y)), W = Wh + Wl, h = act(u), hh = fp16(h):
//
//     u   = yh W1h + fp6(yh) fp6(W1l) + fp6(yl) fp6(W1) + b1        (fp16 MFMA + two fp6 x fp6 block-scaled MFMAs: 1.5 matrix units)
//     out = hh W2h + fp6(hh) fp6(W2l) + b2 + x                       (1.25 matrix units)
//
// The fp6 operands (e2m3) carry one E8M0 scale per lane and 64-k group — the lane's 32 values ARE a scale block of
// v_mfma_scale_f32_32x32x64_f8f6f4 (gemm_h8_astat.hip's "h6" terms).  Per-site emulation (tools/experiments/precision_search.py,
// profiles/r03_precision_search.txt): dropping ONLY mlp.2's activation term moves the network's F_x from 6.6e-5 to 3.6e-4; dropping
// mlp.0's as well gives 6.2e-4 (over the 5e-4 bar of this mode), so mlp.0 keeps it.
//
// Structure.  Neither "output stationary" (32 x 384 accumulators = 192 registers, beside 96 + 36 of y) nor any split of the
// products over two waves of a SIMD fits 256 registers per wave, so a block is 4 waves of 512 registers, ONE
// per SIMD, 32 rows each, and the work of a 128-row tile is ordered so that the big operand changes between two phases:
//   phase 1  (12 hidden tiles of 64 columns): y stationary (96 + 36 registers), W1 streams; the tile's activation is kept as the
//            fp16 fragments of mlp.2's ROW operand — an accumulator of v_mfma_32x32 holds one point per lane and 16 hidden
//            columns in its registers, which is an operand fragment as it stands (k order 8 (e >> 2) + 4 h + (e & 3), the W2
//            stream is written in that order) — 192 registers for all 768 hidden columns, parked in the accumulator file;
//   phase 2  (12 output blocks of 32 columns): h stationary, W2 streams, 16 accumulator registers per block; the block's
//            residual rows arrive by LDS-DMA in the wave's 4 KiB tile while its products run; at the block's end accumulator +
//            bias + residual go through that tile (GroupNorm column sums on the way) and leave as 16-byte row pieces whose stores
//            stay in flight under the next block (two forms of running this epilogue UNDER the next block's matrix
//            instructions were measured and lost: profiles/r05b_negative_results.txt).
// One instruction stream per SIMD has no partner to hide anything: fragment reads of set n + 1 are issued before the matrix
// instructions of set n and fenced there (sched_barrier: left alone the scheduler serialises read -> wait -> MFMA to save
// registers), the LDS-DMA pieces of a stage are spread over its sets (one to two between matrix groups), the stream is
// FRAGMENT-MAJOR — every operand of every matrix instruction is a contiguous 1 KiB chunk, lane l at byte 16 l — so a fragment read is
// base + immediate, conflict-free, with no swizzle arithmetic, and the activation of tile t runs under the matrix work it does
// not depend on.  Ring: 3 slots of 44 KiB; a stage is half a unit (hidden tile / output block): 44 chunks in phase 1, 36 in
// phase 2; one block barrier per stage (30 - 36 matrix instructions).  Blocks are persistent (grid = CUs): the stream wraps,
// so the first stages of the next row tile arrive during the last output blocks of this one.
//
// Other widths (round 6).  Everything above is counted in 64-k groups NG = feature_dim / 64 (WCfg<NG>): 2, 4 and 6 groups run the same
// schedule with GS = NG / 2 groups per stage.  NG = 8 (feature_dim 512) does not fit 512 registers at once (y 176, the hidden fragments 256,
// two accumulator pairs 64, two operand sets 64): it runs TWO PASSES over the hidden width with y kept — phase 1 / phase 2 over hidden tiles
// 0 .. 7, the partial product + bias + residual written to `out`; then phase 1 / phase 2 over tiles 8 .. 15, the second partial product
// added to the first's output (read back like the residual rows), statistics there — with quarter-tile stages (3 x 32 KiB) and the
// hidden tiles' fp6 forms made where they are used: C4's point MLP 409 + 316 us as two launches -> 617 us.
//
// Probe with per-ingredient switches and stamps: tools/probe/mlpw_probe.hip (main loops), tools/probe/mlpfw_probe.hip (this kernel
// against a float64 reference).
//
// Translation units: this file alone is feature_dim 384 + the entry points (MFW_PART 0); mlp_fused_w_p1.hip (512) and mlp_fused_w_p2.hip
// (128, 256) include it with MFW_PART 1 / 2 — sixteen instantiations of a fully unrolled kernel built side by side instead of in a row.
#include "common.h"
#include "h8_scales.h"
#include "kernels.h"

#ifndef MFW_PART
#define MFW_PART 0
#endif

#include <stdlib.h>

#include <utility>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x6 __attribute__((ext_vector_type(6)));

// Shape of an instantiation: feature_dim = 64 NG (NG = 2, 4, 6, 8: 128, 256, 384, 512), width = 2 feature_dim.
// NG = 8 does not fit the register file in one go (y 128 + 48, the hidden fragments 256, two accumulator pairs 64 and two operand sets 64 are
// 560 of 512): it runs TWO PASSES over the hidden width with y kept — pass 0: hidden tiles 0 .. 7, then every output block's partial product
// + bias + residual written to `out`; pass 1: hidden tiles 8 .. 15, the blocks' second partial products added to what pass 0 left (read back
// the way the residual rows are: LDS-DMA), statistics there.  Its stages are a quarter of a hidden tile (2 groups: 3 x 32 KiB of ring).
template <int NG_>
struct WCfg {
    static constexpr int NG = NG_;                 // 64-k groups of mlp.0
    static constexpr int C = 64 * NG, WD = 2 * C;
    static constexpr int NT = 2 * NG;              // 64-column hidden tiles = 64-k groups of mlp.2
    static constexpr int NB = 2 * NG;              // 32-column output blocks
    static constexpr int NPASS = NG == 8 ? 2 : 1;  // passes over the hidden width
    static constexpr int NTP = NT / NPASS;         // hidden tiles of a pass
    static constexpr int GS = NG == 8 ? 2 : NG / 2;   // groups of a phase-1 stage = tile pairs of a phase-2 stage (half a block's NTP / 2 tiles)
    static constexpr int SPT = NG / GS;            // phase-1 stages per hidden tile (2; 4 at NG = 8)
    static constexpr int NSETS = 2 * GS;           // sets of matrix instructions per stage: two per group / one per hidden tile
    static constexpr int CH1 = (2 + 14 * GS + 3) & ~3, CH2 = (1 + 11 * GS + 3) & ~3;   // chunks of a stage, phase 1 / phase 2 (44 / 36 at NG = 6)
    static constexpr int SLOT = (CH1 > CH2 ? CH1 : CH2) * 1024;                        // ring slot (bytes)
    static constexpr int NP1 = CH1 / 4, NP2 = CH2 / 4;                                  // 1 KiB pieces per wave and stage
    static constexpr int NST1 = NTP * SPT, NST2 = 2 * NB;                               // stages of a pass, phase 1 / phase 2
    static constexpr size_t STREAM = (size_t)NPASS * ((size_t)NST1 * CH1 + (size_t)NST2 * CH2) * 1024;   // 1920 KiB per layer at NG = 6
    static constexpr int HB1 = 4 * GS;             // scale bytes per lane in a phase-1 stage header
    static constexpr int HDR_BYTES = NPASS * (NST1 * 64 * HB1 + NST2 * 64 * 8);
    // the previous hidden tile's activation (8 register quads) rides in the first ACT_SETS of a tile's TSETS = 2 NG sets, the next tile's bias
    // (whose stage is the next one only during the tile's LAST stage) in its last BIAS_SETS
    static constexpr int TSETS = NSETS * SPT;
    static constexpr int ACT_SETS = NG >= 6 ? 8 : NG, QA = 8 / ACT_SETS;
    static constexpr int BIAS_SETS = NG == 2 ? 2 : 4, QB = 8 / BIAS_SETS, BIAS_START = TSETS - BIAS_SETS;
    static constexpr int COLP = 4 * 2 * C * 4;
    static constexpr int LDS = 3 * SLOT + 4 * 4096 + COLP;
    static_assert(NG == 2 || NG == 4 || NG == 6 || NG == 8, "feature_dim 128, 256, 384 or 512");
    static_assert(NTP == 4 * GS && TSETS == 2 * NG && BIAS_START >= ACT_SETS && BIAS_SETS <= NSETS, "stage structure");
    static_assert(LDS <= 160 * 1024 && COLP >= 2 * C * 4, "one block per CU");
};
constexpr int W_NS = 3;
constexpr int W_STG = 4096;                // wave-private staging tile of the y build: [32 rows][64 fp16]
// LDS (bytes): ring | 4 wave-private tiles of 4 KiB (the y build's staging; in phase 2 the [32 rows][32 columns] fp32 tile through which the
// residual rows come in and the results leave in 16-byte pieces) | the tile's column partials [4 waves][2][C] floats, whose first 2 C floats hold
// the sample's AdaGN coefficients pa | po while the y build runs.  The biases ride in the weight stream's stage headers.
static_assert(W_STG == 32 * 32 * 4, "the phase-2 tile");

constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 0xF) | (0x7 << 4) | ((lgkm & 0xF) << 8) | ((vm >> 4) << 14); }
template <int N>
__device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(waitcnt_imm(N, 0xF)); }
__device__ __forceinline__ void wait_lgkm0() { __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0)); }

template <int... I, class F>
__device__ __forceinline__ void static_for_w(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { static_for_w(std::make_integer_sequence<int, N>{}, f); }
#define W_IC(v) std::integral_constant<int, (v)>{}

#ifdef MFW_DIAG_NOMFMA
__device__ __forceinline__ f32x16 w_keep16(f16x8 a, f16x8 b, f32x16 c) { asm volatile("" ::"v"(a), "v"(b)); return c; }
__device__ __forceinline__ f32x16 w_keep6(i32x8 a, i32x8 b, f32x16 c, int sa, int sb) { asm volatile("" ::"v"(a), "v"(b), "v"(sa), "v"(sb)); return c; }
#define W_MFMA16(a, b, c) w_keep16(a, b, c)
#define W_MFMA6_(a, b, c, osa, sa, osb, sb) w_keep6(a, b, c, sa, sb)
#else
#define W_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define W_MFMA6_(a, b, c, osa, sa, osb, sb) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, osa, sa, osb, sb)
#endif
#define W_MFMA6(a, b, c, ...) W_MFMA6_(a, b, c, __VA_ARGS__)   // (the scale arguments come out of W_SB: expanded first)
#define W_SCHED() __builtin_amdgcn_sched_barrier(0)
// a scale byte of a register holding four: selected by the instruction (op_sel), or (diagnostic build) shifted down first
#ifdef MFW_DIAG_OPSEL0
#define W_SB(reg, byte) 0, ((reg) >> (8 * (byte)))
#else
#define W_SB(reg, byte) (byte), (reg)
#endif

#ifdef MFW_STAMPS
__device__ unsigned long long g_mfw_stamps[1024 * 8];
#define WSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_mfw_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
// accumulated ticks of the last tile: [6] waits + barriers of the stage entries, [7] the activation blocks
#define WACC_DECL unsigned long long wacc_enter = 0, wacc_act = 0, wacc_epi = 0, wacc_t0 = 0
#define WACC_RESET() do { wacc_enter = 0; wacc_act = 0; wacc_epi = 0; } while (0)
#define WACC_BEGIN() do { wacc_t0 = __builtin_amdgcn_s_memtime(); } while (0)
#define WACC_END(which) do { which += __builtin_amdgcn_s_memtime() - wacc_t0; } while (0)
#define WACC_STORE() do { if (threadIdx.x == 0 && blockIdx.x < 1024) { g_mfw_stamps[blockIdx.x * 8 + 6] = wacc_enter; g_mfw_stamps[blockIdx.x * 8 + 7] = wacc_act | (wacc_epi << 32); } } while (0)
#else
#define WSTAMP(i)
#define WACC_DECL
#define WACC_RESET()
#define WACC_BEGIN()
#define WACC_END(which)
#define WACC_STORE()
#endif

// E8M0 byte of the block scale for a block whose largest magnitude is m: m / 2^(byte - 127) in (3.75, 7.5] (e2m3's top binades)
__device__ __forceinline__ int w_scale_byte(float m) {
    const int e = (int)(__float_as_uint(m * (16.0f / 15.0f)) >> 23) - 2;
    return m > 0.f ? (e < 1 ? 1 : e) : 127;
}
__device__ __forceinline__ float w_scale_of(int byte) { return __uint_as_float((unsigned)byte << 23); }
template <int K>
__device__ __forceinline__ f16x8 w_sub(const f16x32& v) {
    return __builtin_shufflevector(v, v, 8 * K, 8 * K + 1, 8 * K + 2, 8 * K + 3, 8 * K + 4, 8 * K + 5, 8 * K + 6, 8 * K + 7);
}
// largest magnitude of a lane's 32 halves (gemm_h8_astat.hip's h6_absmax32: sign bits masked, packed fp16 maxima)
__device__ __forceinline__ float w_absmax32(const f16x32& v) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    auto ab = [](f16x8 a) {
        u32x4 u = __builtin_bit_cast(u32x4, a);
        u &= 0x7fff7fffu;
        return __builtin_bit_cast(f16x8, u);
    };
    const f16x8 m = __builtin_elementwise_max(__builtin_elementwise_max(ab(w_sub<0>(v)), ab(w_sub<1>(v))), __builtin_elementwise_max(ab(w_sub<2>(v)), ab(w_sub<3>(v))));
    const h2 m2 = __builtin_elementwise_max(__builtin_elementwise_max(h2{m[0], m[1]}, h2{m[2], m[3]}), __builtin_elementwise_max(h2{m[4], m[5]}, h2{m[6], m[7]}));
    return fmaxf((float)m2[0], (float)m2[1]);
}
__device__ __forceinline__ i32x8 w_op6(const u32x4& a, const u32x2& b) {
    return i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], 0, 0};
}
__device__ __forceinline__ i32x8 w_op6(const u32x6& a) { return i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)a[4], (int)a[5], 0, 0}; }

// hidden column (inside its 64-column tile) / k of mlp.2 (inside its 64-k group) of element i (0 .. 31) of a lane of half h: the order in
// which two 32 x 32 accumulators (blocks j = i >> 4) hold a point's 32 values of the tile
__host__ __device__ __forceinline__ int w_kmap(int h, int i) { return 32 * (i >> 4) + 16 * ((i >> 3) & 1) + 8 * ((i >> 2) & 1) + 4 * h + (i & 3); }

// ---------------------------------------------------------------------------------------------------------------------
// The weight stream of a layer (WCfg::STREAM bytes), in consumption order.  1 KiB chunks; lane l = 32 h + r.  GS = NG / 2 (written for NG = 6:
// GS = 3; the smaller shapes have GS = 2 / 1 groups per stage, their stages padded with zero chunks to a multiple of four).
// Phase 1, stage (t, half), groups g = GS half + gi:  chunk 0 header: byte 4 gi + 2 term + j of lane l = scale byte of the lo operand
//   (term 0: W1 - fp16(W1), term 1: W1; hidden block j) of that lane; chunk 1: (half 0) floats 0 .. 63 = mlp.0's bias of the tile's 64
//   hidden columns (x the activation's argument scale, like the weights); group gi at chunk 2 + 14 gi:
//     + 2 s + j (s = 0 .. 3): fp16(W1[64 t + 32 j + r][64 g + 16 s + 8 h + e]), e = 0 .. 7
//     + 8 + j: dwords 0 - 3 of the term-0 operand of block j;  + 10: its dwords 4 - 5, [j][lane] 8 bytes each
//     + 11 + j, + 13: the same for term 1
//   a lo operand = 32 values X[64 t + 32 j + r][64 g + 16 (i >> 3) + 8 h + (i & 7)] / 2^(scale - 127) as fp6 (e2m3), element i at bit 6 i
// Phase 2, stage (nb, half), hidden tiles t = NG half + 2 pi + tt:  chunk 0 header: byte 2 pi + tt = scale byte of tile t's lo operand, dword 2
//   of lane l = mlp.2's bias of column 32 nb + (l & 31);
//   pair pi at chunk 1 + 11 pi:  + 5 tt + s: fp16(W2[32 nb + r][64 t + kmap(h, 8 s + e)]);  + 5 tt + 4: dwords 0 - 3 of the lo operand
//   (W2 - fp16(W2) at [32 nb + r][64 t + kmap(h, i)]);  + 10: dwords 4 - 5, [tt][lane];  the chunks past 1 + 11 GS unused.
// One thread per 16-byte item.
struct WLo {
    u32x6 pk;
    int sb;
};
__device__ __forceinline__ WLo w_lo_pack(const float (&v)[32]) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) m = fmaxf(m, fabsf(v[i]));
    WLo r;
    r.sb = w_scale_byte(m);
    const float inv = __uint_as_float((unsigned)(254 - r.sb) << 23);   // 2^(127 - sb): exact
    f16x32 vh;
#pragma unroll
    for (int i = 0; i < 32; ++i) vh[i] = (_Float16)(v[i] * inv);
    r.pk = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vh, 1.0f);
    return r;
}
// the lo operand of phase 1: (t, g, j, term) of lane (r, h)
__device__ __forceinline__ WLo w_lo1(const float* __restrict__ W1, int C, float ws, int t, int g, int j, int term, int r, int h) {
    const float* src = W1 + (size_t)(64 * t + 32 * j + r) * C + 64 * g + 8 * h;
    float v[32];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(src + 16 * s + 4 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // ONE rounded product feeds the hi rounding and the lo difference in every copy of this function (header, 16-byte part, 8-byte
                // part are written by different threads: a contracted multiply-subtract in one of them moves a block maximum across a binade)
                float w = w4[e] * ws;
                asm volatile("" : "+v"(w));
                v[8 * s + 4 * c + e] = term == 0 ? w - (float)(_Float16)w : w;
            }
        }
    return w_lo_pack(v);
}
// the lo operand of phase 2: (nb, t) of lane (r, h)
__device__ __forceinline__ WLo w_lo2(const float* __restrict__ W2, int WD, int nb, int t, int r, int h) {
    const float* src = W2 + (size_t)(32 * nb + r) * WD + 64 * t + 4 * h;
    float v[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {   // elements 4 q .. 4 q + 3: k = 8 q + 4 h + e
        const f32x4 w = *reinterpret_cast<const f32x4*>(src + 8 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = w[e] - (float)(_Float16)w[e];
    }
    return w_lo_pack(v);
}

// `ws`: mlp.0's weights enter the stream multiplied by the Gaussian activation's argument scale (w_act_scale): the kernel's pre-activations
// are s u, its activation exp2(-(s u)^2) — one multiply per hidden value less in a loop that is bound by its instruction count
__device__ __forceinline__ float w_act_scale(const float* alpha, int act) {
    return (act == 1 || act == 2) ? 0.84932180028801907f / fabsf(alpha[0]) : 1.f;   // sqrt(log2(e) / 2) / |alpha|
}
struct MlpwImageJobs {     // the layers of one launch (blockIdx.y)
    enum { MAX = 16 };
    const float* W1[MAX];
    const float* b1[MAX];
    const float* W2[MAX];
    const float* b2[MAX];
    unsigned* img[MAX];
    const float* alpha[MAX];
    int act, n;
};
// Threads: one per 16-byte item of the stream, then one per scale BYTE of the stage headers (a header item alone would form its lane's 12
// lo operands one after the other — 12 x 32 dependent loads: the launch's critical path, 18 us for 1.9 MB).
template <int NG>
__global__ void mlpw_image_kernel(MlpwImageJobs jobs) {
    typedef WCfg<NG> K;
    constexpr int C = K::C, WD = K::WD, CH1 = K::CH1, CH2 = K::CH2, GS = K::GS, HB1 = K::HB1, SPT = K::SPT, NTP = K::NTP, NSETS = K::NSETS;
    constexpr size_t P1_ITEMS = (size_t)K::NST1 * CH1 * 64, PASS_ITEMS = P1_ITEMS + (size_t)K::NST2 * CH2 * 64;   // items of a pass (phase 1 | phase 2)
    constexpr int P1_HB = K::NST1 * 64 * HB1, PASS_HB = P1_HB + K::NST2 * 64 * 8;                                 // header bytes of a pass
    const int li = blockIdx.y;
    const float* __restrict__ W1 = jobs.W1[li];
    const float* __restrict__ b1 = jobs.b1[li];
    const float* __restrict__ W2 = jobs.W2[li];
    const float* __restrict__ b2 = jobs.b2[li];
    unsigned* __restrict__ img = jobs.img[li];
    const float ws = w_act_scale(jobs.alpha[li], jobs.act);
    const size_t items = K::STREAM / 16;
    for (size_t hb = (size_t)blockIdx.x * blockDim.x + threadIdx.x; hb >= items && hb < items + K::HDR_BYTES; hb += (size_t)gridDim.x * blockDim.x) {
        const int pass = (int)(hb - items) / PASS_HB, q = (int)(hb - items) % PASS_HB;
        unsigned char* out8 = reinterpret_cast<unsigned char*>(img) + (size_t)pass * PASS_ITEMS * 16;
        if (q < P1_HB) {
            const int stage = q / (64 * HB1), l = (q / HB1) & 63, b = q % HB1, t = pass * NTP + stage / SPT, sq = stage % SPT;
            const int gi = b >> 2, term = (b >> 1) & 1, j = b & 1;
            out8[((size_t)stage * CH1 * 64 + l) * 16 + b] = (unsigned char)w_lo1(W1, C, ws, t, GS * sq + gi, j, term, l & 31, l >> 5).sb;
        } else {
            const int q2 = q - P1_HB;
            const int stage = q2 / (64 * 8), l = (q2 / 8) & 63, b = q2 & 7, nb = stage >> 1, half = stage & 1;
            out8[(P1_ITEMS + (size_t)stage * CH2 * 64 + l) * 16 + b] =
                b < 2 * GS ? (unsigned char)w_lo2(W2, WD, nb, pass * NTP + NSETS * half + b, l & 31, l >> 5).sb : (unsigned char)0;
        }
    }
    for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (size_t)gridDim.x * blockDim.x) {
        u32x4 out = {0u, 0u, 0u, 0u};
        const int pass = (int)(it / PASS_ITEMS);
        const size_t ip = it % PASS_ITEMS;
        if (ip < P1_ITEMS) {
            const int stage = (int)(ip / (CH1 * 64)), ci = (int)(ip % (CH1 * 64));
            const int chunk = ci >> 6, l = ci & 63, t = pass * NTP + stage / SPT, sq = stage % SPT;
            if (chunk == 0) {   // bytes 0 .. HB1 - 1: the byte threads above; the dwords past them are zero
#pragma unroll
                for (int d = GS; d < 4; ++d) img[it * 4 + d] = 0u;
                continue;
            } else if (chunk == 1) {
                if (sq == 0 && l < 16 && b1) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(b1 + 64 * t + 4 * l);
#pragma unroll
                    for (int e = 0; e < 4; ++e) out[e] = __float_as_uint(bb[e] * ws);
                }
            } else if (chunk < 2 + 14 * GS) {
                const int gi = (chunk - 2) / 14, c = (chunk - 2) % 14, g = GS * sq + gi;
                if (c < 8) {
                    const int s = c >> 1, j = c & 1, r = l & 31, h = l >> 5;
                    const float* src = W1 + (size_t)(64 * t + 32 * j + r) * C + 64 * g + 16 * s + 8 * h;
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(src), w1 = *reinterpret_cast<const f32x4*>(src + 4);
                    f16x8 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = w0[e] * ws, b = w1[e] * ws;
                        asm volatile("" : "+v"(a), "+v"(b));
                        v[e] = (_Float16)a;
                        v[4 + e] = (_Float16)b;
                    }
                    out = __builtin_bit_cast(u32x4, v);
                } else if (c == 8 || c == 9 || c == 11 || c == 12) {
                    const int term = c >= 11, j = (c - (term ? 11 : 8));
                    const WLo o = w_lo1(W1, C, ws, t, g, j, term, l & 31, l >> 5);
                    out = u32x4{o.pk[0], o.pk[1], o.pk[2], o.pk[3]};
                } else {   // c == 10 / 13: [j][lane] 8 bytes; this item = lanes 2 q, 2 q + 1 of block j
                    const int term = c == 13, j = l >> 5, q = l & 31;
                    const WLo a = w_lo1(W1, C, ws, t, g, j, term, (2 * q) & 31, (2 * q) >> 5), b = w_lo1(W1, C, ws, t, g, j, term, (2 * q + 1) & 31, (2 * q + 1) >> 5);
                    out = u32x4{a.pk[4], a.pk[5], b.pk[4], b.pk[5]};
                }
            }
        } else {
            const size_t i2 = ip - P1_ITEMS;
            const int stage = (int)(i2 / (CH2 * 64)), ci = (int)(i2 % (CH2 * 64));
            const int chunk = ci >> 6, l = ci & 63, nb = stage >> 1, half = stage & 1;
            if (chunk == 0) {   // bytes 0 .. 7: the byte threads above; mlp.2's bias rides in the first pass only
                img[it * 4 + 2] = __float_as_uint(b2 && pass == 0 ? b2[32 * nb + (l & 31)] : 0.f);
                img[it * 4 + 3] = 0u;
                continue;
            } else if (chunk < 1 + 11 * GS) {
                const int pi = (chunk - 1) / 11, q = (chunk - 1) % 11;
                if (q == 10) {
                    const int tt = l >> 5, ql = l & 31, t = pass * NTP + NSETS * half + 2 * pi + tt;
                    const WLo a = w_lo2(W2, WD, nb, t, (2 * ql) & 31, (2 * ql) >> 5), b = w_lo2(W2, WD, nb, t, (2 * ql + 1) & 31, (2 * ql + 1) >> 5);
                    out = u32x4{a.pk[4], a.pk[5], b.pk[4], b.pk[5]};
                } else {
                    const int tt = q / 5, s = q % 5, t = pass * NTP + NSETS * half + 2 * pi + tt, r = l & 31, h = l >> 5;
                    if (s == 4) {
                        const WLo o = w_lo2(W2, WD, nb, t, r, h);
                        out = u32x4{o.pk[0], o.pk[1], o.pk[2], o.pk[3]};
                    } else {
                        // element e of k-step s: k = kmap(h, 8 s + e) = 32 (s >> 1) + 16 (s & 1) + 8 (e >> 2) + 4 h + (e & 3)
                        const float* src = W2 + (size_t)(32 * nb + r) * WD + 64 * t + 32 * (s >> 1) + 16 * (s & 1) + 4 * h;
                        const f32x4 w0 = *reinterpret_cast<const f32x4*>(src), w1 = *reinterpret_cast<const f32x4*>(src + 8);
                        f16x8 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[e] = (_Float16)w0[e];
                            v[4 + e] = (_Float16)w1[e];
                        }
                        out = __builtin_bit_cast(u32x4, v);
                    }
                }
            }
        }
        *reinterpret_cast<u32x4*>(img + it * 4) = out;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct WBuf {       // the operands of one set of matrix instructions: up to six 16-byte and four 8-byte fragment reads
    u32x4 q[6];
    u32x2 d[4];
};

// pieces [a, b) of the NP pieces a wave contributes to a stage that are issued with set I of its NSETS sets: spread over the first
// NSETS - 2 sets, so that the youngest piece has two sets (~400 cycles) to land before the next stage entry waits for it
template <int I, int NSETS, int NP>
struct WSpan {
    static constexpr int F = NSETS > 2 ? NSETS - 2 : 1;
    static constexpr int a = I < F ? I * NP / F : NP, b = I < F ? (I + 1) * NP / F : NP;
};
// the interleave of one set (a scheduling region): NM matrix instructions, each followed by its share of the ND fragment reads of the
// next set and of the NV LDS-DMA pieces — one instruction stream per SIMD hides nothing that is not placed between two matrix instructions
// NA > 0: the set also carries a share of the previous hidden tile's activation — NA vector-ALU and one transcendental instruction behind
// each matrix instruction (their issue cycles lie inside the 32 the matrix pipe is busy)
template <int NM, int ND, int NV, int NA = 0, int NX = 1>
__device__ __forceinline__ void w_interleave() {
    sfor<NM>([&](auto I) {
        constexpr int i = decltype(I)::value;
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        constexpr int nd = (i + 1) * ND / NM - i * ND / NM, nv = (i + 1) * NV / NM - i * NV / NM;
        if constexpr (nd > 0) __builtin_amdgcn_sched_group_barrier(0x100, nd, 0);
        if constexpr (nv > 0) __builtin_amdgcn_sched_group_barrier(0x20, nv, 0);
        if constexpr (NA > 0) {
            __builtin_amdgcn_sched_group_barrier(0x2, NA, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, NX, 0);
        }
    });
}

template <int ACT, int NG>
__global__ __launch_bounds__(256, 1) void mlp_fused_w_kernel(MlpWArgs g) {
    typedef WCfg<NG> K;
    constexpr int W_C = K::C, W_WD = K::WD, W_NG = K::NG, W_NB = K::NB, W_SLOT = K::SLOT, W_CH1 = K::CH1, W_CH2 = K::CH2;
    constexpr int W_NP1 = K::NP1, W_NP2 = K::NP2, NSETS = K::NSETS, GS = K::GS, SPT = K::SPT, NTP = K::NTP, NPASS = K::NPASS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ring = smem;
    char* const stg = ring + W_NS * W_SLOT;
    float* const colp = reinterpret_cast<float*>(stg + 4 * W_STG);   // [4 waves][2][C] column partials of a tile
    float* const pro_lds = colp;                                      // pa[0 .. C) | po[0 .. C) while the y build runs

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tilesM = g.rows >> 7, ntiles = g.B * tilesM;

    WSTAMP(0);

    // ---- the weight stream: a wave's pieces of a stage are NP consecutive chunks; the stream wraps (the next tile's first stages)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.w_img), 0, 0x7fffffff, 0x00020000);
    const unsigned voff16 = (unsigned)lane * 16u;
    unsigned soff = 0;          // stream offset of the stage being issued
    int islot = 0;              // its ring slot
    int istage = 0;             // its index in its pass's NST1 + NST2 stages: < NST1 phase 1
    int ipass = 0;
    auto issue_piece = [&](int np, int p) {
#ifndef MFW_DIAG_NODMA
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(ring + islot * W_SLOT + (wave * np + p) * 1024), 16, voff16,
                                                 soff + (unsigned)((wave * np + p) * 1024), 0, 0);
#endif
    };
    auto issue_advance = [&]() {
        soff += (istage < K::NST1 ? W_CH1 : W_CH2) * 1024u;
        istage = istage + 1;
        if (istage == K::NST1 + K::NST2) {
            istage = 0;
            if constexpr (NPASS > 1) {
                ipass = ipass + 1 == NPASS ? 0 : ipass + 1;
                if (ipass == 0) soff = 0;
            } else {
                soff = 0;
            }
        }
        islot = islot + 1 == W_NS ? 0 : islot + 1;
    };
    // pieces of the stage being issued that go behind set I of a stage of NSETS sets; TYPE1: the issued stage is a phase-1 stage
    auto issue_after = [&](auto I, auto NSETS, auto TYPE1) {
        constexpr int i = decltype(I)::value, nsets = decltype(NSETS)::value;
        constexpr int np = decltype(TYPE1)::value ? W_NP1 : W_NP2;
        typedef WSpan<i, nsets, np> S;
#pragma unroll
        for (int p = S::a; p < S::b; ++p) issue_piece(np, p);
        if constexpr (i == nsets - 1) issue_advance();
    };
#pragma unroll
    for (int s = 0; s < W_NS - 1; ++s) {
#pragma unroll
        for (int p = 0; p < W_NP1; ++p) issue_piece(W_NP1, p);
        issue_advance();
    }

    // ---- fragment reads (LDS-typed pointers throughout: a generic pointer behind an opaque asm turns its reads into flat loads, which
    // count on BOTH wait counters and drain the DMA queue)
    typedef const __attribute__((address_space(3))) char* lds_cptr;
    const lds_cptr lb16 = (lds_cptr)ring + lane * 16;
    const lds_cptr lb8 = (lds_cptr)ring + lane * 8;
    int rslot = 0;              // slot of the stage being computed
    lds_cptr sb16 = lb16, sb8 = lb8, nb16 = lb16, nb8 = lb8;   // bases of the stage being read / of the next one
    lds_cptr sb8b = lb8 + 512, nb8b = lb8 + 512;                 // + 512: the second block's 8-byte parts through their own base register (two
                                                                // reads off one base become ds_read2st64_b64 + four moves into the operand tuples)
#ifdef MFW_DIAG_NOREAD    // (diagnostic: no fragment reads — what the LDS array's load costs the loop)
    auto rd16 = [&](lds_cptr base, int chunk) { unsigned v = (unsigned)(size_t)base + chunk; asm volatile("" : "+v"(v)); return u32x4{v, v, v, v}; };
    auto rd8 = [&](lds_cptr base, int chunk, int half) { unsigned v = (unsigned)(size_t)base + chunk + half; asm volatile("" : "+v"(v)); return u32x2{v, v}; };
#else
    auto rd16 = [&](lds_cptr base, int chunk) { return *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(base + chunk * 1024); };
    auto rd8 = [&](lds_cptr base, int chunk, int half) { return *reinterpret_cast<const __attribute__((address_space(3))) u32x2*>(base + chunk * 1024 + half * 512); };
#endif
    // phase 1: set k (0 / 1) of group gi (0 .. 2) of the stage at (b16, b8) — six matrix instructions each (192 cycles: the time the next
    // set's fragment reads have to arrive): k-steps 0 .. 2 of both hidden blocks | k-step 3 and the four fp6 operands
    auto load_p1 = [&](lds_cptr b16, lds_cptr b8, lds_cptr b8b, auto GI, auto K, WBuf& f) {
        constexpr int c0 = 2 + 14 * decltype(GI)::value, k = decltype(K)::value;
        if constexpr (k == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) f.q[i] = rd16(b16, c0 + i);
        } else {
            f.q[0] = rd16(b16, c0 + 6);
            f.q[1] = rd16(b16, c0 + 7);
            f.q[2] = rd16(b16, c0 + 8);
            f.d[0] = rd8(b8, c0 + 10, 0);
            f.q[3] = rd16(b16, c0 + 9);
            f.d[1] = rd8(b8b, c0 + 10, 0);
            f.q[4] = rd16(b16, c0 + 11);
            f.d[2] = rd8(b8, c0 + 13, 0);
            f.q[5] = rd16(b16, c0 + 12);
            f.d[3] = rd8(b8b, c0 + 13, 0);
        }
    };
    // phase 2: tile tt (0 / 1) of pair pi (0 .. 2)
    auto load_p2 = [&](lds_cptr b16, lds_cptr b8, auto PI, auto TT, WBuf& f) {
        constexpr int c0 = 1 + 11 * decltype(PI)::value, tt = decltype(TT)::value;
#pragma unroll
        for (int s = 0; s < 5; ++s) f.q[s] = rd16(b16, c0 + 5 * tt + s);
        f.d[0] = rd8(b8, c0 + 10, tt);
    };
    // entering a stage: the NEXT stage has landed for every wave (its first fragments are read across the boundary), the slot of the
    // previous one may be refilled.  ALLOW: vector-memory operations younger than the awaited pieces that may stay in flight
    WACC_DECL;
    auto stage_enter = [&](auto ALLOW) {
        WACC_BEGIN();
        wait_vm<decltype(ALLOW)::value>();
        __builtin_amdgcn_s_barrier();
        WACC_END(wacc_enter);
        sb16 = lb16 + rslot * W_SLOT;
        sb8 = lb8 + rslot * W_SLOT;
        rslot = rslot + 1 == W_NS ? 0 : rslot + 1;
        nb16 = lb16 + rslot * W_SLOT;
        nb8 = lb8 + rslot * W_SLOT;
        sb8b = sb8 + 512;
        nb8b = nb8 + 512;
        asm volatile("" : "+v"(sb8b), "+v"(nb8b));
    };

    wait_vm<0>();
    wait_lgkm0();
    __builtin_amdgcn_s_barrier();          // the first two stages are in the ring, the bias tables are written
    WBuf bufA, bufB;
    load_p1(lb16, lb8, lb8 + 512, W_IC(0), W_IC(0), bufA);

    // per-lane LDS bases behind an opaque asm: left as constants (beyond the 16-bit offset field) every distinct address becomes a register
    lds_cptr probase = (lds_cptr)(reinterpret_cast<const char*>(pro_lds)) + 16 * (lane & 15);
    asm volatile("" : "+v"(probase));
    // the wave's phase-2 tile [32 rows][32 floats]: register e of an accumulator <-> row (e & 3) + 8 (e >> 2) + 4 h, column r (4-byte
    // accesses: a lane half covers one row's 128 bytes); 16-byte pieces <-> row (lane >> 3) + 8 i, columns 4 (lane & 7) ..
    typedef __attribute__((address_space(3))) char* lds_ptr;
    lds_ptr tile4 = (lds_ptr)(stg + wave * W_STG) + (4 * h) * 128 + 4 * r;
    lds_ptr tile16 = (lds_ptr)(stg + wave * W_STG) + lane * 16;
    asm volatile("" : "+v"(tile4), "+v"(tile16));
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, 0x7fffffff, 0x00020000);   // (two passes: the second adds to the first's output)
    const f32x16 z16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int opq = 0;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / tilesM, rt = tile - b * tilesM;
        const size_t row0 = (size_t)b * g.rows + (size_t)rt * 128 + wave * 32;
        WSTAMP(1);
        WACC_RESET();
        // ---- AdaGN coefficients of the sample (every wave has left the previous tile's reduction: the barrier that ended it)
        for (int i = tid; i < W_C; i += 256) {
            pro_lds[i] = g.pro_a[(size_t)b * W_C + i];
            pro_lds[W_C + i] = g.pro_o[(size_t)b * W_C + i];
        }
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();

        // ---- y = AdaGN(x) of the wave's 32 rows: fa[g] = fp16(y) as the fragments of the four k-steps of group g (element 8 s + e of lane
        // (r, h): k = 64 g + 16 s + 8 h + e), yl6[g] = fp6(2^11 (y - fp16(y)) / block scale), by[g] / yls[g] the scale bytes of the fp6 forms
        f16x32 fa[W_NG];
        u32x6 yl6[W_NG];
        float byf[NPASS > 1 ? 1 : W_NG];   // 2^(scale byte - 127) of fp6(yh): the conversions' scale operand (two passes: re-formed from byp where used)
        int byp[2] = {0, 0};        // the same bytes, four per register (the matrix instruction selects one: op_sel)
        int ylp[2] = {0, 0};        // scale bytes of yl6 x 2^-11
        {
            const float* xw = g.x + row0 * W_C;
            char* sw = stg + wave * W_STG;
            const int lrow = lane >> 4, c16 = lane & 15;
            f32x4 xs[2][8];   // (all 48 loads in flight at once measured slower: 21.5 K ticks against 14.8 K for this two-slab ring)
#pragma unroll
            for (int i = 0; i < 8; ++i) xs[0][i] = *reinterpret_cast<const f32x4*>(xw + (size_t)(4 * i + lrow) * W_C + 4 * c16);
            sfor<W_NG>([&](auto G) {
                constexpr int gg = decltype(G)::value;
                if constexpr (gg + 1 < W_NG) {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        xs[(gg + 1) & 1][i] = *reinterpret_cast<const f32x4*>(xw + (size_t)(4 * i + lrow) * W_C + 64 * (gg + 1) + 4 * c16);
                }
                const f32x4 pa4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(probase + 256 * gg);
                const f32x4 po4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(probase + 4 * W_C + 256 * gg);
                f16x4 lo16[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = 4 * i + lrow;
                    f16x4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float y = h8_clamp(__builtin_fmaf(xs[gg & 1][i][e], pa4[e], po4[e]));
                        asm volatile("" : "+v"(y));   // ONE rounded fp32 value feeds the hi rounding and the lo difference
                        hv[e] = (_Float16)y;
                        lo16[i][e] = (_Float16)((y - (float)hv[e]) * H8_AL_SCALE);
                    }
                    *reinterpret_cast<u32x2*>(sw + row * 128 + (((c16 >> 1) ^ (row & 7)) << 4) + (c16 & 1) * 8) = __builtin_bit_cast(u32x2, hv);
                }
                __builtin_amdgcn_wave_barrier();   // a wave's LDS operations execute in order
                u32x4 fr[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) fr[s] = *reinterpret_cast<const u32x4*>(sw + r * 128 + (((2 * s + h) ^ (r & 7)) << 4));
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = 4 * i + lrow;
                    *reinterpret_cast<u32x2*>(sw + row * 128 + (((c16 >> 1) ^ (row & 7)) << 4) + (c16 & 1) * 8) = __builtin_bit_cast(u32x2, lo16[i]);
                }
                __builtin_amdgcn_wave_barrier();
                u32x4 lr[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) lr[s] = *reinterpret_cast<const u32x4*>(sw + r * 128 + (((2 * s + h) ^ (r & 7)) << 4));
                __builtin_amdgcn_wave_barrier();
                f16x32 la;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const f16x8 a = __builtin_bit_cast(f16x8, fr[s]), l8 = __builtin_bit_cast(f16x8, lr[s]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        fa[gg][8 * s + e] = a[e];
                        la[8 * s + e] = l8[e];
                    }
                }
                const int bl = w_scale_byte(w_absmax32(la));
                yl6[gg] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(la, w_scale_of(bl));
                ylp[gg >> 2] |= (bl > H8_AL_EXP ? bl - H8_AL_EXP : 0) << (8 * (gg & 3));
                const int bh = w_scale_byte(w_absmax32(fa[gg]));
                if constexpr (NPASS == 1) byf[gg] = w_scale_of(bh);
                byp[gg >> 2] |= bh << (8 * (gg & 3));
                asm volatile("" : "+v"(yl6[gg]));
            });
        }
        WSTAMP(2);

        // ---- the passes over the hidden width (one; two at feature_dim 512: WCfg)
        sfor<NPASS>([&](auto PASS) {
        constexpr int pass = decltype(PASS)::value;
        // ---- phase 1.  Two accumulator pairs in turn: one takes the matrix instructions of hidden tile t (it starts as the tile's bias, read
        // from chunk 1 of the tile's first stage) while the other's — tile t - 1, complete — goes through the activation BETWEEN those matrix
        // instructions: with one wave per SIMD nothing else would fill the matrix pipe's busy cycles.
        f16x32 hf[NTP];     // tile t of the pass: element i = act(u)[point r][64 (pass NTP + t) + kmap(h, i)]
        int hsp[(NTP + 3) / 4];          // block scale bytes of their fp6 forms, four per register
#pragma unroll
        for (int i = 0; i < (NTP + 3) / 4; ++i) hsp[i] = 0;
        f32x16 au0[2], au1[2];
        float mact = 0.f;
        // registers 4 qq .. 4 qq + 3 of block j (q = 4 j + qq): columns 32 j + 8 qq + 4 h + e of the tile
        auto bias_quad = [&](lds_cptr b16, auto Q, f32x16 (&a)[2]) {
            constexpr int q = decltype(Q)::value, j = q >> 2, qq = q & 3;
            const f32x4 bs = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(b16 + (1024 + 4 * (32 * j + 8 * qq)));
#pragma unroll
            for (int e = 0; e < 4; ++e) a[j][4 * qq + e] = bs[e];
        };
        // the same four registers -> fp16 fragment elements 16 j + 4 qq .. of mlp.2's row operand (two packed conversions)
        auto act_quad = [&](auto Q, f32x16 (&a)[2], f16x32& hft) {
            constexpr int q = decltype(Q)::value, j = q >> 2, qq = q & 3;
            float y[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float uu = a[j][4 * qq + e];       // s u
                if constexpr (ACT == 1 || ACT == 2) {
                    const float E = __builtin_amdgcn_exp2f(uu * -uu);
                    y[e] = ACT == 1 ? __builtin_fmaf(E, 1.0f / 0.28f, -2.5f) : E;
                } else if constexpr (ACT == 3) {
                    y[e] = h8_clamp(fmaxf(uu, 0.f));
                } else {
                    y[e] = h8_clamp(uu);
                }
            }
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 p0 = {(_Float16)y[0], (_Float16)y[1]}, p1 = {(_Float16)y[2], (_Float16)y[3]};
            hft[16 * j + 4 * qq] = p0[0];
            hft[16 * j + 4 * qq + 1] = p0[1];
            hft[16 * j + 4 * qq + 2] = p1[0];
            hft[16 * j + 4 * qq + 3] = p1[1];
            if constexpr (ACT == 0 || ACT == 3) mact = fmaxf(fmaxf(mact, fmaxf(fabsf(y[0]), fabsf(y[1]))), fmaxf(fabsf(y[2]), fabsf(y[3])));
        };
        // block scale of a tile's fp6 form: the bounded activations have a fixed one (|h| <= 2.5: 2^-1; exp(.) <= 1: 2^-2)
        auto act_done = [&](auto T) {
            constexpr int t = decltype(T)::value;
            hsp[t >> 2] |= (ACT == 1 ? 126 : ACT == 2 ? 125 : w_scale_byte(mact)) << (8 * (t & 3));
            mact = 0.f;
        };
        // (diagnostics; the never-taken branch also keeps the register allocator from merging the tiles' live ranges: without one per
        // tile the first hidden tiles' fragments go to scratch)
        auto dbg_dump = [&](auto T, f32x16 (&a)[2]) {
            constexpr int t = decltype(T)::value;
            if (g.dbg_u) {
                float* du = g.dbg_u + (row0 + r) * W_WD + 64 * (pass * NTP + t) + 4 * h;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) *reinterpret_cast<f32x4*>(du + 32 * j + 8 * qq) = f32x4{a[j][4 * qq], a[j][4 * qq + 1], a[j][4 * qq + 2], a[j][4 * qq + 3]};
            }
        };
        {
            // tile 0's bias: its stage is the next one to be entered (landed: the previous row tile's last stage entry, or the prologue's wait)
            lds_cptr bb = nb16 + (16 * h - 16 * lane);
            sfor<8>([&](auto Q) { bias_quad(bb, Q, au0); });
        }
        sfor<NTP>([&](auto T) {
            constexpr int t = decltype(T)::value;
            f32x16 (&au)[2] = (t & 1) ? au1 : au0;      // this tile's accumulators
            f32x16 (&ap)[2] = (t & 1) ? au0 : au1;      // the previous tile's, then the next tile's bias
            asm volatile("" : "+s"(opq));
            if constexpr (t > 0) dbg_dump(W_IC(t > 0 ? t - 1 : 0), ap);
            sfor<SPT>([&](auto SQ) {
                constexpr int sq = decltype(SQ)::value;      // stage of the tile: groups GS sq ..
                stage_enter(W_IC(0));
                const u32x4 hdr = rd16(sb16, 0);
                lds_cptr nbias = nb16 + (16 * h - 16 * lane);
                sfor<NSETS>([&](auto I) {
                    constexpr int i = decltype(I)::value, gi = i >> 1, k = i & 1, gg = GS * sq + gi, i12 = NSETS * sq + i;
                    WBuf& bc = (i & 1) ? bufB : bufA;
                    WBuf& bn = (i & 1) ? bufA : bufB;
                    // the next set: of this stage, of the next stage, or (last set of phase 1) the first of phase 2
                    if constexpr (i < NSETS - 1) load_p1(sb16, sb8, sb8b, W_IC((i + 1) >> 1), W_IC((i + 1) & 1), bn);
                    else if constexpr (t == NTP - 1 && sq == SPT - 1) load_p2(nb16, nb8, W_IC(0), W_IC(0), bn);
                    else load_p1(nb16, nb8, nb8b, W_IC(0), W_IC(0), bn);
                    if constexpr (k == 0) {
                        sfor<3>([&](auto S) {
                            constexpr int s = decltype(S)::value;
#pragma unroll
                            for (int j = 0; j < 2; ++j) au[j] = W_MFMA16(__builtin_bit_cast(f16x8, bc.q[2 * s + j]), w_sub<s>(fa[gg]), au[j]);
                        });
                    } else {
#pragma unroll
                        for (int j = 0; j < 2; ++j) au[j] = W_MFMA16(__builtin_bit_cast(f16x8, bc.q[j]), w_sub<3>(fa[gg]), au[j]);
                        // yh Wl: fp6(yh / block scale), one conversion instruction (the scale behind an opaque asm keeps it inside the tile)
                        float sc;
                        if constexpr (NPASS == 1) sc = byf[gg];
                        else sc = w_scale_of((byp[gg >> 2] >> (8 * (gg & 3))) & 0xff);
                        asm volatile("" : "+v"(sc), "+s"(opq));
                        const u32x6 y6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(fa[gg], sc);
#ifndef MFW_DIAG_NOT0
                        au[0] = W_MFMA6(w_op6(bc.q[2], bc.d[0]), w_op6(y6), au[0], W_SB((int)hdr[gi], 0), W_SB(byp[gg >> 2], gg & 3));
                        au[1] = W_MFMA6(w_op6(bc.q[3], bc.d[1]), w_op6(y6), au[1], W_SB((int)hdr[gi], 1), W_SB(byp[gg >> 2], gg & 3));
#endif
#ifndef MFW_DIAG_NOT1
                        // yl W
                        au[0] = W_MFMA6(w_op6(bc.q[4], bc.d[2]), w_op6(yl6[gg]), au[0], W_SB((int)hdr[gi], 2), W_SB(ylp[gg >> 2], gg & 3));
                        au[1] = W_MFMA6(w_op6(bc.q[5], bc.d[3]), w_op6(yl6[gg]), au[1], W_SB((int)hdr[gi], 3), W_SB(ylp[gg >> 2], gg & 3));
#endif
                    }
                    // the previous tile's activation, QA quads per set over the tile's first ACT_SETS sets; then the next tile's bias into the
                    // freed registers, QB quads per set: its stage is the next one (landed since this stage's entry)
                    constexpr bool acting = t > 0 && i12 < K::ACT_SETS;
                    constexpr bool biasing = t < NTP - 1 && i12 >= K::BIAS_START;
                    if constexpr (acting) sfor<K::QA>([&](auto Q) { act_quad(W_IC((K::QA * i12 + decltype(Q)::value) & 7), ap, hf[t > 0 ? t - 1 : 0]); });
                    if constexpr (biasing) sfor<K::QB>([&](auto Q) { bias_quad(nbias, W_IC((K::QB * (i12 - K::BIAS_START) + decltype(Q)::value) & 7), ap); });
                    // the stage two ahead: phase-2 stages from the last two stages of the pass's last hidden tile on
                    constexpr bool ahead1 = !(t == NTP - 1 && sq >= SPT - 2);
                    issue_after(I, W_IC(NSETS), W_IC(ahead1 ? 1 : 0));
                    {
                        constexpr int np = ahead1 ? W_NP1 : W_NP2;
                        typedef WSpan<i, NSETS, np> SP;
                        w_interleave<6, ((i < NSETS - 1 && ((i + 1) & 1)) ? 10 : 6) + (biasing ? K::QB : 0), SP::b - SP::a, acting ? 2 * K::QA : 0, K::QA>();
                    }
                    W_SCHED();
                    if constexpr (t > 0 && i12 == K::ACT_SETS - 1) {
                        act_done(W_IC(t > 0 ? t - 1 : 0));
                        asm volatile("" : "+a"(hf[t > 0 ? t - 1 : 0]), "+v"(hsp[(t > 0 ? t - 1 : 0) >> 2]));   // parked in the accumulator file
                        W_SCHED();
                    }
                });
            });
        });
        // the last tile's activation has no matrix instructions to sit between
        {
            WACC_BEGIN();
            dbg_dump(W_IC(NTP - 1), au1);
            sfor<8>([&](auto Q) { act_quad(Q, au1, hf[NTP - 1]); });
            act_done(W_IC(NTP - 1));
            asm volatile("" : "+a"(hf[NTP - 1]), "+v"(hsp[(NTP - 1) >> 2]));
            WACC_END(wacc_act);
            W_SCHED();
        }
        WSTAMP(3);

        // ---- the fp6 forms of the hidden tiles (one conversion each, from the parked fragments).  Two passes (feature_dim 512): y stays live
        // beside them, so the forms are made where they are used instead (one conversion per hidden tile and output block: 6 registers
        // instead of 48)
        constexpr bool H6_FLY = NPASS > 1;
        u32x6 h6[H6_FLY ? 1 : NTP];
        if constexpr (!H6_FLY) {
            sfor<NTP>([&](auto T) {
                constexpr int t = decltype(T)::value;
                h6[t] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hf[t], w_scale_of((hsp[t >> 2] >> (8 * (t & 3))) & 0xff));
                asm volatile("" : "+v"(h6[t]));   // these stay in the vector file (the accumulator file holds the 192 registers of hf)
            });
        }

        // ---- phase 2
        // 16-byte pieces of the block's rows: lane l <-> row (l >> 3) + 8 i, columns 32 nb + 4 (l & 7) ..
        const unsigned xoff16 = (unsigned)(((row0 + (lane >> 3)) * W_C + 4 * (lane & 7)) * 4);
        float* xout = g.out + (row0 + (lane >> 3)) * W_C + 4 * (lane & 7);
        float* cp = colp + wave * 2 * W_C + h * W_C + r;          // lane half 0 writes the column sums, half 1 the sums of squares
        auto unit2 = [&](auto FIRST, auto LAST, int nb) {
            constexpr bool first = decltype(FIRST)::value != 0, last = decltype(LAST)::value != 0;
            f32x16 acc;
            float bias = 0.f;
            // the loop's big invariants keep their register files (left alone the allocator rotates the fp6 forms through the accumulator
            // file: 12 moves per hidden tile and output block)
            sfor<NTP>([&](auto T) {
                constexpr int t = decltype(T)::value;
                f16x32& a = hf[t];
                if constexpr (H6_FLY) {
                    asm volatile("" : "+a"(a));
                } else {
                    u32x6& c = h6[t];
                    asm volatile("" : "+a"(a), "+v"(c));
                }
            });
            sfor<2>([&](auto HALF) {
                constexpr int half = decltype(HALF)::value;
                // pieces awaited: the next stage's.  Younger than them: the previous block's 4 stores (first stage of a block but the first)
                stage_enter(W_IC((half == 0 && !first) ? 4 : 0));
                const u32x4 hdr = rd16(sb16, 0);
                if constexpr (half == 0) {
                    // the block's residual rows into the wave's tile, row-major: four 1 KiB LDS-DMA pieces of 8 rows x 128 bytes (landed by the
                    // next stage entry, whose wait covers them; the previous block's last reads of the tile fed its stores, issued before these)
#ifndef MFW_DIAG_NODMA
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        // (a later pass reads what the pass before stored from this very CU: system-scope loads (sc0 sc1) never take a line the
                        // vector cache still holds from before the store)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(pass == 0 ? xrsrc : orsrc, (__attribute__((address_space(3))) void*)(stg + wave * W_STG + i * 1024), 16, xoff16,
                                                                 (unsigned)((8 * i * W_C + 32 * nb) * 4), 0, pass == 0 ? 0 : 17);
#endif
                } else {
                    bias = __uint_as_float(hdr[2]);
                }
                W_SCHED();
                sfor<NSETS>([&](auto I) {
                    constexpr int i = decltype(I)::value, pi = i >> 1, tt = i & 1, t = NSETS * half + i;
                    WBuf& bc = (i & 1) ? bufB : bufA;
                    WBuf& bn = (i & 1) ? bufA : bufB;
                    if constexpr (i < NSETS - 1) load_p2(sb16, sb8, W_IC((i + 1) >> 1), W_IC((i + 1) & 1), bn);
                    else if constexpr (half == 0 || !last) load_p2(nb16, nb8, W_IC(0), W_IC(0), bn);
                    else load_p1(nb16, nb8, nb8b, W_IC(0), W_IC(0), bn);      // the next row tile's first set
                    sfor<4>([&](auto S) {
                        constexpr int s = decltype(S)::value;
                        acc = W_MFMA16(w_sub<s>(hf[t]), __builtin_bit_cast(f16x8, bc.q[s]), (t == 0 && s == 0) ? z16 : acc);
                    });
#ifndef MFW_DIAG_NOT2
                    u32x6 c6;
                    if constexpr (H6_FLY) {
                        float sc6 = w_scale_of((hsp[t >> 2] >> (8 * (t & 3))) & 0xff);
                        asm volatile("" : "+v"(sc6));   // (keeps the conversion inside its set)
                        c6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hf[t], sc6);
                    } else {
                        c6 = h6[t];
                    }
                    acc = W_MFMA6(w_op6(c6), w_op6(bc.q[4], bc.d[0]), acc, W_SB(hsp[t >> 2], t & 3), W_SB((int)hdr[(2 * pi + tt) >> 2], (2 * pi + tt) & 3));
#endif
                    // the stage two ahead: phase-1 stages (of the next row tile) from the last output block on
                    issue_after(I, W_IC(NSETS), W_IC(last ? 1 : 0));
                    {
                        typedef WSpan<i, NSETS, last ? W_NP1 : W_NP2> SP;
                        w_interleave<5, 6, SP::b - SP::a>();
                    }
                    W_SCHED();
                });
            });
            // ---- the block's epilogue: accumulator + bias + residual through the tile (4-byte accesses in the accumulator's layout), column
            // sums there, then out in 16-byte row pieces (8 rows x 128 bytes per instruction)
            {
                WACC_BEGIN();
                float s1 = 0.f, s2 = 0.f;
                float rr[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) rr[e] = *reinterpret_cast<const __attribute__((address_space(3))) float*>(tile4 + ((e & 3) + 8 * (e >> 2)) * 128);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = (acc[e] + bias) + rr[e];
                    *reinterpret_cast<__attribute__((address_space(3))) float*>(tile4 + ((e & 3) + 8 * (e >> 2)) * 128) = v;
                    s1 += v;
                    s2 = __builtin_fmaf(v, v, s2);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 o = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(tile16 + i * 1024);
#ifdef MFW_DIAG_NOSTORE   // (diagnostic: what the output stores' write acknowledgements cost the stage waits behind them)
                    asm volatile("" ::"v"(o));
#else
                    if constexpr (pass + 1 < NPASS) *reinterpret_cast<f32x4*>(xout + (size_t)(8 * i) * W_C + 32 * nb) = o;   // read back by the next pass
                    else GECCO_NT_STORE(o, reinterpret_cast<f32x4*>(xout + (size_t)(8 * i) * W_C + 32 * nb));
#endif
                }
                if (pass + 1 == NPASS && g.stats) {
                    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s1), false, false);
                    const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
                    const float t1 = __uint_as_float(a[0]) + __uint_as_float(a[1]), t2 = __uint_as_float(c[0]) + __uint_as_float(c[1]);
                    cp[32 * nb] = h ? t2 : t1;
                }
                WACC_END(wacc_epi);
            }
            W_SCHED();
        };
        unit2(W_IC(1), W_IC(0), 0);
        for (int nb = 1; nb < W_NB - 1; ++nb) unit2(W_IC(0), W_IC(0), nb);
        unit2(W_IC(0), W_IC(1), W_NB - 1);
        });   // passes
        WSTAMP(4);
        // ---- column partials of the tile: the four waves' sums in a fixed order
        if (g.stats) {
            wait_lgkm0();
            __builtin_amdgcn_s_barrier();
            for (int i = tid; i < 2 * W_C; i += 256) {
                const float t = ((colp[i] + colp[2 * W_C + i]) + colp[4 * W_C + i]) + colp[6 * W_C + i];
                g.stats[((size_t)b * tilesM + rt) * 2 * W_C + i] = t;
            }
            wait_lgkm0();
        }
        __builtin_amdgcn_s_barrier();      // the coefficient table (= the partials' first 3 KiB) may be rewritten
        WSTAMP(5);
        WACC_STORE();
    }
    wait_vm<0>();   // the wrapped stream's last pieces still target this block's LDS
}

template <int ACT, int NG>
int mfw_launch_a(const MlpWArgs& g, hipStream_t st) {
    static bool attr = false;
    static int cus = 0, forced = 0;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_w_kernel<ACT, NG>), hipFuncAttributeMaxDynamicSharedMemorySize, WCfg<NG>::LDS);
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (cus <= 0) cus = 256;
        if (const char* e = getenv("GECCO_MLPW_CUS")) { const int v = atoi(e); if (v > 0 && v <= cus) forced = v; }
        attr = true;
    }
    // Blocks of a launch: a block of this kernel fills its CU.  When an evaluation runs as two half batches on two streams (hip_ops.py sets
    // option "mlpwshare" around it -> g.share) three quarters of the CUs per launch let the other stream's kernels find free CUs beside
    // it — C2 4.53 -> 4.42 ms per evaluation (192 of 256; 128: 4.43); alone on the device a launch takes every CU (1024 tiles: 4 rounds
    // instead of 6).  Short launches (fewer than two tiles per CU) take a CU per tile.  GECCO_MLPW_CUS overrides.
    const int ntiles = g.B * (g.rows / 128);
    const int want = forced ? forced : (g.share && ntiles >= 2 * cus ? cus * 3 / 4 : cus);
    const int grid = ntiles < want ? ntiles : want;
    hipLaunchKernelGGL((mlp_fused_w_kernel<ACT, NG>), dim3(grid), dim3(256), WCfg<NG>::LDS, st, g);
    return (int)hipGetLastError();
}

template <int NG>
int mfw_launch_n(const MlpWArgs& g, hipStream_t st) {
    switch (g.act) {
#ifndef MFW_DEV   // development builds: the GaussianActivation instantiation only (a full build takes over a minute)
        case 0: return mfw_launch_a<0, NG>(g, st);
        case 2: return mfw_launch_a<2, NG>(g, st);
        case 3: return mfw_launch_a<3, NG>(g, st);
#endif
        case 1: return mfw_launch_a<1, NG>(g, st);
        default: return -9;
    }
}

template <int NG>
int mfw_images_n(const MlpWImageJob* jobs, int n, int act, hipStream_t st) {
    constexpr int threads = (int)(WCfg<NG>::STREAM / 16) + WCfg<NG>::HDR_BYTES;
    for (int i0 = 0; i0 < n; i0 += MlpwImageJobs::MAX) {
        MlpwImageJobs j{};
        j.act = act;
        j.n = n - i0 < MlpwImageJobs::MAX ? n - i0 : MlpwImageJobs::MAX;
        for (int i = 0; i < j.n; ++i) {
            const MlpWImageJob& s = jobs[i0 + i];
            if ((act == 1 || act == 2) && !s.alpha) return -6;
            j.W1[i] = s.W0; j.b1[i] = s.b0; j.W2[i] = s.W2; j.b2[i] = s.b2; j.img[i] = static_cast<unsigned*>(s.img); j.alpha[i] = s.alpha;
        }
        hipLaunchKernelGGL(mlpw_image_kernel<NG>, dim3((threads + 255) / 256, j.n), dim3(256), 0, st, j);
        const int rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    return 0;
}

}  // namespace

// the widths built in the other translation units
int mfw_launch_ng2(const MlpWArgs& g, hipStream_t st);
int mfw_launch_ng4(const MlpWArgs& g, hipStream_t st);
int mfw_launch_ng8(const MlpWArgs& g, hipStream_t st);
int mfw_images_ng2(const MlpWImageJob* jobs, int n, int act, hipStream_t st);
int mfw_images_ng4(const MlpWImageJob* jobs, int n, int act, hipStream_t st);
int mfw_images_ng8(const MlpWImageJob* jobs, int n, int act, hipStream_t st);

#if MFW_PART == 1
int mfw_launch_ng8(const MlpWArgs& g, hipStream_t st) { return mfw_launch_n<8>(g, st); }
int mfw_images_ng8(const MlpWImageJob* jobs, int n, int act, hipStream_t st) { return mfw_images_n<8>(jobs, n, act, st); }
#elif MFW_PART == 2
int mfw_launch_ng2(const MlpWArgs& g, hipStream_t st) { return mfw_launch_n<2>(g, st); }
int mfw_launch_ng4(const MlpWArgs& g, hipStream_t st) { return mfw_launch_n<4>(g, st); }
int mfw_images_ng2(const MlpWImageJob* jobs, int n, int act, hipStream_t st) { return mfw_images_n<2>(jobs, n, act, st); }
int mfw_images_ng4(const MlpWImageJob* jobs, int n, int act, hipStream_t st) { return mfw_images_n<4>(jobs, n, act, st); }
#else
#ifdef MFW_PROBE   // tools/probe builds: this file alone (feature_dim 384); the other widths are absent
int mfw_launch_ng2(const MlpWArgs&, hipStream_t) { return -9; }
int mfw_launch_ng4(const MlpWArgs&, hipStream_t) { return -9; }
int mfw_launch_ng8(const MlpWArgs&, hipStream_t) { return -9; }
int mfw_images_ng2(const MlpWImageJob*, int, int, hipStream_t) { return -9; }
int mfw_images_ng4(const MlpWImageJob*, int, int, hipStream_t) { return -9; }
int mfw_images_ng8(const MlpWImageJob*, int, int, hipStream_t) { return -9; }
#endif

// feature_dim 128, 256, 384 or 512 (NG = 2, 4, 6, 8 groups of 64), width 2 feature_dim, whole 128-row tiles
bool mlp_fused_w_supported(int C, int Wd, int rows) {
    return (C == 128 || C == 256 || C == 384 || C == 512) && Wd == 2 * C && rows >= 128 && rows % 128 == 0;
}

size_t mlp_fused_w_image_bytes(int C, int Wd) {
    if (!mlp_fused_w_supported(C, Wd, 128)) return 0;
    return C == 512 ? WCfg<8>::STREAM : C == 384 ? WCfg<6>::STREAM : C == 256 ? WCfg<4>::STREAM : WCfg<2>::STREAM;
}

int mlp_fused_w_images_launch(const MlpWImageJob* jobs, int n, int C, int Wd, int act, hipStream_t st) {
    if (!mlp_fused_w_supported(C, Wd, 128)) return -9;
    return C == 512   ? mfw_images_ng8(jobs, n, act, st)
           : C == 384 ? mfw_images_n<6>(jobs, n, act, st)
           : C == 256 ? mfw_images_ng4(jobs, n, act, st)
                      : mfw_images_ng2(jobs, n, act, st);
}

int mlp_fused_w_image_launch(const float* W0, const float* b0, const float* W2, const float* b2, void* img, int C, int Wd, const float* alpha, int act,
                             hipStream_t st) {
    const MlpWImageJob job{W0, b0, W2, b2, img, alpha};
    return mlp_fused_w_images_launch(&job, 1, C, Wd, act, st);
}

int mlp_fused_w_launch(const MlpWArgs& g, int C, int Wd, hipStream_t st) {
    if (!mlp_fused_w_supported(C, Wd, g.rows) || !g.x || !g.out || !g.pro_a || !g.pro_o || !g.w_img) return -9;
    if ((size_t)g.B * g.rows * C * sizeof(float) >= ((size_t)1 << 31)) return -9;   // the residual rows come in through 32-bit buffer offsets
    if ((g.act == 1 || g.act == 2) && !g.alpha) return -6;
    return C == 512 ? mfw_launch_ng8(g, st) : C == 384 ? mfw_launch_n<6>(g, st) : C == 256 ? mfw_launch_ng4(g, st) : mfw_launch_ng2(g, st);
}
#endif
