// A-stationary first linear of the point MLP in "fp16 + fp8 cross terms" arithmetic (h8), gfx950:
//
//   u[b, m, n] = act( sum_k y[b, m, k] W[n, k] + bias[n] ),   y = x * pa[b] + po[b]   (AdaGN apply, never materialised)
//
// written as the TILED SPLIT IMAGE (GemmArgs::c_img layout: bf16 hi | lo planes) that mlp.2's register-fed kernel
// (gemm_x3_areg.hip) consumes.  Reference: x + mlp(mlp_norm(x)), models/set_transformer.py:164-166; models/mlp.py:5-39;
// models/activation.py:17-24; models/normalization.py:36-44.
//
// Arithmetic.  Both operands of this product have to carry more than fp16's 11 significant bits (the product feeds the
// residual stream: tools/experiments/fp16_site_sensitivity.py — one-term fp16 on either side leaves 3e-4 .. 4.5e-4 on F_x,
// two terms on both 6e-5).  With y = yh + yl (yh = fp16(y)) and W = Wh + Wl:
//
//       y W  =  yh Wh          fp16 MFMA            (v_mfma_f32_32x32x16_f16, 32 cycles per 16 k)
//            +  yh Wl          fp8 scaled MFMA      (v_mfma_scale_f32_32x32x64_f8f6f4, 64 cycles per 64 k: fp8(yh 2^-3) x fp8(2^16 Wl))
//            +  yl W           fp8 scaled MFMA      (fp8(2^11 yl) x fp8(2^5 W); the scales and their ranges: h8_scales.h)
//            +  yl Wl          dropped (2^-24)
//
// — the cross terms are 2^-12 of the product, so 4 significant bits of them suffice: 2 matrix-pipe units per product
// instead of the 3 of split-bf16, at the same accuracy (emulated 6.0e-5 vs 6.7e-5 on the C2 network).
//
// Structure.  A block (4 waves; two blocks per CU) owns 128 rows of x; every wave keeps ITS 32 rows in registers for the whole kernel — yh as
// the fp16 fragments of all K / 32 k-steps (96 VGPRs at K = 384), fp8(2^14 yl) as 48 more — built once from coalesced
// reads through a wave-private staging tile.  The waves then walk the output columns 64 at a time: every wave
// multiplies its rows with the SAME 64-column W tile (a byte of W brought into the LDS serves 128 rows; one 256-row block of
// 8 waves per CU measured the same and is not instantiated).  W streams from a pre-tiled image in consumption
// order — per (64-column tile, 64-k group) one 8 KiB stage of fp16 Wh (two 32-k sub-tiles) and one of fp8 (Wl | W) —
// through a ring of 4 - 6 stages by buffer_load ... lds, two 1 KiB pieces per wave and stage; a stage is two
// sub-steps of 128 matrix-pipe cycles per wave, the fragments of the next sub-step are read while one runs.
// The MFMAs take W as the row operand and y as the column operand: the accumulator then holds, per lane, ONE point and
// 16 output columns, so the epilogue (bias, activation, hi / lo split) needs no LDS transpose — one
// v_permlane32_swap per register pair gives every lane 8 consecutive columns = 16 bytes of the hi plane and 16 of the
// lo plane of the image, a wave-instruction writes 1 KiB of consecutive bytes.
#include "gemm_dma_common.h"
#include "h8_scales.h"

#include <stdlib.h>

#include <utility>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int H_BN = 64;           // columns per W tile
inline bool act_gauss_host(int act) { return act == 1 || act == 2; }
// the training forms' activations: GaussianActivation (1 normalized / 2 raw) and ReLU (3) — common.h's expressions without the GELU
// branch (its erf expansion on 32 values per tile does not fit beside the stationary operand)
__device__ __forceinline__ float tr_act(float u, float neg_inv_2a2, int act) {
    return act == 3 ? fmaxf(u, 0.f) : gauss_act(u, neg_inv_2a2, act == 1);
}
__device__ __forceinline__ float tr_act_prime(float u, float neg_inv_2a2, float inv_a2, int kind, float& dalpha) {
    dalpha = 0.f;
    if (kind == 3) return u > 0.f ? 1.f : 0.f;
    const float E = __expf(u * u * neg_inv_2a2) * (kind == 1 ? 1.0f / 0.28f : 1.0f);
    dalpha = E * (u * u * inv_a2);
    return E * (-u * inv_a2);
}
constexpr int H_STAGE = 2048;      // floats per 8 KiB ring stage (two 4 KiB sub-tiles)
constexpr int H_STG = 1536;        // floats of a wave's staging tile: [32][64] fp16 (4 KiB) + [32][64] fp8 (2 KiB)
constexpr int H_STORES = 8;        // store instructions of a wave's epilogue per column tile
constexpr float YL_SCALE = H8_AL_SCALE, W8_SCALE = H8_W8_SCALE, WL_SCALE = H8_WL_SCALE;   // h8_scales.h

__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

// Diagnostic builds (tools/probe/h8_probe.hip): -DH8_STAMPS per-block s_memtime stamps; -DH8_DIAG_NOMFMA / _NODMA / _NOACT /
// _NOSTORE / _NOEPI remove one ingredient each (results are then garbage; only the time is of interest)
#ifdef H8_DIAG_NOACT
#define H8_ACT_ON false
#else
#define H8_ACT_ON true
#endif
#ifdef H8_STAMPS
__device__ unsigned long long g_h8_stamps[1024 * 4];
#define HSTAMP(i)                                                                                              \
    do {                                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_h8_stamps[blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define HSTAMP(i)
#endif

#ifdef H8_DIAG_NOMFMA
__device__ __forceinline__ f32x16 h8_keep16(f16x8 a, f16x8 b, f32x16 c) {
    asm volatile("" ::"v"(a), "v"(b));
    return c;
}
__device__ __forceinline__ f32x16 h8_keep8(i32x8 a, i32x8 b, f32x16 c) {
    asm volatile("" ::"v"(a), "v"(b));
    return c;
}
#define H8_MFMA16(a, b, c) h8_keep16(a, b, c)
#define H8_MFMA8(a, b, c, sa, sb) h8_keep8(a, b, c)
#else
#define H8_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#ifdef H8_DIAG_CROSSFMT   // timing only: the cross terms' instruction with both operands read as fp6 (2) / fp4 (4) — garbage values
#define H8_MFMA8(a, b, c, sa, sb) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, H8_DIAG_CROSSFMT, H8_DIAG_CROSSFMT, 0, sa, 0, sb)
#else
#define H8_MFMA8(a, b, c, sa, sb) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb)
#endif
#endif

__device__ __forceinline__ float clamp448(float v) { return __builtin_fminf(__builtin_fmaxf(v, -448.f), 448.f); }

// ---- "h6": the two cross terms in fp6 (e2m3) with one E8M0 scale per lane and 64-k group (OCP-MX block scaling: the lane's 32 values
// ARE a scale block of v_mfma_scale_f32_32x32x64_f8f6f4) instead of fp8 with fixed power-of-two scales.  Both operands 6 bits wide:
// the instruction takes half the matrix-pipe cycles of the fp8 form (tools/probe/fp6_rate.hip: 1.65x the rate on random operands), and
// a block's scale follows its own maximum, so there is no range to leave (h8_scales.h's clamps become irrelevant for these terms).
// e2m3: 3 mantissa bits like e4m3, normal range [1, 7.5], subnormal step 0.125.  Semantics pinned by tools/probe/fp6_mfma_probe.hip:
// v_cvt_scalef32_pk32_fp6_f16 divides by its scale operand and packs element i at bit 6 i — the layout the MFMA reads; scale operand =
// E8M0 byte (2^(byte - 127)) of the selected byte of a VGPR, per lane.
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned int u32x6 __attribute__((ext_vector_type(6)));
// E8M0 byte of the block scale for a block whose largest magnitude is m: m / 2^(byte - 127) in (3.75, 7.5]
__device__ __forceinline__ int h6_scale_byte(float m) {
    const int e = (int)(__float_as_uint(m * (16.0f / 15.0f)) >> 23) - 2;
    return m > 0.f ? (e < 1 ? 1 : e) : 127;
}
__device__ __forceinline__ float h6_scale_of(int byte) { return __uint_as_float((unsigned)byte << 23); }
// largest magnitude of four fp16 fragments (32 values)
__device__ __forceinline__ float h6_absmax32(f16x8 a, f16x8 b, f16x8 c, f16x8 d) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    auto ab = [](f16x8 v) {
        u32x4 u = __builtin_bit_cast(u32x4, v);
        u &= 0x7fff7fffu;
        return __builtin_bit_cast(f16x8, u);
    };
    f16x8 m = __builtin_elementwise_max(__builtin_elementwise_max(ab(a), ab(b)), __builtin_elementwise_max(ab(c), ab(d)));
    const h2 m2 = __builtin_elementwise_max(__builtin_elementwise_max(h2{m[0], m[1]}, h2{m[2], m[3]}),
                                            __builtin_elementwise_max(h2{m[4], m[5]}, h2{m[6], m[7]}));
    return fmaxf((float)m2[0], (float)m2[1]);
}
__device__ __forceinline__ u32x6 h6_pack32(f16x8 a, f16x8 b, f16x8 c, f16x8 d, float scale) {
    const f16x32 v = __builtin_shufflevector(__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15),
                                             __builtin_shufflevector(c, d, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15), 0, 1, 2, 3, 4, 5,
                                             6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31);
    return __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, scale);
}
#define H6_MFMA(a, b, c, sa, sb) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, sa, 0, sb)

// four floats -> four fp8 (e4m3) bytes, k order
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    int pk = 0;
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, pk, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, pk, true);
    return (unsigned)pk;
}

// ---------------------------------------------------------------------------------------------------------------------
// W image.  Stage s = (ct * NG + g) * 2 + kind of 8 KiB each, ct = 64-column tile, g = 64-k group:
//   kind 0 (H): sub-tile a = fp16(W[n, 64 g + 0 .. 31]), sub-tile b = fp16(W[n, 64 g + 32 .. 63]); a row n (0 .. 63) is 64 bytes =
//               four 16-byte chunks, logical chunk q = k / 8 stored at physical chunk q ^ ((n >> 2) & 3)
//               (the [128][32] fp16 tile layout of gemm_f16_dma.hip, 64 rows of it);
//   kind 1 (L): sub-tile a = fp8(2^16 (W - fp16(W))), sub-tile b = fp8(2^5 W) (h8_scales.h), both [64 n][64 k] bytes: logical chunk q = 2 h + t
//               holds k = 64 g + 32 t + 16 h + 0 .. 15 — the order in which a lane half h packs its fp16 fragments of two
//               k-steps (t) into the fp8 operand; physical chunk as above.
// One thread per 16-byte chunk: 1024 chunks per (ct, g).
// PERM (unpool_outproj_h8.hip: the stationary operand comes out of an attention accumulator): the k order inside a 32-k
// sub-tile is 16 c + 8 (e >> 2) + 4 h + (e & 3) for element e of the fragment (c, lane half h) instead of 16 h + 8 c + e.
// F6 (the "h6" stream, BN = 64, not PERM): an L sub-tile row n keeps its 64 bytes, the lane half h's 32-byte slot (its chunks (h, 0), (h, 1))
// holds 24 bytes of fp6 — element 16 t + j = k 64 g + 32 t + 16 h + j of Wl (sub-tile a) or W (sub-tile b), divided by the block's scale
// — then the scale's E8M0 byte in the low byte of dword 6: the fragment read of the kernel (two 16-byte chunks) brings operand and scale.
template <int BN, bool PERM = false, bool F6 = false>
__device__ __forceinline__ void h8_image_item(const float* __restrict__ W, float* __restrict__ img, int Nout, int K, int ldw, size_t i) {
    constexpr int LB = BN == 64 ? 6 : 7;               // log2(BN); a stage is 2 sub-tiles of [BN][16 floats]
    const int NG = K / 64;
    const int pc = (int)(i & 3), n = (int)((i >> 2) & (BN - 1)), sub = (int)((i >> (2 + LB)) & 1), kind = (int)((i >> (3 + LB)) & 1);
    const size_t cg = i >> (4 + LB);
    const int g = (int)(cg % NG), ct = (int)(cg / NG);
    const int q = pc ^ ((n >> 2) & 3);
    const int nn = min(ct * BN + n, Nout - 1);
    u32x4 out;
    if (kind == 0) {
        // chunk q = 2 h + c of the sub-tile
        const float* src = W + (size_t)nn * ldw + 64 * g + 32 * sub + (PERM ? 16 * (q & 1) + 4 * (q >> 1) : 8 * q);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(src), w1 = *reinterpret_cast<const f32x4*>(src + (PERM ? 8 : 4));
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (_Float16)w0[e];
            v[4 + e] = (_Float16)w1[e];
        }
        out = __builtin_bit_cast(u32x4, v);
    } else if constexpr (F6) {
        static_assert(!PERM && BN == 64, "h6 stream: 64-column tiles in the plain k order");
        const int h = q >> 1, t = q & 1;
        if (t) return;   // the thread of chunk (h, 0) writes the lane half's whole slot
        const float* src = W + (size_t)nn * ldw + 64 * g + 16 * h;
        float v[32];
        float m = 0.f;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(src + 32 * tt + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = sub == 0 ? w[e] - (float)(_Float16)w[e] : w[e];
                    v[16 * tt + 4 * c + e] = x;
                    m = fmaxf(m, fabsf(x));
                }
            }
        const int sb = h6_scale_byte(m);
        const float inv = __uint_as_float((unsigned)(254 - sb) << 23);   // 2^(127 - sb): exact
        f16x32 vh;
#pragma unroll
        for (int e = 0; e < 32; ++e) vh[e] = (_Float16)(v[e] * inv);
        const u32x6 pk = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vh, 1.0f);
        float* base = img + ((cg * 2 + kind) * (2 * BN * 16)) + sub * (BN * 16) + n * 16;
        *reinterpret_cast<u32x4*>(base + ((2 * h) ^ ((n >> 2) & 3)) * 4) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        *reinterpret_cast<u32x4*>(base + ((2 * h + 1) ^ ((n >> 2) & 3)) * 4) = u32x4{pk[4], pk[5], (unsigned)sb, 0u};
        return;
    } else {
        const int h = q >> 1, t = q & 1;
        const float* src = W + (size_t)nn * ldw + 64 * g + 32 * t + (PERM ? 4 * h : 16 * h);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // byte 4 c + e of the chunk (h, t): k = 32 t + 16 h + 4 c + e, PERM: 32 t + 16 (c >> 1) + 8 (c & 1) + 4 h + e
            const f32x4 w = *reinterpret_cast<const f32x4*>(src + (PERM ? 16 * (c >> 1) + 8 * (c & 1) : 4 * c));
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = clamp448(sub == 0 ? (w[e] - (float)(_Float16)w[e]) * WL_SCALE : w[e] * W8_SCALE);
            out[c] = pack_fp8x4(v[0], v[1], v[2], v[3]);
        }
    }
    // stage base: ((ct * NG + g) * 2 + kind) * 2 BN * 16 floats; sub-tile: + BN * 16 floats; row n: 16 floats; chunk: 4 floats
    *reinterpret_cast<u32x4*>(img + ((cg * 2 + kind) * (2 * BN * 16)) + sub * (BN * 16) + n * 16 + pc * 4) = out;
}

// kv_proj | q_proj stream (gemm_kvq_astat_kernel).  Per 64-column tile ct: the H stages of all NG groups; a tile inside
// [lo_begin, lo_end) (the V projection) is followed by NG / 2 L stages, each [fp8(2^16 Wl) of group 2 i | of group 2 i + 1] — the
// two-term weights of the mixed mode where their rounding reaches the output, one-term elsewhere.
// Stage index of tile ct: ct * NG + (NG / 2) * clamp(ct - lo_begin, 0, lo_end - lo_begin).  1024 chunks per (ct, pair of H stages
// or L stage): item i -> (tile, stage-in-tile, sub-tile, row, physical chunk).
// TR (the training path's dX products, one-term only): W is (K, ldw) and the stream is that of W^T — eight strided reads per chunk
// Head-aligned column order at head dim 48 (GemmArgs::kvq_perm): column n (0 .. 63) of tile t (0 .. 5) of a 384-column segment is the
// segment's logical column 48 t + n (head t) for n < 48, else the (t % 3)-th third of head 6 + t / 3.
__host__ __device__ __forceinline__ int kvq_perm48_col(int ct, int n) {
    const int seg = ct / 6, t = ct - 6 * seg;
    return 384 * seg + (n < 48 ? 48 * t + n : 48 * (6 + t / 3) + 16 * (t % 3) + (n - 48));
}

template <bool TR = false, bool P48 = false>
__device__ __forceinline__ void kvq_image_item(const float* __restrict__ W, float* __restrict__ img, int Nout, int K, int ldw, int lo_begin,
                                               int lo_end, size_t i) {
    const int NG = K / 64;
    const int pc = (int)(i & 3), n = (int)((i >> 2) & 63), sub = (int)((i >> 8) & 1);
    const size_t st = i >> 9;                          // stage index in the stream
    // invert the stage index: tiles before lo_begin have NG stages, inside 3 NG / 2, after NG
    const int per_lo = NG + NG / 2;
    const size_t s_lo0 = (size_t)lo_begin * NG, s_lo1 = s_lo0 + (size_t)(lo_end - lo_begin) * per_lo;
    int ct, kt;
    if (st < s_lo0) { ct = (int)(st / NG); kt = (int)(st % NG); }
    else if (st < s_lo1) { ct = lo_begin + (int)((st - s_lo0) / per_lo); kt = (int)((st - s_lo0) % per_lo); }
    else { ct = lo_end + (int)((st - s_lo1) / NG); kt = (int)((st - s_lo1) % NG); }
    const int q = pc ^ ((n >> 2) & 3);
    const int nn = P48 ? kvq_perm48_col(ct, n) : min(ct * H_BN + n, Nout - 1);
    u32x4 out;
    if (kt < NG) {
        f32x4 w0, w1;
        if (TR) {
            const float* src = W + (size_t)(64 * kt + 32 * sub + 8 * q) * ldw + nn;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w0[e] = src[(size_t)e * ldw];
                w1[e] = src[(size_t)(4 + e) * ldw];
            }
        } else {
            const float* src = W + (size_t)nn * ldw + 64 * kt + 32 * sub + 8 * q;
            w0 = *reinterpret_cast<const f32x4*>(src);
            w1 = *reinterpret_cast<const f32x4*>(src + 4);
        }
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (_Float16)w0[e];
            v[4 + e] = (_Float16)w1[e];
        }
        out = __builtin_bit_cast(u32x4, v);
    } else {
        const int g = 2 * (kt - NG) + sub, h = q >> 1, t = q & 1;
        const float* src = W + (size_t)nn * ldw + 64 * g + 32 * t + 16 * h;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(src + 4 * c);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = clamp448((w[e] - (float)(_Float16)w[e]) * WL_SCALE);
            out[c] = pack_fp8x4(v[0], v[1], v[2], v[3]);
        }
    }
    *reinterpret_cast<u32x4*>(img + st * H_STAGE + sub * 1024 + n * 16 + pc * 4) = out;
}

// SplitJob::pad_ = 0: the h8 stream of mlp.0 (64-column tiles), 32: its h6 form (fp6 cross terms with block scales); 2: the same in 128-column tiles (gemm_h8_areg.hip); 16: 64-column
// tiles in the attention accumulator's k order (unpool_outproj_h8.hip);
// pad_ = 1 | lo_begin << 8 | lo_end << 20 (64-column tiles): the kv | q stream; | 4: of W^T, from W (K, ldw) (one-term)
__global__ void h8_image_multi_kernel(SplitJobs jobs) {
    const SplitJob j = jobs.job[blockIdx.y];
    if (j.pad_ & 1) {
        const int lb = (j.pad_ >> 8) & 0xFFF, le = (j.pad_ >> 20) & 0xFFF, NG = j.K / 64;
        const size_t total = ((size_t)(j.Nout / H_BN) * NG + (size_t)(le - lb) * (NG / 2)) * 512;
        if (j.pad_ & 4) {   // the stream of W^T from W (K, ldw): one-term (lb == le)
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
                kvq_image_item<true>(j.W, j.img, j.Nout, j.K, j.ldw, lb, lb, i);
            return;
        }
        if (j.pad_ & 64) {   // head-aligned column order (head dim 48; Nout a multiple of 384)
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
                kvq_image_item<false, true>(j.W, j.img, j.Nout, j.K, j.ldw, lb, le, i);
            return;
        }
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
            kvq_image_item(j.W, j.img, j.Nout, j.K, j.ldw, lb, le, i);
        return;
    }
    if (j.pad_ & 16) {   // 64-column tiles in the attention accumulator's k order (unpool_outproj_h8.hip)
        const size_t total = (size_t)(j.Nout / H_BN) * (j.K / 64) * 1024;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
            h8_image_item<64, true>(j.W, j.img, j.Nout, j.K, j.ldw, i);
        return;
    }
    if (j.pad_ & 2) {   // 128-column tiles (gemm_h8_areg.hip); Nout padded up to whole tiles (the pad rows repeat the last row)
        const size_t total = (size_t)((j.Nout + 127) / 128) * (j.K / 64) * 2048;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
            h8_image_item<128>(j.W, j.img, j.Nout, j.K, j.ldw, i);
        return;
    }
    const size_t total = (size_t)(j.Nout / H_BN) * (j.K / 64) * 1024;
    if (j.pad_ & 32) {   // the "h6" stream: fp6 cross-term operands with block scales
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
            h8_image_item<64, false, true>(j.W, j.img, j.Nout, j.K, j.ldw, i);
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        h8_image_item<64>(j.W, j.img, j.Nout, j.K, j.ldw, i);
}

// ---------------------------------------------------------------------------------------------------------------------
// NG = K / 64; NW = waves per block (rows = 32 NW); NS ring slots with (2 NG) % NS == 0, so the slot of a stage is its position
// in the column tile mod NS — static.  ACT: the epilogue's activation is a template parameter (a runtime code costs a scalar
// branch per VALUE here: the compiler does not hoist it out of the unrolled epilogue).
// IMG2: the output is the h8 activation image (GemmArgs::c_img == 2, gemm_h8_areg.hip: fp16 hi + fp8 lo, 3 bytes per element)
// instead of the tiled split image (bf16 hi | lo planes, 4 bytes) of gemm_x3_areg.hip.
// OUT (the training path's FORWARD in the split-bf16 training arithmetic — h8 is as accurate, two matrix units instead of three, and
// A-stationary; fp32 tensors, row-major): 1  C (| C2) = y W^T + bias — AdaGN(x) -> K | V, q; 2  pre_out = u = y W^T + bias and
// C = act(u), both fp32 — the first linear of an MLP with the pre-activation its backward needs (models/mlp.py:5-39).  A lane holds 4
// consecutive columns of one row per accumulator quad: plain 16-byte stores.  (The backward products keep split-bf16: unscaled
// gradients do not fit the fp16 / fp8 operands.)
// F6: the "h6" arithmetic — the two cross terms as fp6 x fp6 with per-lane block scales (above); the W stream is the h6 form of the image.
// K = 512 (NG = 8): the stationary operand alone is 176 - 192 registers — one block of 4 waves per CU, a wave per SIMD with its whole
// register file (the w2 kernel's regime, mlp_fused_w.hip) instead of two blocks of 256-register waves
template <int NG, int NW, int NS, int ACT, bool IMG2, int OUT = 0, bool F6 = false>
__global__ __launch_bounds__(64 * NW, NG > 6 ? 1 : 2) void gemm_h8_astat_kernel(GemmArgs g) {
    constexpr int NKT = 2 * NG, K = 64 * NG, NT = 64 * NW, ROWS = 32 * NW, PW = 8 / NW;
    constexpr int STORES = OUT == 1 ? 8 : OUT == 2 ? 16 : (IMG2 ? 6 : H_STORES);
    static_assert(NS >= 4 && NKT % NS == 0 && (NW == 4 || NW == 8), "static slots; lookahead NS - 1 >= 3 stages");
    static_assert(NS * H_STAGE * 4 <= 65536 || NS % 2 == 0, "ring addressed from two bases");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ring = smem;                                // [NS][H_STAGE]
    float* stg = ring + NS * H_STAGE;                  // [NW][H_STG] wave-private staging of the A build
    float* bias_lds = stg + NW * H_STG;                // [Nout]
    float* pro_lds = bias_lds + g.Nout;                // pa[0 .. K) | po[0 .. K)

    const int tilesM = g.rows / ROWS, tilesN = g.Nout / H_BN;
    const int bid = g.h8_rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int b = bid / tilesM, rt = bid % tilesM, m0 = rt * ROWS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // rows m0 + 32 wave .. + 31
    const int r = lane & 31, h = lane >> 5;

    HSTAMP(0);
    if constexpr (OUT != 0) {
        for (int n = tid; n < g.Nout; n += NT) {
            const bool seg2 = g.C2 != nullptr && n >= g.n_split;
            const float* bp = seg2 ? g.bias2 : g.bias;
            bias_lds[n] = bp ? bp[seg2 ? n - g.n_split : n] : 0.f;
        }
    } else {
        for (int n = tid; n < g.Nout; n += NT) bias_lds[n] = g.bias ? g.bias[n] : 0.f;
    }
    {
        const bool has_pro = g.pro_a != nullptr;
        const float* pa = has_pro ? g.pro_a + (size_t)b * K : nullptr;
        const float* po = has_pro ? g.pro_o + (size_t)b * K : nullptr;
        for (int i = tid; i < K; i += NT) {
            pro_lds[i] = has_pro ? pa[i] : 1.f;
            pro_lds[K + i] = has_pro ? po[i] : 0.f;
        }
    }
    // two blocks per CU: the second of a pair starts late, so that one's epilogue (vector work) meets the other's matrix work
    if (g.h8_stagger > 0 && (((blockIdx.x >> 3) / g.h8_pair) & 1)) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)g.h8_stagger) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();   // before the first DMA: a block barrier drains the vector-memory queue

    // ---- W stream: PW 1 KiB pieces per wave and stage; the image is consumed front to back.  Past its end the last stage is
    // fetched again (into slots nobody reads any more): every step issues, so every wait below is the same count
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.w_img), 0, 0x7fffffff, 0x00020000);
    const unsigned voff = (unsigned)(wave * PW * 256 + lane * 4) * 4u;
    const unsigned soff_last = (unsigned)(tilesN * NKT - 1) * (H_STAGE * 4u);
    unsigned soff = 0;
    auto issue = [&](int slot) {
#ifndef H8_DIAG_NODMA
#pragma unroll
        for (int p = 0; p < PW; ++p)
            dma16_buf(wrsrc, voff + p * 1024u, soff, ring + slot * H_STAGE + (wave * PW + p) * 256);
#endif
        soff = soff < soff_last ? soff + H_STAGE * 4u : soff_last;
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p) issue(p);

    // ---- the A operand, wave-private: yh = fp16(x pa + po) as the fragments fa[kt][c] (lane (r, h): row r, k = 32 kt + 16 h +
    // 8 c .. + 7) and fp8(2^14 (y - yh)) as alo[g] (byte 16 t + e of the lane's 32: k = 64 g + 32 t + 16 h + e).  Per 64-k slab:
    // eight coalesced 16-byte loads per lane (4 rows x 256 bytes per wave-instruction), affine, rounding, into the staging
    // tile (16-byte chunks XOR-swizzled by row), from where the lane takes its fragments.
    f16x8 fa[2 * NG][2];
    i32x8 alo[NG];
    {
        const float* xw = g.A + ((size_t)b * g.rows + m0 + wave * 32) * g.lda;
        char* sw = reinterpret_cast<char*>(stg + wave * H_STG);
        const int lrow = lane >> 4, c16 = lane & 15;
        f32x4 xs[2][8];
        f16x4 lo16[F6 ? 8 : 1];
#pragma unroll
        for (int i = 0; i < 8; ++i) xs[0][i] = *reinterpret_cast<const f32x4*>(xw + (size_t)(4 * i + lrow) * g.lda + 4 * c16);
        static_for(std::make_integer_sequence<int, NG>{}, [&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (s + 1 < NG) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    xs[(s + 1) & 1][i] = *reinterpret_cast<const f32x4*>(xw + (size_t)(4 * i + lrow) * g.lda + 64 * (s + 1) + 4 * c16);
            }
            const f32x4 pa4 = *reinterpret_cast<const f32x4*>(pro_lds + 64 * s + 4 * c16);
            const f32x4 po4 = *reinterpret_cast<const f32x4*>(pro_lds + K + 64 * s + 4 * c16);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 4 * i + lrow;
                f16x4 hv;
                float lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = h8_clamp(__builtin_fmaf(xs[s & 1][i][e], pa4[e], po4[e]));
                    asm volatile("" : "+v"(y));   // one rounded fp32 value for the hi rounding and the lo difference (see the epilogue)
                    hv[e] = (_Float16)y;
                    lo[e] = F6 ? (y - (float)hv[e]) * YL_SCALE : clamp448((y - (float)hv[e]) * YL_SCALE);
                }
                *reinterpret_cast<u32x2*>(sw + row * 128 + (((c16 >> 1) ^ (row & 7)) << 4) + (c16 & 1) * 8) = __builtin_bit_cast(u32x2, hv);
                if constexpr (F6) {   // 2^11 (y - yh) as fp16 (exact to 2^-11 of itself), through the same tile once the hi fragments are out
#pragma unroll
                    for (int e = 0; e < 4; ++e) lo16[i][e] = (_Float16)lo[e];
                } else {
                    *reinterpret_cast<unsigned*>(sw + 4096 + row * 64 + (((c16 >> 2) ^ ((row >> 1) & 3)) << 4) + (c16 & 3) * 4) =
                        pack_fp8x4(lo[0], lo[1], lo[2], lo[3]);
                }
            }
            __builtin_amdgcn_wave_barrier();   // a wave's LDS operations execute in order: its reads below see these writes
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int cq = 4 * t + 2 * h + c;
                    fa[2 * s + t][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(sw + r * 128 + ((cq ^ (r & 7)) << 4)));
                }
                if constexpr (!F6) {
                    const int nc = 2 * t + h;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(sw + 4096 + r * 64 + ((nc ^ ((r >> 1) & 3)) << 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) alo[s][4 * t + e] = (int)v[e];
                }
            }
            __builtin_amdgcn_wave_barrier();
            if constexpr (F6) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = 4 * i + lrow;
                    *reinterpret_cast<u32x2*>(sw + row * 128 + (((c16 >> 1) ^ (row & 7)) << 4) + (c16 & 1) * 8) = __builtin_bit_cast(u32x2, lo16[i]);
                }
                __builtin_amdgcn_wave_barrier();
                f16x8 la[2][2];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const int cq = 4 * t + 2 * h + c;
                        la[t][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(sw + r * 128 + ((cq ^ (r & 7)) << 4)));
                    }
                __builtin_amdgcn_wave_barrier();
                // the lane's 32 values of this group: element 16 t + 8 c + e = k 64 s + 32 t + 16 h + 8 c + e (the W stream's order)
                const int bl = h6_scale_byte(h6_absmax32(la[0][0], la[0][1], la[1][0], la[1][1]));
                const int bh = h6_scale_byte(h6_absmax32(fa[2 * s][0], fa[2 * s][1], fa[2 * s + 1][0], fa[2 * s + 1][1]));
                const u32x6 pk = h6_pack32(la[0][0], la[0][1], la[1][0], la[1][1], h6_scale_of(bl));
#pragma unroll
                for (int e = 0; e < 6; ++e) alo[s][e] = (int)pk[e];
                alo[s][6] = bl > H8_AL_EXP ? bl - H8_AL_EXP : 0;   // the MFMA's scale byte of the lo term: the block scale x 2^-11
                alo[s][7] = bh;                                    // the hi term's block scale (its fp6 form is made per column tile)
            }
        });
    }

    // ---- per-lane addressing of the W fragments and of the output image.  ds_read offsets are 16-bit: a ring of more than
    // 64 KiB is addressed from two bases, slots [0, NS / 2) and [NS / 2, NS)
    constexpr bool TWO = NS * H_STAGE * 4 > 65536;
    constexpr int HALF = TWO ? NS / 2 : NS;
    const float* bptr[2][2][2];   // [ring half][j][c]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = j * 32 + r;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bptr[0][j][c] = ring + n * 16 + (((2 * h + c) ^ ((n >> 2) & 3)) << 2);
            bptr[1][j][c] = bptr[0][j][c] + HALF * H_STAGE;
        }
    }
    i32x8 fbA[2], fbB[2];
    // fragments of sub-tile `sub` (0 / 1) of ring slot `slot` (compile-time): base register + immediate offset
    auto load_f = [&](auto SLOT, auto SUB, i32x8(&f)[2]) {
        constexpr int slot = decltype(SLOT)::value, sub = decltype(SUB)::value;
        constexpr int hf = slot >= HALF ? 1 : 0;
        constexpr int off = (slot - hf * HALF) * H_STAGE + sub * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(bptr[hf][j][c] + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) f[j][4 * c + e] = (int)v[e];
            }
    };
    const int T128 = g.rows >> 7;
    const int mrow = m0 + wave * 32 + r, ml = mrow & 127;
    unsigned short* lane_dst = reinterpret_cast<unsigned short*>(g.C) +
                               ((size_t)b * T128 + (mrow >> 7)) * (size_t)(g.Nout >> 4) * 4096 + ml * 16 + ((h ^ ((ml >> 3) & 1)) << 3);

    // exp(-u^2 / (2 a^2)) = exp2(u^2 c2): log2(e) folded into the constant (one multiply less per value than gauss_act)
    const float c2 = (ACT == 1 || ACT == 2) ? -1.4426950408889634f / (2.0f * g.alpha[0] * g.alpha[0]) : 0.f;
    f32x16 acc[2];

    // ---- epilogue of one 64-column tile, from registers.  acc[j][4 q + e] = u[row r][n0 + 32 j + 8 q + 4 h + e].
    // h8 activation image: this tile is 64-k group ct of the consumer; block (sample, 128-row tile, ct) of 6144 floats
    float* img2_base = g.C + (((size_t)b * T128 + (mrow >> 7)) * (size_t)(g.Nout >> 6)) * 6144 + ((mrow & 127) >> 5) * 1024;
    auto epilogue = [&](int ct) {
        const int n0 = ct * H_BN;
        if constexpr (OUT != 0) {
            const bool s2 = g.C2 != nullptr && n0 >= g.n_split;
            float* Cf = s2 ? g.C2 : g.C;
            const int ldcf = s2 ? g.ldc2 : g.ldc;
            float* dst = Cf + ((size_t)b * g.rows + mrow) * ldcf + (s2 ? n0 - g.n_split : n0) + 4 * h;
            float* dpre = OUT == 2 ? g.pre_out + ((size_t)b * g.rows + mrow) * g.Nout + n0 + 4 * h : nullptr;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(bias_lds + n0 + 32 * j + 8 * q + 4 * h);
                    f32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = acc[j][4 * q + e] + bs[e];
                    if constexpr (OUT == 2) {
                        *reinterpret_cast<f32x4*>(dpre + 32 * j + 8 * q) = w;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (ACT == 3) w[e] = fmaxf(w[e], 0.f);
                            if (ACT == 1 || ACT == 2) {
                                const float y = __builtin_amdgcn_exp2f(w[e] * w[e] * c2);
                                w[e] = ACT == 1 ? (y - 0.7f) * (1.0f / 0.28f) : y;
                            }
                        }
                    }
                    *reinterpret_cast<f32x4*>(dst + 32 * j + 8 * q) = w;
                }
            return;
        }
        if constexpr (IMG2) {
            float* blk = img2_base + (size_t)ct * 6144;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 bs[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) bs[q] = *reinterpret_cast<const f32x4*>(bias_lds + n0 + 32 * j + 8 * q + 4 * h);
                unsigned hq[4][2], lq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4], lo[4];
                    f16x2 a, c;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[j][4 * q + e] + bs[q][e];
                        if (H8_ACT_ON && ACT == 3) v[e] = fmaxf(v[e], 0.f);
                        if (H8_ACT_ON && (ACT == 1 || ACT == 2)) {
                            const float y = __builtin_amdgcn_exp2f(v[e] * v[e] * c2);
                            v[e] = ACT == 1 ? (y - 0.7f) * (1.0f / 0.28f) : y;
                        }
                        if (ACT == 0 || ACT == 3) v[e] = h8_clamp(v[e]);   // unbounded hidden layers (identity, ReLU): h8_scales.h
                        // ONE fp32 value feeds both the hi rounding and the lo difference: left to itself the compiler forms the
                        // lo path from the unrounded product (v_fma_mixlo_f16 / v_fma_mix_f32) and the stored hi from the rounded
                        // one — near a tie the two hi differ by an fp16 ulp and hi + lo is off by that ulp
                        asm volatile("" : "+v"(v[e]));
                    }
                    a[0] = (_Float16)v[0];
                    a[1] = (_Float16)v[1];
                    c[0] = (_Float16)v[2];
                    c[1] = (_Float16)v[3];
                    lo[0] = clamp448((v[0] - (float)a[0]) * YL_SCALE);
                    lo[1] = clamp448((v[1] - (float)a[1]) * YL_SCALE);
                    lo[2] = clamp448((v[2] - (float)c[0]) * YL_SCALE);
                    lo[3] = clamp448((v[3] - (float)c[1]) * YL_SCALE);
                    hq[q][0] = __builtin_bit_cast(unsigned, a);
                    hq[q][1] = __builtin_bit_cast(unsigned, c);
                    lq[q] = pack_fp8x4(lo[0], lo[1], lo[2], lo[3]);
                }
                // hi: quads (2 p, 2 p + 1) -> the lane half h holds columns 16 p + 8 h + 0 .. 7 of block j = fragment (sub j, c h) of
                // image lane (r, p)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    u32x4 H;
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const auto sh = __builtin_amdgcn_permlane32_swap(hq[2 * p][d], hq[2 * p + 1][d], false, false);
                        H[d] = sh[0];
                        H[2 + d] = sh[1];
                    }
#ifdef H8_DIAG_NOSTORE
                    asm volatile("" ::"v"(H));
#else
                    GECCO_NT_STORE(H, reinterpret_cast<u32x4*>(blk + (2 * j + h) * 256 + (32 * p + r) * 4));
#endif
                }
                // lo: quads (0, 2) and (1, 3) swapped -> the lane half h holds the 16 fp8 of columns 16 h + 0 .. 15 of block j =
                // half t = j of image lane (r, h)
                const auto s02 = __builtin_amdgcn_permlane32_swap(lq[0], lq[2], false, false);
                const auto s13 = __builtin_amdgcn_permlane32_swap(lq[1], lq[3], false, false);
                const u32x4 Lo = {s02[0], s02[1], s13[0], s13[1]};
#ifdef H8_DIAG_NOSTORE
                asm volatile("" ::"v"(Lo));
#else
                GECCO_NT_STORE(Lo, reinterpret_cast<u32x4*>(blk + 4096 - ((mrow & 127) >> 5) * 512 + j * 256 + lane * 4));
#endif
            }
            return;
        }
        unsigned short* tdst = lane_dst + (size_t)(n0 >> 4) * 4096;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 bs[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bs[q] = *reinterpret_cast<const f32x4*>(bias_lds + n0 + 32 * j + 8 * q + 4 * h);
            unsigned hq[4][2], lq[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
                unsigned u[4];
                float lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[j][4 * q + e] + bs[q][e];
                    if (H8_ACT_ON && ACT == 3) v[e] = fmaxf(v[e], 0.f);
                    if (H8_ACT_ON && (ACT == 1 || ACT == 2)) {
                        const float y = __builtin_amdgcn_exp2f(v[e] * v[e] * c2);
                        v[e] = ACT == 1 ? (y - 0.7f) * (1.0f / 0.28f) : y;
                    }
                    u[e] = __float_as_uint(v[e]);
                    lo[e] = v[e] - __uint_as_float(u[e] & 0xFFFF0000u);
                }
                hq[q][0] = __builtin_amdgcn_perm(u[1], u[0], 0x07060302u);
                hq[q][1] = __builtin_amdgcn_perm(u[3], u[2], 0x07060302u);
                bf16x2 l0, l1;
                l0[0] = (__bf16)lo[0];
                l0[1] = (__bf16)lo[1];
                l1[0] = (__bf16)lo[2];
                l1[1] = (__bf16)lo[3];
                lq[q][0] = __builtin_bit_cast(unsigned, l0);
                lq[q][1] = __builtin_bit_cast(unsigned, l1);
            }
            // quads (2 p, 2 p + 1): after the swap the lower lane half holds columns 16 p + 0 .. 7 of its row, the upper half
            // 16 p + 8 .. 15 — chunk h of k-step (n0 + 32 j) / 16 + p of the image
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                u32x4 H, Lo;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto sh = __builtin_amdgcn_permlane32_swap(hq[2 * p][d], hq[2 * p + 1][d], false, false);
                    const auto sl = __builtin_amdgcn_permlane32_swap(lq[2 * p][d], lq[2 * p + 1][d], false, false);
                    H[d] = sh[0];
                    H[2 + d] = sh[1];
                    Lo[d] = sl[0];
                    Lo[2 + d] = sl[1];
                }
                unsigned short* dst = tdst + (size_t)(2 * j + p) * 4096;
#ifdef H8_DIAG_NOSTORE
                asm volatile("" ::"v"(H), "v"(Lo), "v"(dst));
#else
                GECCO_NT_STORE(H, reinterpret_cast<u32x4*>(dst));
                GECCO_NT_STORE(Lo, reinterpret_cast<u32x4*>(dst + 2048));
#endif
            }
        }
    };

    HSTAMP(1);
    // every wave's pieces of the first NS - 1 stages (older than the x loads, long landed) are in the ring
    dma::wait_vm_lgkm0<0>();
    __builtin_amdgcn_s_barrier();
    load_f(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fbA);
    float one = H8_AH_DIV;   // the fp8 conversions' scale operand (fp8(yh / 8), h8_scales.h) behind an opaque asm: keeps them inside the column-tile loop
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int opq = 0;
    for (int ct = 0; ct < tilesN; ++ct) {
        const bool first = ct == 0;
        asm volatile("" : "+s"(one));
        asm volatile("" : "+s"(opq));
#ifdef H8_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        static_for(std::make_integer_sequence<int, NKT>{}, [&](auto KT) {
            constexpr int kt = decltype(KT)::value;
            constexpr bool lstage = (kt & 1) != 0;
            constexpr int gq = kt >> 1;
            // own pieces of stage kt + 1 landed.  Younger vector-memory operations that may stay in flight: the pieces of the
            // NS - 3 stages after it (every step issues one stage, the stream's end included) and, while the awaited pieces
            // are older than them (kt <= NS - 3), the 8 stores of the previous tile's epilogue
            constexpr int young = (NS - 3) * PW;
            constexpr bool st_young = kt <= NS - 3;
            if (st_young && !first) dma::wait_vm_lgkm0<young + STORES>();
            else dma::wait_vm_lgkm0<young>();
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(fbA[j]));
            __builtin_amdgcn_s_barrier();
            // stage kt + NS - 1 goes to the slot of stage kt - 1, whose last fragment reads (this step's sub-step a set, read
            // during the previous step) every wave has completed before the barrier
            issue((kt + NS - 1) % NS);
            // sub-step a: fragments of sub-step b are read while it runs
            load_f(std::integral_constant<int, kt % NS>{}, std::integral_constant<int, 1>{}, fbB);
            if constexpr (!lstage) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const i32x4 wc = c == 0 ? __builtin_shufflevector(fbA[j], fbA[j], 0, 1, 2, 3) : __builtin_shufflevector(fbA[j], fbA[j], 4, 5, 6, 7);
                        // the first product of a column tile starts from zero: no accumulator clearing pass
                        acc[j] = H8_MFMA16(__builtin_bit_cast(f16x8, wc), fa[2 * gq][c], (kt == 0 && c == 0) ? zero16 : acc[j]);
                    }
            } else if constexpr (F6) {
                // yh Wl: fp6(yh / block scale) of the group's 32 values in the image's order, one conversion instruction; the scale's
                // exponent goes through `opq` (an opaque zero) so that the conversions of all groups are not hoisted out of the tile loop
                const int bh = alo[gq][7] + opq;
                const u32x6 pk = h6_pack32(fa[2 * gq][0], fa[2 * gq][1], fa[2 * gq + 1][0], fa[2 * gq + 1][1], h6_scale_of(bh));
                const i32x8 a6 = {(int)pk[0], (int)pk[1], (int)pk[2], (int)pk[3], (int)pk[4], (int)pk[5], 0, 0};
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = H6_MFMA(fbA[j], a6, acc[j], fbA[j][6], bh);
            } else {
                // yh Wl: fp8(yh) of the group's two k-steps, bytes in the image's k order (16 t + 8 c + e)
                i32x8 a8;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const f16x8 v = fa[2 * gq + t][c];
                        s16x2 p0 = {0, 0}, p1 = {0, 0};
                        p0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p0, f16x2{v[0], v[1]}, one, false);
                        p0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p0, f16x2{v[2], v[3]}, one, true);
                        p1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p1, f16x2{v[4], v[5]}, one, false);
                        p1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p1, f16x2{v[6], v[7]}, one, true);
                        a8[4 * t + 2 * c] = __builtin_bit_cast(int, p0);
                        a8[4 * t + 2 * c + 1] = __builtin_bit_cast(int, p1);
                    }
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = H8_MFMA8(fbA[j], a8, acc[j], H8_SC_WL, H8_SC_AH);
            }
            // sub-step b: the first fragments of the next stage are read while it runs
            load_f(std::integral_constant<int, (kt + 1) % NS>{}, std::integral_constant<int, 0>{}, fbA);
            if constexpr (!lstage) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const i32x4 wc = c == 0 ? __builtin_shufflevector(fbB[j], fbB[j], 0, 1, 2, 3) : __builtin_shufflevector(fbB[j], fbB[j], 4, 5, 6, 7);
                        acc[j] = H8_MFMA16(__builtin_bit_cast(f16x8, wc), fa[2 * gq + 1][c], acc[j]);
                    }
            } else if constexpr (F6) {
                // yl W
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = H6_MFMA(fbB[j], alo[gq], acc[j], fbB[j][6], alo[gq][6]);
            } else {
                // yl W
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] = H8_MFMA8(fbB[j], alo[gq], acc[j], H8_SC_W8, H8_SC_AL);
            }
        });
#ifdef H8_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#ifndef H8_DIAG_NOEPI
        epilogue(ct);
#endif
    }
    HSTAMP(2);
#ifdef H8_DIAG_NOEPI
    if (acc[0][0] == 123.456f) epilogue(0);
#endif
    dma::wait_vm_lgkm0<0>();   // the re-fetched tail stages still target this block's LDS: land them before it is released
}

// ---------------------------------------------------------------------------------------------------------------------
// kv_proj | q_proj of the mixed mode on the same structure (reference: models/set_transformer.py:49-52, 65-70 — `kv_proj`, the
// `q` rows of nn.MultiheadAttention's in_proj — over y = AdaGN(x), normalization.py:36-44):
//   C1 (K | V) and C2 (q) = fp16( fp16(y) Wh^T [+ fp8(y) Wl^T for the V columns] + bias ),  head-major or row-major fp16 outputs.
// The A operand is yh alone (96 registers at K = 384; its rounding does not reach the output, section 5 of DESIGN.md), the W
// stream is kvq_image_item's.  Tiles have NG stages (K, q) or 3 NG / 2 (V), so the ring slot is a running counter (scalar
// arithmetic; the fragment addresses take one vector add per sub-step).  Epilogue: fp16 pairs, one v_permlane32_swap per
// register pair, 16-byte stores (8 consecutive columns of one row = one piece of a head's (rows, hd) slab).
//
// OUT (the training path under the reference's autocast(float16) setting, autograd.py `_lin_precision`; fp32 tensors, one-term weights):
//   1  C (| C2) = y W^T + bias as fp32, row-major — AdaGN(x) -> K | V and q (set_transformer.py:161-162 -> :49, :112) and any other
//      product out of a <= 512-wide operand: a lane holds 4 consecutive columns of ONE row per accumulator quad, so the fp32
//      epilogue is plain 16-byte stores (lane halves write adjacent pieces), no exchange;
//   2  pre_out = u = y W^T + bias (fp32) and C = fp16(act(u)): the first linear of an MLP in training (models/mlp.py:5-39), its
//      hidden layer stored as halves (the matrix pipe reads it again as fp16 either way);
//   3  C = (y W^T) * act'(u), u = mul_u (fp32), + the alpha-gradient partial of the block: the dX product through an activation
//      (autograd of mlp.py's Linear -> act); mul_kind 0: C = y W^T + mul_u, a dX product added onto another gradient of the same
//      tensor.  The u rows of the NEXT tile are loaded during the epilogue of this one.
template <int NG, int NS, int OUT = 0, bool P48 = false>
__global__ __launch_bounds__(256, 2) void gemm_kvq_astat_kernel(GemmArgs g) {
    constexpr int K = 64 * NG, NT = 256, NW = 4, ROWS = 128, PW = 2;
    static_assert(!P48 || OUT == 0 || OUT == 3, "the 4th parameter: OUT 0 the head-aligned column order; OUT 3 the result stored as fp16 (C16)");
    constexpr bool C16 = P48 && OUT == 3;   // du = (dy W) act'(u) as halves: its two consumers (the weight gradient and the next dX product) round it to fp16 anyway
    constexpr int KV_STORES = OUT == 0 ? 4 : OUT == 1 ? 8 : OUT == 2 ? 12 : (OUT == 3 && P48) ? 12 : 16;   // vector-memory instructions of one epilogue (OUT 3: 8 u loads + 8 stores, or + 4 as halves)
    static_assert(NS >= 4 && NG % 2 == 0 && NS - 2 <= NG, "lookahead NS - 1 >= 3 stages; L stages hold two groups; one epilogue's stores in flight");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ring = smem;                                // [NS][H_STAGE]
    float* stg = ring + NS * H_STAGE;                  // [NW][1024] wave-private staging: [32][64] fp16
    float* bias_lds = stg + NW * 1024;                 // [Nout] (zeros where a segment has no bias)
    float* pro_lds = bias_lds + g.Nout;                // pa | po

    const int tilesM = g.rows / ROWS, tilesN = g.Nout / H_BN;
    const int bid = g.h8_rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int b = bid / tilesM, rt = bid % tilesM, m0 = rt * ROWS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int lo_begin = g.lo_begin, lo_end = g.lo_tiles;   // 64-column tiles with L stages
    HSTAMP(0);

    for (int n = tid; n < g.Nout; n += NT) {
        const int nl = (P48 && OUT == 0) ? kvq_perm48_col(n >> 6, n & 63) : n;   // the column tile position n computes (bias_lds is in tile order)
        const bool seg2 = g.C2 != nullptr && nl >= g.n_split;
        const float* bp = seg2 ? g.bias2 : g.bias;
        bias_lds[n] = bp ? bp[seg2 ? nl - g.n_split : nl] : 0.f;
    }
    {
        const bool has_pro = g.pro_a != nullptr;
        const float* pa = has_pro ? g.pro_a + (size_t)b * K : nullptr;
        const float* po = has_pro ? g.pro_o + (size_t)b * K : nullptr;
        for (int i = tid; i < K; i += NT) {
            pro_lds[i] = has_pro ? pa[i] : 1.f;
            pro_lds[K + i] = has_pro ? po[i] : 0.f;
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.w_img), 0, 0x7fffffff, 0x00020000);
    const unsigned voff = (unsigned)(wave * PW * 256 + lane * 4) * 4u;
    const int n_stages = tilesN * NG + (lo_end - lo_begin) * (NG / 2);
    const unsigned soff_last = (unsigned)(n_stages - 1) * (H_STAGE * 4u);
    unsigned soff = 0;
    int islot = 0;                                     // slot of the next stage to issue
    auto issue = [&]() {
#ifndef H8_DIAG_NODMA
#pragma unroll
        for (int p = 0; p < PW; ++p)
            dma16_buf(wrsrc, voff + p * 1024u, soff, ring + islot * H_STAGE + (wave * PW + p) * 256);
#endif
        soff = soff < soff_last ? soff + H_STAGE * 4u : soff_last;
        islot = islot + 1 == NS ? 0 : islot + 1;
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p) issue();

    f16x8 fa[2 * NG][2];
    // h8_scales.h: an operand that meets an fp8 term is clamped to +-3584 (the scaled conversions return NaN beyond the format).  A RAW
    // operand (no AdaGN apply) of a one-term stream is a gradient of the training path (autograd.py `_linear_dx`, the activation
    // backward): it keeps fp16's whole range — a loss-scaled value beyond 3584 must not saturate silently, and one beyond 65504 becomes
    // the inf the GradScaler looks for (round 6; until then every operand was clamped)
    const float a_lim = (g.pro_a != nullptr || lo_end > lo_begin) ? H8_A_MAX : __builtin_inff();
    {
        const float* xw = g.A + ((size_t)b * g.rows + m0 + wave * 32) * g.lda;
        char* sw = reinterpret_cast<char*>(stg + wave * 1024);
        const int lrow = lane >> 4, c16 = lane & 15;
        f32x4 xs[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xs[0][i] = *reinterpret_cast<const f32x4*>(xw + (size_t)(4 * i + lrow) * g.lda + 4 * c16);
        static_for(std::make_integer_sequence<int, NG>{}, [&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (s + 1 < NG) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    xs[(s + 1) & 1][i] = *reinterpret_cast<const f32x4*>(xw + (size_t)(4 * i + lrow) * g.lda + 64 * (s + 1) + 4 * c16);
            }
            const f32x4 pa4 = *reinterpret_cast<const f32x4*>(pro_lds + 64 * s + 4 * c16);
            const f32x4 po4 = *reinterpret_cast<const f32x4*>(pro_lds + K + 64 * s + 4 * c16);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 4 * i + lrow;
                f16x4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) hv[e] = (_Float16)__builtin_amdgcn_fmed3f(__builtin_fmaf(xs[s & 1][i][e], pa4[e], po4[e]), -a_lim, a_lim);
                *reinterpret_cast<u32x2*>(sw + row * 128 + (((c16 >> 1) ^ (row & 7)) << 4) + (c16 & 1) * 8) = __builtin_bit_cast(u32x2, hv);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int cq = 4 * t + 2 * h + c;
                    fa[2 * s + t][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(sw + r * 128 + ((cq ^ (r & 7)) << 4)));
                }
            if (g.y16_out) {   // (launch-uniform) the 64-column group of the wave's 32 rows as it stands in the tile: 128 contiguous bytes per row
                _Float16* yb = static_cast<_Float16*>(g.y16_out) + ((size_t)b * g.rows + m0 + wave * 32) * K + 64 * s;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int idx = i * 64 + lane, row = idx >> 3, ch = idx & 7;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(sw + row * 128 + ((ch ^ (row & 7)) << 4));
                    *reinterpret_cast<u32x4*>(yb + (size_t)row * K + ch * 8) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        });
    }

    // fragment addressing: rows r and 32 + r of a sub-tile share the swizzle ((32 + r) >> 2 & 3 == r >> 2 & 3): two lane offsets
    // (c = 0, 1), the second 32-column block 2 KiB further
    int boffc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) boffc[c] = r * 16 + (((2 * h + c) ^ ((r >> 2) & 3)) << 2);
    i32x8 fbA[2], fbB[2];
    auto load_f = [&](const float* sub, i32x8(&f)[2]) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* pc = sub + boffc[c];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(pc + j * 512);
#pragma unroll
                for (int e = 0; e < 4; ++e) f[j][4 * c + e] = (int)v[e];
            }
        }
    };

    const int mrow = m0 + wave * 32 + r;
    const unsigned hm_magic = g.hm_hd ? (1u << 20) / (unsigned)g.hm_hd + 1u : 0u;
    f32x16 acc[2];
    // OUT 3: the pre-activation rows of the tile being multiplied (8 x 16 bytes per lane), its activation constants, the lane's
    // share of the alpha gradient
    f32x4 uq[OUT == 3 ? 8 : 1];
    float ga = 0.f;
    const int act_code = OUT == 2 ? g.act : OUT == 3 ? g.mul_kind : 0;
    const float alpha0 = (OUT >= 2 && act_is_gauss(act_code)) ? g.alpha[0] : 1.f;
    const float neg_inv_2a2 = -1.0f / (2.0f * alpha0 * alpha0), inv_a2 = 1.0f / (alpha0 * alpha0);
    auto load_u = [&](int ct) {
        if constexpr (OUT == 3) {
            const int ctc = min(ct, tilesN - 1);   // past the last tile: a harmless re-read (keeps the instruction count static)
            const float* ub = g.mul_u + ((size_t)b * g.rows + mrow) * g.ldc + ctc * H_BN + 4 * h;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) uq[4 * j + q] = *reinterpret_cast<const f32x4*>(ub + 32 * j + 8 * q);
        }
    };
    auto epilogue = [&](int ct) {
        const int n0 = ct * H_BN;
        if constexpr (OUT != 0) {
            const bool s2 = g.C2 != nullptr && n0 >= g.n_split;
            float* Cf = s2 ? g.C2 : g.C;
            const int ldcf = s2 ? g.ldc2 : g.ldc;
            const int ns0 = s2 ? n0 - g.n_split : n0;
            float* dstf = (OUT == 2 ? g.pre_out : Cf) + ((size_t)b * g.rows + mrow) * (OUT == 2 ? g.Nout : ldcf) + (OUT == 2 ? n0 : ns0) + 4 * h;
            f32x4 v[OUT == 2 ? 8 : 1];
            f32x4 w16[C16 ? 4 : 1];   // C16: the four quads of a 32-column block, packed and stored as halves after the block
            // one accumulator quad at a time (bias, the activation's derivative, store): the scheduler would otherwise put all 32
            // exponentials of the tile in flight at once, ~100 registers the stationary operand does not leave
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(bias_lds + n0 + 32 * j + 8 * q + 4 * h);
                    f32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = acc[j][4 * q + e] + bs[e];
                    if constexpr (OUT == 3) {
                        if (act_code == 0) {   // (launch-uniform) mul_kind 0: mul_u is a RESIDUAL — another gradient contribution to the same tensor
#pragma unroll
                            for (int e = 0; e < 4; ++e) w[e] += uq[4 * j + q][e];
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float da;
                                const float f = tr_act_prime(uq[4 * j + q][e], neg_inv_2a2, inv_a2, act_code, da);
                                ga += w[e] * da;
                                w[e] *= f;
                            }
                        }
                    }
                    if constexpr (C16) w16[q] = w;
                    else *reinterpret_cast<f32x4*>(dstf + 32 * j + 8 * q) = w;
                    if constexpr (OUT == 2) v[4 * j + q] = w;
                    if constexpr (OUT == 3) __builtin_amdgcn_sched_barrier(0);
                    if constexpr (C16) {
                        if (q == 3) {   // (unrolled: q is a constant) the fp16 epilogue's exchange: 8 consecutive columns per lane, 16-byte stores
                            unsigned pk[4][2];
#pragma unroll
                            for (int qq = 0; qq < 4; ++qq) {
                                f16x2 a2, c2;
                                a2[0] = (_Float16)w16[qq][0]; a2[1] = (_Float16)w16[qq][1];
                                c2[0] = (_Float16)w16[qq][2]; c2[1] = (_Float16)w16[qq][3];
                                pk[qq][0] = __builtin_bit_cast(unsigned, a2);
                                pk[qq][1] = __builtin_bit_cast(unsigned, c2);
                            }
                            _Float16* Hd = reinterpret_cast<_Float16*>(g.C) + ((size_t)b * g.rows + mrow) * g.ldc + n0 + 32 * j + 8 * h;
#pragma unroll
                            for (int pp = 0; pp < 2; ++pp) {
                                u32x4 O;
#pragma unroll
                                for (int d = 0; d < 2; ++d) {
                                    const auto sh = __builtin_amdgcn_permlane32_swap(pk[2 * pp][d], pk[2 * pp + 1][d], false, false);
                                    O[d] = sh[0];
                                    O[2 + d] = sh[1];
                                }
                                *reinterpret_cast<u32x4*>(Hd + 16 * pp) = O;
                            }
                        }
                    }
                }
            if constexpr (OUT == 3) load_u(ct + 1);
            if constexpr (OUT == 2) {   // act(u) as halves: the fp16 epilogue's exchange (8 consecutive columns per lane, 16-byte stores)
                _Float16* Hb = reinterpret_cast<_Float16*>(g.C);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    unsigned pk[4][2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f16x2 a, c;
                        a[0] = (_Float16)tr_act(v[4 * j + q][0], neg_inv_2a2, act_code);
                        a[1] = (_Float16)tr_act(v[4 * j + q][1], neg_inv_2a2, act_code);
                        c[0] = (_Float16)tr_act(v[4 * j + q][2], neg_inv_2a2, act_code);
                        c[1] = (_Float16)tr_act(v[4 * j + q][3], neg_inv_2a2, act_code);
                        pk[q][0] = __builtin_bit_cast(unsigned, a);
                        pk[q][1] = __builtin_bit_cast(unsigned, c);
                    }
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        u32x4 O;
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            const auto sh = __builtin_amdgcn_permlane32_swap(pk[2 * p][d], pk[2 * p + 1][d], false, false);
                            O[d] = sh[0];
                            O[2 + d] = sh[1];
                        }
                        *reinterpret_cast<u32x4*>(Hb + ((size_t)b * g.rows + mrow) * g.Nout + n0 + 32 * j + 16 * p + 8 * h) = O;
                    }
                }
            }
            return;
        }
        const bool seg2 = g.C2 != nullptr && n0 >= g.n_split;
        _Float16* Cb = reinterpret_cast<_Float16*>(seg2 ? g.C2 : g.C);
        const int ldc = seg2 ? g.ldc2 : g.ldc;
        const int nseg0 = seg2 ? n0 - g.n_split : n0;
        const int nseg = seg2 ? g.Nout - g.n_split : (g.C2 ? g.n_split : g.Nout);
        // P48: heads of this tensor before the tile's 384-column segment, the tile's place in the segment; the wave's LDS tile (idle since
        // the prologue) takes the full head's (32 rows, 48) slab exactly as it lies in memory: row pitch 96 bytes
        const int p_hb = P48 ? (nseg0 / 384) * 8 : 0, p_t = P48 ? (nseg0 % 384) >> 6 : 0;
        char* p_sw = reinterpret_cast<char*>(stg + wave * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned pk[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bs = *reinterpret_cast<const f32x4*>(bias_lds + n0 + 32 * j + 8 * q + 4 * h);
                f16x2 a, c;
                a[0] = (_Float16)(acc[j][4 * q + 0] + bs[0]);
                a[1] = (_Float16)(acc[j][4 * q + 1] + bs[1]);
                c[0] = (_Float16)(acc[j][4 * q + 2] + bs[2]);
                c[1] = (_Float16)(acc[j][4 * q + 3] + bs[3]);
                pk[q][0] = __builtin_bit_cast(unsigned, a);
                pk[q][1] = __builtin_bit_cast(unsigned, c);
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                u32x4 O;
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const auto sh = __builtin_amdgcn_permlane32_swap(pk[2 * p][d], pk[2 * p + 1][d], false, false);
                    O[d] = sh[0];
                    O[2 + d] = sh[1];
                }
                if constexpr (P48) {
                    if (2 * j + p < 3) {   // a third of the full head: into the slab image
                        *reinterpret_cast<u32x4*>(p_sw + r * 96 + (2 * j + p) * 32 + h * 16) = O;
                    } else {               // the stray third of head 6 + t / 3: 32-byte pieces, as before
                        _Float16* ds = Cb + ((size_t)(b * (nseg / 48) + p_hb + 6 + p_t / 3) * g.rows + mrow) * 48 + 16 * (p_t % 3) + 8 * h;
                        *reinterpret_cast<u32x4*>(ds) = O;
                    }
                    continue;
                }
                const int nb = nseg0 + 32 * j + 16 * p + 8 * h;   // first of this lane's 8 columns, inside its segment
                _Float16* dst;
                if (g.hm_hd) {
                    const int grp = (int)(((unsigned)nb * hm_magic) >> 20);   // nb / hd (exact: nb < 2^20 / hd)
                    dst = Cb + ((size_t)(b * (nseg / g.hm_hd) + grp) * g.rows + mrow) * g.hm_hd + (nb - grp * g.hm_hd);
                } else {
                    dst = Cb + ((size_t)b * g.rows + mrow) * ldc + nb;
                }
#ifdef H8_DIAG_NOSTORE
                asm volatile("" ::"v"(O), "v"(dst));
#else
#ifdef KVQ_NT_STORE
                GECCO_NT_STORE(O, reinterpret_cast<u32x4*>(dst));
#else
                *reinterpret_cast<u32x4*>(dst) = O;   // default policy: the attention kernels read K | V and q next
#endif
#endif
            }
        }
        if constexpr (P48) {   // the slab: 3 KiB contiguous in head-major memory — three whole-wave 1 KiB stores
            __builtin_amdgcn_wave_barrier();
            char* slab = reinterpret_cast<char*>(Cb + ((size_t)(b * (nseg / 48) + p_hb + p_t) * g.rows + m0 + wave * 32) * 48);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(p_sw + i * 1024 + lane * 16);
                *reinterpret_cast<u32x4*>(slab + i * 1024 + lane * 16) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    };

    HSTAMP(1);
    load_u(0);
    dma::wait_vm_lgkm0<0>();
    __builtin_amdgcn_s_barrier();
    int slot = 0;                                      // slot of the current stage
    load_f(ring, fbA);
    float one = H8_AH_DIV;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // one stage: wait / barrier / issue, two sub-steps.  LST: an L stage (two groups' Wl); G2: its first group (L) or the group (H)
    auto stage = [&](auto KT, auto LST, auto G2, bool first) {
        constexpr int kt = decltype(KT)::value, gq = decltype(G2)::value;
        constexpr bool lst = decltype(LST)::value;
        constexpr int young = (NS - 3) * PW;
        constexpr bool st_young = kt <= NS - 3;
        if (st_young && !first) dma::wait_vm_lgkm0<young + KV_STORES>();
        else dma::wait_vm_lgkm0<young>();
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(fbA[j]));
        __builtin_amdgcn_s_barrier();
        issue();
        const float* cur = ring + slot * H_STAGE;
        slot = slot + 1 == NS ? 0 : slot + 1;
        const float* nxt = ring + slot * H_STAGE;
        load_f(cur + 1024, fbB);
        if constexpr (!lst) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const i32x4 wc = c == 0 ? __builtin_shufflevector(fbA[j], fbA[j], 0, 1, 2, 3) : __builtin_shufflevector(fbA[j], fbA[j], 4, 5, 6, 7);
                    acc[j] = H8_MFMA16(__builtin_bit_cast(f16x8, wc), fa[2 * gq][c], (kt == 0 && c == 0) ? zero16 : acc[j]);
                }
        } else {
            i32x8 a8;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f16x8 v = fa[2 * gq + t][c];
                    s16x2 p0 = {0, 0}, p1 = {0, 0};
                    p0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p0, f16x2{v[0], v[1]}, one, false);
                    p0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p0, f16x2{v[2], v[3]}, one, true);
                    p1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p1, f16x2{v[4], v[5]}, one, false);
                    p1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p1, f16x2{v[6], v[7]}, one, true);
                    a8[4 * t + 2 * c] = __builtin_bit_cast(int, p0);
                    a8[4 * t + 2 * c + 1] = __builtin_bit_cast(int, p1);
                }
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = H8_MFMA8(fbA[j], a8, acc[j], H8_SC_WL, H8_SC_AH);
        }
        load_f(nxt, fbA);
        if constexpr (!lst) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const i32x4 wc = c == 0 ? __builtin_shufflevector(fbB[j], fbB[j], 0, 1, 2, 3) : __builtin_shufflevector(fbB[j], fbB[j], 4, 5, 6, 7);
                    acc[j] = H8_MFMA16(__builtin_bit_cast(f16x8, wc), fa[2 * gq + 1][c], acc[j]);
                }
        } else {
            i32x8 a8;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f16x8 v = fa[2 * (gq + 1) + t][c];
                    s16x2 p0 = {0, 0}, p1 = {0, 0};
                    p0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p0, f16x2{v[0], v[1]}, one, false);
                    p0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p0, f16x2{v[2], v[3]}, one, true);
                    p1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p1, f16x2{v[4], v[5]}, one, false);
                    p1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p1, f16x2{v[6], v[7]}, one, true);
                    a8[4 * t + 2 * c] = __builtin_bit_cast(int, p0);
                    a8[4 * t + 2 * c + 1] = __builtin_bit_cast(int, p1);
                }
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = H8_MFMA8(fbB[j], a8, acc[j], H8_SC_WL, H8_SC_AH);
        }
    };
    for (int ct = 0; ct < tilesN; ++ct) {
        const bool first = ct == 0;
        const bool has_lo = ct >= lo_begin && ct < lo_end;
        asm volatile("" : "+s"(one));
        static_for(std::make_integer_sequence<int, NG>{}, [&](auto KT) {
            stage(KT, std::false_type{}, KT, first);
        });
        if (has_lo) {
            static_for(std::make_integer_sequence<int, NG / 2>{}, [&](auto LT) {
                constexpr int lt = decltype(LT)::value;
                stage(std::integral_constant<int, NG + lt>{}, std::true_type{}, std::integral_constant<int, 2 * lt>{}, first);
            });
        }
#ifndef H8_DIAG_NOEPI
        epilogue(ct);
#endif
    }
    HSTAMP(2);
#ifdef H8_DIAG_NOEPI
    if (acc[0][0] == 123.456f) epilogue(0);
#endif
    dma::wait_vm_lgkm0<0>();
    if constexpr (OUT == 3) {
        if (g.agrad && act_is_gauss(act_code)) {   // (block-uniform) lanes, then the four waves, in a fixed order: one partial per block
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) ga += __shfl_xor(ga, o, 64);
            __syncthreads();                       // the ring is dead
            if (lane == 0) ring[wave] = ga;
            __syncthreads();
            if (tid == 0) g.agrad[bid] = (((ring[0] + ring[1]) + ring[2]) + ring[3]) / alpha0;
        }
    }
}

template <int NG, int NS, int OUT = 0, bool P48 = false>
int kvq_launch_t(const GemmArgs& g, hipStream_t st) {
    const size_t lds = ((size_t)NS * H_STAGE + 4 * 1024 + g.Nout + 2 * g.K) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kvq_astat_kernel<NG, NS, OUT, P48>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL((gemm_kvq_astat_kernel<NG, NS, OUT, P48>), dim3(g.B * (g.rows / 128)), dim3(256), lds, st, g);
    return (int)hipGetLastError();
}

constexpr int h8_ns(int NG, int NW) { return NW == 8 ? 2 * NG : (NG % 3 == 0 ? 6 : 4); }   // NW = 4: 48 / 32 KiB rings, two blocks per CU

template <int NG, int NW, int ACT, bool IMG2>
int h8_launch_i(const GemmArgs& g, hipStream_t st) {
    constexpr int NS = h8_ns(NG, NW);
    const size_t lds = ((size_t)NS * H_STAGE + NW * H_STG + g.Nout + 2 * g.K) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h8_astat_kernel<NG, NW, NS, ACT, IMG2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    if constexpr (IMG2 && NW == 4) {
        if (g.h6) {   // the "h6" arithmetic (w_img is the h6 stream)
            static size_t attr6 = 0;
            if (lds > attr6) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h8_astat_kernel<NG, NW, NS, ACT, IMG2, 0, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr6 = lds;
            }
            hipLaunchKernelGGL((gemm_h8_astat_kernel<NG, NW, NS, ACT, IMG2, 0, true>), dim3(g.B * (g.rows / (32 * NW))), dim3(64 * NW), lds, st, g);
            return (int)hipGetLastError();
        }
    } else if (g.h6) {
        return -9;
    }
    hipLaunchKernelGGL((gemm_h8_astat_kernel<NG, NW, NS, ACT, IMG2>), dim3(g.B * (g.rows / (32 * NW))), dim3(64 * NW), lds, st, g);
    return (int)hipGetLastError();
}
template <int NG, int NW, int ACT>
int h8_launch_a(const GemmArgs& g, hipStream_t st) {
    return g.c_img == 2 ? h8_launch_i<NG, NW, ACT, true>(g, st) : h8_launch_i<NG, NW, ACT, false>(g, st);
}
template <int NG, int NW>
int h8_launch_t(const GemmArgs& g, hipStream_t st) {
    switch (g.act) {
        case 0: return h8_launch_a<NG, NW, 0>(g, st);
        case 1: return h8_launch_a<NG, NW, 1>(g, st);
        case 2: return h8_launch_a<NG, NW, 2>(g, st);
        case 3: return h8_launch_a<NG, NW, 3>(g, st);
        default: return -9;
    }
}

int h8_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

}  // namespace

size_t h8_image_bytes(int Nout, int K) { return (size_t)((Nout + H_BN - 1) / H_BN) * H_BN * K * 4; }

int h8_image_multi_launch(const SplitJobs& jobs, hipStream_t st) {
    if (jobs.n <= 0) return 0;
    for (int i = 0; i < jobs.n; ++i)
        if (jobs.job[i].K % 64 || (!(jobs.job[i].pad_ & 2) && jobs.job[i].Nout % H_BN) || (jobs.job[i].ldw & 3)) return -9;
    hipLaunchKernelGGL(h8_image_multi_kernel, dim3(48, jobs.n), dim3(256), 0, st, jobs);
    return (int)hipGetLastError();
}

// GECCO_H8_K512=0: d = 512 keeps mlp.0 on the split-bf16 LDS-DMA GEMM (A/B runs)
static bool h8_k512_on() {
    static const int on = h8_env("GECCO_H8_K512", 1);
    return on != 0;
}

// c_img output only (1: tiled split image, 2: h8 activation image): whole 128-row blocks of one sample, 64-column tiles,
// K = 128 / 256 / 384 (the stationary operand is 3 K / 8 registers per lane); K = 512 at one block per CU
bool gemm_h8_astat_supported(const GemmArgs& g) {
    return g.c_img && !g.a_img && !g.a_f16 && !g.c_f16 && !g.residual && !g.stats && !g.C2 && g.w_img && g.rows >= 128 &&
           !(g.rows % 128) && !(g.Nout % H_BN) && g.Nout >= 2 * H_BN && g.Nout <= 4096 &&
           (g.K == 128 || g.K == 256 || g.K == 384 || (g.K == 512 && h8_k512_on())) &&
           !(g.lda & 3) && ((g.pro_a == nullptr) == (g.pro_o == nullptr)) && !g.mul_u && !g.pre_out && g.act >= 0 && g.act <= 3;
}

// kv_proj | q_proj stream: bytes of the image of one job (Nout columns, `lo_cols` of them with L stages)
size_t kvq_image_bytes(int Nout, int K, int lo_cols) { return ((size_t)(Nout / H_BN) * (K / 64) + (size_t)(lo_cols / H_BN) * (K / 128)) * H_STAGE * 4; }

bool kvq_perm48_ok(int hm_hd, int K, int n_first, int n_second) {
    return hm_hd == 48 && K == 384 && n_first > 0 && n_first % 384 == 0 && n_second % 384 == 0;
}

// fp16 outputs (one or two segments, row- or head-major), 128-row blocks, 64-column tiles; lo_begin / lo_tiles in 64-column tiles
bool gemm_kvq_astat_supported(const GemmArgs& g) {
    return g.c_f16 && !g.a_f16 && !g.a_img && !g.c_img && !g.residual && !g.stats && g.w_img && !g.act && g.rows >= 128 && !(g.rows % 128) &&
           !(g.Nout % H_BN) && g.Nout >= 2 * H_BN && g.Nout <= 4096 && (g.K == 128 || g.K == 256 || g.K == 384 || g.K == 512) && !(g.lda & 3) && !(g.ldc & 7) &&
           (!g.C2 || (!(g.n_split % H_BN) && !(g.ldc2 & 7) && g.n_split > 0 && g.n_split < g.Nout)) &&
           ((g.pro_a == nullptr) == (g.pro_o == nullptr)) && g.lo_begin >= 0 && g.lo_tiles >= g.lo_begin && g.lo_tiles <= g.Nout / H_BN &&
           (!g.hm_hd || (g.hm_hd >= 8 && !(g.hm_hd & 7) && g.Nout < (1 << 20) / g.hm_hd && !((g.C2 ? g.n_split : g.Nout) % g.hm_hd) &&
                         !((g.C2 ? g.Nout - g.n_split : 0) % g.hm_hd)));
}

// gemm_h8_astat_kernel's OUT forms: the training forward in h8 arithmetic (w_img = the h8 stream of the weight, or of W1 | W2)
bool gemm_h8_train_supported(const GemmArgs& g) {
    const bool keep = g.pre_out != nullptr;
    return !g.c_f16 && !g.a_f16 && !g.a_img && !g.c_img && !g.residual && !g.stats && !g.mul_u && g.w_img && g.rows >= 128 && !(g.rows % 128) &&
           !(g.Nout % H_BN) && g.Nout >= 2 * H_BN && g.Nout <= 4096 && (g.K == 128 || g.K == 256 || g.K == 384) && !(g.lda & 3) && !(g.ldc & 3) &&
           (!g.C2 || (!keep && !(g.n_split % H_BN) && !(g.ldc2 & 3) && g.n_split > 0 && g.n_split < g.Nout)) &&
           ((g.pro_a == nullptr) == (g.pro_o == nullptr)) && (keep ? (g.act >= 1 && g.act <= 3 && g.ldc == g.Nout) : g.act == 0) &&
           (!(keep && act_gauss_host(g.act)) || g.alpha);
}

template <int NG, int NS>
int h8_train_launch_t(const GemmArgs& g, hipStream_t st) {
    const size_t lds = ((size_t)NS * H_STAGE + 4 * H_STG + g.Nout + 2 * g.K) * sizeof(float);
    const dim3 grid(g.B * (g.rows / 128));
#define H8T(ACT_, OUT_)                                                                                                          \
    do {                                                                                                                         \
        static size_t attr = 0;                                                                                                  \
        if (lds > attr) {                                                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h8_astat_kernel<NG, 4, NS, ACT_, false, OUT_>),          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                     \
            attr = lds;                                                                                                          \
        }                                                                                                                        \
        hipLaunchKernelGGL((gemm_h8_astat_kernel<NG, 4, NS, ACT_, false, OUT_>), grid, dim3(256), lds, st, g);                    \
    } while (0)
    if (!g.pre_out) H8T(0, 1);
    else if (g.act == 1) H8T(1, 2);
    else if (g.act == 2) H8T(2, 2);
    else H8T(3, 2);
#undef H8T
    return (int)hipGetLastError();
}

int gemm_h8_train_launch(const GemmArgs& g0, hipStream_t st) {
    if (!gemm_h8_train_supported(g0)) return -9;
    GemmArgs g = g0;
    g.h8_rev = 0; g.h8_stagger = 0; g.h8_pair = 32;
    switch (g.K) {
        case 128: return h8_train_launch_t<2, 4>(g, st);
        case 256: return h8_train_launch_t<4, 4>(g, st);
        case 384: return h8_train_launch_t<6, 6>(g, st);
        default: return -9;
    }
}

// the training forms (OUT 1 / 2 / 3 above): fp32 C (| C2), or pre_out + fp16 C, or mul_u; one-term weights, row-major
bool gemm_astat_train_supported(const GemmArgs& g) {
    const bool keep = g.pre_out != nullptr, abw = g.mul_u != nullptr;
    if (keep && abw) return false;
    if (g.c_f16 && !(abw && g.mul_kind >= 1 && !(g.ldc & 7))) return false;   // fp16 result: the activation-backward form only
    return !g.a_f16 && !g.a_img && !g.c_img && !g.residual && !g.stats && g.w_img && g.rows >= 128 && !(g.rows % 128) && !(g.Nout % H_BN) &&
           g.Nout >= 2 * H_BN && g.Nout <= 4096 && (g.K == 128 || g.K == 256 || g.K == 384 || g.K == 512) && !(g.lda & 3) && !(g.ldc & 3) &&
           (!g.C2 || (!keep && !abw && !(g.n_split % H_BN) && !(g.ldc2 & 3) && g.n_split > 0 && g.n_split < g.Nout)) &&
           ((g.pro_a == nullptr) == (g.pro_o == nullptr)) && g.lo_begin == 0 && g.lo_tiles == 0 && !g.hm_hd &&
           (keep ? (g.act >= 1 && g.act <= 3 && g.ldc == g.Nout) : g.act == 0) &&
           (!abw || (g.mul_kind >= 0 && g.mul_kind <= 3 && !g.bias && !g.pro_a && g.ldc == g.Nout)) &&
           (!((keep && act_gauss_host(g.act)) || (abw && act_gauss_host(g.mul_kind))) || g.alpha);
}

template <int OUT, bool VAR = false>
int astat_train_launch_o(const GemmArgs& g, hipStream_t st) {
    switch (g.K) {
        case 128: return kvq_launch_t<2, 4, OUT, VAR>(g, st);
        case 256: return kvq_launch_t<4, 6, OUT, VAR>(g, st);
        case 384: return kvq_launch_t<6, 6, OUT, VAR>(g, st);
        case 512: return kvq_launch_t<8, 6, OUT, VAR>(g, st);
        default: return -9;
    }
}

int gemm_astat_train_launch(const GemmArgs& g0, hipStream_t st) {
    if (!gemm_astat_train_supported(g0)) return -9;
    GemmArgs g = g0;
    g.h8_rev = 0;
    if (g.pre_out) return astat_train_launch_o<2>(g, st);
    if (g.mul_u) return g.c_f16 ? astat_train_launch_o<3, true>(g, st) : astat_train_launch_o<3>(g, st);
    return astat_train_launch_o<1>(g, st);
}

int gemm_kvq_astat_launch(const GemmArgs& g0, hipStream_t st) {
    if (!gemm_kvq_astat_supported(g0)) return -9;
    static const int rev = h8_env("GECCO_H8_REV", 0);
    GemmArgs g = g0;
    g.h8_rev = rev;
    switch (g.K) {
        case 128: return kvq_launch_t<2, 4>(g, st);   // NS - 2 <= stages of the shortest tile: the wait counts assume one epilogue in flight
        case 256: return kvq_launch_t<4, 6>(g, st);
#ifdef KVQ_NS
        case 384: return kvq_launch_t<6, KVQ_NS>(g, st);
#else
        case 384:
            if (g.kvq_perm) {
                if (!kvq_perm48_ok(g.hm_hd, g.K, g.C2 ? g.n_split : g.Nout, g.C2 ? g.Nout - g.n_split : 0)) return -9;
                return kvq_launch_t<6, 6, 0, true>(g, st);
            }
            return kvq_launch_t<6, 6>(g, st);
#endif
        case 512: return kvq_launch_t<8, 6>(g, st);   // d = 512: 128 registers of A fragments, no scratch
        default: return -9;
    }
}

// Two 128-row blocks of 4 waves per CU (one 256-row block of 8 waves measured the same, 192 vs 194 us, and is not instantiated).
// GECCO_H8_STAGGER: start offset (s_memtime ticks) of every second block of a CU, GECCO_H8_PAIR: blocks per XCD between partners
// (measured: no gain, default 0)
int gemm_h8_astat_launch(const GemmArgs& g0, hipStream_t st) {
    if (!gemm_h8_astat_supported(g0)) return -9;
    static const int stagger = h8_env("GECCO_H8_STAGGER", 0), pair = h8_env("GECCO_H8_PAIR", 32), rev = h8_env("GECCO_H8_REV", 0);
    GemmArgs g = g0;
    g.h8_rev = rev;
    g.h8_stagger = stagger;
    g.h8_pair = pair > 0 ? pair : 32;
    switch (g.K) {
        case 128: return h8_launch_t<2, 4>(g, st);
        case 256: return h8_launch_t<4, 4>(g, st);
        case 384: return h8_launch_t<6, 4>(g, st);
        case 512: return h8_launch_t<8, 4>(g, st);
        default: return -9;
    }
}
