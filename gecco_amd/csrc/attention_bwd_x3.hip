// Backward of the inducing-point attention in split-bf16 arithmetic (training path, precision "bf16x3") — gfx950.
//
// Same contract, grid, partials and orientation as attention_bwd_f32.hip (reference: autograd through
// F.scaled_dot_product_attention, models/set_transformer.py:55-63, and nn.MultiheadAttention, :112, under
// loss.backward(), diffusion.py:213-222), but every product is three v_mfma_f32_32x32x16_bf16 on hi | lo operands
// (a = hi + lo, hi = the top 16 bits, lo = bf16(a - hi): gemm_f32_dma.hip) instead of eight fp32 MFMAs of twice the
// cycles: 108 matrix instructions of 32 cycles per 32-row tile instead of 288 of 64, which takes both kernels off the
// matrix pipe.
//
// ONE LDS layout serves every operand: bf16 planes of 4-row x 32-column blocks (256 B), the layout of gemm_tn_x3.hip and
// of attention_x3.hip's value tile.  It is read three ways:
//   * row fragment (contraction over the columns: S^T = K Q^T, dP^T = V dO^T): 8 consecutive columns of a row are 16
//     contiguous bytes of a block — ds_read_b128 (4-way bank conflicts, 12 such reads per tile);
//   * transposed fragment (contraction over the rows: everything else): ds_read_b64_tr_b16, which hands lane half h the
//     rows 16s + 8(j>>2) + 4h + (j&3), j = 0..7, of its column — exactly the row order of accumulator registers
//     8s .. 8s+7, so a dS^T accumulator is the B fragment as it is (dQ^T = K^T dS^T, dq^T = k^T dS^T), and two
//     transposed fragments pair up with each other (dV = P dO, dK = dS Q; dv = P^T dO, dk = dS^T q: the "TN" products);
//   * the P / dS tile goes back to LDS for those TN products as hi | lo planes, 8 bytes (4 consecutive rows of the
//     accumulator = 4 consecutive columns of the plane) per store.
// Head dims 16, 32, 48, 64; anything else runs attention_bwd_f32.hip.
//
// IO16 (round 6, with F16): the point-stream tensors are fp16 TENSORS — K | V / q, dO in, dK | dV / dq out — as the reference's
// autocast(float16) backward holds them (autograd of F.scaled_dot_product_attention / nn.MultiheadAttention under Lightning's
// precision="16-mixed").  The planes take the halves as they are (the bits the fp32 tensors were rounded to on the way in), the
// results are rounded once on the way out — where their consumers (the dX / dW products of kv_proj | q_proj) rounded them before: half
// the bytes of six crossings of HBM per layer.  Inducer-side tensors (queries, k | v of the inducers, their gradients) stay fp32.
#include "common.h"
#include "kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// F16 (the reference's autocast(float16) trainer arithmetic, autograd.py `_lin_precision`: torch runs SDPA / MultiheadAttention and
// their backward with fp16 operands there): ONE fp16 plane per operand, one v_mfma_f32_32x32x16_f16 per product.  The kernels keep
// their layout; the lo planes are neither written nor multiplied (their fragment reads are dead code).
// 4 fp32 -> 4 bf16 hi (top 16 bits) and 4 bf16 lo = rne(x - hi), each packed in two dwords
template <bool F16>
__device__ __forceinline__ void split4(const f32x4& x, u32x2& hi, u32x2& lo) {
    if (F16) {
        f16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (_Float16)x[e];
        hi = __builtin_bit_cast(u32x2, v);
        lo = hi;
        return;
    }
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

// accumulator registers e0 .. e0+7 -> the hi / lo fragments of one 16-row chunk
template <bool F16>
__device__ __forceinline__ void split_acc8(const f32x16& s, int e0, u32x4& hi, u32x4& lo) {
    if (F16) {
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)s[e0 + e];
        hi = __builtin_bit_cast(u32x4, v);
        lo = hi;
        return;
    }
    u32x4 hb;
    bf16x8 l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = s[e0 + 2 * p], c = s[e0 + 2 * p + 1];
        const unsigned ua = __float_as_uint(a), uc = __float_as_uint(c);
        hb[p] = __builtin_amdgcn_perm(uc, ua, 0x07060302u);
        l[2 * p] = (__bf16)(a - __uint_as_float(ua & 0xFFFF0000u));
        l[2 * p + 1] = (__bf16)(c - __uint_as_float(uc & 0xFFFF0000u));
    }
    hi = hb;
    lo = __builtin_bit_cast(u32x4, l);
}

template <bool F16>
__device__ __forceinline__ f32x16 mfma3(const u32x4& ahi, const u32x4& alo, const u32x4& bhi, const u32x4& blo, f32x16 acc) {
    if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ahi), __builtin_bit_cast(f16x8, bhi), acc, 0, 0, 0);
    const bf16x8 ah = __builtin_bit_cast(bf16x8, ahi), al = __builtin_bit_cast(bf16x8, alo);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, bhi), bl = __builtin_bit_cast(bf16x8, blo);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// element offset of (row, col) in a plane of 4-row x 32-column blocks, NB column blocks per row group
template <int NB>
__device__ __forceinline__ int blk_off(int row, int col) {
    // inside a 256-byte block (= all 64 banks) a row takes one of four 64-byte slots and an 8-column piece one of its four
    // 16-byte chunks.  Both are permuted: the slot rotates with the column block (the column blocks one row of a store
    // instruction touches land on different bank groups) and the chunk is XORed with the 4-row group index — a row fragment
    // read (32 lanes = 32 rows = 8 row groups, the same 16 bytes of each) hit 4 banks 8-fold in the plain layout
    // (SQ_LDS_BANK_CONFLICT 74 % / 63 % of the LDS-active cycles of the two kernels); now groups g and g + 4 share a chunk,
    // 2-fold.  A transposed read (16 lanes: 4 rows of one group x 32 bytes) still covers whole slots.
    const int slot = (row + (col >> 5)) & 3, chunk = ((col & 31) >> 3) ^ ((row >> 2) & 3);
    return ((row >> 2) * NB + (col >> 5)) * 128 + slot * 32 + chunk * 8 + (col & 7);
}

// columns 16c + 8h .. + 7 of `row`: the fragment of a product that contracts over the columns
template <int NB>
__device__ __forceinline__ u32x4 rowfrag(const u16* plane, int row, int c, int h) {
    return *reinterpret_cast<const u32x4*>(plane + blk_off<NB>(row, c * 16 + 8 * h));
}

// rows 16s + 8(j>>2) + 4h + (j&3) of column 32 blk + (lane & 31): the fragment of a product that contracts over the rows
template <int NB>
__device__ __forceinline__ u32x4 trfrag(const u16* plane, int sg, int blk, int lane) {
    typedef __attribute__((address_space(3))) s16x4* lp;
    const int h = lane >> 5, tq = (lane & 15) >> 2, tp = lane & 3, tcol = 16 * ((lane >> 4) & 1) + 4 * tp;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(plane + blk_off<NB>(16 * sg + 4 * h + tq, blk * 32 + tcol)));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(plane + blk_off<NB>(16 * sg + 8 + 4 * h + tq, blk * 32 + tcol)));
    return __builtin_bit_cast(u32x4, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
}

// a [rows][HD] fp32 tile held as f32x4 pieces -> hi | lo planes (piece f: row f / CH, columns 4 (f % CH) ..)
template <int NB, bool F16>
__device__ __forceinline__ void put4(u16* hi_plane, u16* lo_plane, int row, int col, const f32x4& v) {
    u32x2 hi, lo;
    split4<F16>(v, hi, lo);
    const int o = blk_off<NB>(row, col);
    *reinterpret_cast<u32x2*>(hi_plane + o) = hi;
    if (!F16) *reinterpret_cast<u32x2*>(lo_plane + o) = lo;
}

// IO16: a [rows][HD] fp16 tile held as 8-byte pieces (4 halves) goes into the hi plane as it is
template <int NB>
__device__ __forceinline__ void put4h(u16* hi_plane, int row, int col, const u32x2& v) {
    *reinterpret_cast<u32x2*>(hi_plane + blk_off<NB>(row, col)) = v;
}
__device__ __forceinline__ f32x4 as_f32x4(const u32x2& v) {   // (register type of a loaded piece: two dwords used, two idle)
    return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), 0.f, 0.f};
}
__device__ __forceinline__ u32x2 as_u32x2(const f32x4& v) { return u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}; }

// ------------------------------------------------------------------------------------- pool
template <int HD, bool F16, bool IO16 = false>
__global__ __launch_bounds__(256) void pool_attn_bwd_x3_kernel(const float* __restrict__ KV, const float* __restrict__ Qind,
                                                               const float* __restrict__ Omerged, const float* __restrict__ lse,
                                                               const float* __restrict__ dO, float* __restrict__ dKV,
                                                               float* __restrict__ dQpart, int B, int N, int C, int H, int nsplit) {
    static_assert(!IO16 || F16, "fp16 tensors carry the fp16 arithmetic");
    constexpr int DT = (HD + 31) / 32, CH = HD / 4, LD_IT = (32 * CH + 63) / 64, NC = HD / 16;
    constexpr int P64 = 16 * DT * 128;   // elements of a 64-row plane with DT column blocks
    constexpr int P32 = 8 * DT * 128;    // 32-row plane
    constexpr int PT = 16 * 128;         // the P / dS plane: 64 rows (queries) x 32 columns (keys)
    // head dims 48 / 64 (DT = 2): the P / dS planes lie over the wave's V planes — V is dead once dP^T is formed, the planes
    // have the same size, and the next tile's V is stored after the last read of T.  96.5 KB instead of 128.5 KB per block: a
    // 64 KB block of the weight-gradient kernel (side stream) fits next to it on the CU
    constexpr bool T_OVER_V = DT == 2;
    static_assert(!T_OVER_V || PT == P32, "the aliased planes have one size");
    // F16: one plane per operand — the lo planes do not exist (their pointers alias the hi planes: never written, their fragment reads are
    // dead code): half the LDS, two to three blocks per CU instead of one (round 6: the kernel is latency-bound at one wave per SIMD)
    constexpr int NPL = F16 ? 1 : 2;
    constexpr int WAVE_E = NPL * (2 * P32 + (T_OVER_V ? 0 : PT));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16* lds = reinterpret_cast<u16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
    const int b = bh / H, hh = bh % H;

    u16* Qhi = lds;                 // inducer queries of this head (raw), [64][DT blocks]
    u16* Qlo = F16 ? Qhi : Qhi + P64;
    u16* Ghi = Qlo + P64;           // dO rows of (b, head)
    u16* Glo = F16 ? Ghi : Ghi + P64;
    float* Ls = reinterpret_cast<float*>(Glo + P64);   // [64] lse2, [64] D
    u16* Khi = reinterpret_cast<u16*>(Ls + 128) + wave * WAVE_E;
    u16* Klo = F16 ? Khi : Khi + P32;
    u16* Vhi = Klo + P32;
    u16* Vlo = F16 ? Vhi : Vhi + P32;
    u16* Thi = T_OVER_V ? Vhi : Vlo + P32;   // P, then dS, as [query][key]
    u16* Tlo = F16 ? Thi : (T_OVER_V ? Vlo : Thi + PT);

    const int ks = (((N + nsplit - 1) / nsplit) + 31) / 32 * 32;
    const int k_begin = split * ks, k_end = min(N, k_begin + ks);
    const int ntiles = k_end > k_begin ? (k_end - k_begin + 31) / 32 : 0;
    const int nit = (ntiles + 3) / 4;

    // zero everything once: the padding columns of the last block (HD = 16, 48) enter products whose results are dropped,
    // but must not hold NaN patterns a later pass could propagate
    for (int f = tid; f < (int)((2 * NPL * P64 + 4 * WAVE_E) / 2 + 128); f += 256) reinterpret_cast<unsigned*>(lds)[f] = 0u;
    __syncthreads();
    for (int f = tid; f < 64 * CH; f += 256) {
        const int row = f / CH, ch = f % CH;
        put4<DT, F16>(Qhi, Qlo, row, ch * 4, *reinterpret_cast<const f32x4*>(Qind + ((size_t)hh * 64 + row) * HD + ch * 4));
        put4<DT, F16>(Ghi, Glo, row, ch * 4, *reinterpret_cast<const f32x4*>(dO + ((size_t)b * 64 + row) * C + hh * HD + ch * 4));
    }
    if (tid < 64) {
        const float* o = Omerged + ((size_t)b * 64 + tid) * C + hh * HD;
        const float* g = dO + ((size_t)b * 64 + tid) * C + hh * HD;
        float d = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < CH; ++c4) {
            const f32x4 ov = *reinterpret_cast<const f32x4*>(o + c4 * 4), gv = *reinterpret_cast<const f32x4*>(g + c4 * 4);
            d += ov[0] * gv[0] + ov[1] * gv[1] + ov[2] * gv[2] + ov[3] * gv[3];
        }
        Ls[64 + tid] = d;
        Ls[tid] = lse[(size_t)bh * 64 + tid];
    }

    const size_t ldkv = 2 * (size_t)C;
    const float* Kg = KV + (size_t)b * N * ldkv + hh * HD;
    const float* Vg = Kg + C;
    float* dKg = dKV + (size_t)b * N * ldkv + hh * HD;
    float* dVg = dKg + C;

    f32x4 rk[LD_IT], rv[LD_IT];
    auto load_tile = [&](int tile) {
        const int base = k_begin + tile * 32;
#pragma unroll
        for (int it = 0; it < LD_IT; ++it) {
            const int f = it * 64 + lane, row = f / CH, ch = f % CH, key = base + row;
            f32x4 zk = {0.f, 0.f, 0.f, 0.f}, zv = {0.f, 0.f, 0.f, 0.f};
            if (f < 32 * CH && tile < ntiles && key < k_end) {
                if (IO16) {
                    const _Float16* K16 = reinterpret_cast<const _Float16*>(KV) + (size_t)b * N * ldkv + hh * HD;
                    zk = as_f32x4(*reinterpret_cast<const u32x2*>(K16 + key * ldkv + ch * 4));
                    zv = as_f32x4(*reinterpret_cast<const u32x2*>(K16 + C + key * ldkv + ch * 4));
                } else {
                    zk = *reinterpret_cast<const f32x4*>(Kg + key * ldkv + ch * 4);
                    zv = *reinterpret_cast<const f32x4*>(Vg + key * ldkv + ch * 4);
                }
            }
            rk[it] = zk;
            rv[it] = zv;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < LD_IT; ++it) {
            const int f = it * 64 + lane, row = f / CH, ch = f % CH;
            if (f < 32 * CH) {
                if (IO16) {
                    put4h<DT>(Khi, row, ch * 4, as_u32x2(rk[it]));
                    put4h<DT>(Vhi, row, ch * 4, as_u32x2(rv[it]));
                } else {
                    put4<DT, F16>(Khi, Klo, row, ch * 4, rk[it]);
                    put4<DT, F16>(Vhi, Vlo, row, ch * 4, rv[it]);
                }
            }
        }
    };

    f32x16 dQ[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) dQ[dt][j][e] = 0.f;

    const float sc = rsqrtf((float)HD), scale2 = LOG2E * sc;
    load_tile(wave);
    __syncthreads();   // Q, dO planes and Ls complete
    float lsej[2], Dj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        lsej[j] = Ls[32 * j + r];
        Dj[j] = Ls[64 + 32 * j + r];
    }

    for (int it = 0; it < nit; ++it) {
        const int tile = wave + 4 * it;
        store_tile();
        wave_lds_sync();
        load_tile(tile + 4);
        if (tile < ntiles) {
            const int kbase = k_begin + tile * 32;
            f32x16 p[2], dp[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) { p[j][e] = 0.f; dp[j][e] = 0.f; }
            // S^T = K Q^T and dP^T = V dO^T (keys on the row index, queries on the lane)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const u32x4 kh = rowfrag<DT>(Khi, r, c, h), kl = rowfrag<DT>(Klo, r, c, h);
                const u32x4 vh = rowfrag<DT>(Vhi, r, c, h), vl = rowfrag<DT>(Vlo, r, c, h);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    p[j] = mfma3<F16>(kh, kl, rowfrag<DT>(Qhi, 32 * j + r, c, h), rowfrag<DT>(Qlo, 32 * j + r, c, h), p[j]);
                    dp[j] = mfma3<F16>(vh, vl, rowfrag<DT>(Ghi, 32 * j + r, c, h), rowfrag<DT>(Glo, 32 * j + r, c, h), dp[j]);
                }
            }
            // P^T, then dS^T (in dp)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const bool valid = kbase + mfma_row(e, h) < k_end;
                    const float pv = valid ? exp2f(p[j][e] * scale2 - lsej[j]) : 0.f;
                    p[j][e] = pv;
                    dp[j][e] = pv * (dp[j][e] - Dj[j]) * sc;
                }
            // dQ^T[d, i] += sum_key K[key, d] dS^T[key, i]: transposed K fragment x the dS^T registers
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
                u32x4 sh[2], sl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) split_acc8<F16>(dp[j], 8 * sg, sh[j], sl[j]);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const u32x4 ah = trfrag<DT>(Khi, sg, dt, lane), al = trfrag<DT>(Klo, sg, dt, lane);
#pragma unroll
                    for (int j = 0; j < 2; ++j) dQ[dt][j] = mfma3<F16>(ah, al, sh[j], sl[j], dQ[dt][j]);
                }
            }
            // dV[key, d] = sum_i P[i, key] dO[i, d];  dK[key, d] = sum_i dS[i, key] Q[i, d]   (TN products)
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x16& src = pass == 0 ? p[j] : dp[j];
                        const f32x4 v = {src[4 * g4], src[4 * g4 + 1], src[4 * g4 + 2], src[4 * g4 + 3]};
                        put4<1, F16>(Thi, Tlo, 32 * j + r, 8 * g4 + 4 * h, v);   // keys mfma_row(4 g4 .. 4 g4 + 3, h)
                    }
                wave_lds_sync();
                const u16* Bh = pass == 0 ? Ghi : Qhi;
                const u16* Bl = pass == 0 ? Glo : Qlo;
                f32x16 acc[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[dt][e] = 0.f;
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {   // 16-query chunks
                    const u32x4 ah = trfrag<1>(Thi, sg, 0, lane), al = trfrag<1>(Tlo, sg, 0, lane);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        acc[dt] = mfma3<F16>(ah, al, trfrag<DT>(Bh, sg, dt, lane), trfrag<DT>(Bl, sg, dt, lane), acc[dt]);
                }
                float* dst = pass == 0 ? dVg : dKg;
                _Float16* dst16 = reinterpret_cast<_Float16*>(dKV) + (size_t)b * N * ldkv + hh * HD + (pass == 0 ? C : 0);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kbase + mfma_row(e, h), d = dt * 32 + r;
                        if (key < k_end && d < HD) {
                            if (IO16) dst16[key * ldkv + d] = (_Float16)acc[dt][e];
                            else dst[key * ldkv + d] = acc[dt][e];
                        }
                    }
                wave_lds_sync();   // the tile's reads are done before the next pass / tile overwrites it
            }
        }
        wave_lds_sync();
    }
    __syncthreads();

    // ---- sum the four waves' dQ^T in wave order and emit the partial of this (b, head, split)
    float* Dw = smem;   // [4][HD][64]
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int d = dt * 32 + mfma_row(e, h);
                if (d < HD) Dw[(wave * HD + d) * 64 + 32 * j + r] = dQ[dt][j][e];
            }
    __syncthreads();
    float* out = dQpart + (((size_t)b * nsplit + split) * H + hh) * 64 * HD;
    for (int f = tid; f < 64 * HD; f += 256) {
        const int i = f / HD, d = f % HD;
        out[f] = ((Dw[(0 * HD + d) * 64 + i] + Dw[(1 * HD + d) * 64 + i]) + Dw[(2 * HD + d) * 64 + i]) + Dw[(3 * HD + d) * 64 + i];
    }
}

// ----------------------------------------------------------------------------------- unpool
template <int HD, bool F16, bool IO16 = false>
__global__ __launch_bounds__(256) void unpool_attn_bwd_x3_kernel(const float* __restrict__ q, const float* __restrict__ kvh,
                                                                 const float* __restrict__ dO, float* __restrict__ dq,
                                                                 float* __restrict__ dkv_part, int B, int N, int C, int H,
                                                                 int tiles_per_wave, int nchunk) {
    constexpr int DT = (HD + 31) / 32, CH = HD / 4, LD_IT = (32 * CH + 63) / 64, NC = HD / 16, KP = HD + 4;
    constexpr int P64 = 16 * DT * 128, P32 = 8 * DT * 128;
    constexpr int PT = 8 * 2 * 128;      // the P / dS plane: 32 rows (queries) x 64 columns (inducers)
    constexpr int NPL = F16 ? 1 : 2;     // F16: one plane per operand (see the pool kernel): 64 KiB per block instead of 128
    constexpr int WAVE_E = NPL * (PT + 2 * P32);
    // the fp32 staging tile of the dq rows lies over the T planes and the q planes (F16, hd = 64: and the first bytes of the dO plane):
    // all dead by then, and rewritten in full before the next tile reads them
    static_assert(32 * KP * 4 <= WAVE_E * 2, "the dq staging tile fits over the dead T | q | dO planes");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16* lds = reinterpret_cast<u16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int chunk = blockIdx.x % nchunk, bh = blockIdx.x / nchunk;
    const int b = bh / H, hh = bh % H;

    u16* Khi = lds;                 // inducer keys / values of (b, head), [64][DT blocks]
    u16* Klo = F16 ? Khi : Khi + P64;
    u16* Vhi = Klo + P64;
    u16* Vlo = F16 ? Vhi : Vhi + P64;
    u16* Thi = Vlo + P64 + wave * WAVE_E;   // P, then dS, as [query][inducer]
    u16* Tlo = F16 ? Thi : Thi + PT;
    u16* Qhi = Tlo + PT;            // the tile's raw queries
    u16* Qlo = F16 ? Qhi : Qhi + P32;
    u16* Ghi = Qlo + P32;           // the tile's dO rows
    u16* Glo = F16 ? Ghi : Ghi + P32;
    float* St = reinterpret_cast<float*>(Thi);   // [32][KP] fp32 staging of dq rows, over the dead T | q planes

    for (int f = tid; f < (int)((2 * NPL * P64 + 4 * WAVE_E) / 2); f += 256) reinterpret_cast<unsigned*>(lds)[f] = 0u;
    __syncthreads();
    for (int f = tid; f < 64 * CH; f += 256) {
        const int row = f / CH, ch = f % CH;
        const float* src = kvh + ((size_t)b * 64 + row) * 2 * C + hh * HD + ch * 4;
        put4<DT, F16>(Khi, Klo, row, ch * 4, *reinterpret_cast<const f32x4*>(src));
        put4<DT, F16>(Vhi, Vlo, row, ch * 4, *reinterpret_cast<const f32x4*>(src + C));
    }
    const float sc = rsqrtf((float)HD), scale2 = LOG2E * sc;
    const float* qb = q + (size_t)b * N * C + hh * HD;
    const float* gb = dO + (size_t)b * N * C + hh * HD;
    float* dqb = dq + (size_t)b * N * C + hh * HD;

    f32x4 rq[LD_IT], rg[LD_IT];
    auto load_q = [&](int it) {
        const int q0 = (chunk * tiles_per_wave + it) * 128 + wave * 32;
#pragma unroll
        for (int ld = 0; ld < LD_IT; ++ld) {
            const int f = ld * 64 + lane, row = f / CH, ch = f % CH, n = q0 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
            if (f < 32 * CH && it < tiles_per_wave && n < N) {
                if (IO16) {
                    const size_t o = ((size_t)b * N + n) * C + hh * HD + ch * 4;
                    v = as_f32x4(*reinterpret_cast<const u32x2*>(reinterpret_cast<const _Float16*>(q) + o));
                    g = as_f32x4(*reinterpret_cast<const u32x2*>(reinterpret_cast<const _Float16*>(dO) + o));
                } else {
                    v = *reinterpret_cast<const f32x4*>(qb + (size_t)n * C + ch * 4);
                    g = *reinterpret_cast<const f32x4*>(gb + (size_t)n * C + ch * 4);
                }
            }
            rq[ld] = v;
            rg[ld] = g;
        }
    };

    f32x16 dk[2][DT], dv[2][DT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dk[rt][dt][e] = 0.f; dv[rt][dt][e] = 0.f; }

    load_q(0);
    for (int it = 0; it < tiles_per_wave; ++it) {
        const int q0 = (chunk * tiles_per_wave + it) * 128 + wave * 32;
#pragma unroll
        for (int ld = 0; ld < LD_IT; ++ld) {
            const int f = ld * 64 + lane, row = f / CH, ch = f % CH;
            if (f < 32 * CH) {
                if (IO16) {
                    put4h<DT>(Qhi, row, ch * 4, as_u32x2(rq[ld]));
                    put4h<DT>(Ghi, row, ch * 4, as_u32x2(rg[ld]));
                } else {
                    put4<DT, F16>(Qhi, Qlo, row, ch * 4, rq[ld]);
                    put4<DT, F16>(Ghi, Glo, row, ch * 4, rg[ld]);
                }
            }
        }
        load_q(it + 1);
        if (it == 0) __syncthreads();   // K, V planes complete
        wave_lds_sync();
        if (q0 < N) {
            f32x16 p[2], dp[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { p[rt][e] = 0.f; dp[rt][e] = 0.f; }
            // S^T = K q^T, dP^T = V dO^T (inducers on the row index, the tile's queries on the lane)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const u32x4 qh = rowfrag<DT>(Qhi, r, c, h), ql = rowfrag<DT>(Qlo, r, c, h);
                const u32x4 gh = rowfrag<DT>(Ghi, r, c, h), gl = rowfrag<DT>(Glo, r, c, h);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    p[rt] = mfma3<F16>(rowfrag<DT>(Khi, 32 * rt + r, c, h), rowfrag<DT>(Klo, 32 * rt + r, c, h), qh, ql, p[rt]);
                    dp[rt] = mfma3<F16>(rowfrag<DT>(Vhi, 32 * rt + r, c, h), rowfrag<DT>(Vlo, 32 * rt + r, c, h), gh, gl, dp[rt]);
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { p[rt][e] *= scale2; mx = fmaxf(mx, p[rt][e]); }
            mx = fmaxf(mx, xor32(mx));
            float ls = 0.f;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { p[rt][e] = exp2f(p[rt][e] - mx); ls += p[rt][e]; }
            ls += xor32(ls);
            const float inv = 1.0f / ls;
            float Dn = 0.f;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { p[rt][e] *= inv; Dn += p[rt][e] * dp[rt][e]; }
            Dn += xor32(Dn);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) dp[rt][e] = p[rt][e] * (dp[rt][e] - Dn) * sc;   // dS^T
            // dq^T[d, n] = sum_i k[i, d] dS^T[i, n]: transposed k fragment x the dS^T registers
            f32x16 O[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) O[dt][e] = 0.f;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int sg = 0; sg < 2; ++sg) {
                    u32x4 sh, sl;
                    split_acc8<F16>(dp[rt], 8 * sg, sh, sl);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        O[dt] = mfma3<F16>(trfrag<DT>(Khi, 2 * rt + sg, dt, lane), trfrag<DT>(Klo, 2 * rt + sg, dt, lane), sh, sl, O[dt]);
                }
            // dv[i, d] += sum_n P[n, i] dO[n, d];  dk[i, d] += sum_n dS[n, i] q[n, d]   (TN products)
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x16& src = pass == 0 ? p[rt] : dp[rt];
                        const f32x4 v = {src[4 * g4], src[4 * g4 + 1], src[4 * g4 + 2], src[4 * g4 + 3]};
                        put4<2, F16>(Thi, Tlo, r, 32 * rt + 8 * g4 + 4 * h, v);   // inducers 32 rt + mfma_row(4 g4 .. + 3, h)
                    }
                wave_lds_sync();
                const u16* Bh = pass == 0 ? Ghi : Qhi;
                const u16* Bl = pass == 0 ? Glo : Qlo;
#pragma unroll
                for (int sg = 0; sg < 2; ++sg) {   // 16-query chunks of the tile
                    u32x4 bh_[DT], bl_[DT];
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        bh_[dt] = trfrag<DT>(Bh, sg, dt, lane);
                        bl_[dt] = trfrag<DT>(Bl, sg, dt, lane);
                    }
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        const u32x4 ah = trfrag<2>(Thi, sg, rt, lane), al = trfrag<2>(Tlo, sg, rt, lane);
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            if (pass == 0) dv[rt][dt] = mfma3<F16>(ah, al, bh_[dt], bl_[dt], dv[rt][dt]);
                            else dk[rt][dt] = mfma3<F16>(ah, al, bh_[dt], bl_[dt], dk[rt][dt]);
                        }
                    }
                }
                wave_lds_sync();
            }
            // dq rows: transpose dq^T (query on the lane) through LDS, then coalesced row stores
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d = dt * 32 + 8 * g4 + 4 * h;
                    if (d < HD) {
                        f32x4 v = {O[dt][4 * g4], O[dt][4 * g4 + 1], O[dt][4 * g4 + 2], O[dt][4 * g4 + 3]};
                        *reinterpret_cast<f32x4*>(St + r * KP + d) = v;
                    }
                }
            wave_lds_sync();
#pragma unroll
            for (int ld = 0; ld < LD_IT; ++ld) {
                const int f = ld * 64 + lane, row = f / CH, ch = f % CH, n = q0 + row;
                if (f < 32 * CH && n < N) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(St + row * KP + ch * 4);
                    if (IO16) {
                        const f16x4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                        *reinterpret_cast<u32x2*>(reinterpret_cast<_Float16*>(dq) + ((size_t)b * N + n) * C + hh * HD + ch * 4) = __builtin_bit_cast(u32x2, hv);
                    } else {
                        *reinterpret_cast<f32x4*>(dqb + (size_t)n * C + ch * 4) = v;
                    }
                }
            }
            wave_lds_sync();
        }
        wave_lds_sync();
    }
    __syncthreads();

    // ---- sum the four waves' dk | dv in a fixed order, (w0 + w2) + (w1 + w3): partial of this (chunk, b, head).  Two rounds through
    // TWO wave slots (round 6: half the LDS of four slots — with one plane per operand it was what kept a second block off the CU)
    float* Dw = smem;   // [2 slots][2 (k, v)][64][HD]
    auto slot_io = [&](int slot, bool add, bool store) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = rt * 32 + mfma_row(e, h), d = dt * 32 + r;
                    if (d < HD) {
                        float* pk = Dw + ((slot * 2 + 0) * 64 + i) * HD + d;
                        float* pv = Dw + ((slot * 2 + 1) * 64 + i) * HD + d;
                        if (add) { dk[rt][dt][e] += *pk; dv[rt][dt][e] += *pv; }
                        if (store) { *pk = dk[rt][dt][e]; *pv = dv[rt][dt][e]; }
                    }
                }
    };
    if (wave >= 2) slot_io(wave - 2, false, true);
    __syncthreads();
    if (wave < 2) slot_io(wave, true, true);     // a lane adds what the same lane of wave + 2 stored, then stores the sum
    __syncthreads();
    float* out = dkv_part + ((size_t)chunk * B + b) * 64 * 2 * C + hh * HD;
    for (int f = tid; f < 2 * 64 * HD; f += 256) {
        const int kv = f / (64 * HD), i = (f / HD) % 64, d = f % HD;
        const int o = (kv * 64 + i) * HD + d;
        out[(size_t)i * 2 * C + kv * C + d] = Dw[o] + Dw[o + 2 * 64 * HD];
    }
}

template <int HD, bool F16, bool IO16 = false>
int pool_bwd_x3_t(const float* KV, const float* ind, const float* O, const float* lse, const float* dO, float* dKV, float* dQp,
                  int B, int N, int C, int H, int nsplit, hipStream_t st) {
    constexpr int DT = (HD + 31) / 32, NPL = F16 ? 1 : 2;
    const size_t a = (size_t)NPL * (2 * 16 * DT * 128 + 4 * (2 * 8 * DT * 128 + (DT == 2 ? 0 : 16 * 128))) * 2 + 128 * 4, c = (size_t)4 * HD * 64 * 4;
    const size_t lds = a > c ? a : c;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pool_attn_bwd_x3_kernel<HD, F16, IO16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((pool_attn_bwd_x3_kernel<HD, F16, IO16>), dim3(B * H * nsplit), dim3(256), lds, st, KV, ind, O, lse, dO, dKV, dQp, B, N, C, H, nsplit);
    return (int)hipGetLastError();
}

template <int HD, bool F16, bool IO16 = false>
int unpool_bwd_x3_t(const float* q, const float* kvh, const float* dO, float* dq, float* part, int B, int N, int C, int H, int tpw,
                    int nchunk, hipStream_t st) {
    constexpr int DT = (HD + 31) / 32, NPL = F16 ? 1 : 2;
    const size_t a = (size_t)NPL * (2 * 16 * DT * 128 + 4 * (8 * 2 * 128 + 2 * 8 * DT * 128)) * 2, c = (size_t)4 * 64 * HD * 4;
    const size_t lds = a > c ? a : c;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(unpool_attn_bwd_x3_kernel<HD, F16, IO16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((unpool_attn_bwd_x3_kernel<HD, F16, IO16>), dim3(B * H * nchunk), dim3(256), lds, st, q, kvh, dO, dq, part, B, N, C, H, tpw, nchunk);
    return (int)hipGetLastError();
}

}  // namespace

bool attn_bwd_x3_supported(int HD) { return HD == 16 || HD == 32 || HD == 48 || HD == 64; }

int pool_attn_bwd_x3_launch(const float* KV, const float* inducers, const float* merged, const float* lse, const float* dO,
                            float* dKV, float* dQpart, int B, int N, int C, int H, int nsplit, hipStream_t st, int f16) {
#define POOL_BWD(HD_)                                                                                              \
    return f16 == 2 ? pool_bwd_x3_t<HD_, true, true>(KV, inducers, merged, lse, dO, dKV, dQpart, B, N, C, H, nsplit, st) \
           : f16 ? pool_bwd_x3_t<HD_, true>(KV, inducers, merged, lse, dO, dKV, dQpart, B, N, C, H, nsplit, st)      \
               : pool_bwd_x3_t<HD_, false>(KV, inducers, merged, lse, dO, dKV, dQpart, B, N, C, H, nsplit, st)
    switch (C / H) {
        case 16: POOL_BWD(16);
        case 32: POOL_BWD(32);
        case 48: POOL_BWD(48);
        case 64: POOL_BWD(64);
        default: return -4;
    }
#undef POOL_BWD
}

int unpool_attn_bwd_x3_launch(const float* q, const float* kvh, const float* dO, float* dq, float* dkv_part, int B, int N, int C,
                              int H, int tpw, int nchunk, hipStream_t st, int f16) {
#define UNPOOL_BWD(HD_)                                                                                    \
    return f16 == 2 ? unpool_bwd_x3_t<HD_, true, true>(q, kvh, dO, dq, dkv_part, B, N, C, H, tpw, nchunk, st) \
           : f16 ? unpool_bwd_x3_t<HD_, true>(q, kvh, dO, dq, dkv_part, B, N, C, H, tpw, nchunk, st)        \
               : unpool_bwd_x3_t<HD_, false>(q, kvh, dO, dq, dkv_part, B, N, C, H, tpw, nchunk, st)
    switch (C / H) {
        case 16: UNPOOL_BWD(16);
        case 32: UNPOOL_BWD(32);
        case 48: UNPOOL_BWD(48);
        case 64: UNPOOL_BWD(64);
        default: return -4;
    }
#undef UNPOOL_BWD
}
