// Weight gradients of the training path in the reference's autocast(float16) trainer arithmetic, gfx950:
//
//     C[g, n, k] = sum over the samples z of group g, sum over rows m:  fp16(A[z, m, n]) * fp16(B'[z, m, k])        (fp32 accumulation)
//
// A = dY (Z, R, N), B = X (Z, R, K), both row-major; B' = B * pro_a[z] + pro_o[z] (the AdaGN apply of a linear whose input was
// AdaGN(x)) or B itself, fp32 or — B16 — an fp16 tensor already (the hidden layer gecco_linear_act_keep_h16 stored that way).
// Reference: autograd of every nn.Linear under Lightning's precision="16-mixed" (example_configs/*.py; diffusion.py:213-222), where
// torch forms dW as an fp16 matmul; here the partials and their fixed-order sum stay fp32 (bit-reproducible).
//
// Same scheme as gemm_tn_x3.hip — 32-row slabs of both operands converted while they are staged, one fp16 plane each in 4-row x
// 32-column blocks, BOTH MFMA fragments ds_read_b64_tr_b16 transposed reads, two LDS stages, the next slab in registers — with the
// block tile as a template parameter.  With ONE matrix instruction per product this kernel is bound by moving its operand slabs
// global -> registers -> LDS (tools/probe/tn_probe.hip, profiles/r04p: removing the loads 185 -> 82 us, the matrix instructions
// 185 -> 161), and every output tile re-reads its slabs: the 128 x 128 tile moves 32 wave-loads per 32 matrix instructions, a
// 256 x 128 / 128 x 256 tile (2 x 2 waves of 128 x 64 / 64 x 128) 48 per 64, and the 768 x 384 / 384 x 768 gradients of the shipped
// model become 9 tiles per sample instead of 18 — 432 blocks, one round on the chip's 512 slots instead of 1.7 (opt-in: tn_f16_shape).
// Slab loads are raw buffer loads (one per-lane offset per operand + scalar offsets), the AdaGN coefficients sit in LDS.
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

__device__ __forceinline__ u32x2 cvt4(const f32x4& x) {
    f16x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (_Float16)x[e];
    return __builtin_bit_cast(u32x2, v);
}

// element offset of (row, col) in a [32][32 NCB] plane: 4-row x 32-column blocks of 128 elements (256 B = all 64 banks); the row's
// 64-byte slot inside its block rotates with the column block (gemm_tn_x3.hip: the column blocks one row of a store instruction
// touches land on different bank groups); a transposed read still covers whole blocks
template <int NCB>
__device__ __forceinline__ int toff(int row, int col) {
    return ((row >> 2) * NCB + (col >> 5)) * 128 + ((row + (col >> 5)) & 3) * 32 + (col & 31);
}

// WNT x WKT: 32 x 32 accumulator tiles per wave (2 x 2 waves): block tile 64 WNT x 64 WKT.  PRO: AdaGN apply on B.  B16: fp16 B tensor.
// A16: fp16 A tensor (TnArgs::a_f16): half the loads, no conversion; the bias gradient's column sums are formed from the halves.
template <int WNT, int WKT, bool PRO, bool B16, bool A16 = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_f16_kernel(TnArgs g) {
    static_assert(!(PRO && B16), "the AdaGN apply reads the fp32 operand");
    constexpr int TN = 64 * WNT, TK = 64 * WKT, NCA = TN / 32, NCB = TK / 32;
    constexpr int PA = 32 * TN, PB = 32 * TK;                 // u16 per plane
    constexpr int LA = TN / 32, LB = TK / 32, LB8 = TK / 64;  // 16-byte loads per thread and slab: A, B (fp32), B (fp16: 8 halves each)
    constexpr int LA8 = TN / 64;                              // A (fp16)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16* lds = reinterpret_cast<u16*>(smem);                  // [2 stages][A | B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int tilesK = (g.K + TK - 1) / TK;
    // the output tiles of one sample group read the same rows of dY / X: the XCD-aware virtual grid puts them on ONE XCD
    const int ntile = (int)gridDim.x;
    const int vb = g.xcd ? xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y)) : (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int bx = vb % ntile, by = vb / ntile;
    const int n0 = (bx / tilesK) * TN, k0 = (bx % tilesK) * TK;
    // a thread's columns are the same in every slab: one predicate per operand zero-fills what lies beyond the matrix
    const int ca = A16 ? tid % (TN / 8) : tid % (TN / 4), cb = B16 ? tid % (TK / 8) : tid % (TK / 4);
    const bool aok = n0 + ca * (A16 ? 8 : 4) < g.N, bok = k0 + cb * (B16 ? 8 : 4) < g.K;
    const int z0 = by * g.group, z1 = min(g.Z, z0 + g.group);
    const int msteps = g.R / 32, nsteps = (z1 - z0) * msteps;

    // slab loads as raw buffer loads: ONE per-lane byte offset per operand (out-of-range for the lanes beyond the matrix: they read
    // zeros) and scalar offsets for the sample / slab / row group — with per-load 64-bit addresses the 12 loads of a wide tile cost
    // 24 address registers next to 128 accumulator and 48 staging registers, and the kernel spilled
    f32x4 ra[A16 ? 1 : LA], rb[B16 ? 1 : LB];
    u32x4 rb8[B16 ? LB8 : 1], ra8[A16 ? LA8 : 1];
    // PRO: the AdaGN coefficients of the block's K columns live in LDS ([2 sample parities][pa | po][TK]) and are read at the
    // store — as registers (8 per thread, live across the matrix phase) they were what made the wide tiles spill
    float* ptab = smem + (size_t)2 * (PA + PB) / 2;
    const unsigned va = !aok ? 0x7fffffffu : A16 ? (unsigned)(((tid / (TN / 8)) * g.lda + ca * 8) * 2) : (unsigned)(((tid / (TN / 4)) * g.lda + ca * 4) * 4);
    const unsigned vbo = !bok ? 0x7fffffffu : B16 ? (unsigned)(((tid / (TK / 8)) * g.ldb + cb * 8) * 2) : (unsigned)(((tid / (TK / 4)) * g.ldb + cb * 4) * 4);
    auto load = [&](int s) {
        const int z = z0 + s / msteps, m0 = (s % msteps) * 32;
        if (PRO && s % msteps == 0 && tid < TK / 4) {   // first slab of a sample: its coefficients into the table of its parity
            const bool ok = k0 + tid * 4 < g.K;
            float* t = ptab + ((s / msteps) & 1) * 2 * TK;
            reinterpret_cast<f32x4*>(t)[tid] = ok ? *reinterpret_cast<const f32x4*>(g.pro_a + (size_t)z * g.K + k0 + tid * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
            reinterpret_cast<f32x4*>(t + TK)[tid] = ok ? *reinterpret_cast<const f32x4*>(g.pro_o + (size_t)z * g.K + k0 + tid * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (A16) {
            const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<_Float16*>(reinterpret_cast<const _Float16*>(g.A) + (size_t)z * g.sA), 0, 0x7fffffff, 0x00020000);
            const unsigned sa0 = (unsigned)((m0 * g.lda + n0) * 2);
#pragma unroll
            for (int i = 0; i < LA8; ++i) ra8[i] = __builtin_amdgcn_raw_buffer_load_b128(ars, va, sa0 + (unsigned)(i * (2048 / TN) * g.lda * 2), 0);
        } else {
            const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A + (size_t)z * g.sA), 0, 0x7fffffff, 0x00020000);
            const unsigned sa0 = (unsigned)((m0 * g.lda + n0) * 4);
#pragma unroll
            for (int i = 0; i < LA; ++i)
                ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, va, sa0 + (unsigned)(i * (1024 / TN) * g.lda * 4), 0));
        }
        if (B16) {
            const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<_Float16*>(reinterpret_cast<const _Float16*>(g.Bm) + (size_t)z * g.sB), 0, 0x7fffffff, 0x00020000);
            const unsigned sb0 = (unsigned)((m0 * g.ldb + k0) * 2);
#pragma unroll
            for (int i = 0; i < LB8; ++i) rb8[i] = __builtin_amdgcn_raw_buffer_load_b128(brs, vbo, sb0 + (unsigned)(i * (2048 / TK) * g.ldb * 2), 0);
        } else {
            const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.Bm + (size_t)z * g.sB), 0, 0x7fffffff, 0x00020000);
            const unsigned sb0 = (unsigned)((m0 * g.ldb + k0) * 4);
#pragma unroll
            for (int i = 0; i < LB; ++i)
                rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, vbo, sb0 + (unsigned)(i * (1024 / TK) * g.ldb * 4), 0));
        }
    };
    // bias gradient = column sums of dY: the blocks of the first K tile add up the rows they stage anyway
    const bool want_cs = g.colsum != nullptr && k0 == 0;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f}, cs2 = {0.f, 0.f, 0.f, 0.f};   // (A16: a thread's 8 columns)
    // LDS offsets of a thread's pieces: rows row0 + i RPI with RPI a multiple of 4 — whole 4-row groups apart, the same slot: base + constant
    constexpr int RA = A16 ? 2048 / TN : 1024 / TN, RB = B16 ? 2048 / TK : 1024 / TK;
    static_assert(RA % 4 == 0 && RB % 4 == 0, "pieces of a thread lie whole row groups apart");
    const int oa = A16 ? toff<NCA>(tid / (TN / 8), ca * 8) : toff<NCA>(tid / (TN / 4), ca * 4);
    const int ob = B16 ? toff<NCB>(tid / (TK / 8), cb * 8) : toff<NCB>(tid / (TK / 4), cb * 4);
    auto store = [&](int stage, int s) {
        u16* sa = lds + stage * (PA + PB) + oa;
        u16* sb = lds + stage * (PA + PB) + PA + ob;
        f32x4 pa4, po4;
        if (PRO) {
            const float* t = ptab + ((s / msteps) & 1) * 2 * TK;
            pa4 = reinterpret_cast<const f32x4*>(t)[cb];
            po4 = reinterpret_cast<const f32x4*>(t + TK)[cb];
        }
        if (A16) {
#pragma unroll
            for (int i = 0; i < LA8; ++i) {
                if (want_cs) {
                    const f16x8 hv = __builtin_bit_cast(f16x8, ra8[i]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        cs[e] += (float)hv[e];
                        cs2[e] += (float)hv[4 + e];
                    }
                }
                *reinterpret_cast<u32x4*>(sa + i * (RA / 4) * NCA * 128) = ra8[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                if (want_cs) cs += ra[i];
                *reinterpret_cast<u32x2*>(sa + i * (RA / 4) * NCA * 128) = cvt4(ra[i]);
            }
        }
        if (B16) {
#pragma unroll
            for (int i = 0; i < LB8; ++i) *reinterpret_cast<u32x4*>(sb + i * (RB / 4) * NCB * 128) = rb8[i];
        } else {
#pragma unroll
            for (int i = 0; i < LB; ++i)   // the AdaGN apply here — where the loaded values are consumed a step after their loads were issued
                *reinterpret_cast<u32x2*>(sb + i * (RB / 4) * NCB * 128) = cvt4(PRO ? rb[i] * pa4 + po4 : rb[i]);
        }
    };
    // transposed-read addressing (attention_x3.hip): lane 4q+p of a 16-lane group points at row q, columns 4p .. 4p+3
    const int tq = (lane & 15) >> 2, tp = lane & 3, tcol = 16 * ((lane >> 4) & 1) + 4 * tp;
    auto frag = [&](const u16* plane, auto NC, int sg, int blk) -> f16x8 {
        constexpr int nc = decltype(NC)::value;
        typedef __attribute__((address_space(3))) s16x4* lp;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(plane + toff<nc>(16 * sg + 4 * h + tq, blk * 32 + tcol)));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(plane + toff<nc>(16 * sg + 8 + 4 * h + tq, blk * 32 + tcol)));
        return __builtin_bit_cast(f16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x16 acc[WNT][WKT];
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
        for (int j = 0; j < WKT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (nsteps > 0) load(0);
    for (int s = 0; s < nsteps; ++s) {
        const int stage = s & 1;
        // (block-uniform) the first slab of a sample: its coefficient table was written by load(s) an iteration ago, after that
        // iteration's barrier — order it before the reads below
        if (PRO && s % msteps == 0) __syncthreads();
#ifdef TN_DIAG_NOSTORE   // diagnostic builds (tools/probe/tn_probe.hip): one ingredient removed each; results are garbage, only the time matters
        if (s < 2) store(stage, s);
        else {
#pragma unroll
            for (int i = 0; i < LA; ++i) asm volatile("" ::"v"(ra[i]));
        }
#else
        store(stage, s);
#endif
        __syncthreads();   // stage complete; every wave is past its reads of the other stage's previous contents
#ifdef TN_DIAG_NOLOAD
        if (s == 0) load(1 < nsteps ? 1 : 0);
#else
        if (s + 1 < nsteps) load(s + 1);
#endif
        const u16* sa = lds + stage * (PA + PB);
        const u16* sb = sa + PA;
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            // the fragments of the narrow side are held, those of the wide side read one at a time (4 registers instead of 16)
            if constexpr (WNT >= WKT) {
                f16x8 bb[WKT];
#pragma unroll
                for (int j = 0; j < WKT; ++j) bb[j] = frag(sb, std::integral_constant<int, NCB>{}, sg, wk * WKT + j);
#pragma unroll
                for (int i = 0; i < WNT; ++i) {
                    const f16x8 a = frag(sa, std::integral_constant<int, NCA>{}, sg, wn * WNT + i);
#pragma unroll
                    for (int j = 0; j < WKT; ++j)
#ifdef TN_DIAG_NOMFMA
                        acc[i][j][0] += (float)a[0] + (float)bb[j][0];
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb[j], acc[i][j], 0, 0, 0);
#endif
                }
            } else {
                f16x8 a[WNT];
#pragma unroll
                for (int i = 0; i < WNT; ++i) a[i] = frag(sa, std::integral_constant<int, NCA>{}, sg, wn * WNT + i);
#pragma unroll
                for (int j = 0; j < WKT; ++j) {
                    const f16x8 bj = frag(sb, std::integral_constant<int, NCB>{}, sg, wk * WKT + j);
#pragma unroll
                    for (int i = 0; i < WNT; ++i)
#ifdef TN_DIAG_NOMFMA
                        acc[i][j][0] += (float)a[i][0] + (float)bj[0];
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], bj, acc[i][j], 0, 0, 0);
#endif
                }
            }
        }
    }
    if (want_cs) {   // (block-uniform) the threads of a column group, summed in thread order
        constexpr int NGRP = A16 ? TN / 8 : TN / 4, PER = 256 / NGRP;
        __syncthreads();
        f32x4* red = reinterpret_cast<f32x4*>(smem);
        if (A16) {
            red[2 * tid] = cs;
            red[2 * tid + 1] = cs2;
        } else {
            red[tid] = cs;
        }
        __syncthreads();
        if (A16) {
            if (tid < 2 * NGRP && n0 + tid * 4 < g.N) {   // thread t: columns 4 t .. 4 t + 3 = half (t & 1) of column group t >> 1
                f32x4 sum = red[tid];
#pragma unroll
                for (int j = 1; j < PER; ++j) sum += red[tid + 2 * NGRP * j];
                *reinterpret_cast<f32x4*>(g.colsum + (size_t)by * g.N + n0 + tid * 4) = sum;
            }
        } else if (tid < NGRP && n0 + tid * 4 < g.N) {
            f32x4 sum = red[tid];
#pragma unroll
            for (int j = 1; j < PER; ++j) sum += red[tid + NGRP * j];
            *reinterpret_cast<f32x4*>(g.colsum + (size_t)by * g.N + n0 + tid * 4) = sum;
        }
    }
    float* Cb = g.C + (size_t)by * g.N * g.K;
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
        for (int j = 0; j < WKT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + (wn * WNT + i) * 32 + mfma_row(e, h), k = k0 + (wk * WKT + j) * 32 + r;
                if (n < g.N && k < g.K) Cb[(size_t)n * g.K + k] = acc[i][j][e];
            }
    if (g.counters == nullptr) return;
    // ---- the fixed-order sum of the tile's group partials, by the LAST group block to finish it (no reduce_batch launch): every block
    // publishes its partial (release fence), takes a ticket, and the holder of the last one reads all G partials behind an acquire
    // fence and adds them in group order — exactly reduce_batch_kernel's 0 + p0 + p1 + ..: bit-identical, run-to-run deterministic
    const int G = (int)gridDim.y;
    __threadfence();
    __syncthreads();
    unsigned* tick = reinterpret_cast<unsigned*>(smem);
    if (tid == 0) *tick = atomicAdd(g.counters + bx, 1u);
    __syncthreads();
    if (*tick != (unsigned)(G - 1)) return;
    __threadfence();
    const size_t NK = (size_t)g.N * g.K;
    for (int e = tid; e < TN * (TK / 4); e += 256) {
        const int n = n0 + e / (TK / 4), k = k0 + (e % (TK / 4)) * 4;
        if (n >= g.N || k >= g.K) continue;     // (K % 4 == 0: a piece is whole or absent)
        const float* src = g.C + (size_t)n * g.K + k;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 8 <= G; z += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)(z + u) * NK));
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; z < G; ++z) sum += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)z * NK));
        *reinterpret_cast<f32x4*>(g.out + (size_t)n * g.K + k) = sum;
    }
    if (g.colsum != nullptr && g.colsum_out != nullptr && k0 == 0) {
        for (int e = tid; e < TN / 4; e += 256) {
            const int n = n0 + e * 4;
            if (n >= g.N) continue;
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
            for (int z = 0; z < G; ++z) sum += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.colsum + (size_t)z * g.N + n));
            *reinterpret_cast<f32x4*>(g.colsum_out + n) = sum;
        }
    }
    if (tid == 0) g.counters[bx] = 0u;   // the slot is zero again for whoever takes it next
}

// ---------------------------------------------------------------------------------------------------------------------
// BOTH operands fp16 tensors, whole tiles (round 6: dY = the fp16 gradient of an `_io16_ok` / `_du16_ok` product, X = the fp16 operand its
// forward kernel stored, GemmArgs::y16_out): the 32-row slabs go global -> LDS by DMA (buffer_load ... lds, 16 bytes per lane) straight
// into the planes' 4-row x 32-column block layout — the LDS side of a DMA instruction is linear (lane l at byte 16 l of a 1 KiB piece =
// four blocks), the GLOBAL side is per lane, so each lane fetches the 8 halves its slot of the layout holds.  No register staging, no
// conversion, no ds_write: the path that bound the register-staged form.  One barrier per slab; the next slab's DMA flies under this
// slab's matrix instructions.  The bias gradient (column sums of dY) is one more matrix instruction per A fragment against a fragment
// of ones (exact products, fp32 accumulation), by the waves of the first K tile's first wave column.
// (a non-template device function: inside the kernel template the host pass would have to accept the 16-byte form of the builtin, which
// only the gfx950 target has, and drops the whole instantiation without a word)
__device__ __forceinline__ void tn_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, u16* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

template <int WNT, int WKT>
__global__ __launch_bounds__(256, 2) void gemm_tn_f16_dma_kernel(TnArgs g) {
    constexpr int TN = 64 * WNT, TK = 64 * WKT, NCA = TN / 32, NCB = TK / 32;
    constexpr int PA = 32 * TN, PB = 32 * TK;                 // u16 per plane
    constexpr int NPA = PA / 512, NPB = PB / 512, NPW = (NPA + NPB) / 4;   // 1 KiB pieces per plane, per wave
    static_assert((NPA + NPB) % 4 == 0 && NPA % NPW == 0, "a wave's pieces lie in one plane");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16* lds = reinterpret_cast<u16*>(smem);                  // [2 stages][A | B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int tilesK = g.K / TK;
    const int ntile = (int)gridDim.x;
    const int vb = g.xcd ? xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y)) : (int)(blockIdx.y * gridDim.x + blockIdx.x);
    const int bx = vb % ntile, by = vb / ntile;
    const int n0 = (bx / tilesK) * TN, k0 = (bx % tilesK) * TK;
    const int z0 = by * g.group, z1 = min(g.Z, z0 + g.group);
    const int msteps = g.R / 32, nsteps = (z1 - z0) * msteps;

    // this wave's pieces p = NPW wave .. + NPW - 1 of the list [A pieces | B pieces]; piece q of a plane with NC column blocks: block
    // 4 q + (lane >> 4) = (row group rg, column block cb), slot (lane & 15) >> 2 = (row + cb) & 3, 16-byte chunk lane & 3
    const bool isB = NPW * wave >= NPA;
    const int q0 = isB ? NPW * wave - NPA : NPW * wave;
    unsigned voff[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int bi = 4 * (q0 + i) + (lane >> 4), nc = isB ? NCB : NCA;
        const int rg = bi / nc, cb = bi % nc, slot = (lane & 15) >> 2, ch = lane & 3;
        const int row = 4 * rg + ((slot - cb) & 3), col = cb * 32 + ch * 8;
        voff[i] = (unsigned)((row * (isB ? g.ldb : g.lda) + col) * 2);
    }
    auto issue = [&](int s) {
        const int z = z0 + s / msteps, m0 = (s % msteps) * 32;
        const _Float16* base = isB ? reinterpret_cast<const _Float16*>(g.Bm) + (size_t)z * g.sB : reinterpret_cast<const _Float16*>(g.A) + (size_t)z * g.sA;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(base), 0, 0x7fffffff, 0x00020000);
        const unsigned so = (unsigned)((m0 * (isB ? g.ldb : g.lda) + (isB ? k0 : n0)) * 2);
        u16* dst = lds + (s & 1) * (PA + PB) + (isB ? PA : 0) + q0 * 512;
#pragma unroll
        for (int i = 0; i < NPW; ++i)
            tn_dma16(rs, voff[i], so, dst + i * 512);
    };
    const int tq = (lane & 15) >> 2, tp = lane & 3, tcol = 16 * ((lane >> 4) & 1) + 4 * tp;
    auto frag = [&](const u16* plane, auto NC, int sg, int blk) -> f16x8 {
        constexpr int nc = decltype(NC)::value;
        typedef __attribute__((address_space(3))) s16x4* lp;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(plane + toff<nc>(16 * sg + 4 * h + tq, blk * 32 + tcol)));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(plane + toff<nc>(16 * sg + 8 + 4 * h + tq, blk * 32 + tcol)));
        return __builtin_bit_cast(f16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    };

    f32x16 acc[WNT][WKT], accs[WNT];
#pragma unroll
    for (int i = 0; i < WNT; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) accs[i][e] = 0.f;
#pragma unroll
        for (int j = 0; j < WKT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    const bool want_cs = g.colsum != nullptr && k0 == 0 && wk == 0;   // (wave-uniform)
    const f16x8 ones = {(_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f, (_Float16)1.f};

    if (nsteps > 0) issue(0);
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of slab s have landed
        __syncthreads();   // everybody's pieces landed; everybody is past its reads of the other stage
        if (s + 1 < nsteps) issue(s + 1);
        const u16* sa = lds + (s & 1) * (PA + PB);
        const u16* sb = sa + PA;
#pragma unroll
        for (int sg = 0; sg < 2; ++sg) {
            f16x8 bb[WKT];
#pragma unroll
            for (int j = 0; j < WKT; ++j) bb[j] = frag(sb, std::integral_constant<int, NCB>{}, sg, wk * WKT + j);
#pragma unroll
            for (int i = 0; i < WNT; ++i) {
                const f16x8 a = frag(sa, std::integral_constant<int, NCA>{}, sg, wn * WNT + i);
#pragma unroll
                for (int j = 0; j < WKT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb[j], acc[i][j], 0, 0, 0);
                if (want_cs) accs[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, ones, accs[i], 0, 0, 0);
            }
        }
    }
    if (want_cs && r == 0) {   // every column of accs holds the sums: lanes 0 and 32 own the tile's 32 rows between them
#pragma unroll
        for (int i = 0; i < WNT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) g.colsum[(size_t)by * g.N + n0 + (wn * WNT + i) * 32 + mfma_row(e, h)] = accs[i][e];
    }
    float* Cb = g.C + (size_t)by * g.N * g.K;
#pragma unroll
    for (int i = 0; i < WNT; ++i)
#pragma unroll
        for (int j = 0; j < WKT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + (wn * WNT + i) * 32 + mfma_row(e, h), k = k0 + (wk * WKT + j) * 32 + r;
                Cb[(size_t)n * g.K + k] = acc[i][j][e];
            }
}

template <int WNT, int WKT>
int tn_f16_launch_t(const TnArgs& g, hipStream_t st) {
    constexpr int TN = 64 * WNT, TK = 64 * WKT;
    const int G = (g.Z + g.group - 1) / g.group;
    const size_t lds = (size_t)2 * 32 * (TN + TK) * 2 + (g.pro_a ? (size_t)4 * TK * sizeof(float) : 0);   // 32 - 52 KiB: no attribute needed
    const dim3 grid(((g.N + TN - 1) / TN) * ((g.K + TK - 1) / TK), G);
    if (g.a_f16 && g.b_f16) return -9;   // (the DMA form: gemm_tn_f16_launch)
    if (g.a_f16) {   // the MLP backward's du as halves: with the AdaGN apply on x (mlp.0), or plain
        if (g.b_f16) return -9;
        if (g.pro_a) hipLaunchKernelGGL((gemm_tn_f16_kernel<WNT, WKT, true, false, true>), grid, dim3(256), lds, st, g);
        else hipLaunchKernelGGL((gemm_tn_f16_kernel<WNT, WKT, false, false, true>), grid, dim3(256), lds, st, g);
    } else if (g.b_f16) hipLaunchKernelGGL((gemm_tn_f16_kernel<WNT, WKT, false, true>), grid, dim3(256), lds, st, g);
    else if (g.pro_a) hipLaunchKernelGGL((gemm_tn_f16_kernel<WNT, WKT, true, false>), grid, dim3(256), lds, st, g);
    else hipLaunchKernelGGL((gemm_tn_f16_kernel<WNT, WKT, false, false>), grid, dim3(256), lds, st, g);
    return (int)hipGetLastError();
}

// block tile for an (N, K) gradient: the wide side on the dimension that fills whole 256-column tiles (768 = 3 x 256), 128 x 128
// otherwise (384 x 384: 9 tiles already; the conditioner's 96 / 192 / 48 / 672-wide layers)
int tn_f16_shape(int N, int K) {
    // Measured (profiles/r04p): alone, the wide tiles take 134 / 132 us against 168 / 174 for the 768 x 384 / 384 x 768 gradients
    // (AdaGN form 161 / 152 against 198 / 190); inside the training step, where the weight gradients share the CUs with the dX chain
    // of the main stream (autograd.py: side stream), their 250 registers and 48 KiB pack worse beside the other kernels and the step
    // is 19.2 ms against 19.0 (without the side stream: 19.1 against 19.35).  Round 6: with fp16 tensors between the training kernels the
    // balance tips — 16.46 - 16.53 ms with the wide tiles against 16.63 - 16.68 (three runs each, one box): the DEFAULT since;
    // GECCO_TN_F16_WIDE=0 selects 128 x 128.
    static int wide = -1;
    if (wide < 0) {
        const char* e = getenv("GECCO_TN_F16_WIDE");
        wide = (e && atoi(e) == 0) ? 0 : 1;
    }
    if (!wide) return 0;
    if (K % 256 == 0 && (N % 256 != 0 || K >= N)) return 2;   // 128 x 256
    if (N % 256 == 0) return 1;                              // 256 x 128
    return 0;
}

}  // namespace

bool gemm_tn_f16_supported(const TnArgs& g) {
    // one sample's rows are addressed with 31-bit byte offsets (buffer loads)
    return g.Z > 0 && g.group > 0 && g.R >= 32 && g.R % 32 == 0 && g.N > 0 && g.K > 0 && !(g.N & 3) && !(g.K & 3) && !(g.lda & 3) &&
           !(g.ldb & 3) && (!g.b_f16 || (!g.pro_a && !(g.K & 7) && !(g.ldb & 7))) && (!g.a_f16 || (!(g.N & 7) && !(g.lda & 7))) &&
           (!(g.a_f16 && g.b_f16) || (!(g.N % 128) && !(g.K % 128) && !g.counters)) &&
           (size_t)g.R * g.lda * 4 < 0x7fffffffu &&
           (size_t)g.R * g.ldb * 4 < 0x7fffffffu;
}

// output tiles per sample group (the host sizes the groups so that groups x tiles fill the chip)
int gemm_tn_f16_tiles(int N, int K) {
    const int sh = tn_f16_shape(N, K), tn = sh == 1 ? 256 : 128, tk = sh == 2 ? 256 : 128;
    return ((N + tn - 1) / tn) * ((K + tk - 1) / tk);
}

int gemm_tn_f16_launch(const TnArgs& g, hipStream_t st) {
    if (!gemm_tn_f16_supported(g)) return -9;
    static int xcd = -1;
    if (xcd < 0) {
        const char* e = getenv("GECCO_TN_XCD");   // 0: plain dispatch order (A/B runs)
        xcd = (e && atoi(e) == 0) ? 0 : 1;
    }
    TnArgs ga = g;
    ga.xcd = xcd;
    if (g.a_f16 && g.b_f16) {   // both operands fp16 tensors: the DMA form, whole 128 x 128 tiles (gemm_tn_f16_supported checked)
        const int G = (g.Z + g.group - 1) / g.group;
        const dim3 grid((g.N / 128) * (g.K / 128), G);
        hipLaunchKernelGGL((gemm_tn_f16_dma_kernel<2, 2>), grid, dim3(256), (size_t)2 * 32 * (128 + 128) * 2, st, ga);
        return (int)hipGetLastError();
    }
    switch (tn_f16_shape(g.N, g.K)) {
        case 1: return tn_f16_launch_t<4, 2>(ga, st);
        case 2: return tn_f16_launch_t<2, 4>(ga, st);
        default: return tn_f16_launch_t<2, 2>(ga, st);
    }
}
