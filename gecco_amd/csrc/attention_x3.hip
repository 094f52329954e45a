// Inducing-point attention in split-bf16 arithmetic ("bf16x3", see gemm_f32_dma.hip), gfx950.  Same contract and
// the same S^T = K Q^T / O^T = V^T P^T orientation as attention_f32.hip, but every product runs as three
// v_mfma_f32_32x32x16_bf16 (a_lo b_hi + a_hi b_lo + a_hi b_hi, fp32 accumulate): 42 matrix instructions of 32
// cycles per 32-row tile instead of 112 fp32 ones of 64 cycles, which moves both kernels from the fp32 matrix pipe
// to the HBM stream of their (B, N, C) operand.
//   * operands live in LDS as bf16 hi | lo planes, rows padded to an odd number of 16-byte chunks (conflict-free
//     ds_read_b128 fragments: lane half h takes k = 8h .. 8h+7 of a 16-wide chunk);
//   * the probability tile never leaves registers: accumulator registers 8s .. 8s+7 of lane half h are keys
//     16s + 8(j>>2) + 4h + (j&3), j = 0..7, and become the B fragment of key chunk s after the hi / lo split, so the
//     V^T fragment is delivered in that key order:
//       unpool — the 64 inducer values are block-constant: stored transposed and key-permuted once per block;
//       pool   — the streamed value tile is stored row-major in 4-key x 32-column blocks of 256 B and read with
//                ds_read_b64_tr_b16 (the hardware transpose), one conflict-free block per 32-lane half.
// Head dims: multiples of 16 up to 64; anything else stays on the fp32 kernels.
// The same kernels serve the fp16 mode (precision 2, template flag F16): one fp16 plane instead of hi | lo, one
// v_mfma_f32_32x32x16_f16 per product, operands rounded to nearest even.
#include "common.h"
#include "h8_scales.h"
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

// 4 fp32 -> 4 bf16 hi (top 16 bits) and 4 bf16 lo = rne(x - hi), each packed in two dwords
// (F16: hi = the 4 values rounded to fp16, lo unused)
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
        hi[p] = __builtin_amdgcn_perm(uc, ua, 0x07060302u);  // {hi16(x[2p+1]), hi16(x[2p])}
        l[2 * p] = (__bf16)(x[2 * p] - __uint_as_float(ua & 0xFFFF0000u));
        l[2 * p + 1] = (__bf16)(x[2 * p + 1] - __uint_as_float(uc & 0xFFFF0000u));
    }
    lo = __builtin_bit_cast(u32x2, l);
}

// accumulator registers e0 .. e0+7 -> the hi / lo fragments of one 16-key chunk (F16: one fp16 fragment)
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

__device__ __forceinline__ u32x4 frag(const u16* p) { return *reinterpret_cast<const u32x4*>(p); }

template <bool F16>
__device__ __forceinline__ f32x16 mfma3(const u32x4& ahi, const u32x4& alo, const u32x4& bhi, const u32x4& blo,
                                        f32x16 acc) {
    if (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ahi), __builtin_bit_cast(f16x8, bhi), acc, 0, 0, 0);
    const bf16x8 ah = __builtin_bit_cast(bf16x8, ahi), al = __builtin_bit_cast(bf16x8, alo);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, bhi), bl = __builtin_bit_cast(bf16x8, blo);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// ------------------------------------------------------------------------------------- pool
// element offset of value (key, d) in a wave's value tile: 4-key x 32-column blocks of 128 elements (256 B).
// (The 16-byte tile stores of the fp16 tensors conflict 2-way here — a key's columns 32 .. land on the banks of its columns 0 .., stores
// are banked modulo 128 B — 23 % of the kernel's LDS-active cycles by the counter; rotating the odd blocks by half a block did not move
// it, a conflict-free placement is row-dependent, and all the kernel's LDS stores together are 12 of its 50 us: DESIGN.md section 5b.)
template <int DT>
__device__ __forceinline__ int vt_off(int key, int d) {
    return ((key >> 2) * DT + (d >> 5)) * 128 + (key & 3) * 32 + (d & 31);
}

// IO16 (fp16 mode only): KV is an fp16 tensor (the kv_proj GEMM stored it that way) — tiles are copied to LDS as they are
template <int HD, bool F16, bool IO16>
__global__ __launch_bounds__(256, 2) void pool_attn_x3_kernel(const float* __restrict__ KV,
                                                           const float* __restrict__ Qind,
                                                           float* __restrict__ part_o, float* __restrict__ part_ml,
                                                           int B, int N, int C, int H, int nsplit, int hm) {
    constexpr int KS = HD + 8;            // bf16 elements per K / Q row: an odd number of 16-byte chunks
    constexpr int DT = (HD + 31) / 32;
    constexpr int CH = HD / 4;
    constexpr int LD_IT = (32 * CH + 63) / 64;
    constexpr int NC = HD / 16;           // 16-wide k chunks of the head dim
    constexpr int VT = 8 * DT * 128;      // elements per value plane of one wave tile
    constexpr int NP = F16 ? 1 : 2;       // planes per operand
    constexpr int WAVE_E = NP * (32 * KS + VT);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16* lds = reinterpret_cast<u16*>(smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // newest data first: K | V were written in ascending (sample, row) order by a kernel whose output (300 MB) exceeds the
    // 256 MB Infinity Cache — walking them backwards meets what is still cached instead of chasing the eviction front
    const int bid = gridDim.x - 1 - blockIdx.x;
    const int bh = bid / nsplit, split = bid % nsplit;
    const int b = bh / H, hh = bh % H;

    u16* Qhi = lds;
    u16* Qlo = F16 ? Qhi : lds + 64 * KS;
    u16* Khi = lds + NP * 64 * KS + wave * WAVE_E;
    u16* Klo = F16 ? Khi : Khi + 32 * KS;
    u16* Vhi = Khi + NP * 32 * KS;
    u16* Vlo = F16 ? Vhi : Vhi + VT;

    const int ks = (((N + nsplit - 1) / nsplit) + 31) / 32 * 32;
    const int k_begin = split * ks;
    const int k_end = min(N, k_begin + ks);
    const int ntiles = k_end > k_begin ? (k_end - k_begin + 31) / 32 : 0;
    const int nit = (ntiles + 3) / 4;

    const float scale = LOG2E * rsqrtf((float)HD);
    for (int f = tid; f < 64 * CH; f += 256) {
        const int row = f / CH, ch = f % CH;
        const f32x4 v = *reinterpret_cast<const f32x4*>(Qind + ((size_t)hh * 64 + row) * HD + ch * 4) * scale;
        u32x2 hi, lo;
        split4<F16>(v, hi, lo);
        *reinterpret_cast<u32x2*>(Qhi + row * KS + ch * 4) = hi;
        if (!F16) *reinterpret_cast<u32x2*>(Qlo + row * KS + ch * 4) = lo;
    }

    const size_t ldkv = 2 * (size_t)C;
    const float* Kg = KV + (size_t)b * N * ldkv + hh * HD;
    const float* Vg = Kg + C;

    static_assert(!IO16 || F16, "fp16 tensors exist in fp16 mode only");
    constexpr int CH8 = HD / 8, LD8 = (32 * CH8 + 63) / 64;   // 16-byte chunks of 8 fp16
    // hm (IO16 only): head-major K | V, one contiguous (N, HD) slab per (sample, K or V, head) — gemm_f16_astat.hip
    const size_t ld16 = hm ? (size_t)HD : ldkv;
    const _Float16* Kg16 = reinterpret_cast<const _Float16*>(KV) +
                           (hm ? ((size_t)b * 2 * H + hh) * N * HD : (size_t)b * N * ldkv + hh * HD);
    const _Float16* Vg16 = Kg16 + (hm ? (size_t)H * N * HD : (size_t)C);
    f32x4 rk[IO16 ? 1 : LD_IT], rv[IO16 ? 1 : LD_IT];
    u32x4 rk8[IO16 ? LD8 : 1], rv8[IO16 ? LD8 : 1];
    auto load_tile = [&](int tile) {
        const int base = k_begin + tile * 32;
        if (IO16) {
#pragma unroll
            for (int it = 0; it < LD8; ++it) {
                const int f = it * 64 + lane, row = f / CH8, c8 = f % CH8, key = base + row;
                u32x4 zk = {0u, 0u, 0u, 0u}, zv = {0u, 0u, 0u, 0u};
                if (f < 32 * CH8 && tile < ntiles && key < k_end) {
                    zk = *reinterpret_cast<const u32x4*>(Kg16 + key * ld16 + c8 * 8);
                    zv = *reinterpret_cast<const u32x4*>(Vg16 + key * ld16 + c8 * 8);
                }
                rk8[it] = zk;
                rv8[it] = zv;
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < LD_IT; ++it) {
            const int f = it * 64 + lane, row = f / CH, ch = f % CH, key = base + row;
            f32x4 zk = {0.f, 0.f, 0.f, 0.f}, zv = {0.f, 0.f, 0.f, 0.f};
            if (f < 32 * CH && tile < ntiles && key < k_end) {
                zk = *reinterpret_cast<const f32x4*>(Kg + key * ldkv + ch * 4);
                zv = *reinterpret_cast<const f32x4*>(Vg + key * ldkv + ch * 4);
            }
            rk[it] = zk;
            rv[it] = zv;
        }
    };
    auto store_tile = [&]() {
        if (IO16) {
#pragma unroll
            for (int it = 0; it < LD8; ++it) {
                const int f = it * 64 + lane, row = f / CH8, c8 = f % CH8;
                if (f < 32 * CH8) {
                    *reinterpret_cast<u32x4*>(Khi + row * KS + c8 * 8) = rk8[it];
                    *reinterpret_cast<u32x4*>(Vhi + vt_off<DT>(row, c8 * 8)) = rv8[it];
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < LD_IT; ++it) {
            const int f = it * 64 + lane, row = f / CH, ch = f % CH;
            if (f < 32 * CH) {
                u32x2 hi, lo;
                split4<F16>(rk[it], hi, lo);
                *reinterpret_cast<u32x2*>(Khi + row * KS + ch * 4) = hi;
                if (!F16) *reinterpret_cast<u32x2*>(Klo + row * KS + ch * 4) = lo;
                split4<F16>(rv[it], hi, lo);
                *reinterpret_cast<u32x2*>(Vhi + vt_off<DT>(row, ch * 4)) = hi;
                if (!F16) *reinterpret_cast<u32x2*>(Vlo + vt_off<DT>(row, ch * 4)) = lo;
            }
        }
    };
    // transposed-read addressing: lane 4q+p of a 16-lane group points at row q, columns 4p .. 4p+3 of its block
    const int tq = (lane & 15) >> 2, tp = lane & 3, tcol = 16 * ((lane >> 4) & 1) + 4 * tp;

    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    f32x16 O[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) O[dt][j][e] = 0.f;

    load_tile(wave);
    __syncthreads();  // the shared query planes are complete
    for (int it = 0; it < nit; ++it) {
        const int tile = wave + 4 * it;
#ifdef PA_DIAG_NOSTORE   // diagnostic builds (tools/probe/pool_probe.hip): one ingredient removed each; only the time matters
        if (it == 0)
#endif
        store_tile();
        wave_lds_sync();
#ifdef PA_DIAG_NOLOAD
        if (it == 0)
#endif
        load_tile(tile + 4);
        if (tile < ntiles) {
            f32x16 s[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[j][e] = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const u32x4 ah = frag(Khi + r * KS + c * 16 + 8 * h);
                const u32x4 al = F16 ? ah : frag(Klo + r * KS + c * 16 + 8 * h);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32x4 qh = frag(Qhi + (j * 32 + r) * KS + c * 16 + 8 * h);
                    const u32x4 ql = F16 ? qh : frag(Qlo + (j * 32 + r) * KS + c * 16 + 8 * h);
#ifdef PA_DIAG_NOMFMA
                    s[j][0] += (float)ah[0] + (float)qh[0];
#else
                    s[j] = mfma3<F16>(ah, al, qh, ql, s[j]);
#endif
                }
            }
            const int kbase = k_begin + tile * 32;
            const bool ragged = kbase + 32 > k_end;   // wave-uniform: only a split's last tile masks keys
#ifndef PA_DIAG_NOSOFTMAX
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float mx = -INFINITY;
                if (ragged) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (kbase + mfma_row(e, h) >= k_end) s[j][e] = -INFINITY;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[j][e]);
                mx = fmaxf(mx, xor32(mx));
                const float mn = fmaxf(m[j], mx);  // finite: the tile holds >= 1 valid key
                // v_exp_f32 directly: arguments are <= 0, results in (0, 1]; what it flushes is below 2^-126
                const float alpha = __builtin_amdgcn_exp2f(m[j] - mn);
                float ps = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    s[j][e] = __builtin_amdgcn_exp2f(s[j][e] - mn);
                    ps += s[j][e];
                }
                l[j] = l[j] * alpha + ps;
                m[j] = mn;
                // the running maximum settles after a few tiles: rescale the accumulators only when some query's moved
                if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) O[dt][j][e] *= alpha;
                }
            }
#endif
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {   // the tile's two 16-key chunks
                u32x4 ph[2], pl[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) split_acc8<F16>(s[j], 8 * sg, ph[j], pl[j]);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const int o0 = vt_off<DT>(16 * sg + 4 * h + tq, dt * 32 + tcol);
                    const int o1 = vt_off<DT>(16 * sg + 8 + 4 * h + tq, dt * 32 + tcol);
                    typedef __attribute__((address_space(3))) s16x4* lp;
                    const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(Vhi + o0));
                    const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(Vhi + o1));
                    const u32x4 vh = __builtin_bit_cast(u32x4, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                    u32x4 vl = vh;
                    if (!F16) {
                        const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(Vlo + o0));
                        const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(Vlo + o1));
                        vl = __builtin_bit_cast(u32x4, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#ifdef PA_DIAG_NOMFMA
                        O[dt][j][0] += (float)vh[0] + (float)ph[j][0];
#else
                        O[dt][j] = mfma3<F16>(vh, vl, ph[j], pl[j], O[dt][j]);
#endif
                }
            }
        }
        wave_lds_sync();  // this wave's reads of its K / V tile are done before it overwrites them
    }
    __syncthreads();      // every wave is done with its staging area: the combine below reuses the LDS

    // ---- combine the four waves' (m, l, O) and emit one partial per (b, head, split)
    float* Ow = smem;                 // [4][HD][64]
    float* Mw = smem + 4 * HD * 64;   // [4][64]
    float* Lw = Mw + 256;             // [4][64]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float lt = l[j] + xor32(l[j]);
        if (h == 0) {
            Mw[wave * 64 + j * 32 + r] = m[j];
            Lw[wave * 64 + j * 32 + r] = lt;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int d = dt * 32 + mfma_row(e, h);
                if (d < HD) Ow[(wave * HD + d) * 64 + j * 32 + r] = O[dt][j][e];
            }
    }
    __syncthreads();
    {
        const int q = tid & 63, part = tid >> 6;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, Mw[w * 64 + q]);
        float f[4], L = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = Mw[w * 64 + q];
            f[w] = (mw == -INFINITY) ? 0.f : exp2f(mw - M);
            L += f[w] * Lw[w * 64 + q];
        }
        const size_t pbase = ((size_t)bh * nsplit + split) * 64 + q;
        for (int d = part; d < HD; d += 4) {
            float o = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) o += f[w] * Ow[(w * HD + d) * 64 + q];
            part_o[pbase * HD + d] = o;
        }
        if (part == 0) {
            part_ml[pbase * 2 + 0] = M;
            part_ml[pbase * 2 + 1] = L;
        }
    }
}

// ----------------------------------------------------------------------------------- unpool
// IO16 (fp16 mode only): q and out are fp16 tensors (written by the q projection / read by out_proj in that form)
template <int HD, bool F16, bool IO16>
__global__ __launch_bounds__(256, 2) void unpool_attn_x3_kernel(const float* __restrict__ q,
                                                             const float* __restrict__ kvh, float* __restrict__ out,
                                                             int B, int N, int C, int H, int tiles_per_wave,
                                                             int nchunk, int hm, int out32) {   // out32: 1 (IO16) fp32 output; 2: tiled split image
    constexpr int KS = HD + 8;            // bf16 elements per K / Q row
    constexpr int VS = 64 + 8;            // bf16 elements per V^T row (64 permuted keys + pad)
    constexpr int DT = (HD + 31) / 32;
    constexpr int CH = HD / 4;
    constexpr int LD_IT = (32 * CH + 63) / 64;
    constexpr int NC = HD / 16;
    constexpr int OP = HD + 4;            // fp32 row stride of the output transpose tile (aliases the Q planes)
    constexpr int NP = F16 ? 1 : 2;       // planes per operand
    constexpr int QW = 32 * OP * 2 > NP * 32 * KS ? 32 * OP * 2 : NP * 32 * KS;   // u16 per wave: Q planes, later the fp32 output tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16* lds = reinterpret_cast<u16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int chunk = blockIdx.x % nchunk, bh = blockIdx.x / nchunk;
    const int b = bh / H, hh = bh % H;

    u16* Khi = lds;                        // [64][KS]
    u16* Klo = F16 ? Khi : Khi + 64 * KS;
    u16* Vthi = Khi + NP * 64 * KS;        // [DT * 32][VS]: row = head-dim index, keys permuted inside 16-chunks
    u16* Vtlo = F16 ? Vthi : Vthi + DT * 32 * VS;
    u16* Qhi = Vthi + NP * DT * 32 * VS + wave * QW;
    u16* Qlo = F16 ? Qhi : Qhi + 32 * KS;
    float* Ot = reinterpret_cast<float*>(Qhi);   // [32][OP] fp32, after the Q fragments are consumed

    for (int f = tid; f < DT * 32 * VS / 2; f += 256) {   // zero the value planes: padded head-dim rows stay finite
        reinterpret_cast<unsigned*>(Vthi)[f] = 0u;
        if (!F16) reinterpret_cast<unsigned*>(Vtlo)[f] = 0u;
    }
    __syncthreads();
    for (int f = tid; f < 64 * CH; f += 256) {
        const int key = f / CH, ch = f % CH;
        const float* src = kvh + ((size_t)b * 64 + key) * 2 * C + hh * HD + ch * 4;
        u32x2 hi, lo;
        split4<F16>(*reinterpret_cast<const f32x4*>(src), hi, lo);
        *reinterpret_cast<u32x2*>(Khi + key * KS + ch * 4) = hi;
        if (!F16) *reinterpret_cast<u32x2*>(Klo + key * KS + ch * 4) = lo;
        split4<F16>(*reinterpret_cast<const f32x4*>(src + C), hi, lo);
        // key 16c + 8a + 4g + i sits at position 16c + 8g + 4a + i: lane half g reads its 8 keys contiguously
        const int pos = (key & ~15) + 8 * ((key >> 2) & 1) + 4 * ((key >> 3) & 1) + (key & 3);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Vthi[(ch * 4 + e) * VS + pos] = (u16)(hi[e >> 1] >> (16 * (e & 1)));
            if (!F16) Vtlo[(ch * 4 + e) * VS + pos] = (u16)(lo[e >> 1] >> (16 * (e & 1)));
        }
    }
    const float scale = LOG2E * rsqrtf((float)HD);
    const float* qb = q + (size_t)b * N * C + hh * HD;
    float* ob = out + (size_t)b * N * C + hh * HD;

    static_assert(!IO16 || F16, "fp16 tensors exist in fp16 mode only");
    constexpr int CH8 = HD / 8, LD8 = (32 * CH8 + 63) / 64;
    // hm (IO16 only): head-major q, one contiguous (N, HD) slab per (sample, head)
    const size_t ldq16 = hm ? (size_t)HD : (size_t)C;
    const _Float16* qb16 = reinterpret_cast<const _Float16*>(q) +
                           (hm ? ((size_t)b * H + hh) * N * HD : (size_t)b * N * C + hh * HD);
    _Float16* ob16 = reinterpret_cast<_Float16*>(out) + (size_t)b * N * C + hh * HD;
    f32x4 rq[IO16 ? 1 : LD_IT];
    u32x4 rq8[IO16 ? LD8 : 1];
    auto load_q = [&](int it) {
        const int q0 = (chunk * tiles_per_wave + it) * 128 + wave * 32;
        if (IO16) {
#pragma unroll
            for (int ld = 0; ld < LD8; ++ld) {
                const int f = ld * 64 + lane, row = f / CH8, c8 = f % CH8, n = q0 + row;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (f < 32 * CH8 && it < tiles_per_wave && n < N) v = *reinterpret_cast<const u32x4*>(qb16 + (size_t)n * ldq16 + c8 * 8);
                rq8[ld] = v;
            }
            return;
        }
#pragma unroll
        for (int ld = 0; ld < LD_IT; ++ld) {
            const int f = ld * 64 + lane, row = f / CH, ch = f % CH, n = q0 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (f < 32 * CH && it < tiles_per_wave && n < N) v = *reinterpret_cast<const f32x4*>(qb + (size_t)n * C + ch * 4);
            rq[ld] = v;
        }
    };
    load_q(0);
    for (int it = 0; it < tiles_per_wave; ++it) {
        const int q0 = (chunk * tiles_per_wave + it) * 128 + wave * 32;
        // fp16 mode: the queries are rounded as they are and the softmax scale multiplies S in fp32 (so a q tensor
        // already stored as fp16 gives the same bits); split-bf16 mode: scaled before the split
        if (IO16) {
#pragma unroll
            for (int ld = 0; ld < LD8; ++ld) {
                const int f = ld * 64 + lane, row = f / CH8, c8 = f % CH8;
                if (f < 32 * CH8) *reinterpret_cast<u32x4*>(Qhi + row * KS + c8 * 8) = rq8[ld];
            }
        } else {
#pragma unroll
            for (int ld = 0; ld < LD_IT; ++ld) {
                const int f = ld * 64 + lane, row = f / CH, ch = f % CH;
                if (f < 32 * CH) {
                    u32x2 hi, lo;
                    split4<F16>(F16 ? rq[ld] : rq[ld] * scale, hi, lo);
                    *reinterpret_cast<u32x2*>(Qhi + row * KS + ch * 4) = hi;
                    if (!F16) *reinterpret_cast<u32x2*>(Qlo + row * KS + ch * 4) = lo;
                }
            }
        }
        load_q(it + 1);
        if (it == 0) __syncthreads();  // the shared key / value planes are complete
        wave_lds_sync();
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const u32x4 qh = frag(Qhi + r * KS + c * 16 + 8 * h);
            const u32x4 ql = F16 ? qh : frag(Qlo + r * KS + c * 16 + 8 * h);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const u32x4 kh = frag(Khi + (kt * 32 + r) * KS + c * 16 + 8 * h);
                const u32x4 kl = F16 ? kh : frag(Klo + (kt * 32 + r) * KS + c * 16 + 8 * h);
                s[kt] = mfma3<F16>(kh, kl, qh, ql, s[kt]);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (F16) s[kt][e] *= scale;
                mx = fmaxf(mx, s[kt][e]);
            }
        mx = fmaxf(mx, xor32(mx));
        float ls = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[kt][e] = __builtin_amdgcn_exp2f(s[kt][e] - mx);
                ls += s[kt][e];
            }
        ls += xor32(ls);
        const float inv = 1.0f / ls;
        f32x16 O[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) O[dt][e] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
                u32x4 ph, pl;
                split_acc8<F16>(s[kt], 8 * sg, ph, pl);
                const int c16 = 2 * kt + sg;   // 16-key chunk of the 64 inducers
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const u32x4 vh = frag(Vthi + (dt * 32 + r) * VS + c16 * 16 + 8 * h);
                    const u32x4 vl = F16 ? vh : frag(Vtlo + (dt * 32 + r) * VS + c16 * 16 + 8 * h);
                    O[dt] = mfma3<F16>(vh, vl, ph, pl, O[dt]);
                }
            }
        wave_lds_sync();   // the Q fragments are consumed: their planes become the fp32 output tile
        // transpose O^T (query on the lane) back to rows
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = dt * 32 + 8 * g4 + 4 * h;
                if (d < HD) {
                    f32x4 v = {O[dt][4 * g4], O[dt][4 * g4 + 1], O[dt][4 * g4 + 2], O[dt][4 * g4 + 3]};
                    *reinterpret_cast<f32x4*>(Ot + r * OP + d) = v * inv;
                }
            }
        wave_lds_sync();
        if (out32 == 2) {
            // the output as the tiled split image a split-bf16 out_proj loads straight into registers (GemmArgs::a_img,
            // gemm_x3_areg.hip): per (sample, 128-row tile, 16-k step) an 8 KiB block of bf16 hi | lo planes [128][16]
            const int t128 = (N + 127) >> 7, nkc = C >> 4;
            unsigned short* img = reinterpret_cast<unsigned short*>(out);
#pragma unroll
            for (int ld = 0; ld < LD8; ++ld) {
                const int f = ld * 64 + lane, row = f / CH8, c8 = f % CH8, n = q0 + row;
                if (f < 32 * CH8 && n < N) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(Ot + row * OP + c8 * 8);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(Ot + row * OP + c8 * 8 + 4);
                    u32x4 hi, lo;
                    split_acc8<false>(f32x16{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3], 0, 0, 0, 0, 0, 0, 0, 0}, 0, hi, lo);
                    const int col = hh * HD + c8 * 8, ml = n & 127;
                    unsigned short* dst = img + (((size_t)b * t128 + (n >> 7)) * nkc + (col >> 4)) * 4096 + ml * 16 +
                                          ((((col & 15) >> 3) ^ ((ml >> 3) & 1)) << 3);
                    *reinterpret_cast<u32x4*>(dst) = hi;
                    *reinterpret_cast<u32x4*>(dst + 2048) = lo;
                }
            }
        } else if (out32 == 3) {
            // the output as the h8 activation image an h8 out_proj loads straight into registers (GemmArgs::a_img == 2,
            // gemm_h8_areg.hip): per (sample, 128-row tile, 64-k group) 6144 floats — fp16 hi fragments [32-row tile][sub][c][lane],
            // then fp8(2^11 lo) halves [32-row tile][t][lane]; a row's 8 columns are one 16-byte hi chunk and 8 lo bytes
            const int t128 = (N + 127) >> 7, ngk = C >> 6;
#pragma unroll
            for (int ld = 0; ld < LD8; ++ld) {
                const int f = ld * 64 + lane, row = f / CH8, c8 = f % CH8, n = q0 + row;
                if (f < 32 * CH8 && n < N) {
                    f32x4 v0 = *reinterpret_cast<const f32x4*>(Ot + row * OP + c8 * 8);
                    f32x4 v1 = *reinterpret_cast<const f32x4*>(Ot + row * OP + c8 * 8 + 4);
                    f16x8 hv;
                    float lo[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v0[e] = h8_clamp(v0[e]);       // h8_scales.h: an h8 operand is finite in every term
                        v1[e] = h8_clamp(v1[e]);
                        hv[e] = (_Float16)v0[e];       // v0 / v1 come from LDS: one fp32 value for the hi rounding and the lo difference
                        hv[4 + e] = (_Float16)v1[e];
                        lo[e] = __builtin_fminf(__builtin_fmaxf((v0[e] - (float)hv[e]) * H8_AL_SCALE, -448.f), 448.f);
                        lo[4 + e] = __builtin_fminf(__builtin_fmaxf((v1[e] - (float)hv[4 + e]) * H8_AL_SCALE, -448.f), 448.f);
                    }
                    int p0 = 0, p1 = 0;
                    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], p0, false);
                    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], p0, true);
                    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[4], lo[5], p1, false);
                    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[6], lo[7], p1, true);
                    const int col = hh * HD + c8 * 8, kk = col & 63, ml = n & 127, rt = ml >> 5, rr = ml & 31;
                    const int sub = kk >> 5, hhalf = (kk >> 4) & 1, cc = (kk >> 3) & 1;
                    float* blk = out + (((size_t)b * t128 + (n >> 7)) * ngk + (col >> 6)) * 6144;
                    *reinterpret_cast<u32x4*>(blk + rt * 1024 + (2 * sub + cc) * 256 + (32 * hhalf + rr) * 4) = __builtin_bit_cast(u32x4, hv);
                    typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<u32x2_*>(blk + 4096 + rt * 512 + sub * 256 + (32 * hhalf + rr) * 4 + 2 * cc) = u32x2_{(unsigned)p0, (unsigned)p1};
                }
            }
        } else if (IO16 && !out32) {
#pragma unroll
            for (int ld = 0; ld < LD8; ++ld) {
                const int f = ld * 64 + lane, row = f / CH8, c8 = f % CH8, n = q0 + row;
                if (f < 32 * CH8 && n < N) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(Ot + row * OP + c8 * 8);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(Ot + row * OP + c8 * 8 + 4);
                    f16x8 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hv[e] = (_Float16)v0[e];
                        hv[4 + e] = (_Float16)v1[e];
                    }
                    *reinterpret_cast<u32x4*>(ob16 + (size_t)n * C + c8 * 8) = __builtin_bit_cast(u32x4, hv);
                }
            }
        } else {
#pragma unroll
            for (int ld = 0; ld < LD_IT; ++ld) {
                const int f = ld * 64 + lane, row = f / CH, ch = f % CH, n = q0 + row;
                if (f < 32 * CH && n < N)
                    *reinterpret_cast<f32x4*>(ob + (size_t)n * C + ch * 4) =
                        *reinterpret_cast<const f32x4*>(Ot + row * OP + ch * 4);
            }
        }
        wave_lds_sync();
    }
}

template <int HD, bool F16, bool IO16>
int pool_x3_launch_t(const float* KV, const float* ind, float* po, float* pml, int B, int N, int C, int H, int nsplit,
                     hipStream_t st, int hm) {
    constexpr int KS = HD + 8, DT = (HD + 31) / 32, VT = 8 * DT * 128, NP = F16 ? 1 : 2;
    const size_t a = ((size_t)NP * 64 * KS + 4 * NP * (32 * KS + VT)) * 2, c = ((size_t)4 * HD * 64 + 512) * 4;
    const size_t lds = a > c ? a : c;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pool_attn_x3_kernel<HD, F16, IO16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((pool_attn_x3_kernel<HD, F16, IO16>), dim3(B * H * nsplit), dim3(256), lds, st, KV, ind, po, pml, B, N, C,
                       H, nsplit, hm);
    return (int)hipGetLastError();
}

template <int HD, bool F16, bool IO16>
int unpool_x3_launch_t(const float* q, const float* kvh, float* out, int B, int N, int C, int H, hipStream_t st, int hm, int out32 = 0) {
    constexpr int KS = HD + 8, VS = 72, DT = (HD + 31) / 32, NP = F16 ? 1 : 2, OP = HD + 4;
    constexpr int QW = 32 * OP * 2 > NP * 32 * KS ? 32 * OP * 2 : NP * 32 * KS;
    const size_t lds = ((size_t)NP * 64 * KS + NP * DT * 32 * VS + 4 * QW) * 2;
    const int tiles = (N + 127) / 128;
    int tpw = 1;
    while (tpw < 4 && (long)B * H * ((tiles + tpw * 2 - 1) / (tpw * 2)) >= 2048) tpw *= 2;
    const int nchunk = (tiles + tpw - 1) / tpw;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(unpool_attn_x3_kernel<HD, F16, IO16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((unpool_attn_x3_kernel<HD, F16, IO16>), dim3(B * H * nchunk), dim3(256), lds, st, q, kvh, out, B, N, C, H,
                       tpw, nchunk, hm, out32);
    return (int)hipGetLastError();
}

}  // namespace

bool attn_x3_supported(int HD) { return HD == 16 || HD == 32 || HD == 48 || HD == 64; }

int pool_attn_x3_partials_launch(const float* KV, const float* inducers, float* part_o, float* part_ml, int B, int N,
                                 int C, int H, int nsplit, hipStream_t st, int precision, int io16, int hm) {
    if (io16 && precision != 2) return -9;
    if (hm && !io16) return -9;
#define POOL_CASE(HD)                                                                                                    \
    case HD:                                                                                                             \
        return io16 ? pool_x3_launch_t<HD, true, true>(KV, inducers, part_o, part_ml, B, N, C, H, nsplit, st, hm)            \
               : precision == 2 ? pool_x3_launch_t<HD, true, false>(KV, inducers, part_o, part_ml, B, N, C, H, nsplit, st, 0) \
                                : pool_x3_launch_t<HD, false, false>(KV, inducers, part_o, part_ml, B, N, C, H, nsplit, st, 0)
    switch (C / H) {
        POOL_CASE(16);
        POOL_CASE(32);
        POOL_CASE(48);
        POOL_CASE(64);
        default: return -4;
    }
#undef POOL_CASE
}

int unpool_attn_x3_launch(const float* q, const float* kvh, float* out, int B, int N, int C, int H, hipStream_t st,
                          int precision, int io16, int hm, int out_img) {
    if (io16 && precision != 2) return -9;
    if (hm && !io16) return -9;
    if (out_img && (C % 16 || (out_img == 2 && C % 64) || (io16 && io16 != 2))) return -9;
    // io16 = 2: q is an fp16 tensor, the output stays fp32 (operand of a split-bf16 out_proj); out_img: as a tiled split image
    const int out32 = out_img == 2 ? 3 : out_img ? 2 : io16 == 2;   // out_img 2: h8 activation image (gemm_h8_areg.hip)
#define UNPOOL_CASE(HD)                                                                                \
    case HD:                                                                                           \
        return io16 ? unpool_x3_launch_t<HD, true, true>(q, kvh, out, B, N, C, H, st, hm, out32)       \
               : precision == 2 ? unpool_x3_launch_t<HD, true, false>(q, kvh, out, B, N, C, H, st, 0)  \
                                : unpool_x3_launch_t<HD, false, false>(q, kvh, out, B, N, C, H, st, 0, out_img == 2 ? 3 : out_img ? 2 : 0)
    switch (C / H) {
        UNPOOL_CASE(16);
        UNPOOL_CASE(32);
        UNPOOL_CASE(48);
        UNPOOL_CASE(64);
        default: return -4;
    }
#undef UNPOOL_CASE
}
