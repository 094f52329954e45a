// A-stationary fused linear of the fp16 mode: AdaGN apply + fp16 rounding + every column tile of the layer's wide
// projections in ONE pass over x.
//
//   C16[b, m, n] = fp16( act( sum_k fp16(x[b, m, k] * pa[b, k] + po[b, k]) * fp16(W[n, k]) + bias[n] ) )
//
// for the two launches that read an AdaGN-normalised residual stream and write fp16 intermediates: kv_proj | q_proj
// (Nout = 3d, two output tensors) and mlp.0 + GaussianActivation (Nout = 2d).  Same arithmetic, bit for bit, as
// gecco_affine_cast_f16 followed by the fp16 GEMM (gemm_f16_dma.hip) — but:
//   * a block (4 waves) owns 128 rows of x and every wave keeps ITS 32 rows in REGISTERS: it reads the fp32 rows once
//     from HBM, applies the affine, rounds to fp16 and holds them as the MFMA A fragments of all K / 32 K-steps
//     (K = 384: 96 VGPRs).  No fp16 copy of AdaGN(x) in HBM, no cast pass, no A tile re-fetched by every column tile
//     (the tile fill of the streaming kernel was its bound: tools/probe/dma_rate.hip, ~43 B/clk/CU into LDS) and no A
//     fragment re-read from LDS by every MFMA (an LDS-resident panel with 32 x 64 wave tiles needed 1.5 KiB of LDS
//     reads per MFMA — the LDS port, not the matrix pipe, set its pace);
//   * the block walks ALL column tiles with 32 x 128 wave tiles; W streams from its pre-tiled fp16 image through a
//     6-slot LDS-DMA ring (two 1 KiB pieces per wave per K-step), its fragments double-buffered in registers; the
//     K-steps of consecutive column tiles form ONE pipeline, so the next tile's stages are in flight while a tile's
//     epilogue stores drain.  48 KiB of LDS and <= 256 VGPRs: two independent blocks per CU, whose build, MFMA and
//     store phases interleave;
//   * the epilogue needs no LDS: bias and activation in registers, neighbouring lanes exchange one value (DPP) so
//     every lane stores a packed pair of fp16 (rows stay 64-byte contiguous per 32-column tile).
// vmcnt bookkeeping: a wave's epilogue issues exactly 32 stores (full tiles only: rows % 128 == 0, Nout % 128 == 0),
// which sit in the in-order vmcnt queue behind the DMA pieces already in flight; the K-steps after an epilogue whose
// awaited piece was issued before it wait with that count added instead of draining the stores.
#include "gemm_dma_common.h"
#include "h8_scales.h"

#include <stdlib.h>

#include <utility>

namespace {

using dma::dma16;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int SBK = 32;            // k per stage
constexpr int S_TILE = 2048;       // floats of one [128][32] fp16 W tile (8 KiB)
constexpr int S_NT = 256;          // threads
constexpr int S_PW = 2;            // 1 KiB pieces of a W stage per wave
constexpr int S_STORES = 32;       // store instructions a wave's epilogue issues (32 x 128 wave tile)

// value of the neighbouring lane (lane ^ 1): one DPP move (quad_perm [1, 0, 3, 2])
__device__ __forceinline__ unsigned swap_pair(unsigned v) {
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}

// W stage pieces by buffer_load ... lds: the per-lane offset is a register computed once, the stage offset a scalar
__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

#ifdef ASTAT_STAMPS   // diagnostic build (tools/probe): per-block s_memtime stamps
__device__ unsigned long long g_astat_stamps[4096 * 8];
#define ASTAMP(i)                                                                                          \
    do {                                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_astat_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define ASTAMP(i)
#endif

template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
// the same with a second compile-time tag handed through (no wrapper lambda around f: the accumulators stay in registers)
template <int TAG, int... I, class F>
__device__ __forceinline__ void static_for_tag(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}, std::integral_constant<int, TAG>{}), ...);
}

// NK = K / 32 K-steps (compile-time: the A fragments are registers, indexed statically); NS ring slots with
// NK % NS == 0, so the ring slot of K-step kt of ANY column tile is kt % NS — every LDS address in the loop is static.
// WS = 2 ("mixed" mode): two-term fp16 weights W = W_hi + W_lo.  The lo image (g.w_img2) is streamed as NK further
// K-steps of every column tile, multiplied with the SAME register-resident A fragments: x . W_hi + x . W_lo in one
// accumulator.  The rounding of the weights — coherent over all points of a cloud, the dominant error of the fp16
// mode (tools/experiments/fp16_site_sensitivity.py) — drops from 2^-12 to 2^-23; the A side is unchanged.
// WS = 3: the lo term on the fp8 matrix instruction.  The lo image holds fp8(2^16 (W - fp16(W))) (h8_scales.h) in 64-k stages
// (f8lo_image_item): NK / 2 further stages per column tile, each ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 32-column block
// (scale_b = 2^-19) on the A fragments of two K-steps converted to fp8 on the fly — half the LDS bytes and half the
// matrix cycles of the fp16 lo term for the same result (the lo term is 2^-12 of the product: 3 mantissa bits of it suffice;
// tools/experiments/fp16_site_sensitivity.py scheme x2a_v8).
template <int NK, int NS, int WS = 1>
__global__ __launch_bounds__(S_NT, 2) void gemm_f16_astat_kernel(GemmArgs g) {
    static_assert(NK % 2 == 0 && NK % NS == 0 && NS >= 4, "static slots; the A build stages 4 K-steps in the ring");
    static_assert(WS != 3 || (NK / 2) % NS == 0, "fp8 lo stages keep the static slot arithmetic");
    constexpr int NL = WS == 3 ? NK / 2 : (WS == 2 ? NK : 0);   // lo stages per column tile
    constexpr int NKW = NK + NL;   // W stages per column tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ASTAMP(0);
    float* ring = smem;                                // [NS][S_TILE]; first the staging area of the A build
    float* bias_lds = ring + NS * S_TILE;              // [Nout] (zeros where a segment has no bias)
    float* pro_lds = bias_lds + g.Nout;                // pa[0..K) | po[0..K)
    constexpr int K = NK * SBK;

    const int tilesM = g.rows / 128, tilesN = g.Nout / 128;
    const int b = blockIdx.x / tilesM, rt = blockIdx.x % tilesM, m0 = rt * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // rows 32 wave .. 32 wave + 31
    const int r = lane & 31, h = lane >> 5;

    for (int n = tid; n < g.Nout; n += S_NT) {
        const bool seg2 = g.C2 != nullptr && n >= g.n_split;
        const float* bp = seg2 ? g.bias2 : g.bias;
        bias_lds[n] = bp ? bp[seg2 ? n - g.n_split : n] : 0.f;
    }
    const bool has_pro = g.pro_a != nullptr;
    if (has_pro) {
        const float* pa = g.pro_a + (size_t)b * K;
        const float* po = g.pro_o + (size_t)b * K;
        for (int i = tid; i < K; i += S_NT) {
            pro_lds[i] = pa[i];
            pro_lds[K + i] = po[i];
        }
    }

    // ---- the A operand: fp16(x * pa + po).  Built 4 K-steps (128 k) at a time: coalesced 32-byte reads of the 128
    // rows (16 threads per row), affine + rounding, into the ring area laid out as four [128][32] fp16 stages (the
    // A16 layout of gemm_f16_dma.hip), from where every wave takes the fragments of ITS 32 rows into registers:
    // lane (r, h) holds row r, k = 32 kt + 16 h + 8 c .. + 7 in fa[kt][c].
    f16x8 fa[NK][2];
    {
        const float* xb = g.A + ((size_t)b * g.rows + m0) * g.lda;
        const int ra = wave * 32 + r;
        int aoff[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) aoff[c] = ra * 16 + (((2 * h + c) ^ ((ra >> 2) & 3)) << 2);
#pragma unroll
        for (int k0 = 0; k0 < NK; k0 += 4) {
            constexpr int ITEMS = 128 * 16 / S_NT;   // (row, 8-k chunk) items per thread per 128-k slab
            f32x4 x0[ITEMS], x1[ITEMS];
#pragma unroll
            for (int u = 0; u < ITEMS; ++u) {
                const int i = tid + u * S_NT, row = i >> 4, c8 = i & 15;
                const bool in = k0 * SBK + c8 * 8 < K;
                const float* src = xb + (size_t)row * g.lda + k0 * SBK + (in ? c8 * 8 : 0);
                x0[u] = *reinterpret_cast<const f32x4*>(src);   // default policy: x was written by the previous kernel
                x1[u] = *reinterpret_cast<const f32x4*>(src + 4);
            }
            __syncthreads();   // coefficients in LDS (first slab) / every wave done reading the previous slab
#pragma unroll
            for (int u = 0; u < ITEMS; ++u) {
                const int i = tid + u * S_NT, row = i >> 4, c8 = i & 15;
                if (k0 * SBK + c8 * 8 >= K) continue;
                if (has_pro) {
                    const float* ap = pro_lds + k0 * SBK + c8 * 8;
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
                    const f32x4 o0 = *reinterpret_cast<const f32x4*>(ap + K), o1 = *reinterpret_cast<const f32x4*>(ap + K + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x0[u][e] = __builtin_fmaf(x0[u][e], a0[e], o0[e]);
                        x1[u][e] = __builtin_fmaf(x1[u][e], a1[e], o1[e]);
                    }
                }
                f16x8 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // WS == 3: this operand also goes through the scaled fp8 conversion, which returns NaN beyond the format (h8_scales.h)
                    v[e] = (_Float16)(WS == 3 ? h8_clamp(x0[u][e]) : x0[u][e]);
                    v[4 + e] = (_Float16)(WS == 3 ? h8_clamp(x1[u][e]) : x1[u][e]);
                }
                const int sub = c8 >> 2, ch = (c8 & 3) ^ ((row >> 2) & 3);
                *reinterpret_cast<u32x4*>(ring + sub * S_TILE + row * 16 + ch * 4) = __builtin_bit_cast(u32x4, v);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u < NK) {
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        fa[k0 + u][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(ring + u * S_TILE + aoff[c]));
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();   // the staging area becomes the W ring
    }
    ASTAMP(1);

    // ---- W image: buffer_load ... lds, wave w moves pieces 2w, 2w+1 (1 KiB each) of every 8 KiB stage
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.w_img), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(WS >= 2 ? g.w_img2 : g.w_img), 0, 0x7fffffff, 0x00020000);
    const unsigned voff = (unsigned)((S_PW * wave) * 256 + lane * 4) * 4u;
    unsigned soff = 0;                                 // byte offset of the next stage to issue in its image (hi or lo)
    unsigned soff_hi = 0;                              // WS >= 2: where the hi image continues after a lo part
    int part_left = NK;                                // WS >= 2: stages left in the current part (hi, then lo) of the column tile
    bool lo_part = false;
    int it_tile = 0;                                   // column tile of the next stage to issue
    // WS == 2: column tiles [lo_begin, lo_tiles) carry a lo part — the V projection; the rounding of the K and q
    // projections' weights does not reach the output (tools/experiments/fp16_site_sensitivity.py): streamed hi only
    const int lo_tiles = WS >= 2 ? (g.lo_tiles > 0 ? g.lo_tiles : tilesN) : 0;
    const int lo_begin = WS >= 2 ? g.lo_begin : 0;
    auto issue = [&](int slot) {
        if (WS >= 2 && lo_part) {
            dma16_buf(wrsrc2, voff, soff, ring + slot * S_TILE + (S_PW * wave) * 256);
            dma16_buf(wrsrc2, voff + 1024u, soff, ring + slot * S_TILE + (S_PW * wave + 1) * 256);
        } else {
            dma16_buf(wrsrc, voff, soff, ring + slot * S_TILE + (S_PW * wave) * 256);
            dma16_buf(wrsrc, voff + 1024u, soff, ring + slot * S_TILE + (S_PW * wave + 1) * 256);
        }
        soff += S_TILE * 4u;
        if (WS >= 2 && --part_left == 0) {             // hi part done: this tile's stages of the lo image; lo done: next tile's hi
            if (!lo_part && it_tile >= lo_begin && it_tile < lo_tiles) {
                soff_hi = soff;
                soff = (unsigned)it_tile * (unsigned)(NL * S_TILE * 4);
                part_left = NL;
                lo_part = true;
            } else {
                if (lo_part) soff = soff_hi;
                part_left = NK;
                lo_part = false;
                ++it_tile;
            }
        }
    };
#pragma unroll
    for (int p = 0; p < NS; ++p) issue(p);             // tilesN * NK >= NK >= NS stages exist

    const bool has_act = g.act != 0;
    const int act_mode = g.act;
    const float neg_inv_2a2 = act_is_gauss(g.act) ? -1.0f / (2.0f * g.alpha[0] * g.alpha[0]) : 0.f;
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // W fragment addressing (float offsets inside a stage): rows of the four 32-column tiles
    int boff[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rb = j * 32 + r;
#pragma unroll
        for (int c = 0; c < 2; ++c) boff[j][c] = rb * 16 + (((2 * h + c) ^ ((rb >> 2) & 3)) << 2);
    }
    // the two 16-byte chunks of a 32-column block as ONE 8-register tuple: the fp16 MFMAs take its halves (sub-registers),
    // the fp8 lo stage (WS = 3) the whole tuple as its 32-byte B operand — no copies either way
    i32x8 fb[2][4];
    auto load_b = [&](int slot, int f) {
        const float* st = ring + slot * S_TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c)
            {
                const u32x4 v = *reinterpret_cast<const u32x4*>(st + boff[j][c]);
#pragma unroll
                for (int e = 0; e < 4; ++e) fb[f][j][4 * c + e] = (int)v[e];
            }
    };

    // ---- epilogue of one column tile, from registers: bias, activation, a packed fp16 pair per lane.
    // Lane (column n, rows in registers) converts registers e0, e0+1 (rows R, R+1 of its column) to one dword, swaps it
    // with its neighbour (DPP) and permutes: the even lane ends up with row R of columns n, n+1, the odd lane with row
    // R+1 of columns n-1, n — four bytes of one output row each.
    const bool odd = lane & 1;
    const unsigned psel = odd ? 0x03020706u : 0x05040100u;   // v_perm_b32 over {neighbour, own}
    const unsigned hm_magic = g.hm_hd ? (1u << 20) / (unsigned)g.hm_hd + 1u : 0u;
    auto epilogue = [&](int ct) {
        const int n0 = ct * 128;
        const bool seg2 = g.C2 != nullptr && n0 >= g.n_split;
        _Float16* Cb = reinterpret_cast<_Float16*>(seg2 ? g.C2 : g.C);
        const int ldc = seg2 ? g.ldc2 : g.ldc;
        const int nseg0 = seg2 ? n0 - g.n_split : n0;
        _Float16* lanebase = Cb + ((size_t)b * g.rows + m0 + wave * 32 + 4 * h + (odd ? 1 : 0)) * ldc + nseg0 + (r & ~1);
        // head-major output: row stride = head dim; the (sample, head) slab of this lane's column pair, per j
        const int rstride = g.hm_hd ? g.hm_hd : ldc;
        const int nseg = seg2 ? g.Nout - g.n_split : (g.C2 ? g.n_split : g.Nout);
        const int hm_row = m0 + wave * 32 + 4 * h + (odd ? 1 : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float bias = bias_lds[n0 + j * 32 + r];
            _Float16* jbase = lanebase + j * 32;
            if (g.hm_hd) {
                const int n = nseg0 + j * 32 + (r & ~1);
                const int grp = (int)(((unsigned)n * hm_magic) >> 20);   // n / hd (exact: n < 2^20 / hd)
                jbase = Cb + ((size_t)(b * (nseg / g.hm_hd) + grp) * g.rows + hm_row) * g.hm_hd + (n - grp * g.hm_hd);
            }
#pragma unroll
            for (int e0 = 0; e0 < 16; e0 += 2) {
                float v0 = acc[j][e0] + bias, v1 = acc[j][e0 + 1] + bias;
                acc[j][e0] = 0.f;
                acc[j][e0 + 1] = 0.f;
                if (has_act) {
                    v0 = act_apply(v0, neg_inv_2a2, act_mode);
                    v1 = act_apply(v1, neg_inv_2a2, act_mode);
                }
                f16x2 pk;
                pk[0] = (_Float16)v0;
                pk[1] = (_Float16)v1;
                const unsigned own = __builtin_bit_cast(unsigned, pk);
                const unsigned out = __builtin_amdgcn_perm(swap_pair(own), own, psel);
                constexpr int dummy = 0;
                (void)dummy;
                const int rowoff = (e0 & 3) + 8 * (e0 >> 2);   // + 4h + odd in lanebase
                *reinterpret_cast<unsigned*>(jbase + (size_t)rowoff * rstride) = out;
            }
        }
    };

    // first stages landed
    dma::wait_vm_lgkm0<(NS - 1) * S_PW>();
    __builtin_amdgcn_s_barrier();
    load_b(0, 0);
    float one = H8_AH_DIV;   // the fp8 conversions' scale operand: fp8(a / 8), h8_scales.h (see the lo stage)
    for (int ct = 0; ct < tilesN; ++ct) {
        const bool first = ct == 0, last = ct == tilesN - 1;
        const bool has_lo = WS >= 2 && ct >= lo_begin && ct < lo_tiles;   // this column tile runs NK + NL stages (hi, lo), else NK
        if (WS == 3) asm volatile("" : "+s"(one));
        static_for(std::make_integer_sequence<int, NKW>{}, [&](auto KT) {
            constexpr int kt = decltype(KT)::value;
            constexpr int cur = kt & 1;
            if (WS >= 2 && kt >= NK && !has_lo) return;   // wave-uniform: a tile without a lo part ends after NK stages
            constexpr bool early = kt <= NS - 2;   // the previous tile's epilogue stores still queue behind the awaited piece
            // steps left after this one in the last tile, for either length
            constexpr int rem_l = NKW - 1 - kt, rem_s = NK - 1 - kt;
            constexpr int n_l = rem_l >= NS - 1 ? NS - 2 : (rem_l >= 1 ? rem_l - 1 : 0);
            constexpr int n_s = rem_s >= NS - 1 ? NS - 2 : (rem_s >= 1 ? rem_s - 1 : 0);
            // own pieces of the next K-step's stage landed; younger stages (and those stores) may stay in flight
            if (!last) {
                if (early && !first) dma::wait_vm_lgkm0<(NS - 2) * S_PW + S_STORES>();
                else dma::wait_vm_lgkm0<(NS - 2) * S_PW>();
            } else if (WS == 1 || has_lo) {
                if (early && !first) dma::wait_vm_lgkm0<n_l * S_PW + S_STORES>();
                else dma::wait_vm_lgkm0<n_l * S_PW>();
            } else {
                if (early && !first) dma::wait_vm_lgkm0<n_s * S_PW + S_STORES>();
                else dma::wait_vm_lgkm0<n_s * S_PW>();
            }
            // this step's W fragments were read during the previous one and the wait above covered them: an empty asm
            // "redefines" the registers so the compiler's wait-count pass does not park its own lgkmcnt(0) in front of
            // the first MFMA, behind the NEXT step's reads issued below
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(fb[cur][j]));
            __builtin_amdgcn_s_barrier();
            const int len = (WS == 1 || has_lo) ? NKW : NK;
            if (!last || kt + NS < len) issue(kt % NS);                 // the stage NS steps ahead reuses this step's slot
            constexpr bool lo8 = WS == 3 && kt >= NK;
            // fp8 lo stage: its A operand is built in the registers of the idle fragment set, so the next stage's fragments
            // are read right AFTER the four matrix instructions are issued (256 cycles of pipe cover the LDS latency)
            if (!lo8 && (!last || kt + 1 < len)) load_b((kt + 1) % NS, cur ^ 1);
            if constexpr (lo8) {
                // fp8 lo stage S = kt - NK: k = 64 S .. 64 S + 63; lane half h holds k = 64 S + 32 c + 16 h .. + 15 of
                // its row in fa[2 S + c] — converted to 16 fp8 each, bytes in k order, matching the image's chunks
                constexpr int S = kt - NK;
                i32x8 a8;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        // f16 -> fp8 directly, two values per instruction (v_cvt_scalef32_pk_fp8_f16, scale 1)
                        // (`one` is 1.0 behind an opaque per-tile asm: it keeps these loop-invariant conversions from being
                        // hoisted out of the column-tile loop, where 48 more live registers would spill)
                        const f16x8 v = fa[2 * S + c][q];
                        s16x2 p0 = {0, 0}, p1 = {0, 0};
                        p0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p0, f16x2{v[0], v[1]}, one, false);
                        p0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p0, f16x2{v[2], v[3]}, one, true);
                        p1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p1, f16x2{v[4], v[5]}, one, false);
                        p1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p1, f16x2{v[6], v[7]}, one, true);
                        const int w0 = __builtin_bit_cast(int, p0), w1 = __builtin_bit_cast(int, p1);
                        a8[4 * c + 2 * q] = w0;
                        a8[4 * c + 2 * q + 1] = w1;
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, fb[cur][j], acc[j], 0, 0, 0, H8_SC_AH, 0, H8_SC_WL);
                if (!last || kt + 1 < len) load_b((kt + 1) % NS, cur ^ 1);
            } else {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const i32x4 bc = c == 0 ? __builtin_shufflevector(fb[cur][j], fb[cur][j], 0, 1, 2, 3)
                                                : __builtin_shufflevector(fb[cur][j], fb[cur][j], 4, 5, 6, 7);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kt % NK][c], __builtin_bit_cast(f16x8, bc), acc[j], 0, 0, 0);
                    }
            }
        });
        epilogue(ct);
    }
    ASTAMP(2);
}

template <int NK, int NS, int WS>
int astat_launch_ws(const GemmArgs& g, hipStream_t st) {
    const size_t lds = ((size_t)NS * S_TILE + g.Nout + 2 * g.K) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_astat_kernel<NK, NS, WS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL((gemm_f16_astat_kernel<NK, NS, WS>), dim3(g.B * (g.rows / 128)), dim3(S_NT), lds, st, g);
    return (int)hipGetLastError();
}
template <int NK, int NS>
int astat_launch_t(const GemmArgs& g, hipStream_t st) {
    if (g.w_img2 && g.lo_fp8) {
        if constexpr ((NK / 2) % NS == 0) return astat_launch_ws<NK, NS, 3>(g, st);
        else return -9;
    }
    return g.w_img2 ? astat_launch_ws<NK, NS, 2>(g, st) : astat_launch_ws<NK, NS, 1>(g, st);
}

}  // namespace

bool gemm_f16_astat_lo8_supported(int K) { return K == 256 || K == 384; }   // (K / 64) stages in whole ring rounds: <8, 4>, <12, 6>

bool gemm_f16_astat_supported(const GemmArgs& g) {
    const int nk = g.K / SBK;
    return g.c_f16 && !g.a_f16 && !g.residual && !g.stats && g.w_img && g.rows >= 128 && !(g.rows % 128) &&
           !(g.Nout % 128) && !(g.K % SBK) && (nk == 4 || nk == 8 || nk == 12 || nk == 16) && !(g.lda & 3) && !(g.ldc & 1) &&
           (!g.C2 || (!(g.n_split % 128) && !(g.ldc2 & 1) && g.n_split > 0 && g.n_split < g.Nout)) &&
           ((g.pro_a == nullptr) == (g.pro_o == nullptr)) &&
           (!g.hm_hd || (g.hm_hd >= 8 && !(g.hm_hd & 1) && g.Nout < (1 << 20) / g.hm_hd &&
                         !((g.C2 ? g.n_split : g.Nout) % g.hm_hd) && !((g.C2 ? g.Nout - g.n_split : 0) % g.hm_hd)));
}

int gemm_f16_astat_launch(const GemmArgs& g, hipStream_t st) {
    switch (g.K / SBK) {
        case 4: return astat_launch_t<4, 4>(g, st);     // d = 128
        case 8: return astat_launch_t<8, 4>(g, st);     // d = 256
        case 12: return astat_launch_t<12, 6>(g, st);   // d = 384 (the shipped configs)
        case 16: return astat_launch_t<16, 4>(g, st);   // d = 512
        default: return -9;
    }
}
