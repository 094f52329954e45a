// A-stationary fused linear of the fp16 mode: AdaGN apply + fp16 rounding + every column tile of the layer's wide
// projections in ONE pass over x.
//
//   C16[b, m, n] = fp16( act( sum_k fp16(x[b, m, k] * pa[b, k] + po[b, k]) * fp16(W[n, k]) + bias[n] ) )
//
// for the two launches that read an AdaGN-normalised residual stream and write fp16 intermediates: kv_proj | q_proj
// (Nout = 3d, two output tensors) and mlp.0 + GaussianActivation (Nout = 2d).  Same arithmetic, bit for bit, as
// gecco_affine_cast_f16 followed by the fp16 GEMM (gemm_f16_dma.hip) — but:
//   * a block owns a 128-row tile of x: it reads the fp32 rows ONCE from HBM, applies the affine, rounds to fp16 and
//     parks the whole [128][K] operand in LDS (K <= 384: 96 KiB) — no fp16 copy of AdaGN(x) in HBM, no cast pass,
//     and the A tile is not re-fetched by every column tile (the tile fill of the streaming kernel was its bound:
//     tools/probe/dma_rate.hip, ~43 B/clk/CU into LDS);
//   * the block then walks ALL column tiles: 8 waves as 4 (rows) x 2 (columns), each 32 rows x 64 columns of the
//     current 128-column tile; W streams from its pre-tiled fp16 image through ONE 6-slot LDS-DMA ring (one 1 KiB
//     piece per wave per K-step, up to five stages in flight: with a single block per CU the ring depth is what hides
//     the L2 latency), fragments double-buffered in registers.  The K-steps of consecutive column tiles form ONE
//     pipeline: the DMA of the next tiles' stages is in flight while a tile's epilogue stores drain;
//   * the epilogue needs no LDS: bias and activation in registers, neighbouring lanes exchange one value so every
//     lane stores a packed pair of fp16 (rows stay 64-byte contiguous per 32-column tile).
// vmcnt bookkeeping: a wave's epilogue issues exactly 16 stores (full tiles only: rows % 128 == 0, Nout % 128 == 0),
// which sit in the in-order vmcnt queue behind the DMA pieces already in flight; the K-steps after an epilogue whose
// awaited piece was issued before it wait with that count added instead of draining the stores.
#include "gemm_dma_common.h"

#include <stdlib.h>

namespace {

using dma::dma16;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SBK = 32;            // k per stage
constexpr int S_TILE = 2048;       // floats of one [128][32] fp16 tile (8 KiB)
constexpr int S_STORES = 16;       // store instructions a wave's epilogue issues (32 x 64 wave tile)

// s_waitcnt vmcnt(N + S_STORES * E) lgkmcnt(0) for run-time N in [0, S_NS - 2] pieces and E in [0, 2] epilogues' stores
// (PW = DMA pieces a wave issues per stage)
template <int E, int PW>
__device__ __forceinline__ void wait_pieces_e(int n) {
    switch (n) {
        case 0: dma::wait_vm_lgkm0<S_STORES * E + 0>(); break;
        case 1: dma::wait_vm_lgkm0<S_STORES * E + 1 * PW>(); break;
        case 2: dma::wait_vm_lgkm0<S_STORES * E + 2 * PW>(); break;
        case 3: dma::wait_vm_lgkm0<S_STORES * E + 3 * PW>(); break;
        default: dma::wait_vm_lgkm0<S_STORES * E + 4 * PW>(); break;
    }
}
template <int PW>
__device__ __forceinline__ void wait_pieces(int n, int epilogues) {
    if (epilogues == 0) wait_pieces_e<0, PW>(n);
    else if (epilogues == 1) wait_pieces_e<1, PW>(n);
    else wait_pieces_e<2, PW>(n);
}

// value of the neighbouring lane (lane ^ 1): one DPP move (quad_perm [1, 0, 3, 2])
__device__ __forceinline__ float swap_pair(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
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

// BM rows per block: 64 (4 waves as 2 x 2, two blocks per CU — two independent barrier domains whose phases
// interleave on the CU) or 128 (8 waves as 4 x 2, one block per CU, deeper ring).  Wave tile 32 x 64 either way.
template <int BM, int S_NS>
__global__ __launch_bounds__(BM * 4, 1) void gemm_f16_astat_kernel(GemmArgs g) {
    constexpr int S_NT = BM * 4;       // threads: one wave per 32 rows x 64 columns
    constexpr int MW = BM / 32;        // waves along the rows
    constexpr int PW = 8 / (2 * MW);   // 1 KiB pieces of a W stage per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ASTAMP(0);
    const int nk = g.K / SBK;
    constexpr int P_TILE = BM * 16;                    // floats of one [BM][32] fp16 panel stage
    float* panel = smem;                               // [nk][P_TILE]
    float* ring = smem + nk * P_TILE;                  // [S_NS][S_TILE]
    float* bias_lds = ring + S_NS * S_TILE;            // [Nout] (zeros where a segment has no bias)

    const int tilesM = g.rows / BM, tilesN = g.Nout / 128;
    const int b = blockIdx.x / tilesM, rt = blockIdx.x % tilesM, m0 = rt * BM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / MW, w = wave % MW;           // rows 32w .. 32w+31, columns 64 wn .. 64 wn + 63
    const int r = lane & 31, h = lane >> 5;

    // ---- W image pieces: wave `wave` moves pieces PW * wave .. of every 8 KiB stage
    const float* img = static_cast<const float*>(g.w_img) + (PW * wave) * 256 + lane * 4;
    const int nsteps = tilesN * nk;                    // flattened step s = column tile * nk + kt
    auto issue = [&](int s) {
#pragma unroll
        for (int q = 0; q < PW; ++q)
            dma16(img + (size_t)s * S_TILE + q * 256, ring + (s % S_NS) * S_TILE + (PW * wave + q) * 256);
    };
#pragma unroll
    for (int p = 0; p < S_NS; ++p)
        if (p < nsteps) issue(p);

    // ---- the A panel: fp16(x * pa + po), laid out like a stream of gemm_f16_dma.hip A16 stages
    {
        const float* xb = g.A + ((size_t)b * g.rows + m0) * g.lda;
        const float* pa = g.pro_a ? g.pro_a + (size_t)b * g.K : nullptr;
        const float* po = g.pro_o ? g.pro_o + (size_t)b * g.K : nullptr;
        const int c8n = g.K / 8;
        // batches of 4 items per thread: 8 independent 16-byte loads in flight before the first use
        const int nitems = BM * c8n;
        for (int i0 = tid; i0 < nitems; i0 += 4 * S_NT) {
            f32x4 x0[4], x1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + u * S_NT, nitems - 1);
                const int row = i / c8n, c8 = i - row * c8n;
                const float* src = xb + (size_t)row * g.lda + c8 * 8;
                x0[u] = GECCO_NT_LOAD(reinterpret_cast<const f32x4*>(src));
                x1[u] = GECCO_NT_LOAD(reinterpret_cast<const f32x4*>(src + 4));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * S_NT;
                if (i >= nitems) break;
                const int row = i / c8n, c8 = i - row * c8n;
                if (pa) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(pa + c8 * 8), a1 = *reinterpret_cast<const f32x4*>(pa + c8 * 8 + 4);
                    const f32x4 o0 = *reinterpret_cast<const f32x4*>(po + c8 * 8), o1 = *reinterpret_cast<const f32x4*>(po + c8 * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x0[u][e] = __builtin_fmaf(x0[u][e], a0[e], o0[e]);
                        x1[u][e] = __builtin_fmaf(x1[u][e], a1[e], o1[e]);
                    }
                }
                f16x8 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = (_Float16)x0[u][e];
                    v[4 + e] = (_Float16)x1[u][e];
                }
                const int kt = c8 >> 2, ch = (c8 & 3) ^ ((row >> 2) & 3);
                *reinterpret_cast<u32x4*>(panel + kt * P_TILE + row * 16 + ch * 4) = __builtin_bit_cast(u32x4, v);
            }
        }
        for (int n = tid; n < g.Nout; n += S_NT) {
            const bool seg2 = g.C2 != nullptr && n >= g.n_split;
            const float* bp = seg2 ? g.bias2 : g.bias;
            bias_lds[n] = bp ? bp[seg2 ? n - g.n_split : n] : 0.f;
        }
    }
    const bool has_act = g.act != 0, act_norm = g.act == 1;
    const float neg_inv_2a2 = has_act ? -1.0f / (2.0f * g.alpha[0] * g.alpha[0]) : 0.f;

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // fragment addressing (float offsets): A rows of this wave in a panel stage, W rows of its two column tiles
    const int ra = w * 32 + r;
    int aoff[2], boff[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) aoff[c] = ra * 16 + (((2 * h + c) ^ ((ra >> 2) & 3)) << 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rb = (wn * 2 + j) * 32 + r;
#pragma unroll
        for (int c = 0; c < 2; ++c) boff[j][c] = rb * 16 + (((2 * h + c) ^ ((rb >> 2) & 3)) << 2);
    }
    f16x8 fa[2][2], fb[2][2][2];
    auto load_frags = [&](int kt, int slot, int f) {
        const float* pa_ = panel + kt * P_TILE;
        const float* st = ring + slot * S_TILE;
#pragma unroll
        for (int c = 0; c < 2; ++c) fa[f][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(pa_ + aoff[c]));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                fb[f][j][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(st + boff[j][c]));
    };

    // ---- epilogue of one column tile, from registers: bias, activation, fp16, packed-pair stores
    auto epilogue = [&](int ct) {
        const int n0 = ct * 128;
        const bool seg2 = g.C2 != nullptr && n0 >= g.n_split;
        _Float16* Cb = reinterpret_cast<_Float16*>(seg2 ? g.C2 : g.C);
        const int ldc = seg2 ? g.ldc2 : g.ldc;
        const int nseg0 = seg2 ? n0 - g.n_split : n0;
        _Float16* base = Cb + ((size_t)b * g.rows + m0 + w * 32) * ldc + nseg0 + wn * 64;
        const bool odd = lane & 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float bias = bias_lds[n0 + wn * 64 + j * 32 + r];
            f32x16 v = acc[j];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                v[e] += bias;
                if (has_act) v[e] = gauss_act(v[e], neg_inv_2a2, act_norm);
                acc[j][e] = 0.f;
            }
#pragma unroll
            for (int e0 = 0; e0 < 16; e0 += 2) {
                // even lane keeps register e0 (its column and the right neighbour's), odd lane register e0 + 1
                const float recv = swap_pair(odd ? v[e0] : v[e0 + 1]);
                f16x2 pk;
                pk[0] = (_Float16)(odd ? recv : v[e0]);
                pk[1] = (_Float16)(odd ? v[e0 + 1] : recv);
                const int row = mfma_row(odd ? e0 + 1 : e0, h);
#ifdef ASTAT_NOSTORE   // diagnostic: keep the values alive, write (almost) nothing
                if (pk[0] == (_Float16)123.0f) *reinterpret_cast<f16x2*>(base) = pk;
#else
                *reinterpret_cast<f16x2*>(base + (size_t)row * ldc + j * 32 + (r & ~1)) = pk;
#endif
            }
        }
    };

    ASTAMP(1);
    __syncthreads();   // panel and bias are in LDS; vmcnt drained: the first S_NS stages have landed
    ASTAMP(2);
    load_frags(0, 0, 0);
    // The K-steps of all column tiles as one software pipeline; every index advances incrementally (no division in the
    // loop) and the steady state takes one of three fixed waits.
    unsigned epi_hist = 0;   // bit i: this wave issued epilogue stores at the end of K-step s - 1 - i (in-order vmcnt queue)
    int s = 0, slot_next = 1 % S_NS;   // flattened step; ring slot of stage s + 1
    auto kstep = [&](int ct, int kt, int cur, bool may_end) {
        // own piece of stage s + 1 landed; younger stages, and epilogue stores issued after it, may stay in flight
        // (the awaited piece was issued at the top of K-step s + 1 - S_NS: epilogues since then queue behind it)
        const int rem = nsteps - 1 - s;
        const int epis = __builtin_popcount(epi_hist & ((1u << (S_NS - 1)) - 1u));
        if (rem >= S_NS - 1) {
            if (epis == 0) dma::wait_vm_lgkm0<(S_NS - 2) * PW>();
            else if (epis == 1) dma::wait_vm_lgkm0<(S_NS - 2) * PW + S_STORES>();
            else dma::wait_vm_lgkm0<(S_NS - 2) * PW + 2 * S_STORES>();
        } else {
            wait_pieces<PW>(rem >= 1 ? rem - 1 : 0, epis);   // the tail: fewer stages left than ring slots
        }
        // the fragments of this step were read during the previous one and the wait above covered them: say so in a
        // form the compiler's wait-count pass sees (an empty asm "redefines" the registers), or it parks its own
        // lgkmcnt(0) in front of the first MFMA — behind the NEXT step's reads issued below, exposing their latency
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            asm volatile("" : "+v"(fa[cur][c]));
#pragma unroll
            for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(fb[cur][j][c]));
        }
#ifndef ASTAT_NOBARRIER
        __builtin_amdgcn_s_barrier();
#endif
#ifndef ASTAT_NODMA
        if (rem >= S_NS) issue(s + S_NS);
#endif
#ifndef ASTAT_NOFRAGS
        if (rem >= 1) load_frags(kt + 1 == nk ? 0 : kt + 1, slot_next, cur ^ 1);
#endif
#ifndef ASTAT_NOMFMA
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][c], fb[cur][j][c], acc[j], 0, 0, 0);
#endif
        const bool last = may_end && kt == nk - 1;
        if (last) epilogue(ct);
        epi_hist = (epi_hist << 1) | (last ? 1u : 0u);
        ++s;
        slot_next = slot_next + 1 == S_NS ? 0 : slot_next + 1;
    };
    for (int ct = 0; ct < tilesN; ++ct)
        for (int kt = 0; kt < nk; kt += 2) {   // nk is even: a column tile always ends on the second step of a pair
            kstep(ct, kt, 0, false);
            kstep(ct, kt + 1, 1, true);
        }
    ASTAMP(3);
}

}  // namespace

bool gemm_f16_astat_supported(const GemmArgs& g) {
    return g.c_f16 && !g.a_f16 && !g.residual && !g.stats && g.w_img && g.rows >= 128 && !(g.rows % 128) &&
           !(g.Nout % 128) && !(g.K % (2 * SBK)) && g.K >= 4 * SBK && g.K <= 384 && !(g.lda & 3) && !(g.ldc & 1) &&
           (!g.C2 || (!(g.n_split % 128) && !(g.ldc2 & 1) && g.n_split > 0 && g.n_split < g.Nout)) &&
           ((g.pro_a == nullptr) == (g.pro_o == nullptr));
}

template <int BM, int NS>
static int astat_launch_t(const GemmArgs& g, hipStream_t st) {
    const int nk = g.K / SBK;
    const size_t lds = ((size_t)nk * BM * 16 + NS * S_TILE + g.Nout) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_astat_kernel<BM, NS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL((gemm_f16_astat_kernel<BM, NS>), dim3(g.B * (g.rows / BM)), dim3(BM * 4), lds, st, g);
    return (int)hipGetLastError();
}

int gemm_f16_astat_launch(const GemmArgs& g, hipStream_t st) {
    static int bm = 0;
    if (!bm) {
        const char* e = getenv("GECCO_ASTAT_BM");
        bm = (e && atoi(e) == 128) ? 128 : 64;
    }
    return bm == 128 ? astat_launch_t<128, 6>(g, st) : astat_launch_t<64, 3>(g, st);
}
