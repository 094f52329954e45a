// The 64-inducer chain of one BroadcastingLayer in ONE launch, fp16 mode (precision 2), gfx950.
//
// Between the pool attention and the unpool attention of a layer (reference models/set_transformer.py:99-117) the
// 64 inducer states of every sample go through
//
//     merged = softmax-merge of the pool partials                     (pool_merge_kernel)
//     h0     = merged @ Wpo^T                                          (pool.out_proj)
//     u      = act(AdaGN_1(h0) @ W0^T + b0)                            (norm_1, broadcast.mlp.0 + GaussianActivation)
//     h2     = u @ W2^T + b2                                           (broadcast.mlp.2)
//     h      = AdaGN_2(h2)                                             (norm_2; the state a cached upsample re-uses)
//     kvh    = h @ Wkv^T + bkv                                         (unpool.in_proj, rows C..3C)
//
// — eight launches of 5-14 us each on 64 rows per sample, every one of them latency-bound (a 4096 x 384 x 384 product
// is 1.2 GFLOP).  Everything is per-sample (GroupNorm statistics included), so one block per sample runs the whole
// chain with the activations in LDS / registers and ONE linear stream of weight tiles:
//   * the four fp16 weight images (gemm_f16_dma.hip's 8 KiB blocks, one per 128-column tile and 32-k step) lie
//     back to back in the layer's image in the order the chain consumes them; the block streams them through an
//     11-slot global_load ... lds ring that keeps ~80 KiB in flight across GEMM boundaries and epilogues;
//   * 8 waves (two per SIMD) as 2 x 4 of 32 x 32 per 128-column tile, 64-k steps (two weight blocks per barrier
//     round); the A operand (64 rows x C, fp16, rows padded by 16 B) sits in LDS;
//   * every GEMM's output stays in registers until its last column tile: GroupNorm column sums come from the
//     accumulators (per-lane columns), the AdaGN apply and the fp16 rounding happen on them, neighbouring lanes
//     exchange one DPP move so the next operand is written as packed pairs;
//   * the 2C-wide hidden layer never exists as a whole in LDS: it is held as packed fp16 in 32 registers per column
//     tile and enters the A buffer in K-halves of C; mlp.2 accumulates all its column tiles over half 0, then half 1
//     (ascending k per output element, as in the stand-alone kernels).
// Rounding points are those of the stand-alone chain (fp16 operands formed by one fma + one rounding, fp32
// accumulation in the same k order, fp32 h and kvh); GroupNorm column sums are added in a different order (fp32).
#include "gemm_dma_common.h"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace {

using dma::dma16;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int CH_TILE = 2048;           // floats per 8 KiB weight block
constexpr int CH_NT = 512;              // threads per block
constexpr int CH_LDS_BYTES = 160 * 1024;

constexpr int chain_par_floats(int C, int WD) { return WD + C + 2 * C + 4 * C + 2 * C + 128; }
constexpr int chain_ns(int C, int WD) {
    const int left = CH_LDS_BYTES - 64 * (2 * C + 16) - 4 * chain_par_floats(C, WD);
    return left / (CH_TILE * 4) > 12 ? 12 : left / (CH_TILE * 4);
}
constexpr size_t chain_lds_bytes(int C, int WD) {
    return (size_t)chain_ns(C, WD) * CH_TILE * 4 + 64 * (2 * C + 16) + 4 * chain_par_floats(C, WD);
}

__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

__device__ __forceinline__ unsigned swap_pair(unsigned v) {   // value of lane ^ 1 (DPP quad_perm [1, 0, 3, 2])
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}

// LDS writes of this wave have landed, then the block barrier (no vmcnt wait: the weight ring stays in flight)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// two fp32 -> one dword of two fp16, each rounded on its own (the asm keeps the compiler from folding a preceding
// fma into v_fma_mixlo_f16, which would round once where every other kernel of the path rounds twice)
__device__ __forceinline__ unsigned pack2(float v0, float v1) {
    asm volatile("" : "+v"(v0), "+v"(v1));
    f16x2 p;
    p[0] = (_Float16)v0;
    p[1] = (_Float16)v1;
    return __builtin_bit_cast(unsigned, p);
}

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

#ifdef CHAIN_STAMPS   // diagnostic build (tools/probe): per-block s_memtime stamps of the phases
__device__ unsigned long long g_chain_stamps[256 * 16];
#define CSTAMP(i)                                                                                              \
    do {                                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 256) g_chain_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define CSTAMP(i)
#endif

// TWO (the mixed mode, option "chain2"): two-term fp16 weights W = fp16(W) + fp16(W - fp16(W)).  The stream carries, per column
// tile, the NK blocks of the hi image and then the NK blocks of the lo image (f16_image_item_hilo); a tile's K pass simply runs
// twice over the same A operand, into the same accumulator: x . W_hi + x . W_lo.  The rounding of the WEIGHTS — the same for
// every point of a cloud, the part of this chain's fp16 error that reaches the output (tools/experiments/precision_search.py:
// chain = x2a holds F_x at 1.0e-4 .. 1.3e-4, chain = fp16 at 5e-4) — drops from 2^-12 to 2^-23; the activations stay one-term.
//
// CL (option "chaincl"): a CLUSTER of CL = NT1 blocks per sample.  One block per sample streams all 4 MB of a layer's chain weights through
// ONE CU's fill path (~33 B/clk: 87 us of the 109); in the cluster, block cb computes column tile cb of the C-wide products and tiles
// cb, NT1 + cb of the 2C-wide ones — with NH = 2 every product is a whole number of NT1-tile groups, so block cb's weight stream is tile
// units cb, cb + NT1, ..., cb + 6 NT1 of the same image: a third of the bytes per CU.  What the other blocks computed arrives through L2 as
// REGISTER IMAGES: a block stores the accumulators (or packed fp16 pairs) of its tile exactly as its threads hold them ([tile][quad][thread]
// [4 dwords]: whole 1 KiB rows per store instruction), thread i of every other block loads what thread i of the owner held, and from there on
// the code — GroupNorm column sums, coefficient passes, the A-buffer writes — is the one-block kernel's, on the same values in the same
// order: the same bits.  Three hand-offs per layer (h0, u, h2), each: write-through (sc1) 16-byte stores, every wave's vmcnt(0), the block
// barrier, one agent-scope add on the sample's counter; the readers poll it with one lane, pass a block barrier and load sc1
// (MI355X_MICROARCH.md, inter-workgroup visibility: the counter / sc1 row).  The counters are zeroed by the host once per forward (one
// memset node for all layers) and only ever count up inside a launch.  Blocks of a cluster have consecutive ids: dispatch is in order, so a
// block only ever waits for blocks dispatched before it or right behind it — at most one incomplete cluster per launch holds CUs.
template <int NT1, int NH, bool TWO, int CL = 1>
__global__ __launch_bounds__(CH_NT, 1) void inducer_chain_f16_kernel(ChainArgs g) {
    static_assert(CL == 1 || (CL == NT1 && NH == 2), "cluster = one block per column tile of a C-wide product");
    constexpr int C = 128 * NT1, WD = C * NH, NTW = NT1 * NH, NT2 = 2 * NT1;
    constexpr int NK = C / 32;                  // weight blocks (32 k) per column tile of a K = C operand
    constexpr int RS = 2 * C + 16;              // bytes per A row: 16 B of padding -> conflict-free ds_read_b128
    constexpr int WT = TWO ? 2 : 1;             // K passes (weight terms) per column tile
    constexpr int S1 = WT * NT1 * NK, S2 = WT * NTW * NK, S3 = WT * NTW * NK, S4 = WT * NT2 * NK;
    // AST (cluster form, d <= 384): a wave's A fragments (its 32 rows x C: C / 4 registers) are read from the A buffer ONCE per operand and
    // stay in registers over the hi | lo passes and column tiles that share it — a step then reads only the weight fragments out of LDS
    // (32 KiB instead of 64 KiB per step: the GEMM phases of this kernel are LDS-read-bound, profiles/r04z_chain_cluster_stamps.txt)
    constexpr bool AST = CL > 1 && NT1 <= 3;
    constexpr int SEG = WT * NK;                // blocks of one tile unit of the stream
    constexpr int S_TOTAL = CL == 1 ? S1 + S2 + S3 + S4 : 7 * SEG;   // blocks THIS block streams
    constexpr int NS = chain_ns(C, WD);
    constexpr int AHEAD = NS - 4;               // DMA pieces (one per block per wave) that may stay in flight at a wait
    static_assert(NS >= 6 && S_TOTAL >= NS && NK % 4 == 0, "ring");
    static_assert(C <= CH_NT, "one channel per thread in the coefficient passes");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ring = smem;
    char* abuf = reinterpret_cast<char*>(smem + NS * CH_TILE);
    float* lb0 = reinterpret_cast<float*>(abuf + 64 * RS);   // mlp.0 bias
    float* lb2 = lb0 + WD;                                     // mlp.2 bias
    float* lbk = lb2 + C;                                      // unpool k|v bias
    float* red = lbk + 2 * C;                                  // [2 row halves][sum, sum of squares][C]
    float* coef = red + 4 * C;                                 // a[C] | o[C]
    float* gm = coef + 2 * C;                                  // mean[G] | rstd[G]

    // 8 waves = 2 per SIMD (one wave's issue stalls hide under the other's): 2 row halves x 4 column quarters of a
    // 128-column tile, one 32 x 32 accumulator per wave and tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const int b = CL == 1 ? blockIdx.x : blockIdx.x / CL;
    const int cb = CL == 1 ? 0 : blockIdx.x - b * CL;       // this block's column tile (cluster form)
    const bool odd = lane & 1;
    const unsigned psel = odd ? 0x03020706u : 0x05040100u;   // v_perm_b32 over {neighbour, own}
    CSTAMP(0);

    // ---- the weight stream: consecutive 8 KiB blocks (api.hip writes the mlp.2 image K-half by K-half, so the
    // order the chain consumes is the order in memory); buffer_load ... lds with the block offset in an SGPR, wave w
    // moves piece w (1 KiB) of every block
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w_stream), 0, 0x7fffffff, 0x00020000);
    const unsigned voff = (unsigned)(wave * 256 + lane * 4) * 4u;
    unsigned soff = (unsigned)cb * (SEG * CH_TILE * 4u);     // byte offset of the next block to issue
    int islot = 0;         // its ring slot
    int issued = 0;
    int seg_left = SEG;    // cluster form: blocks left in the tile unit being issued
    auto issue = [&]() {
#ifndef CHAIN_DIAG_NODMA
        dma16_buf(wrsrc, voff, soff, ring + islot * CH_TILE + wave * 256);
#endif
        soff += CH_TILE * 4u;
        if (CL > 1 && --seg_left == 0) {   // on to tile unit + NT1
            seg_left = SEG;
            soff += (unsigned)(CL - 1) * (SEG * CH_TILE * 4u);
        }
        islot = islot + 1 == NS ? 0 : islot + 1;
        ++issued;
    };
    // ---- prologue.  Every global read of it is issued in batches of independent loads (one round trip per batch,
    // not one per element): biases and AdaGN parameters first, then the weight ring's first NS - 2 blocks (younger in
    // the in-order vmcnt, so waiting for the parameters does not wait for them), then the pool partials.
    float sc[2], sh[2];   // AdaGN scale / shift of channel tid, norm_1 and norm_2
    {
        constexpr int NB0 = (WD + CH_NT - 1) / CH_NT, NBK = (2 * C + CH_NT - 1) / CH_NT;
        float v0[NB0], vk[NBK];
#pragma unroll
        for (int q = 0; q < NB0; ++q) v0[q] = (g.b0 && tid + CH_NT * q < WD) ? g.b0[tid + CH_NT * q] : 0.f;
        const float v2 = (g.b2 && tid < C) ? g.b2[tid] : 0.f;
#pragma unroll
        for (int q = 0; q < NBK; ++q) vk[q] = (g.bkv && tid + CH_NT * q < 2 * C) ? g.bkv[tid + CH_NT * q] : 0.f;
#pragma unroll
        for (int nrm = 0; nrm < 2; ++nrm) {
            const float* sb = nrm ? g.n2_scale_b : g.n1_scale_b;
            const float* bb = nrm ? g.n2_bias_b : g.n1_bias_b;
            const bool on = sb && tid < C;
            sc[nrm] = on ? sb[tid] : 1.f;
            sh[nrm] = on ? bb[tid] : 0.f;
        }
        for (int j = 0; j < g.ctx_dim; ++j) {
            const float tj = g.t[(size_t)b * g.ctx_dim + j];
#pragma unroll
            for (int nrm = 0; nrm < 2; ++nrm) {
                const float* sw = nrm ? g.n2_scale_w : g.n1_scale_w;
                const float* bw = nrm ? g.n2_bias_w : g.n1_bias_w;
                if (sw && tid < C) {
                    sc[nrm] += tj * sw[(size_t)tid * g.ctx_dim + j];
                    sh[nrm] += tj * bw[(size_t)tid * g.ctx_dim + j];
                }
            }
        }
#pragma unroll 1
        for (int p = 0; p < NS - 2; ++p) issue();
        CSTAMP(1);
#pragma unroll
        for (int q = 0; q < NB0; ++q)
            if (tid + CH_NT * q < WD) lb0[tid + CH_NT * q] = v0[q];
        if (tid < C) lb2[tid] = v2;
#pragma unroll
        for (int q = 0; q < NBK; ++q)
            if (tid + CH_NT * q < 2 * C) lbk[tid + CH_NT * q] = vk[q];
    }

    // ---- merged[i, hh*HD + d] = sum_s f_s O_s / sum_s f_s l_s (pool_merge_kernel's arithmetic), rounded to fp16 into
    // the A buffer.  Batches of items per thread: all of a batch's loads are independent.
    auto merge = [&](auto nsp_tag) {
        constexpr int NSP = decltype(nsp_tag)::value;
        // all of a thread's items in one round trip while the partials fit the register file (6 NSP registers per item)
        constexpr int ITEMS = 64 * (C / 4) / CH_NT, BATCH = ITEMS * NSP <= 24 ? ITEMS : (NSP <= 4 ? 4 : 2);
        static_assert(64 * (C / 4) % CH_NT == 0 && ITEMS % BATCH == 0, "items per thread");
        const int HD = C / g.H;
        // c / HD for c < 512, HD >= 4 by multiply-shift: exact, the fractional part of c / HD is a multiple of 1 / HD
        const unsigned hd_magic = (1u << 20) / (unsigned)HD + 1u;
#pragma unroll 1
        for (int k0 = 0; k0 < ITEMS; k0 += BATCH) {
            float ml[BATCH][NSP][2];
            f32x4 po[BATCH][NSP];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int idx = tid + CH_NT * (k0 + k);
                const int i = idx / (C / 4), c = (idx % (C / 4)) * 4;
                const int hh = (int)(((unsigned)c * hd_magic) >> 20);
                const size_t base = ((size_t)(b * g.H + hh) * NSP) * 64 + i;
#pragma unroll
                for (int s = 0; s < NSP; ++s) {
                    const size_t p = base + (size_t)s * 64;
                    ml[k][s][0] = g.part_ml[p * 2];
                    ml[k][s][1] = g.part_ml[p * 2 + 1];
                    po[k][s] = *reinterpret_cast<const f32x4*>(g.part_o + p * HD + (c - hh * HD));
                }
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int idx = tid + CH_NT * (k0 + k);
                const int i = idx / (C / 4), c = (idx % (C / 4)) * 4;
                float M = -INFINITY;
#pragma unroll
                for (int s = 0; s < NSP; ++s) M = fmaxf(M, ml[k][s][0]);
                f32x4 num = {0.f, 0.f, 0.f, 0.f};
                float den = 0.f;
#pragma unroll
                for (int s = 0; s < NSP; ++s) {
                    const float f = (ml[k][s][0] == -INFINITY) ? 0.f : exp2f(ml[k][s][0] - M);
                    num += f * po[k][s];
                    den += f * ml[k][s][1];
                }
                f16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (_Float16)(num[e] / den);
                *reinterpret_cast<u32x2*>(abuf + i * RS + 2 * c) = __builtin_bit_cast(u32x2, v);
            }
        }
    };
    switch (g.nsplit) {
        case 1: merge(std::integral_constant<int, 1>{}); break;
        case 2: merge(std::integral_constant<int, 2>{}); break;
        case 4: merge(std::integral_constant<int, 4>{}); break;
        default: merge(std::integral_constant<int, 8>{}); break;
    }
    CSTAMP(2);
    lds_barrier();

    // ---- the step machine.  One weight block = 8 KiB = 32 k of one 128-column tile; a STEP consumes two of them (64 k:
    // 4 MFMAs per wave per barrier round).  The bookkeeping is increments and immediates: the fragments of the next
    // step are read (second register set) before the MFMAs of this one are issued, ring slots and the stream offset
    // advance by increments.
    //   D = issued - s (blocks) is NS at a primed step: the wait for blocks s + 2, s + 3 leaves the NS - 4 younger
    //   pieces in flight (in-order vmcnt; epilogue stores issued in between only make it wait for more), the barrier
    //   frees the slots of blocks s, s + 1 (every wave's reads of them returned before it) and two blocks are issued
    //   into them.  Where the A operand changes the pipeline drains (D = NS - 2) and is primed again: same wait, the
    //   barrier frees blocks s - 2, s - 1, two issues.  Tiles that reach the end of the stream (TAIL) stop issuing
    //   and wait for everything.
    int s = 0;
    int rslot = 0;             // ring slot of the next block to read fragments from
    const char* arow = abuf + (wm * 32 + r) * RS + 32 * h;
    int boff[2];
    {
        const int rb = wn * 32 + r;
#pragma unroll
        for (int c = 0; c < 2; ++c) boff[c] = rb * 16 + (((2 * h + c) ^ ((rb >> 2) & 3)) << 2);
    }
    f16x8 fa[2][2][2], fb[2][2][2];   // [set][block][chunk]
    f16x8 areg[AST ? NK : 1][2];      // AST: the stationary A fragments [block][chunk]
    auto load_areg = [&]() {
#pragma unroll
        for (int k = 0; k < (AST ? NK : 0); ++k)
#pragma unroll
            for (int c = 0; c < 2; ++c) areg[k][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(arow + k * 64 + 16 * c));
    };
    auto load_frags = [&](auto set_tag, int kt) {   // blocks kt, kt + 1 of the tile
        constexpr int set = decltype(set_tag)::value;
#ifdef CHAIN_DIAG_NOFRAGS
        if (kt >= 0) return;
#endif
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float* st = ring + rslot * CH_TILE;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if constexpr (!AST) fa[set][q][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(arow + (kt + q) * 64 + 16 * c));
                fb[set][q][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(st + boff[c]));
            }
            rslot = rslot + 1 == NS ? 0 : rslot + 1;
        }
    };
    auto wait_blocks = [&](auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        if (TAIL && issued >= S_TOTAL) dma::wait_vm_lgkm0<0>();
        else dma::wait_vm_lgkm0<AHEAD>();
    };
    auto issue2 = [&](auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (!TAIL || issued < S_TOTAL) issue();
    };
    // one step: set `cur` holds its fragments; has_next: the following step reads the same A operand
    auto kstep = [&](auto cur_tag, auto tail_tag, bool has_next, int kt_next, f32x16& a0, auto kt_tag) {
        constexpr int cur = decltype(cur_tag)::value;
        constexpr int KT = decltype(kt_tag)::value;   // AST: this step's first block (its A fragments are areg[KT], areg[KT + 1])
        if (has_next) {
            wait_blocks(tail_tag);
            // this step's fragments were read during the previous one and the wait above covered them: "redefine"
            // them so the compiler's wait-count pass does not put an lgkmcnt(0) — which would also wait for the
            // reads issued below — in front of the first MFMA
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if constexpr (!AST) asm volatile("" : "+v"(fa[cur][q][c]));
                    asm volatile("" : "+v"(fb[cur][q][c]));
                }
#ifndef CHAIN_DIAG_NOBARRIER
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
            issue2(tail_tag);
            load_frags(std::integral_constant<int, cur ^ 1>{}, kt_next);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#ifdef CHAIN_DIAG_NOMFMA
                a0[0] += (float)fa[cur][q][c][0] + (float)fb[cur][q][c][0];
#else
                if constexpr (AST) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(areg[KT + q][c], fb[cur][q][c], a0, 0, 0, 0);
                else a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][q][c], fb[cur][q][c], a0, 0, 0, 0);
#endif
            }
    };
    auto tile_steps = [&](auto tail_tag, bool first, bool more, f32x16& a0) {
        constexpr std::integral_constant<int, 0> set0{};
        constexpr std::integral_constant<int, 1> set1{};
        if (first) {   // prime: the fragments of blocks s, s + 1
            wait_blocks(tail_tag);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue2(tail_tag);
            if constexpr (AST) load_areg();
            load_frags(set0, 0);
        }
        if constexpr (AST) {   // every step with its block index at compile time (register-indexed A fragments)
            static_for<NK / 2>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value;
                constexpr bool last = i + 1 == NK / 2;
                kstep(std::integral_constant<int, (i & 1)>{}, tail_tag, last ? more : true, last ? 0 : 2 * (i + 1), a0,
                      std::integral_constant<int, 2 * i>{});
            });
        } else {
            constexpr std::integral_constant<int, 0> k0{};
#pragma unroll 1
            for (int kt = 0; kt < NK - 4; kt += 4) {
                kstep(set0, tail_tag, true, kt + 2, a0, k0);
                kstep(set1, tail_tag, true, kt + 4, a0, k0);
            }
            kstep(set0, tail_tag, true, NK - 2, a0, k0);
            kstep(set1, tail_tag, more, 0, a0, k0);
        }
        s += NK;
    };
    // the NK blocks of one column tile; first: the A operand is new (prime the pipeline); more: another tile over
    // the same A operand follows
    auto run_pass = [&](f32x16& a0, bool first, bool more) {
        if (AST || s + NK + NS > S_TOTAL) tile_steps(std::true_type{}, first, more, a0);   // AST: one (unrolled) form, end-of-stream checks at run time
        else tile_steps(std::false_type{}, first, more, a0);
    };
    auto run_tile = [&](f32x16& a0, bool first, bool more) {
        if (TWO) {   // hi pass, then the lo pass over the same A operand
            run_pass(a0, first, true);
            run_pass(a0, false, more);
        } else {
            run_pass(a0, first, more);
        }
    };
    auto zero = [](f32x16& a) {
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = 0.f;
    };

    // GroupNorm coefficients of the [64 x C] tensor whose per-lane columns are in acc (+ bias already added):
    // column sums -> group mean / rstd (double, as adagn_coeffs_kernel) -> coef = a | o.  Three barriers.
    auto norm_coeffs = [&](f32x16 (&acc)[NT1], int nrm) {
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s1 += acc[t][e];
                s2 += acc[t][e] * acc[t][e];
            }
            s1 += xor32(s1);
            s2 += xor32(s2);
            if (h == 0) {
                const int n = t * 128 + wn * 32 + r;
                red[(wm * 2 + 0) * C + n] = s1;
                red[(wm * 2 + 1) * C + n] = s2;
            }
        }
        lds_barrier();
        const int G = g.G, cpg = C / G;
        if (tid < G) {
            double d1 = 0.0, d2 = 0.0;
            for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
                d1 += (double)(red[c] + red[2 * C + c]);
                d2 += (double)(red[C + c] + red[3 * C + c]);
            }
            const double n = 64.0 * cpg;
            const double mean = d1 / n;
            double var = d2 / n - mean * mean;
            var = var < 0.0 ? 0.0 : var;
            gm[tid] = (float)mean;
            gm[G + tid] = (float)(1.0 / sqrt(var + (double)g.eps));
        }
        lds_barrier();
        if (tid < C) {
            const float mean = gm[tid / cpg], rstd = gm[G + tid / cpg];
            const float sv = nrm ? sc[1] : sc[0], zv = nrm ? sh[1] : sh[0];
            coef[tid] = sv * rstd;
            coef[C + tid] = zv - sv * mean * rstd;
        }
        lds_barrier();
    };
    // packed pair (rows of accumulator registers 2p, 2p+1 of this lane's column) -> (one row, this lane pair's two
    // columns): even lanes keep row 2p, odd lanes row 2p + 1
    auto pair_rows = [&](float v0, float v1) -> unsigned {
        const unsigned own = pack2(v0, v1);
        return __builtin_amdgcn_perm(swap_pair(own), own, psel);
    };
    // dword address of (accumulator pair p, column tile slot tl) in the A buffer
    auto a_dst = [&](int tl, int p) -> unsigned* {
        const int row = wm * 32 + ((2 * p) & 3) + 8 * ((2 * p) >> 2) + 4 * h + (odd ? 1 : 0);
        const int col = tl * 128 + wn * 32 + (r & ~1);
        return reinterpret_cast<unsigned*>(abuf + row * RS + 2 * col);
    };

    // k | v of column tile t (columns 128 t + 32 wn + r of the 2C-wide product, + bias) as the unpool kernel's fp16 image (kernels.h): a lane
    // holds ONE column and 16 keys (rows).  K part (t < NT1): K[key][d], two bytes per key; V part: V^T[d][pos(key)] — the lane's keys
    // 8 q + 4 h + i sit at pos 32 wm + 16 (q >> 1) + 8 h + 4 (q & 1) + i: two runs of 8 halves, two 16-byte stores.  Same roundings as
    // kvh_image_kernel (fp32 sum, one rounding to fp16).
    auto store_kv_img = [&](int t, const f32x16& a0, float bias) {
        const int HD = C / g.H, KS = HD + 8;
        const int n = t * 128 + wn * 32 + r;
        const int c = n < C ? n : n - C;
        const int hh = (int)(((unsigned)c * ((1u << 20) / (unsigned)HD + 1u)) >> 20), d = c - hh * HD;
        unsigned short* base = g.kv_img + ((size_t)b * g.H + hh) * (size_t)(g.kv_img_bytes / 2);
        if (n < C) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                base[(wm * 32 + mfma_row(e, h)) * KS + d] = __builtin_bit_cast(unsigned short, (_Float16)(a0[e] + bias));
        } else {
            unsigned short* vrow = base + 64 * KS + d * 72 + wm * 32 + 8 * h;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                f16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (_Float16)(a0[8 * qq + e] + bias);
                *reinterpret_cast<u32x4*>(vrow + 16 * qq) = __builtin_bit_cast(u32x4, v);
            }
        }
    };

    const bool has_act = g.act != 0;
    const int act_mode = g.act;
    const float neg_inv_2a2 = act_is_gauss(g.act) ? -1.0f / (2.0f * g.alpha[0] * g.alpha[0]) : 0.f;

    if constexpr (CL == 1) {
    // ================= GEMM 1: h0 = merged16 Wpo^T; norm_1; y16 -> A buffer
    f32x16 acc1[NT1];
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
        zero(acc1[t]);
        run_tile(acc1[t], t == 0, t + 1 < NT1);
    }
    CSTAMP(3);
    norm_coeffs(acc1, 0);   // its barriers also order every wave's last A reads before the writes below
    CSTAMP(4);
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
        const int n = t * 128 + wn * 32 + r;
        const float ca = coef[n], co = coef[C + n];
#pragma unroll
        for (int p = 0; p < 8; ++p)
            *a_dst(t, p) = pair_rows(__builtin_fmaf(acc1[t][2 * p], ca, co), __builtin_fmaf(acc1[t][2 * p + 1], ca, co));
    }
    lds_barrier();

    // ================= GEMM 2: u = act(y16 W0^T + b0), kept as packed fp16 pairs in registers
    CSTAMP(5);
    unsigned upk[NTW][8];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        f32x16 a0;
        zero(a0);
        run_tile(a0, t == 0, t + 1 < NTW);
        const float bias = lb0[t * 128 + wn * 32 + r];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float v0 = a0[2 * p] + bias, v1 = a0[2 * p + 1] + bias;
            if (has_act) {
                v0 = act_apply(v0, neg_inv_2a2, act_mode);
                v1 = act_apply(v1, neg_inv_2a2, act_mode);
            }
            upk[t][p] = pair_rows(v0, v1);
        }
    }

    // ================= GEMM 3: h2 = u16 W2^T + b2, K-half by K-half through the A buffer
    CSTAMP(6);
#pragma unroll
    for (int t = 0; t < NT1; ++t) zero(acc1[t]);
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
        lds_barrier();   // every wave is done reading the previous operand
#pragma unroll
        for (int tl = 0; tl < NT1; ++tl)
#pragma unroll
            for (int p = 0; p < 8; ++p) *a_dst(tl, p) = upk[hf * NT1 + tl][p];
        lds_barrier();
#pragma unroll
        for (int t = 0; t < NT1; ++t) run_tile(acc1[t], t == 0, t + 1 < NT1);
    }
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
        const float bias = lb2[t * 128 + wn * 32 + r];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[t][e] += bias;
    }
    CSTAMP(7);
    norm_coeffs(acc1, 1);
    CSTAMP(8);
    // h = AdaGN_2(h2): fp32 to memory (the cacheable inducer state), fp16 into the A buffer
    {
        float* hb = g.h_out + (size_t)b * 64 * C;
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const int n = t * 128 + wn * 32 + r;
            const float ca = coef[n], co = coef[C + n];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float v0 = __builtin_fmaf(acc1[t][2 * p], ca, co);
                const float v1 = __builtin_fmaf(acc1[t][2 * p + 1], ca, co);
                const int row0 = wm * 32 + mfma_row(2 * p, h);
                hb[(size_t)row0 * C + n] = v0;
                hb[(size_t)(row0 + 1) * C + n] = v1;
                *a_dst(t, p) = pair_rows(v0, v1);
            }
        }
    }
    lds_barrier();

    // ================= GEMM 4: kvh = h16 Wkv^T + bkv (fp32, read by the unpool attention)
    CSTAMP(9);
    {
        float* kb = g.kvh + (size_t)b * 64 * 2 * C;
#pragma unroll 1
        for (int t = 0; t < NT2; ++t) {
            f32x16 a0;
            zero(a0);
            run_tile(a0, t == 0, t + 1 < NT2);
            const int n = t * 128 + wn * 32 + r;
            const float bias = lbk[n];
            if (g.kv_img) {
                store_kv_img(t, a0, bias);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) kb[(size_t)(wm * 32 + mfma_row(e, h)) * 2 * C + n] = a0[e] + bias;
            }
        }
    }
    } else {
    // ======================================================================= the cluster form
    typedef unsigned int u32v4 __attribute__((vector_size(16)));
    constexpr int SC1 = 16;   // cache-policy bit of the buffer instructions: sc1 (write-through stores, L1-bypassing loads)
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(g.x1 + (size_t)b * 64 * C, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r3 = __builtin_amdgcn_make_buffer_rsrc(g.x3 + (size_t)b * 64 * C, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(g.xu + (size_t)b * 32 * WD, 0, 0x7fffffff, 0x00020000);
    unsigned* const flags = g.flags + (size_t)b * 8;
    const int tv = tid * 16;
    auto store_acc = [&](const __amdgpu_buffer_rsrc_t& rs, int t, const f32x16& a) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32v4, v), rs, tv, (t * 4 + q) * (CH_NT * 16), SC1);
        }
    };
    auto load_acc = [&](const __amdgpu_buffer_rsrc_t& rs, int t, f32x16& a) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tv, (t * 4 + q) * (CH_NT * 16), SC1));
#pragma unroll
            for (int e = 0; e < 4; ++e) a[4 * q + e] = v[e];
        }
    };
    // every wave's stores have left, the block barrier, one add on the sample's counter k
    auto publish = [&](int k) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(flags + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // all CL blocks of the sample have published k; the barrier stands between the poll and every load of the bytes
    auto await = [&](int k) {
        if (tid == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
            while (__hip_atomic_load(flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)CL) {
                __builtin_amdgcn_s_sleep(4);
                // 20 s (a partner that is merely preempted comes back): counters not zeroed or a lost block — fail the launch, never hang
                if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000000ull) __builtin_trap();
            }
        }
        __syncthreads();
    };

    // ================= GEMM 1: column tile cb of h0 = merged16 Wpo^T; the other tiles from their blocks; norm_1; y16 -> A buffer
    f32x16 acc1[NT1];
    {
        f32x16 own;
        zero(own);
        run_tile(own, true, false);
        CSTAMP(3);
        store_acc(r1, cb, own);
        publish(0);
        await(0);
        CSTAMP(11);
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            if (t == cb) acc1[t] = own;
            else load_acc(r1, t, acc1[t]);
        }
    }
    norm_coeffs(acc1, 0);   // its barriers also order every wave's last A reads before the writes below
    CSTAMP(4);
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
        const int n = t * 128 + wn * 32 + r;
        const float ca = coef[n], co = coef[C + n];
#pragma unroll
        for (int p = 0; p < 8; ++p)
            *a_dst(t, p) = pair_rows(__builtin_fmaf(acc1[t][2 * p], ca, co), __builtin_fmaf(acc1[t][2 * p + 1], ca, co));
    }
    lds_barrier();

    // ================= GEMM 2: tiles cb, NT1 + cb of u = act(y16 W0^T + b0) as packed fp16 pairs; the other tiles from their blocks
    CSTAMP(5);
    unsigned upk[NTW][8];
    {
        unsigned uown[NH][8];
#pragma unroll
        for (int j = 0; j < NH; ++j) {
            f32x16 a0;
            zero(a0);
            run_tile(a0, j == 0, j + 1 < NH);
            const float bias = lb0[(j * NT1 + cb) * 128 + wn * 32 + r];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float v0 = a0[2 * p] + bias, v1 = a0[2 * p + 1] + bias;
                if (has_act) {
                    v0 = act_apply(v0, neg_inv_2a2, act_mode);
                    v1 = act_apply(v1, neg_inv_2a2, act_mode);
                }
                uown[j][p] = pair_rows(v0, v1);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u32v4 v = {uown[j][4 * q], uown[j][4 * q + 1], uown[j][4 * q + 2], uown[j][4 * q + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(v, ru, tv, ((j * NT1 + cb) * 2 + q) * (CH_NT * 16), SC1);
            }
        }
        CSTAMP(14);
        publish(1);
        await(1);
        CSTAMP(12);
#pragma unroll
        for (int j = 0; j < NH; ++j)
#pragma unroll
            for (int tl = 0; tl < NT1; ++tl) {
                if (tl == cb) {
#pragma unroll
                    for (int p = 0; p < 8; ++p) upk[j * NT1 + tl][p] = uown[j][p];
                } else {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const u32v4 v = __builtin_amdgcn_raw_buffer_load_b128(ru, tv, ((j * NT1 + tl) * 2 + q) * (CH_NT * 16), SC1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) upk[j * NT1 + tl][4 * q + e] = v[e];
                    }
                }
            }
    }

    // ================= GEMM 3: column tile cb of h2 = u16 W2^T + b2, K-half by K-half through the A buffer
    CSTAMP(6);
    {
        f32x16 own;
        zero(own);
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
            lds_barrier();   // every wave is done reading the previous operand
#pragma unroll
            for (int tl = 0; tl < NT1; ++tl)
#pragma unroll
                for (int p = 0; p < 8; ++p) *a_dst(tl, p) = upk[hf * NT1 + tl][p];
            lds_barrier();
            run_tile(own, true, false);
        }
        const float bias = lb2[cb * 128 + wn * 32 + r];
#pragma unroll
        for (int e = 0; e < 16; ++e) own[e] += bias;
        CSTAMP(7);
        store_acc(r3, cb, own);
        publish(2);
        await(2);
        CSTAMP(13);
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            if (t == cb) acc1[t] = own;
            else load_acc(r3, t, acc1[t]);
        }
    }
    norm_coeffs(acc1, 1);
    CSTAMP(8);
    // h = AdaGN_2(h2): this block's column tile as fp32 to memory (the cacheable inducer state), all of it as fp16 into the A buffer
    {
        float* hb = g.h_out + (size_t)b * 64 * C;
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const int n = t * 128 + wn * 32 + r;
            const float ca = coef[n], co = coef[C + n];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float v0 = __builtin_fmaf(acc1[t][2 * p], ca, co);
                const float v1 = __builtin_fmaf(acc1[t][2 * p + 1], ca, co);
                if (t == cb) {
                    const int row0 = wm * 32 + mfma_row(2 * p, h);
                    hb[(size_t)row0 * C + n] = v0;
                    hb[(size_t)(row0 + 1) * C + n] = v1;
                }
                *a_dst(t, p) = pair_rows(v0, v1);
            }
        }
    }
    lds_barrier();

    // ================= GEMM 4: column tiles cb, NT1 + cb of kvh = h16 Wkv^T + bkv (fp32, read by the unpool attention)
    CSTAMP(9);
    {
        float* kb = g.kvh + (size_t)b * 64 * 2 * C;
#pragma unroll 1
        for (int j = 0; j < 2; ++j) {
            f32x16 a0;
            zero(a0);
            run_tile(a0, j == 0, j + 1 < 2);
            const int n = (j * NT1 + cb) * 128 + wn * 32 + r;
            const float bias = lbk[n];
            if (g.kv_img) {
                store_kv_img(j * NT1 + cb, a0, bias);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) kb[(size_t)(wm * 32 + mfma_row(e, h)) * 2 * C + n] = a0[e] + bias;
            }
        }
    }
    }
    CSTAMP(10);
}

template <int NT1, int NH, bool TWO, int CL>
int chain_launch_w(const ChainArgs& g, hipStream_t st) {
    constexpr int C = 128 * NT1, WD = C * NH;
    constexpr size_t lds = chain_lds_bytes(C, WD);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(inducer_chain_f16_kernel<NT1, NH, TWO, CL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((inducer_chain_f16_kernel<NT1, NH, TWO, CL>), dim3(g.B * CL), dim3(CH_NT), lds, st, g);
    return (int)hipGetLastError();
}
template <int NT1, int NH>
int chain_launch_t(const ChainArgs& g, hipStream_t st) {
    if constexpr (NT1 > 1) {
        if (g.cluster) {
            if (!g.x1 || !g.x3 || !g.xu || !g.flags) return -9;
            return g.two_term ? chain_launch_w<NT1, NH, true, NT1>(g, st) : chain_launch_w<NT1, NH, false, NT1>(g, st);
        }
    }
    return g.two_term ? chain_launch_w<NT1, NH, true, 1>(g, st) : chain_launch_w<NT1, NH, false, 1>(g, st);
}

}  // namespace

bool inducer_chain_f16_supported(int C, int Wd, int H, int G, int I) {
    if (I != 64 || C % 128 || C > 512 || Wd != 2 * C || H <= 0 || C % H || (C / H) % 4) return false;
    return G > 0 && G <= 64 && C % G == 0;
}

int inducer_chain_f16_launch(const ChainArgs& g, int C, int Wd, hipStream_t st) {
    if (!inducer_chain_f16_supported(C, Wd, g.H, g.G, 64)) return -9;
    if (g.nsplit != 1 && g.nsplit != 2 && g.nsplit != 4 && g.nsplit != 8) return -9;
    switch (C / 128) {
        case 1: return chain_launch_t<1, 2>(g, st);
        case 2: return chain_launch_t<2, 2>(g, st);
        case 3: return chain_launch_t<3, 2>(g, st);
        default: return chain_launch_t<4, 2>(g, st);
    }
}
