// x += mlp.2(GaussianActivation(mlp.0(AdaGN(x)))) in ONE launch, fp16 mode (precision 2), gfx950.
//
// The point-stream MLP of a BroadcastingLayer (reference models/set_transformer.py:165-166, models/mlp.py) was two
// launches — AdaGN + mlp.0 + activation writing the 2C-wide hidden layer as fp16 (gemm_f16_astat.hip), then mlp.2 +
// residual + GroupNorm partials reading it back (gemm_f16_dma.hip): 402 MB of hidden-layer traffic per layer at C2, an
// LDS-DMA A tile per K-step in the second, and two x-sized kernels' worth of launch phases.  Here a block owns 128
// rows from x to x:
//   * y16 = fp16(x * a + o) of the 128 rows is built once into LDS (96 KiB, XOR-swizzled 16-byte chunks);
//   * the hidden layer is produced 128 columns at a time (GEMM a: y16 @ W0[chunk]^T, bias, activation, fp16) and
//     consumed at once as a K-slice of the second product (GEMM b: out += hidden[:, chunk] @ W2[:, chunk]^T, in two
//     K-halves of 64 through an 18 KiB LDS buffer), the 128 x C fp32 result staying in registers (96 per lane);
//   * ONE linear stream of 8 KiB weight blocks (api.hip interleaves the images: W0 tile j | W2 K-slice j half 0 |
//     half 1) through a buffer_load ... lds ring; 8 waves = 4 row groups x 2 column halves, two per SIMD; the step
//     machine of inducer_chain_f16.hip (fragments of block s + 1 read before the MFMAs of block s, counted waits);
//   * epilogue as in gemm_dma_common.h: each wave transposes its 32 x 64 sub-tiles through a private LDS tile (the y16
//     buffer is dead by then) so that the residual x is re-read and the result stored in 16-byte row pieces (4-byte
//     accesses in the accumulator layout cost four times the wave-instructions and made the epilogue as long as a
//     sixth of the matrix work); bias, residual, store, per-(sample, row tile, column) GroupNorm partials.
// Every rounding point, the k order of every accumulation, (sum + b) + x and the order of the partial sums are those of
// the two-launch form: bit-identical x and statistics (tests/test_hip_ops.py::test_mlp_fused_matches_the_two_launch_form).
#include "gemm_dma_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_NT = 512;             // threads per block
constexpr int MF_TILE = 2048;          // floats per 8 KiB weight block
constexpr int MF_RSH = 2 * 64 + 16;    // bytes per row of the hidden K-half buffer (16 B of padding)
constexpr int MF_LDS_BYTES = 160 * 1024;

constexpr int mf_fixed_bytes(int C) { return 128 * 2 * C + 128 * MF_RSH + (2 * C + C) * 4; }
constexpr int mf_ns(int C) {
    const int n = (MF_LDS_BYTES - mf_fixed_bytes(C)) / (MF_TILE * 4);
    return n > 12 ? 12 : n;
}
constexpr size_t mf_lds_bytes(int C) { return (size_t)mf_ns(C) * MF_TILE * 4 + mf_fixed_bytes(C); }

__device__ __forceinline__ void mf_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ unsigned mf_swap_pair(unsigned v) {   // value of lane ^ 1 (DPP quad_perm [1, 0, 3, 2])
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ void mf_lds_barrier() {   // this wave's LDS writes landed, then the block barrier (no vmcnt wait)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// two fp32 -> two fp16 in one dword, each rounded on its own (the asm keeps a preceding fma out of v_fma_mixlo_f16)
__device__ __forceinline__ unsigned mf_pack2(float v0, float v1) {
    asm volatile("" : "+v"(v0), "+v"(v1));
    f16x2 p;
    p[0] = (_Float16)v0;
    p[1] = (_Float16)v1;
    return __builtin_bit_cast(unsigned, p);
}

#ifdef MLPF_STAMPS   // diagnostic build (tools/probe): per-block s_memtime stamps of the phases
__device__ unsigned long long g_mlpf_stamps[2048 * 8];
#define MSTAMP(i)                                                                                                  \
    do {                                                                                                           \
        if (threadIdx.x == 0 && blockIdx.x < 2048) g_mlpf_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define MSTAMP(i)
#endif

template <int NT1>
__global__ __launch_bounds__(MF_NT, 1) void mlp_fused_f16_kernel(MlpArgs g) {
    constexpr int C = 128 * NT1, WD = 2 * C, NCH = WD / 128;
    constexpr int NK = C / 32;                       // weight blocks of one hidden tile (GEMM a)
    constexpr int S_TOTAL = NCH * 2 * NK;            // + 2 halves x NT1 output tiles x 2 blocks = NK per chunk (GEMM b)
    constexpr int NS = mf_ns(C);
    constexpr int AHEAD = NS - 2;                    // DMA pieces (one per block per wave) that may stay in flight at a wait
    constexpr int RSY = 2 * C;                       // bytes per y16 row (swizzled, no padding)
    static_assert(NS >= 4 && S_TOTAL >= NS && AHEAD <= 63, "ring");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ring = smem;
    char* ybuf = reinterpret_cast<char*>(smem + NS * MF_TILE);   // [128][C] fp16, chunk ci of row r at ci ^ (r & 15)
    char* hbuf = ybuf + 128 * RSY;                                 // [128][64] fp16 (+ pad): one K-half of a hidden chunk
    float* lb0 = reinterpret_cast<float*>(hbuf + 128 * MF_RSH);   // mlp.0 bias [WD]
    float* lb2 = lb0 + WD;                                         // mlp.2 bias [C]
    float* lpro = reinterpret_cast<float*>(hbuf);                  // a[C] | o[C] during the A build

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int tilesM = g.rows / 128;
    const int b = blockIdx.x / tilesM, rt = blockIdx.x % tilesM, m0 = rt * 128;
    const bool odd = lane & 1;
    const unsigned psel = odd ? 0x03020706u : 0x05040100u;   // v_perm_b32 over {neighbour, own}
    MSTAMP(0);

    // ---- start-up stagger.  Every block begins with an HBM-bound phase (196 KB of x) followed by ~40 us of matrix
    // work, one block per CU: started together, all 256 CUs ask HBM for 50 MB at once, wait 8 us for it, and leave it
    // idle afterwards — round after round, since every block takes the same time.  The first block of each CU (the
    // first 256 dispatched) is held back by 0 .. 7 x `stagger` cycles, by CU group within its XCD (blockIdx % 8 is the
    // XCD), which spreads each round's burst over 8 offsets; the offsets persist.
    if (g.stagger > 0 && blockIdx.x < 256) {
        const long long until = (long long)__builtin_amdgcn_s_memtime() + (long long)((blockIdx.x >> 3) & 7) * g.stagger;
        while ((long long)__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(8);
    }

    // ---- the weight stream: consecutive 8 KiB blocks, wave w moves piece w (1 KiB) of every block
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w_stream), 0, 0x7fffffff, 0x00020000);
    const unsigned voff = (unsigned)(wave * 256 + lane * 4) * 4u;
    unsigned soff = 0;
    int ioff = 0, issued = 0;   // float offset of the ring slot the next block goes to
    auto issue = [&]() {
#ifndef MLPF_DIAG_NODMA
        mf_dma16(wrsrc, voff, soff, ring + ioff + wave * 256);
#endif
        soff += MF_TILE * 4u;
#ifndef MLPF_DIAG_NOADDR
        ioff = ioff + MF_TILE == NS * MF_TILE ? 0 : ioff + MF_TILE;
#endif
        ++issued;
    };
#pragma unroll 1
    for (int p = 0; p < NS - 1; ++p) issue();

    // ---- per-sample constants, then y16 = fp16(x * a + o): all loads of a thread in flight at once
    {
        const float* pa = g.pro_a + (size_t)b * C;
        const float* po = g.pro_o + (size_t)b * C;
        if (tid < C) {
            lpro[tid] = pa[tid];
            lpro[C + tid] = po[tid];
            lb2[tid] = g.b2 ? g.b2[tid] : 0.f;
        }
        for (int i = tid; i < WD; i += MF_NT) lb0[i] = g.b0 ? g.b0[i] : 0.f;
        constexpr int ITEMS = 128 * (C / 8) / MF_NT;   // (row, 8-k chunk) items per thread
        static_assert(128 * (C / 8) % MF_NT == 0, "items per thread");
        const float* xb = g.x + ((size_t)b * g.rows + m0) * C;
        f32x4 x0[ITEMS], x1[ITEMS];
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const int i = tid + u * MF_NT, row = i / (C / 8), c8 = i % (C / 8);
            const float* src = xb + (size_t)row * C + c8 * 8;
            x0[u] = *reinterpret_cast<const f32x4*>(src);
            x1[u] = *reinterpret_cast<const f32x4*>(src + 4);
        }
        mf_lds_barrier();   // a | o are in LDS
#pragma unroll
        for (int u = 0; u < ITEMS; ++u) {
            const int i = tid + u * MF_NT, row = i / (C / 8), c8 = i % (C / 8);
            const float* ap = lpro + c8 * 8;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 4);
            const f32x4 o0 = *reinterpret_cast<const f32x4*>(ap + C), o1 = *reinterpret_cast<const f32x4*>(ap + C + 4);
            u32x4 pk;
            pk[0] = mf_pack2(__builtin_fmaf(x0[u][0], a0[0], o0[0]), __builtin_fmaf(x0[u][1], a0[1], o0[1]));
            pk[1] = mf_pack2(__builtin_fmaf(x0[u][2], a0[2], o0[2]), __builtin_fmaf(x0[u][3], a0[3], o0[3]));
            pk[2] = mf_pack2(__builtin_fmaf(x1[u][0], a1[0], o1[0]), __builtin_fmaf(x1[u][1], a1[1], o1[1]));
            pk[3] = mf_pack2(__builtin_fmaf(x1[u][2], a1[2], o1[2]), __builtin_fmaf(x1[u][3], a1[3], o1[3]));
            *reinterpret_cast<u32x4*>(ybuf + row * RSY + ((c8 ^ (row & 15)) << 4)) = pk;
        }
    }
    mf_lds_barrier();   // y16 complete; a | o no longer needed (hbuf is free)
    MSTAMP(1);

    // ---- the step machine (inducer_chain_f16.hip): one weight block per step.  D = issued - s is NS at a primed
    // step: the wait for block s + 1 leaves the NS - 2 younger pieces in flight, the barrier frees block s's slot and
    // one block is issued into it; where the A operand changes the pipeline drains (D = NS - 1) and is primed again:
    // same wait, the barrier frees block s - 1's slot, one issue.  Tiles that reach the end of the stream (TAIL) stop
    // issuing and wait for everything.
    int s = 0, roff = 0;   // float offset of the ring slot of the next block to read fragments from
    const int row = wm * 32 + r;
    // y16 fragment of block kt, chunk c: chunk index (kt * 4 + 2 h + c) ^ (row & 15) = ((kt ^ (row >> 2 & 3)) * 4) | ((2 h + c)
    // ^ (row & 3)): a per-lane base per c, plus (kt ^ ym) * 64 bytes
    const int ym = (row >> 2) & 3;
    const char* ybase[2] = {ybuf + row * RSY + (((2 * h) ^ (row & 3)) << 4), ybuf + row * RSY + (((2 * h + 1) ^ (row & 3)) << 4)};
    const char* hrow = hbuf + row * MF_RSH + 32 * h;
    int boff[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rb = (wn * 2 + j) * 32 + r;
#pragma unroll
        for (int c = 0; c < 2; ++c) boff[j][c] = rb * 16 + (((2 * h + c) ^ ((rb >> 2) & 3)) << 2);
    }
    f16x8 fa[2][2], fb[2][2][2];   // [set][chunk], [set][n-block][chunk]
    f16x8 fh[2][2];                // GEMM b: the A fragments of the hidden K-half (blocks 0, 1), read once per half and
                                   // reused by every output tile (LDS reads are the pace-setter: 128 B/clk per CU)
    // A fragment addresses of block kt: GEMM a reads y16 (swizzled), GEMM b the hidden K-half buffer
    auto a_addr = [&](bool gemm_b, int kt, int c) -> const char* {
#ifdef MLPF_DIAG_NOADDR   // timing only: no swizzle, fixed ring slot (wrong results)
        return gemm_b ? hrow + kt * 64 + 16 * c : ybase[c] + kt * 64;
#endif
        return gemm_b ? hrow + kt * 64 + 16 * c : ybase[c] + ((kt ^ ym) << 6);
    };
    auto load_frags = [&](auto set_tag, auto gb_tag, int kt) {
        constexpr int set = decltype(set_tag)::value;
        constexpr bool gemm_b = decltype(gb_tag)::value;
        const float* st = ring + roff;
#ifdef MLPF_DIAG_NOFRAGS
        if (kt >= 0) return;
#endif
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (!gemm_b) fa[set][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(a_addr(false, kt, c)));
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fb[set][j][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(st + boff[j][c]));
        }
#ifndef MLPF_DIAG_NOADDR
        roff = roff + MF_TILE == NS * MF_TILE ? 0 : roff + MF_TILE;
#endif
    };
    auto wait_block = [&](auto tail_tag) {
        if (decltype(tail_tag)::value && issued >= S_TOTAL) dma::wait_vm_lgkm0<0>();
        else dma::wait_vm_lgkm0<AHEAD>();
    };
    auto kstep = [&](auto cur_tag, auto tail_tag, bool has_next, auto gb_tag, int kt_next, f32x16& a0, f32x16& a1) {
        constexpr int cur = decltype(cur_tag)::value;
        constexpr bool gemm_b = decltype(gb_tag)::value;
        if (has_next) {
            wait_block(tail_tag);
            // this step's fragments were read during the previous one and the wait above covered them: "redefine" them
            // so the compiler's wait-count pass does not put an lgkmcnt(0) in front of the first MFMA
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (!gemm_b) asm volatile("" : "+v"(fa[cur][c]));
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(fb[cur][j][c]));
            }
#ifndef MLPF_DIAG_NOBARRIER
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
            if (!decltype(tail_tag)::value || issued < S_TOTAL) issue();
            load_frags(std::integral_constant<int, cur ^ 1>{}, gb_tag, kt_next);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#ifdef MLPF_DIAG_NOMFMA
            a0[0] += (float)fa[cur][c][0] + (float)fb[cur][0][c][0];
            a1[0] += (float)fa[cur][c][1] + (float)fb[cur][1][c][0];
#else
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(gemm_b ? fh[cur][c] : fa[cur][c], fb[cur][0][c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(gemm_b ? fh[cur][c] : fa[cur][c], fb[cur][1][c], a1, 0, 0, 0);
#endif
        }
    };
    // n (even) blocks of one 128-column tile; first: prime the pipeline; more: another tile over the same A follows
    auto tile_steps = [&](auto tail_tag, int n, auto gb_tag, bool first, bool more, f32x16& a0, f32x16& a1) {
        constexpr std::integral_constant<int, 0> set0{};
        constexpr std::integral_constant<int, 1> set1{};
        if (first) {
            wait_block(tail_tag);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!decltype(tail_tag)::value || issued < S_TOTAL) issue();
            load_frags(set0, gb_tag, 0);
        }
#pragma unroll 1
        for (int kt = 0; kt < n - 2; kt += 2) {
            kstep(set0, tail_tag, true, gb_tag, kt + 1, a0, a1);
            kstep(set1, tail_tag, true, gb_tag, kt + 2, a0, a1);
        }
        kstep(set0, tail_tag, true, gb_tag, n - 1, a0, a1);
        kstep(set1, tail_tag, more, gb_tag, 0, a0, a1);
        s += n;
    };
    auto run_tile = [&](int n, auto gb_tag, bool first, bool more, f32x16& a0, f32x16& a1) {
        if (s + n + NS > S_TOTAL) tile_steps(std::true_type{}, n, gb_tag, first, more, a0, a1);
        else tile_steps(std::false_type{}, n, gb_tag, first, more, a0, a1);
    };
    auto zero = [](f32x16& a) {
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = 0.f;
    };

    const bool has_act = g.act != 0;
    const int act_mode = g.act;
    const float neg_inv_2a2 = act_is_gauss(g.act) ? -1.0f / (2.0f * g.alpha[0] * g.alpha[0]) : 0.f;

    f32x16 acc2[NT1][2];   // the 128 x C result: this wave's 32 rows x (64 columns of every output tile)
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
        zero(acc2[t][0]);
        zero(acc2[t][1]);
    }
#pragma unroll 1
    for (int jc = 0; jc < NCH; ++jc) {
        // GEMM a: hidden[:, 128 jc .. + 127] = act(y16 W0[tile jc]^T + b0), this wave's 32 x 64 as packed fp16 pairs
        f32x16 a0, a1;
        zero(a0);
        zero(a1);
        run_tile(NK, std::false_type{}, true, false, a0, a1);
        unsigned hp[2][8];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x16& a = j ? a1 : a0;
            const float bias = lb0[jc * 128 + wn * 64 + j * 32 + r];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float v0 = a[2 * p] + bias, v1 = a[2 * p + 1] + bias;
                if (has_act) {
                    v0 = act_apply(v0, neg_inv_2a2, act_mode);
                    v1 = act_apply(v1, neg_inv_2a2, act_mode);
                }
                const unsigned own = mf_pack2(v0, v1);
                // even lanes keep row 2p of columns (n, n + 1), odd lanes row 2p + 1 of (n - 1, n)
                hp[j][p] = __builtin_amdgcn_perm(mf_swap_pair(own), own, psel);
            }
        }
        // GEMM b: out += hidden[:, chunk] W2[:, chunk]^T, K-half by K-half (half hf = the columns of the wn == hf waves)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (hf) mf_lds_barrier();   // every wave is done reading half 0 (the steps of GEMM a separate chunks)
            if (wn == hf) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p = 0; p < 8; ++p) {
                        const int hr = wm * 32 + ((2 * p) & 3) + 8 * ((2 * p) >> 2) + 4 * h + (odd ? 1 : 0);
                        *reinterpret_cast<unsigned*>(hbuf + hr * MF_RSH + 2 * (j * 32 + (r & ~1))) = hp[j][p];
                    }
            }
            mf_lds_barrier();
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    fh[q][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(a_addr(true, q, c)));
#pragma unroll
            for (int t = 0; t < NT1; ++t) run_tile(2, std::true_type{}, t == 0, t + 1 < NT1, acc2[t][0], acc2[t][1]);
        }
    }
    MSTAMP(2);

    // ---- epilogue (the store phase of dma::epilogue for a 4 x 2 wave grid): per output tile, the wave's 32 x 64
    // sub-tile goes through its private LDS tile and comes back as rows: lane (lr, c4) owns rows it * 4 + lr, columns
    // c4 * 4 .. + 3.  The residual rows of the next tile are fetched while this one is added and stored.
    {
        constexpr int TP = 64 + 4;   // transpose tile row stride (floats)
        // ring | y16 buffer are contiguous and dead: 8 tiles of 8.5 KiB + the partial sums fit for every supported C
        static_assert((8 * 32 * TP + 4 * 2 * C) * 4 <= NS * MF_TILE * 4 + 128 * RSY, "epilogue scratch");
        float* Tt = smem + wave * 32 * TP;
        float* red2 = smem + 8 * 32 * TP;   // [4 row groups][2][C]
        const int lr = lane >> 4, c4 = lane & 15;
        float* xw = g.x + ((size_t)b * g.rows + m0 + wm * 32) * C + wn * 64 + c4 * 4;
        f32x4 rres[2][8];
        auto fetch = [&](int t, int set) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
                rres[set][it] = *reinterpret_cast<const f32x4*>(xw + (size_t)(it * 4 + lr) * C + t * 128);
        };
        fetch(0, 0);
        mf_lds_barrier();   // every wave is done with the ring and the y16 buffer
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const int set = t & 1;
            if (t + 1 < NT1) fetch(t + 1, set ^ 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bias = lb2[t * 128 + wn * 64 + j * 32 + r];
#pragma unroll
                for (int e = 0; e < 16; ++e) Tt[mfma_row(e, h) * TP + j * 32 + r] = acc2[t][j][e] + bias;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tile: the wave's own LDS operations are in order
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                f32x4 v4 = *reinterpret_cast<const f32x4*>(Tt + (it * 4 + lr) * TP + c4 * 4);
                v4 += rres[set][it];
                *reinterpret_cast<f32x4*>(xw + (size_t)(it * 4 + lr) * C + t * 128) = v4;   // default policy: the next kernel re-reads x (201 MB: Infinity Cache)
                s1 += v4;
                // explicit fma, as in dma::epilogue: the kernels sharing this store phase must agree to the bit on the statistics
#pragma unroll
                for (int q = 0; q < 4; ++q) s2[q] = __builtin_fmaf(v4[q], v4[q], s2[q]);
            }
            if (g.stats) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s1[q] += __shfl_xor(s1[q], 16, 64);
                    s1[q] += __shfl_xor(s1[q], 32, 64);
                    s2[q] += __shfl_xor(s2[q], 16, 64);
                    s2[q] += __shfl_xor(s2[q], 32, 64);
                }
                if (lane < 16) {
                    const int cl = t * 128 + wn * 64 + c4 * 4;
                    *reinterpret_cast<f32x4*>(red2 + (wm * 2 + 0) * C + cl) = s1;
                    *reinterpret_cast<f32x4*>(red2 + (wm * 2 + 1) * C + cl) = s2;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is read before the next one overwrites it
        }
        if (g.stats) {
            mf_lds_barrier();
            for (int i = tid; i < 2 * C; i += MF_NT) {
                const int which = i / C, c = i % C;
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) t += red2[(w * 2 + which) * C + c];
                g.stats[(((size_t)b * tilesM + rt) * 2 + which) * C + c] = t;
            }
        }
    }
    MSTAMP(3);
}

template <int NT1>
int mlpf_launch_t(const MlpArgs& g, hipStream_t st) {
    constexpr int C = 128 * NT1;
    constexpr size_t lds = mf_lds_bytes(C);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_f16_kernel<NT1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((mlp_fused_f16_kernel<NT1>), dim3(g.B * (g.rows / 128)), dim3(MF_NT), lds, st, g);
    return (int)hipGetLastError();
}

}  // namespace

bool mlp_fused_f16_supported(int C, int Wd, int rows) {
    return C % 128 == 0 && C >= 128 && C <= 384 && Wd == 2 * C && rows >= 128 && rows % 128 == 0;
}

int mlp_fused_f16_launch(const MlpArgs& g0, int C, int Wd, hipStream_t st) {
    if (!mlp_fused_f16_supported(C, Wd, g0.rows)) return -9;
    static int stagger = -1;   // GECCO_MLP_STAGGER=<cycles> (0: off); 4000 measured best at C2 (0, 2000 .. 16000 tried)
    if (stagger < 0) {
        const char* e = getenv("GECCO_MLP_STAGGER");
        stagger = e ? atoi(e) : 4000;
    }
    MlpArgs g = g0;
    g.stagger = g.B * (g.rows / 128) >= 512 ? stagger : 0;   // only when there is more than one round of blocks
    switch (C / 128) {
        case 1: return mlpf_launch_t<1>(g, st);
        case 2: return mlpf_launch_t<2>(g, st);
        default: return mlpf_launch_t<3>(g, st);
    }
}
