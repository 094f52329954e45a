// Split-bf16 ("bf16x3") linear layer on PRE-SPLIT operands, 8-phase LDS-DMA schedule (gfx950).
//
//   C[m, n] = residual[m, n] + act( sum_k Y[m, k] * W[n, k] + bias[n] )        (+ GroupNorm partial sums of C)
//
// Same contract as gemm_f32_dma.hip's X3 mode (every nn.Linear of the reference on the point stream:
// models/set_transformer.py:49,65,108-112,164-166, models/mlp.py), same arithmetic — each fp32 operand is hi + lo with
// hi = its top 16 bits (a bf16 by truncation) and lo = bf16(x - hi); a*b ~ a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation — but built for the matrix pipe:
//
//  * "planes" operand format.  Both operands arrive already split: a row of K values is K/32 blocks of 128 bytes,
//    each [32 bf16 hi | 32 bf16 lo] (4 bytes per element, like fp32).  Activations are written that way by their
//    producer (affine_split_planes_kernel: the AdaGN apply; or this kernel's own planes epilogue: the MLP hidden
//    layer), weights once per forward (split_planes_image_kernel).  The K loop therefore has NO conversion VALU and
//    both operand tiles travel global -> LDS as whole 128-byte lines by global_load_lds_dwordx4.
//  * 8 waves, two groups of four that run half a phase apart (group 1 passes one extra s_barrier at the start): while
//    one group's waves issue their fragment reads and DMA, the other group's waves — one of each group sits on every
//    SIMD — run their 18 / 24 MFMAs, so the matrix pipe alternates between the groups instead of idling through a
//    wait -> barrier -> read chain per step (what held the 128 x 128 / 4-wave kernels at ~0.9 PFLOP/s executed).
//  * A K-tile (32 k) is four half-tiles — activation rows of the waves' first / second row quadrant (HY0, HY1),
//    weight rows of their first / second column quadrant (HW0, HW1) — consumed by four phases: (Y0,W0) (Y0,W1) (Y1,W1)
//    (Y1,W0); each phase reads at most one Y and one W quadrant (4..12 ds_read_b128), issues the DMA of ONE half-tile
//    two K-tiles ahead into the slot whose last read lies >= 2 phases back, and waits (counted vmcnt, never 0 in the
//    loop) for the half-tile the NEXT phase reads; two K-tile buffers (128 KiB) hold five phases of prefetch distance.
//  * transposed product: the WEIGHT rows are the MFMA's A operand, the points its B operand, so an accumulator lane
//    holds 4 consecutive output channels of ONE point; the weight image stores each 32-channel group so that a pair of
//    tiles gives a lane 8 consecutive channels: the epilogue stores 16-byte pieces straight from registers — fp32
//    (+ residual, + per-column GroupNorm sums) or hi | lo planes (the next GEMM's operand) — with no LDS transpose.
//
// Tile configurations (template): waves 2 (points) x 4 (channels), wave tile (16 TM points) x (16 TN channels):
//   <TM = 4, TN = 6>: block 128 x 384 — Nout = 384, 768, 1152 (d = 384);   <TM = 8, TN = 4>: block 256 x 256 (d = 256, 512).
// Other widths (d = 128's 128-wide outputs) stay on gemm_f32_dma.hip.
// Requires K % 64 == 0, rows % BM == 0 (a row tile never straddles samples when the per-sample row count is a multiple
// of BM), Nout % BN == 0 (per segment).
#include "../../../gecco_amd/csrc/common.h"
#include "x3_experimental.h"

// Epilogue traffic is streamed (nontemporal): the tile walk lives on W and the Y row tiles staying in the XCD's L2
#ifdef X3_PLAIN_EPILOGUE
#define X3_LOAD(p) (*(p))
#define X3_STORE(v, p) (*(p) = (v))
#else
#define X3_LOAD(p) __builtin_nontemporal_load(p)
#define X3_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// 3-bit XOR swizzle of the 16-byte chunk index inside a 128-byte LDS row (rows in tiles of 16): conflict-free for the
// ds_read_b128 lane groups both when a lane reads chunk fq / 4 + fq (hi | lo planes) and chunks 2 fq, 2 fq + 1 (fp32 rows)
__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) ^ (((row >> 2) & 1) * 4) ^ (((row >> 3) & 1) * 6); }

constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 0xF) | (0x7 << 4) | ((lgkm & 0xF) << 8) | ((vm >> 4) << 14); }
// 8 fp32 -> bf16 hi (truncation) and bf16 lo = rne(x - hi), as gemm_f32_dma.hip's split8
__device__ __forceinline__ void split8p(const float (&x)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned ua = __float_as_uint(x[2 * p]), uc = __float_as_uint(x[2 * p + 1]);
        hi[p] = __builtin_amdgcn_perm(uc, ua, 0x07060302u);   // {hi16(x[2p+1]), hi16(x[2p])}
        const float la = x[2 * p] - __uint_as_float(ua & 0xFFFF0000u), lc = x[2 * p + 1] - __uint_as_float(uc & 0xFFFF0000u);
        const __bf16 ba = (__bf16)la, bc = (__bf16)lc;
        lo[p] = (unsigned)__builtin_bit_cast(unsigned short, ba) | ((unsigned)__builtin_bit_cast(unsigned short, bc) << 16);
    }
}

template <int TM, int TN>
struct Cfg {
    static constexpr int BM = 32 * TM, BN = 64 * TN;       // block tile: points x channels
    static constexpr int QM = TM / 2, QN = TN / 2;          // tiles per quadrant
    static constexpr int HY_ROWS = 2 * QM * 16, HW_ROWS = 4 * QN * 16;   // rows of one half-tile (all waves)
    static constexpr int HY_BYTES = HY_ROWS * 128, HW_BYTES = HW_ROWS * 128;
    static constexpr int NY = HY_BYTES / 8192, NW = HW_BYTES / 8192;      // DMA wave-instructions per wave per half-tile
    static constexpr int BUF_BYTES = 2 * HY_BYTES + 2 * HW_BYTES;
    static constexpr int OFF_Y0 = 0, OFF_Y1 = HY_BYTES, OFF_W0 = 2 * HY_BYTES, OFF_W1 = 2 * HY_BYTES + HW_BYTES;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;
    static_assert(HY_BYTES % 8192 == 0 && HW_BYTES % 8192 == 0, "half-tiles are whole rounds of 8 wave-instructions");
    static_assert(NY + NW == 4, "the counted wait assumes 8 DMA instructions per wave per 4 phases");
    static_assert(TN % 2 == 0 && TM % 2 == 0, "quadrants");
};

#ifdef X3_STAMPS   // diagnostic build (tools/probe): s_memtime stamps of wave 0 / wave 4 per output tile
__device__ unsigned long long g_x3_stamps[256 * 2 * 16 * 8];
#define X3_STAMP(slot)                                                                                          \
    do {                                                                                                        \
        if (lane == 0 && (wave & 3) == 0 && it < 16)                                                            \
            g_x3_stamps[((blockIdx.x * 2 + wr) * 16 + it) * 8 + (slot)] = __builtin_amdgcn_s_memtime();        \
    } while (0)
#else
#define X3_STAMP(slot)
#endif

template <int E>
__device__ __forceinline__ void wait_vm_window(int n) {   // s_waitcnt vmcnt(n + E), wave-uniform n in 0..8
    switch (n) {
        case 0: __builtin_amdgcn_s_waitcnt(waitcnt_imm(0 + E, 0xF)); break;
        case 1: __builtin_amdgcn_s_waitcnt(waitcnt_imm(1 + E, 0xF)); break;
        case 2: __builtin_amdgcn_s_waitcnt(waitcnt_imm(2 + E, 0xF)); break;
        case 3: __builtin_amdgcn_s_waitcnt(waitcnt_imm(3 + E, 0xF)); break;
        case 4: __builtin_amdgcn_s_waitcnt(waitcnt_imm(4 + E, 0xF)); break;
        case 5: __builtin_amdgcn_s_waitcnt(waitcnt_imm(5 + E, 0xF)); break;
        case 6: __builtin_amdgcn_s_waitcnt(waitcnt_imm(6 + E, 0xF)); break;
        case 7: __builtin_amdgcn_s_waitcnt(waitcnt_imm(7 + E, 0xF)); break;
        default: __builtin_amdgcn_s_waitcnt(waitcnt_imm(8 + E, 0xF)); break;
    }
}

// PERSISTENT: one block per CU walks output tiles; the DMA pipeline runs on across tile boundaries (the half-tiles of
// the next output tile's first K-tiles are issued during the last phases of the current one) and a wave's epilogue —
// stores straight from its accumulators — sits between its last MFMA phase of one tile and its first of the next, while
// the other wave group is still / already multiplying.  vmcnt counts DMA, loads and stores together in issue order, so
// the counted waits of the two phases after an epilogue allow its E stores to stay in flight as well.
template <int TM, int TN, bool C_PLANES>
__global__ __launch_bounds__(512, 1) void gemm_x3_planes_kernel(X3Args g) {
    using C_ = Cfg<TM, TN>;
    constexpr int BM = C_::BM, BN = C_::BN, QM = C_::QM, QN = C_::QN, NY = C_::NY, NW = C_::NW;
    constexpr int E_STORES = (TN / 2) * TM * 2;   // 16-byte stores of one wave's epilogue
    static_assert(8 + E_STORES <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;         // row group (= stagger group), column quarter
    const int fr = lane & 15, fq = lane >> 4;

    const int tilesN = g.Nout / BN, tilesM = g.rows_total / BM, ntiles = tilesM * tilesN;
    const int nk = g.K / 32;
    const size_t row_bytes = (size_t)g.K * 4;          // a planes row: K/32 blocks of 128 B
    // tile walk: the blocks that share an XCD (blockIdx % 8) take consecutive tiles of every round, so the column tiles
    // of one row tile run at the same time behind the same L2
    const int G = gridDim.x, perx = G >> 3;
    auto tile_id = [&](int it) { return (G & 7) ? it * G + (int)blockIdx.x : it * G + ((int)blockIdx.x & 7) * perx + ((int)blockIdx.x >> 3); };

    // ---- DMA sources, as byte offsets from the tile's first Y row / first image row.  Half-tile rows are 128 B; a
    // wave-instruction fills 8 rows: lane l -> row 8 q + (l >> 3), LDS chunk l & 7, which receives global chunk
    // (l & 7) ^ swz(row) of that row (swizzle on the source side; the fragment reads apply the same involution).
    // HY(mi) row rho: group rho / (QM 16): point m0 + group * 16 TM + mi * 16 QM + rho % (QM 16)
    // HW(ni) row rho: wave column rho / (QN 16): image row n0 + wcol * 16 TN + ni * 16 QN + rho % (QN 16)
    unsigned yoff[2][NY], woff[2][NW];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int q = 0; q < NY; ++q) {
            const int rho = (wave * NY + q) * 8 + (lane >> 3);
            const int grp = rho / (QM * 16), within = rho % (QM * 16);
            const int m = grp * 16 * TM + h * 16 * QM + within;
            yoff[h][q] = (unsigned)(m * (int)row_bytes + (((lane & 7) ^ swz(rho)) * 16));
        }
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const int rho = (wave * NW + q) * 8 + (lane >> 3);
            const int wcol = rho / (QN * 16), within = rho % (QN * 16);
            const int n = wcol * 16 * TN + h * 16 * QN + within;
            woff[h][q] = (unsigned)(n * (int)row_bytes + (((lane & 7) ^ swz(rho)) * 16));
        }
    }
    // item i of the issue sequence (4 per K-tile): 0 HY0, 1 HW0, 2 HW1, 3 HY1 of K-tile kt (buffer kt & 1) of the tile
    // whose first Y row / image row are ybase / wbase
    auto issue = [&](int item, int kt, const unsigned char* ybase, const unsigned char* wbase) -> int {
        unsigned char* buf = lds + (kt & 1) * C_::BUF_BYTES;
        const size_t ko = (size_t)kt * 128;
        if (item == 0 || item == 3) {
            const int h = item == 0 ? 0 : 1;
            unsigned char* dst = buf + (h ? C_::OFF_Y1 : C_::OFF_Y0) + wave * NY * 1024;
#pragma unroll
            for (int q = 0; q < NY; ++q) dma16(ybase + ko + yoff[h][q], dst + q * 1024);
            return NY;
        }
        const int h = item == 1 ? 0 : 1;
        unsigned char* dst = buf + (h ? C_::OFF_W1 : C_::OFF_W0) + wave * NW * 1024;
#pragma unroll
        for (int q = 0; q < NW; ++q) dma16(wbase + ko + woff[h][q], dst + q * 1024);
        return NW;
    };

    // ---- fragment read offsets (bytes inside a half-tile): row = wave's first row of the quadrant + 16 i + fr; the
    // swizzle term depends on fr only (the bases are multiples of 16 rows)
    const int sw = swz(fr);
    const int yrow0 = wr * QM * 16 + fr, wrow0 = wc * QN * 16 + fr;
    const int off_hi = (fq ^ sw) * 16, off_lo = ((4 + fq) ^ sw) * 16;

    f32x4 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    bf16x8 yh[QM], yl[QM], wh[2][QN], wl[2][QN];
    auto read_y = [&](const unsigned char* buf, int mi) {
        const unsigned char* base = buf + (mi ? C_::OFF_Y1 : C_::OFF_Y0) + yrow0 * 128;
#pragma unroll
        for (int i = 0; i < QM; ++i) {
            yh[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + i * 2048 + off_hi));
            yl[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + i * 2048 + off_lo));
        }
    };
    auto read_w = [&](const unsigned char* buf, int ni) {
        const unsigned char* base = buf + (ni ? C_::OFF_W1 : C_::OFF_W0) + wrow0 * 128;
#pragma unroll
        for (int j = 0; j < QN; ++j) {
            wh[ni][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + j * 2048 + off_hi));
            wl[ni][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + j * 2048 + off_lo));
        }
    };
    auto mfmas = [&](int ni, int mi) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < QN; ++j)
#pragma unroll
            for (int i = 0; i < QM; ++i) {
                f32x4& a = acc[ni * QN + j][mi * QM + i];
#ifdef X3_DIAG_NOMFMA   // keep the fragments alive with one VALU op each instead
                a[0] += (float)wl[ni][j][0] + (float)yh[i][0] + (float)wh[ni][j][0] + (float)yl[i][0];
                continue;
#endif
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ni][j], yh[i], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni][j], yl[i], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni][j], yh[i], a, 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
    };

    int it = 0;   // output tiles done by this block
    const bool has_act = g.act != 0;
    const int act_mode = g.act;
    const float neg_inv_2a2 = act_is_gauss(g.act) ? -1.0f / (2.0f * g.alpha[0] * g.alpha[0]) : 0.f;

    // ---- epilogue of one output tile, straight from the accumulators (and their reset).  Tile pair (2u, 2u+1), point
    // tile i: this lane holds the 8 consecutive channels cb + 32 u + 8 fq .. + 7 of point m0 + wr 16 TM + 16 i + fr.
    auto epilogue = [&](int rt, int ct) {
#ifdef X3_DIAG_NOEPI   // the accumulators stay live, no global traffic
        {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) { t += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3]; acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            if (t == 123.456f) g.C[0] = t;
            return;
        }
#endif
        const int m0 = rt * BM, n0 = ct * BN;
        const bool seg2 = g.C2 != nullptr && n0 >= g.n_split;
        const int nbase = seg2 ? n0 - g.n_split : n0;         // first channel of the block inside its segment
        const int nseg = g.C2 ? (seg2 ? g.Nout - g.n_split : g.n_split) : g.Nout;
        float* Cseg = seg2 ? g.C2 : g.C;
        const int ldc = seg2 ? g.ldc2 : g.ldc;
        const float* bias = seg2 ? g.bias2 : g.bias;
        const int mrow0 = m0 + wr * 16 * TM + fr;
        if (!C_PLANES && g.residual) {
            // x += ...: all residual pieces of the wave's tile in flight at once, added into the accumulators
#pragma unroll
            for (int u = 0; u < TN / 2; ++u) {
                const int cg = nbase + wc * 16 * TN + 32 * u + 8 * fq;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#ifdef X3_DIAG_LINSTORE   // timing only: every instruction touches 1 KiB of consecutive bytes (wrong placement)
                    const float* rs = g.residual + (size_t)m0 * g.ldr + (size_t)wave * 16 * TM * 16 * TN + (size_t)((u * TM + i) * 2) * 256 + lane * 4;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rs), r1 = *reinterpret_cast<const f32x4*>(rs + 256);
#else
                    const float* rs = g.residual + (size_t)(mrow0 + 16 * i) * g.ldr + cg;
                    const f32x4 r0 = X3_LOAD(reinterpret_cast<const f32x4*>(rs)), r1 = X3_LOAD(reinterpret_cast<const f32x4*>(rs + 4));
#endif
                    acc[2 * u][i] += r0;
                    acc[2 * u + 1][i] += r1;
                }
            }
        }
        X3_STAMP(2);
#pragma unroll
        for (int u = 0; u < TN / 2; ++u) {
            const int cl = wc * 16 * TN + 32 * u + 8 * fq;    // channel inside the block
            const int cg = nbase + cl;                        // channel inside the segment
            float b8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) b8[e] = bias ? bias[cg + e] : 0.f;
            float s1[8], s2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const size_t m = (size_t)(mrow0 + 16 * i);
                float v8[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v8[e] = acc[2 * u][i][e] + b8[e];
                    v8[4 + e] = acc[2 * u + 1][i][e] + b8[4 + e];
                }
                acc[2 * u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[2 * u + 1][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (has_act) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v8[e] = act_apply(v8[e], neg_inv_2a2, act_mode);
                }
                if (C_PLANES) {
                    // hi | lo planes of the output: row m = nseg / 32 blocks of 128 B; this lane's 8 channels sit in
                    // block cg / 32 at element cg % 32: 16 B of hi, 16 B of lo
                    u32x4 hi, lo;
                    split8p(v8, hi, lo);
                    unsigned char* dst = reinterpret_cast<unsigned char*>(Cseg) + m * (size_t)nseg * 4 + (size_t)(cg >> 5) * 128 + (cg & 31) * 2;
                    X3_STORE(hi, reinterpret_cast<u32x4*>(dst));
                    X3_STORE(lo, reinterpret_cast<u32x4*>(dst + 64));
                } else {
#ifdef X3_DIAG_LINSTORE
                    float* dst = Cseg + (size_t)m0 * ldc + (size_t)wave * 16 * TM * 16 * TN + (size_t)((u * TM + i) * 2) * 256 + lane * 4;
                    *reinterpret_cast<f32x4*>(dst) = f32x4{v8[0], v8[1], v8[2], v8[3]};
                    *reinterpret_cast<f32x4*>(dst + 256) = f32x4{v8[4], v8[5], v8[6], v8[7]};
#else
                    float* dst = Cseg + m * (size_t)ldc + cg;
                    X3_STORE((f32x4{v8[0], v8[1], v8[2], v8[3]}), reinterpret_cast<f32x4*>(dst));
                    X3_STORE((f32x4{v8[4], v8[5], v8[6], v8[7]}), reinterpret_cast<f32x4*>(dst + 4));
#endif
                    if (g.stats) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            s1[e] += v8[e];
                            s2[e] += v8[e] * v8[e];
                        }
                    }
                }
            }
            if (!C_PLANES && g.stats) {
                // sums over the wave's 16 TM points: over the tiles above, then over the 16 lanes of this fq group (fixed
                // butterfly order).  One partial per (row group of 16 TM points): stats (rows_total / (16 TM), 2, Nout)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        s1[e] += __shfl_xor(s1[e], o, 64);
                        s2[e] += __shfl_xor(s2[e], o, 64);
                    }
                }
                if (fr == 0) {
                    float* sd = g.stats + ((size_t)(rt * 2 + wr) * 2) * g.Nout + n0 + cl;
                    *reinterpret_cast<f32x4*>(sd) = f32x4{s1[0], s1[1], s1[2], s1[3]};
                    *reinterpret_cast<f32x4*>(sd + 4) = f32x4{s1[4], s1[5], s1[6], s1[7]};
                    *reinterpret_cast<f32x4*>(sd + g.Nout) = f32x4{s2[0], s2[1], s2[2], s2[3]};
                    *reinterpret_cast<f32x4*>(sd + g.Nout + 4) = f32x4{s2[4], s2[5], s2[6], s2[7]};
                }
            }
        }
    };

    // ---- start-up skew: the blocks of XCD x begin x eighths of a tile late, so that the store bursts of the epilogues
    // (all of a tile's output leaves within a few thousand cycles) of the eight XCDs do not meet in HBM
#ifndef X3_NO_SKEW
    for (int z = ((((int)blockIdx.x >> 3) / tilesN) & 7) * g.skew; z > 0; --z) __builtin_amdgcn_s_sleep(127);
#endif
    // ---- first tile: K-tile 0 whole, HY0 / HW0 of K-tile 1 (the steady-state issue order continued backwards)
    int w_cur = tile_id(0);
    if (w_cur >= ntiles) return;
    int rt = w_cur / tilesN, ct = w_cur % tilesN;
    const unsigned char* ycur = static_cast<const unsigned char*>(g.Y) + (size_t)rt * BM * row_bytes;
    const unsigned char* wcur = static_cast<const unsigned char*>(g.Wimg) + (size_t)ct * BN * row_bytes;
    issue(0, 0, ycur, wcur); issue(1, 0, ycur, wcur); issue(2, 0, ycur, wcur); issue(3, 0, ycur, wcur);
    issue(0, 1, ycur, wcur); issue(1, 1, ycur, wcur);                    // nk >= 2 (K % 64 == 0)
    // the first reads need HY0(0), HW0(0): everything issued after them (8 instructions) may stay in flight
    wait_vm_window<0>(8);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // the stagger: group 1 runs one barrier behind group 0
    int c3 = NW, c2 = NY, c1 = NY, c0 = NW;      // DMA instructions of the last four phases: HW1(0), HY1(0), HY0(1), HW0(1)

    for (;;) {
        const int w_nxt = tile_id(it + 1);
        const bool has_next = w_nxt < ntiles;
        const int rtn = has_next ? w_nxt / tilesN : 0, ctn = has_next ? w_nxt % tilesN : 0;
        const unsigned char* ynxt = static_cast<const unsigned char*>(g.Y) + (size_t)rtn * BM * row_bytes;
        const unsigned char* wnxt = static_cast<const unsigned char*>(g.Wimg) + (size_t)ctn * BN * row_bytes;
        // one phase: reads -> DMA issue -> counted wait for what the next phase reads -> barrier -> MFMAs -> barrier
        auto phase = [&](int kt, int p, bool after_epilogue) {
            const unsigned char* buf = lds + (kt & 1) * C_::BUF_BYTES;
            if (p == 0) { read_w(buf, 0); __builtin_amdgcn_sched_barrier(0); read_y(buf, 0); }
            else if (p == 1) read_w(buf, 1);
            else if (p == 2) read_y(buf, 1);
            // issue sequence: p0 -> HW1(kt+1), p1 -> HY1(kt+1), p2 -> HY0(kt+2), p3 -> HW0(kt+2); past the tile's last
            // K-tile the sequence continues with the next output tile's first K-tiles
            const int item = p == 0 ? 2 : p == 1 ? 3 : p == 2 ? 0 : 1;
            const int ktt = kt + (p < 2 ? 1 : 2);
            int n = 0;
            if (ktt < nk) n = issue(item, ktt, ycur, wcur);
            else if (has_next) n = issue(item, ktt - nk, ynxt, wnxt);
            c3 = c2; c2 = c1; c1 = c0; c0 = n;
            // the next phase's half-tiles were issued four phases ago: whatever was issued since may stay in flight
            // (after an epilogue: its stores too — they are younger than those half-tiles)
            if (p != 2) {
                if (after_epilogue) wait_vm_window<E_STORES>(c3 + c2 + c1 + c0);
                else wait_vm_window<0>(c3 + c2 + c1 + c0);
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(0x3F, 0));   // lgkmcnt(0): this phase's fragments are in
            if (p == 0) mfmas(0, 0);
            else if (p == 1) mfmas(1, 0);
            else if (p == 2) mfmas(1, 1);
            else mfmas(0, 1);
            __builtin_amdgcn_s_barrier();
        };
        const bool ae = it > 0;
        X3_STAMP(0);
        phase(0, 0, ae); phase(0, 1, ae); phase(0, 2, false); phase(0, 3, false);
        phase(1, 0, false); phase(1, 1, false); phase(1, 2, false); phase(1, 3, false);
        for (int kt = 2; kt < nk; kt += 2) {
            phase(kt, 0, false); phase(kt, 1, false); phase(kt, 2, false); phase(kt, 3, false);
            phase(kt + 1, 0, false); phase(kt + 1, 1, false); phase(kt + 1, 2, false); phase(kt + 1, 3, false);
        }
        X3_STAMP(1);
        epilogue(rt, ct);
        X3_STAMP(3);
        if (!has_next) break;
        ++it;
        rt = rtn; ct = ctn; ycur = ynxt; wcur = wnxt;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // group 0 catches up with group 1's extra barrier
}

// ---- operand producers -------------------------------------------------------------------------------------------
// y = a[b, c] * x[b, m, c] + o[b, c] (the AdaGN apply, models/normalization.py:44) -> hi | lo planes; a == nullptr: y = x.
// One thread per 8 consecutive channels: 32 B in, 16 B hi + 16 B lo out.
__global__ __launch_bounds__(256) void affine_split_planes_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                                  const float* __restrict__ o, unsigned char* __restrict__ y,
                                                                  size_t rows_total, int rows_per_sample, int C) {
    const int c8n = C / 8;
    const size_t total = rows_total * c8n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t m = i / c8n;
        const int c = (int)(i % c8n) * 8;
        const float* src = x + m * C + c;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
        float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        if (a) {
            const size_t b = m / rows_per_sample;
            const float* pa = a + b * C + c;
            const float* po = o + b * C + c;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] * pa[e] + po[e];
        }
        u32x4 hi, lo;
        split8p(v, hi, lo);
        unsigned char* dst = y + m * (size_t)C * 4 + (size_t)(c >> 5) * 128 + (c & 31) * 2;
        *reinterpret_cast<u32x4*>(dst) = hi;
        *reinterpret_cast<u32x4*>(dst + 64) = lo;
    }
}

// W (Nout, ldw) fp32 -> the planes image the kernel streams.  Image row R of a 32-row group g = R / 32 (tile e = (R / 16) & 1,
// tile row r = R & 15) holds channel 32 g + 8 (r >> 2) + 4 e + (r & 3): the accumulator lane (fq = r >> 2, reg = r & 3)
// of the tile pair then owns 8 consecutive channels.  One thread per (image row, 8 k).
__device__ __forceinline__ void planes_image_item(const float* __restrict__ W, unsigned char* __restrict__ img, int Nout, int K,
                                                  int ldw, size_t i) {
    const int k8n = K / 8;
    const int R = (int)(i / k8n), k = (int)(i % k8n) * 8;
    const int grp = R >> 5, e = (R >> 4) & 1, r = R & 15;
    const int n = 32 * grp + 8 * (r >> 2) + 4 * e + (r & 3);
    const float* src = W + (size_t)min(n, Nout - 1) * ldw + k;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
    const float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    u32x4 hi, lo;
    split8p(v, hi, lo);
    unsigned char* dst = img + (size_t)R * K * 4 + (size_t)(k >> 5) * 128 + (k & 31) * 2;
    *reinterpret_cast<u32x4*>(dst) = hi;
    *reinterpret_cast<u32x4*>(dst + 64) = lo;
}

__global__ void split_planes_image_kernel(const float* __restrict__ W, unsigned char* __restrict__ img, int Nout, int K, int ldw) {
    const size_t total = (size_t)((Nout + 31) / 32 * 32) * (K / 8);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        planes_image_item(W, img, Nout, K, ldw, i);
}

__global__ void split_planes_image_multi_kernel(SplitJobs jobs) {
    const SplitJob j = jobs.job[blockIdx.y];
    const size_t total = (size_t)((j.Nout + 31) / 32 * 32) * (j.K / 8);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        planes_image_item(j.W, reinterpret_cast<unsigned char*>(j.img), j.Nout, j.K, j.ldw, i);
}

template <int TM, int TN>
int launch_t(const X3Args& g, hipStream_t st) {
    using C_ = Cfg<TM, TN>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_planes_kernel<TM, TN, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C_::LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_planes_kernel<TM, TN, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, C_::LDS_BYTES);
        attr = true;
    }
    const int ntiles = (g.rows_total / C_::BM) * (g.Nout / C_::BN);
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t pr;
        (void)hipGetDevice(&dev);
        cus = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    const dim3 grid(ntiles < cus ? ntiles : cus);   // persistent: one block per CU (128 KiB of LDS each)
    X3Args ga = g;
    if (ga.skew < 0) ga.skew = ntiles > 2 * cus ? (g.K / 32 + 11) / 12 : 0;   // s_sleep(127) units per XCD index: ~1/8 tile
    if (g.c_planes) hipLaunchKernelGGL((gemm_x3_planes_kernel<TM, TN, true>), grid, dim3(512), C_::LDS_BYTES, st, ga);
    else hipLaunchKernelGGL((gemm_x3_planes_kernel<TM, TN, false>), grid, dim3(512), C_::LDS_BYTES, st, ga);
    return (int)hipGetLastError();
}

// block tile for an output width: 384-wide tiles when they divide it (d = 384), else 256, else 128
int pick_cfg(int nout, int n_split) {
    auto ok = [&](int bn) { return nout % bn == 0 && (n_split == 0 || n_split % bn == 0); };
    if (ok(384)) return 0;
    if (ok(256)) return 1;
    return -1;
}
int cfg_bm(int cfg) { return cfg == 1 ? 256 : 128; }

}  // namespace

bool gemm_x3_planes_supported(const X3Args& g) {
    const int cfg = pick_cfg(g.Nout, g.C2 ? g.n_split : 0);
    if (cfg < 0 || g.K % 64 || g.K <= 0) return false;
    if (g.rows_total % cfg_bm(cfg) || g.rows_per_sample % cfg_bm(cfg)) return false;
    if (g.C2 && (g.stats || g.residual || g.c_planes || g.n_split <= 0 || g.n_split >= g.Nout)) return false;
    if (g.c_planes && (g.residual || g.stats || g.Nout % 32)) return false;
    if (g.residual && g.act) return false;   // the residual joins the accumulators before the bias (call sites: act == 0)
    if (!g.c_planes && ((g.ldc & 3) || (g.residual && (g.ldr & 3)) || (g.C2 && (g.ldc2 & 3)))) return false;
    return g.Y && g.Wimg && g.C;
}

int gemm_x3_planes_row_tile(int Nout, int n_split) {   // rows per statistics partial: one wave row group = BM / 2
    const int cfg = pick_cfg(Nout, n_split);
    return cfg < 0 ? 0 : cfg_bm(cfg) / 2;
}

int gemm_x3_planes_launch(const X3Args& g, hipStream_t st) {
    if (!gemm_x3_planes_supported(g)) return -9;
    switch (pick_cfg(g.Nout, g.C2 ? g.n_split : 0)) {
        case 0: return launch_t<4, 6>(g, st);
        default: return launch_t<8, 4>(g, st);
    }
}

size_t planes_image_bytes(int Nout, int K) { return (size_t)((Nout + 31) / 32 * 32) * K * 4; }

int split_planes_image_launch(const float* W, void* img, int Nout, int K, int ldw, hipStream_t st) {
    if (K % 32 || (ldw & 3)) return -9;
    const size_t total = (size_t)((Nout + 31) / 32 * 32) * (K / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(split_planes_image_kernel, dim3(grid), dim3(256), 0, st, W, static_cast<unsigned char*>(img), Nout, K, ldw);
    return (int)hipGetLastError();
}

int split_planes_image_multi_launch(const SplitJobs& jobs, hipStream_t st) {
    if (jobs.n <= 0) return 0;
    hipLaunchKernelGGL(split_planes_image_multi_kernel, dim3(64, jobs.n), dim3(256), 0, st, jobs);
    return (int)hipGetLastError();
}

int affine_split_planes_launch(const float* x, const float* a, const float* o, void* y, size_t rows_total, int rows_per_sample,
                               int C, hipStream_t st) {
    if (C % 32) return -9;
    const size_t total = rows_total * (C / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(affine_split_planes_kernel, dim3(grid), dim3(256), 0, st, x, a, o, static_cast<unsigned char*>(y), rows_total,
                       rows_per_sample, C);
    return (int)hipGetLastError();
}
