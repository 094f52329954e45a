// Fused linear layer, LDS-DMA variant (the fast path for the N-token GEMMs): same contract as gemm_f32.hip
//
//   C[b, m, n] = residual[b, m, n] + act( sum_k (A[b, m, k] * pa[b, k] + po[b, k]) * W[n, k] + bias[n] )  (+ GroupNorm partials)
//
// but both operand tiles travel global -> LDS by `global_load_lds_dwordx4` (no staging registers, no ds_write, no
// staging VALU), four K-steps deep:
//   * a wave-instruction writes 1 KiB lane-linearly; bank-conflict-free `ds_read_b128` fragment reads come from
//     XOR-swizzling the 16-byte chunk index with row bits — applied to the per-lane SOURCE address of the DMA and to
//     the read address (both sides, guide rule 21);
//   * the ring has DNS stages; every K-step: counted `s_waitcnt vmcnt(N)` on the wave's own pieces of the oldest
//     stage, ONE raw `s_barrier`, issue the DMA DNS-1 K-steps ahead, then the MFMAs; `__syncthreads()` is never used
//     while a DMA is in flight (it would drain vmcnt to 0);
//   * the AdaGN affine moves from the staging pass to the A fragment, its per-(b, k) coefficients parked in LDS once
//     per tile, so no ordinary global load sits in the K loop.
// Rows beyond `rows` are clamped at the source (duplicates of the last row) and masked in the epilogue.
//
// Two arithmetic modes share the kernel:
//   X3 = false  exact fp32: v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak), bit-for-bit an fp32 fma chain;
//   X3 = true   split-bf16 ("bf16x3"): every fp32 operand is hi + lo with hi = its top 16 bits (a bf16) and
//               lo = bf16(x - hi); a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 with fp32
//               accumulation: 3 MFMAs at 16x the fp32-MFMA rate, relative error ~2^-16 per product (plain bf16
//               operands: 2^-9, which misses the 1e-3 parity bar).  W is pre-split into two
//               bf16 planes, tiled per (column tile, K-step) (split_bf16_tiled_kernel); A is split on the fragment,
//               after the AdaGN affine.
// Requires K % 16 == 0, Nout % 4 == 0, rows >= 128; everything else runs on gemm_f32.hip.
#include "gemm_dma_common.h"

#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

using dma::DBN;
using dma::DNT;
using dma::D_EPI;
using dma::dma16;
constexpr int DBK = 16;                           // row tile BM: template parameter (128, or 64 in bf16x3 mode)
constexpr int D_TILE = 128 * DBK;                 // floats per operand tile per stage (8 KiB)
constexpr int D_STAGE = 2 * D_TILE;               // A then B (fp32 W tile, or bf16 hi | lo planes: same 8 KiB)
// 256-row tiles (bf16x3): the A tile is [256][16] = 16 KiB, the W block stays 8 KiB
constexpr int a_tile_floats(int bm) { return bm == 256 ? 256 * DBK : D_TILE; }
constexpr int stage_floats(int bm) { return a_tile_floats(bm) + D_TILE; }
constexpr int d_main_floats(int ns, int bm = 128) { return ns * stage_floats(bm) > D_EPI ? ns * stage_floats(bm) : D_EPI; }

// 8 fp32 -> bf16 hi (truncation: the top 16 bits) and bf16 lo = rne(x - hi); x - hi is exact in fp32
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, bf16x8& hi, bf16x8& lo) {
    u32x4 hb;
    float l[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = p < 2 ? x0[2 * p] : x1[2 * p - 4], c = p < 2 ? x0[2 * p + 1] : x1[2 * p - 3];
        const unsigned ua = __float_as_uint(a), uc = __float_as_uint(c);
        hb[p] = __builtin_amdgcn_perm(uc, ua, 0x07060302u);  // {hi16(c), hi16(a)}
        l[2 * p] = a - __uint_as_float(ua & 0xFFFF0000u);
        l[2 * p + 1] = c - __uint_as_float(uc & 0xFFFF0000u);
    }
    hi = __builtin_bit_cast(bf16x8, hb);
#pragma unroll
    for (int e = 0; e < 8; ++e) lo[e] = (__bf16)l[e];
}

// BM = 128 rows per block; BM = 64 (bf16x3 only) serves the 64-inducer GEMMs (rows per sample < 128);
// BM = 256 (bf16x3 only): 4 x 1 waves of 64 x 128 — the split-bf16 loop is bound by LDS bandwidth (per K-step the four
// 32 x 128 wave tiles of a 128-row block read 40 KiB of fragments and the DMA writes 16 KiB for 48 MFMAs: 146 B/clk/CU
// at the full matrix rate against the LDS's 128), and a wave tile twice as tall re-uses every W fragment for two row
// tiles: 0.75 KiB of LDS traffic per MFMA instead of 1.17; 72 KiB of ring = two blocks per CU, so one block's epilogue
// still runs under the other's K loop.
// AIMG: A arrives as the tiled split image of GemmArgs::a_img (written by the producing GEMM's epilogue): its blocks are DMA'd
// as they are — 1 KiB contiguous per wave-instruction like the W image, where fp32 rows give 64-byte pieces — and the
// fragments are read as bf16 hi / lo planes: no split, no VALU in the K loop.
template <int DNS, bool HAS_PRO, bool X3, int BM = 128, bool AIMG = false, bool ACTBWD = false>
__global__ __launch_bounds__(DNT, BM == 256 ? 2 : (DNS <= 3 ? 3 : 2)) void gemm_dma_kernel(GemmArgs g) {
    // ACTBWD: the epilogue multiplies by act'(u) (GemmArgs::mul_u) — the training path's dX product after an activation
    static_assert(BM == 128 || ((BM == 64 || BM == 256) && X3), "64- and 256-row tiles exist in split-bf16 mode only");
    static_assert(!AIMG || (X3 && !HAS_PRO && BM >= 128), "the activation image feeds the split-bf16 kernel without prologue");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int A_TILE = a_tile_floats(BM), STAGE = stage_floats(BM);
    float* pro_lds = smem + d_main_floats(DNS, BM);   // pa[0..K) | po[0..K)

    const dma::Tile T = dma::tile_of_block<BM>(g);
    const int ct = T.ct, b = T.b, m0 = T.m0, nseg0 = T.nseg0, nseg = T.nseg;
    const float* Wseg = T.Wseg;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave layout: fp32 mode 2 x 2 waves of 64 x 64; bf16x3 mode 4 x 1 waves of 32 x 128 — the A fragment costs
    // ~28 VALU per 32-row tile per K-step to split into hi / lo, the (pre-split) B fragment nothing, so the
    // wave tile is made wide in N: half the split work per MFMA of the square layout.
    // (64-row tiles: 2 x 2 waves of 32 x 64)
    constexpr int WMN = (X3 && BM >= 128) ? 4 : 2, WNN = 4 / WMN;
    constexpr int TMW = (BM / 32) / WMN, TNW = 4 / WNN;   // 32 x 32 MFMA tiles per wave in M / N
    constexpr int NAP = BM / 64;                  // A pieces (16 rows x 64 B) per wave per K-step
    constexpr int NPIECE = NAP + 2;               // DMA wave-instructions per wave per K-step
    const int wm = wave / WNN, wn = wave % WNN;
    const int r = lane & 31, h = lane >> 5;

    // ---- DMA source pointers (bytes advance by one K-step = 16 k per iteration)
    // A tile [128][16] fp32: 8 pieces of 16 rows x 64 B; wave w moves pieces {2w, 2w+1}; chunk ^= (row >> 2) & 3.
    // fp32 W tile: the same.  bf16 planes [128][16] bf16: 4 pieces of 32 rows x 32 B each; wave w moves piece w of
    // the hi plane and of the lo plane; chunk ^= (row >> 3) & 1.
    const float* __restrict__ Ab = g.A + (size_t)b * g.rows * g.lda;
    constexpr int NQ = NAP > 2 ? NAP : 2;
    const float* asrc[NQ];
    const void* bsrc[2];
    const int nk = g.K / DBK;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int row = (NAP * wave + (q < NAP ? q : 0)) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        asrc[q] = Ab + (size_t)min(m0 + row, g.rows - 1) * g.lda + c * 4;
        if (AIMG) {
            // piece p of the block tile's image: 8 pieces (hi plane 4, lo plane 4) per 128-row image tile
            const int p = NAP * wave + (q < NAP ? q : 0), t128 = (g.rows + 127) >> 7;
            const int tile = min((m0 >> 7) + (p >> 3), t128 - 1);
            asrc[q] = g.A + ((size_t)b * t128 + tile) * nk * D_TILE + (p & 7) * 256 + lane * 4;
        }
        if (!X3 && q < 2) bsrc[q] = Wseg + (size_t)min(nseg0 + row, nseg - 1) * g.ldw + c * 4;
    }
    if (X3) {
        // pre-tiled W image (split_bf16_tiled_kernel): one 8 KiB block per (column tile, K-step) that IS the LDS
        // image (hi plane | lo plane, swizzle baked in), so a wave-instruction reads 1 KiB of consecutive bytes —
        // measured 2.4x the fill rate of 32-byte row pieces (tools/probe/dma_rate.hip)
        const float* img = static_cast<const float*>(g.w_img) + (size_t)ct * nk * D_TILE + wave * 256 + lane * 4;
        bsrc[0] = img;
        bsrc[1] = img + 1024;
    }
    auto issue = [&](int kt) {   // 4 DMA wave-instructions: this wave's share of K-step kt
#ifdef GEMM_DIAG_NODMA
        return;
#endif
        float* st = smem + (kt % DNS) * STAGE;
#pragma unroll
        for (int q = 0; q < NAP; ++q) dma16(asrc[q] + (AIMG ? (size_t)kt * D_TILE : (size_t)kt * DBK), st + (NAP * wave + q) * 256);
        if (X3) {
            dma16(static_cast<const float*>(bsrc[0]) + (size_t)kt * D_TILE, st + A_TILE + wave * 256);
            dma16(static_cast<const float*>(bsrc[1]) + (size_t)kt * D_TILE, st + A_TILE + 1024 + wave * 256);
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                dma16(static_cast<const float*>(bsrc[q]) + kt * DBK, st + A_TILE + (2 * wave + q) * 256);
        }
    };

    if (HAS_PRO) {  // park the AdaGN coefficients of this sample (ordinary loads, drained before the ring starts)
        const float* pa = g.pro_a + (size_t)b * g.K;
        const float* po = g.pro_o + (size_t)b * g.K;
        for (int i = tid; i < g.K; i += DNT) {
            pro_lds[i] = pa[i];
            pro_lds[g.K + i] = po[i];
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // fp32 mode keeps DNS - 1 K-steps in flight ahead of the one being read; bf16x3 mode holds the K-step being
    // multiplied in registers (software pipeline below), so all DNS ring slots can be in flight or landed ahead
    constexpr int PRE = X3 ? DNS : DNS - 1;
#pragma unroll
    for (int p = 0; p < PRE; ++p)
        if (p < nk) issue(p);

    f32x16 acc[TMW][TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addressing (float offsets inside a stage).  Every lane reads two 16-byte chunks of an fp32 row per
    // K-step: chunks {h, 2 + h} in fp32 mode (k = 8*kk + 4*h + e for kk = 0, 1), chunks {2h, 2h + 1} in bf16x3 mode
    // (k = 8*h + j); row R's global chunk c sits at LDS chunk c ^ ((R >> 2) & 3).
    int aoff[TMW][2], boff[TNW][2];
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        const int ra = (wm * TMW + i) * 32 + r;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = X3 ? (2 * h + q) : (2 * q + h);
            aoff[i][q] = ra * DBK + ((c ^ ((ra >> 2) & 3)) << 2);
        }
        if (AIMG) {   // image tile ra >> 7 of the block tile; bf16 plane row = 8 floats, this lane's chunk = h ^ ((row >> 3) & 1)
            const int rl = ra & 127;
            aoff[i][0] = (ra >> 7) * D_TILE + rl * 8 + ((h ^ ((rl >> 3) & 1)) << 2);   // hi plane
            aoff[i][1] = aoff[i][0] + 1024;                                             // lo plane
        }
    }
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int rb = (wn * TNW + j) * 32 + r;
        if (X3) {  // bf16 plane row = 32 B = 8 floats; this lane's 8 k-values = chunk h ^ ((row >> 3) & 1)
            const int ch = h ^ ((rb >> 3) & 1);
            boff[j][0] = A_TILE + rb * 8 + ch * 4;          // hi plane
            boff[j][1] = A_TILE + 1024 + rb * 8 + ch * 4;   // lo plane
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q) boff[j][q] = A_TILE + rb * DBK + (((2 * q + h) ^ ((rb >> 2) & 3)) << 2);
        }
    }

    if (X3) {
        // Software-pipelined: the fragments of K-step kt sit in registers while its 12 MFMAs issue; under them the
        // wave reads the fragments of K-step kt + 1 (LDS latency, AdaGN affine and the hi / lo split hide under the
        // matrix pipe instead of in front of it) and the DMA runs DNS K-steps ahead.  One 32x32x16 chunk per K-step:
        // lane half h holds k = 8h .. 8h+7 of both operands.  Two fragment sets: set (kt & 1) is multiplied while
        // the other is filled, so every LDS read of the next step issues right after the barrier.
        bf16x8 ahi[2][TMW], alo[2][TMW], bhi[2][TNW], blo[2][TNW];
        auto load_frags = [&](const float* st, int kt, int f) {
            f32x4 pa0, pa1, po0, po1;
            if (HAS_PRO) {
                pa0 = *reinterpret_cast<const f32x4*>(pro_lds + kt * DBK + 8 * h);
                pa1 = *reinterpret_cast<const f32x4*>(pro_lds + kt * DBK + 8 * h + 4);
                po0 = *reinterpret_cast<const f32x4*>(pro_lds + g.K + kt * DBK + 8 * h);
                po1 = *reinterpret_cast<const f32x4*>(pro_lds + g.K + kt * DBK + 8 * h + 4);
            }
            f32x4 x0[TMW], x1[TMW];
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                x0[i] = *reinterpret_cast<const f32x4*>(st + aoff[i][0]);
                x1[i] = *reinterpret_cast<const f32x4*>(st + aoff[i][1]);
                if (AIMG) {   // already hi / lo planes
                    ahi[f][i] = __builtin_bit_cast(bf16x8, x0[i]);
                    alo[f][i] = __builtin_bit_cast(bf16x8, x1[i]);
                }
            }
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                bhi[f][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(st + boff[j][0]));
                blo[f][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(st + boff[j][1]));
            }
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                if (AIMG) continue;
                if (HAS_PRO) {
                    x0[i] = x0[i] * pa0 + po0;
                    x1[i] = x1[i] * pa1 + po1;
                }
#ifdef GEMM_DIAG_NOSPLIT
                ahi[f][i] = __builtin_bit_cast(bf16x8, x0[i]);
                alo[f][i] = __builtin_bit_cast(bf16x8, x1[i]);
#else
                split8(x0[i], x1[i], ahi[f][i], alo[f][i]);
#endif
            }
        };
        // counted waits: NPIECE wave-instructions per K-step in flight (4; 3 with 64-row, 6 with 256-row tiles)
        if (DNS >= 4 && nk >= 4) dma::wait_vm<3 * NPIECE>();
        else if (nk >= 3) dma::wait_vm<2 * NPIECE>();
        else if (nk == 2) dma::wait_vm<NPIECE>();
        else dma::wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        load_frags(smem, 0, 0);   // K-step 0 into set 0
        auto kstep = [&](int kt, int cur) {
            // own reads of stage kt are complete (its slot may be refilled) and own pieces of stage kt + 1 landed
            const int ahead = min(nk - 1, kt + DNS - 1) - (kt + 1);   // K-steps in flight beyond kt + 1
            if (DNS >= 4 && ahead >= 2) dma::wait_vm_lgkm0<2 * NPIECE>();
            else if (ahead >= 1) dma::wait_vm_lgkm0<NPIECE>();
            else dma::wait_vm_lgkm0<0>();
            // this step's fragments were read during the previous one and the wait above covered them: an empty asm
            // "redefines" the registers so the compiler's wait-count pass does not park its own lgkmcnt(0) in front
            // of the first MFMA — behind the NEXT step's reads issued below, exposing their latency every step
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                asm volatile("" : "+v"(ahi[cur][i]));
                asm volatile("" : "+v"(alo[cur][i]));
            }
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                asm volatile("" : "+v"(bhi[cur][j]));
                asm volatile("" : "+v"(blo[cur][j]));
            }
#ifndef GEMM_DIAG_NOBARRIER   // diagnostic build: wrong results, the cost of the per-K-step lock-step
            __builtin_amdgcn_s_barrier();
#endif
            if (kt + DNS < nk) issue(kt + DNS);
            // next K-step's slot; past the end a landed slot is re-read and the values are never used
            const int kn = min(kt + 1, nk - 1);
            load_frags(smem + (kn % DNS) * STAGE, kn, cur ^ 1);
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int i = 0; i < TMW; ++i) {
#ifdef GEMM_DIAG_NOMFMA   // diagnostic: keep the operand reads alive with one VALU op per fragment instead
                    acc[i][j][0] += (float)alo[cur][i][0] + (float)bhi[cur][j][0] + (float)ahi[cur][i][0] + (float)blo[cur][j][0];
#else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[cur][i], bhi[cur][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[cur][i], blo[cur][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[cur][i], bhi[cur][j], acc[i][j], 0, 0, 0);
#endif
                }
        };
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            kstep(kt, 0);
            kstep(kt + 1, 1);
        }
        if (kt < nk) kstep(kt, 0);
    } else {
        for (int kt = 0; kt < nk; ++kt) {
            // own pieces of K-step kt have landed once at most the younger K-steps' DMAs are outstanding
            const int ahead = min(nk - 1 - kt, DNS - 2);   // K-steps issued after kt and not yet waited for
            if (DNS >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // everyone's pieces of kt are in; everyone is done reading stage (kt-1) % DNS
            if (kt + DNS - 1 < nk) issue(kt + DNS - 1);
            const float* st = smem + (kt % DNS) * STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // lane half h holds k = 8*kk + 4*h + e: the same k permutation on both operands
                f32x4 fa[TMW], fb[TNW];
#pragma unroll
                for (int i = 0; i < TMW; ++i) fa[i] = *reinterpret_cast<const f32x4*>(st + aoff[i][kk]);
#pragma unroll
                for (int j = 0; j < TNW; ++j) fb[j] = *reinterpret_cast<const f32x4*>(st + boff[j][kk]);
                if (HAS_PRO) {
                    const f32x4 pa4 = *reinterpret_cast<const f32x4*>(pro_lds + kt * DBK + kk * 8 + 4 * h);
                    const f32x4 po4 = *reinterpret_cast<const f32x4*>(pro_lds + g.K + kt * DBK + kk * 8 + 4 * h);
#pragma unroll
                    for (int i = 0; i < TMW; ++i) fa[i] = fa[i] * pa4 + po4;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TMW; ++i)
#pragma unroll
                        for (int j = 0; j < TNW; ++j) acc[i][j] = mfma32(fa[i][e], fb[j][e], acc[i][j]);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // ring is dead: the epilogue reuses it

    dma::epilogue_t<TMW, TNW, WMN, false, ACTBWD>(g, T, acc, smem, wave, lane, wm, wn);
}

// W (Nout, ldw) fp32 -> the tiled split-bf16 image the X3 kernel streams: for column tile ct (128 rows of W) and
// K-step kt (16 k) the 8 KiB block at float offset (ct * K/16 + kt) * 2048 holds the hi plane (row rb at rb * 8
// floats, its two 8-k chunks swapped when (rb >> 3) & 1 — the read-side swizzle) then the lo plane at +1024.
// Rows past Nout repeat the last row (masked in the GEMM epilogue).  One thread per (block, row, chunk).
__global__ void split_bf16_tiled_kernel(const float* __restrict__ W, float* __restrict__ img, int Nout, int K,
                                        int ldw, size_t total) {
    const int nk = K / DBK;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i & 1), rb = (int)((i >> 1) & 127);
        const size_t blk = i >> 8;
        const int kt = (int)(blk % nk), ct = (int)(blk / nk);
        const int c = ch ^ ((rb >> 3) & 1);
        const float* src = W + (size_t)min(ct * DBN + rb, Nout - 1) * ldw + kt * DBK + c * 8;
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
        bf16x8 hi, lo;
        split8(x0, x1, hi, lo);
        float* dst = img + blk * D_TILE + rb * 8 + ch * 4;
        *reinterpret_cast<u32x4*>(dst) = __builtin_bit_cast(u32x4, hi);
        *reinterpret_cast<u32x4*>(dst + 1024) = __builtin_bit_cast(u32x4, lo);
    }
}

template <int BM>
int dma_launch_aimg(const GemmArgs& g, hipStream_t st) {
    const int tilesM = (g.rows + BM - 1) / BM, tilesN = (g.Nout + DBN - 1) / DBN;
    const size_t lds = (size_t)(d_main_floats(3, BM) + 2 * g.K) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<3, false, true, BM, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL((gemm_dma_kernel<3, false, true, BM, true>), dim3(g.B * tilesM * tilesN), dim3(DNT), lds, st, g);
    return (int)hipGetLastError();
}

template <int DNS, bool X3, int BM = 128>
int dma_launch_t(const GemmArgs& g, hipStream_t st) {
    const int tilesM = (g.rows + BM - 1) / BM, tilesN = (g.Nout + DBN - 1) / DBN;
    const size_t lds = (size_t)(d_main_floats(DNS, BM) + (g.pro_a ? 2 * g.K : 0)) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<DNS, true, X3, BM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<DNS, false, X3, BM>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    const dim3 grid(g.B * tilesM * tilesN);
    if (g.mul_u || g.pre_out || g.dot_x) {   // the training path's epilogue forms: their own kernel instantiations
        if ((g.mul_u && g.pro_a) || (g.pre_out && g.c_img) || (g.dot_x && (g.residual || g.mul_u || g.pre_out || !g.stats || g.C2 || g.c_img))) return -9;
        static size_t attr2 = 0;
        if (lds > attr2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<DNS, false, X3, BM, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<DNS, true, X3, BM, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr2 = lds;
        }
        if (g.pro_a) hipLaunchKernelGGL((gemm_dma_kernel<DNS, true, X3, BM, false, true>), grid, dim3(DNT), lds, st, g);
        else hipLaunchKernelGGL((gemm_dma_kernel<DNS, false, X3, BM, false, true>), grid, dim3(DNT), lds, st, g);
    } else if (g.pro_a) hipLaunchKernelGGL((gemm_dma_kernel<DNS, true, X3, BM>), grid, dim3(DNT), lds, st, g);
    else hipLaunchKernelGGL((gemm_dma_kernel<DNS, false, X3, BM>), grid, dim3(DNT), lds, st, g);
    return (int)hipGetLastError();
}

}  // namespace

bool gemm_f32_dma_supported(const GemmArgs& g, int precision) {
    if (precision < 0) precision = g.precision;
    if (g.C2 && ((g.n_split % DBN) || g.stats || g.residual || (g.ldc2 & 3) || g.n_split <= 0 || g.n_split >= g.Nout))
        return false;
    // K <= 1024 with the AdaGN prologue (its coefficients are parked in LDS: 8 K bytes); without it up to 2048 (the ConvNeXt
    // conditioner's 4 C -> C linears at C = 384: K = 1536)
    return g.rows >= (precision == 1 ? 64 : 128) && g.K % DBK == 0 && g.K <= (g.pro_a ? 1024 : 2048) && !(g.Nout & 3) && !(g.ldc & 3) && !(g.ldr & 3) &&
           !(g.lda & 3) && !(g.ldw & 7);
}

int gemm_f32_dma_launch(const GemmArgs& g, hipStream_t st) {
    static int ns3 = -1;
    if (ns3 < 0) {
        const char* e = getenv("GECCO_GEMM_STAGES");   // 3 stages = 51 KB LDS = three blocks per CU (measured best)
        ns3 = (e && atoi(e) == 4) ? 0 : 1;
    }
    if (g.a_img) {
        if (!(g.precision == 1 && g.w_img && !g.pro_a && g.rows >= 128 && g.rows % 128 == 0)) return -9;
        static int areg = -1;
        if (areg < 0) {
            const char* e = getenv("GECCO_AREG");   // 0: the image through the LDS ring (A/B runs)
            areg = (e && atoi(e) == 0) ? 0 : 1;
        }
        if (areg && gemm_x3_areg_supported(g)) return gemm_x3_areg_launch(g, st);
        return (g.rows >= 256 && g.K >= 512) ? dma_launch_aimg<256>(g, st) : dma_launch_aimg<128>(g, st);
    }
    if (g.c_img && !(g.precision == 1 && g.w_img && !g.residual && !g.stats && !g.C2 && g.rows >= 128 && g.rows % 128 == 0 &&
                     g.Nout % 16 == 0 && (g.c_img != 2 || g.Nout % 64 == 0)))
        return -9;
    if (g.precision == 1 && g.w_img) {
        if (g.rows < 128) return dma_launch_t<3, true, 64>(g, st);
        static int bm256 = -1;
        if (bm256 < 0) {
            const char* e = getenv("GECCO_GEMM_BM256");   // 0: keep the 128-row tiles (A/B runs)
            bm256 = (e && atoi(e) == 0) ? 0 : 1;
        }
        // measured at C2 (B = 64, N = 2048): 768 -> 384 with residual 0.321 -> 0.288 ms; 384 -> 384 and the prologue
        // form 384 -> 768 are 3-8 % slower with the tall tile (fewer, longer blocks; the prologue form spills): long K only
        if (bm256 && g.rows >= 256 && g.rows % 128 == 0 && !g.pro_a && g.K >= 512) return dma_launch_t<3, true, 256>(g, st);
        return ns3 ? dma_launch_t<3, true>(g, st) : dma_launch_t<4, true>(g, st);
    }
    return ns3 ? dma_launch_t<3, false>(g, st) : dma_launch_t<4, false>(g, st);
}

namespace {

// several weights in one launch (blockIdx.y = job): the per-forward weight pass of the set transformer
__global__ void split_bf16_tiled_multi_kernel(SplitJobs jobs) {
    const SplitJob j = jobs.job[blockIdx.y];
    const int nk = j.K / DBK;
    const size_t total = (size_t)((j.Nout + DBN - 1) / DBN) * nk * 256;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i & 1), rb = (int)((i >> 1) & 127);
        const size_t blk = i >> 8;
        const int kt = (int)(blk % nk), ct = (int)(blk / nk);
        const int c = ch ^ ((rb >> 3) & 1);
        f32x4 x0, x1;
        if (j.pad_ == 4) {
            // the image of W^T from W (K, ldw) itself (the dX product of a linear is linear(dY, W^T)): row n of the image is
            // column n of W — eight strided reads, consecutive n in consecutive threads
            const float* src = j.W + (size_t)(kt * DBK + c * 8) * j.ldw + min(ct * DBN + rb, j.Nout - 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x0[e] = src[(size_t)e * j.ldw];
                x1[e] = src[(size_t)(4 + e) * j.ldw];
            }
        } else {
            const float* src = j.W + (size_t)min(ct * DBN + rb, j.Nout - 1) * j.ldw + kt * DBK + c * 8;
            x0 = *reinterpret_cast<const f32x4*>(src);
            x1 = *reinterpret_cast<const f32x4*>(src + 4);
        }
        bf16x8 hi, lo;
        split8(x0, x1, hi, lo);
        float* dst = j.img + blk * D_TILE + rb * 8 + ch * 4;
        *reinterpret_cast<u32x4*>(dst) = __builtin_bit_cast(u32x4, hi);
        *reinterpret_cast<u32x4*>(dst + 1024) = __builtin_bit_cast(u32x4, lo);
    }
}

}  // namespace

int split_bf16_tiled_multi_launch(const SplitJobs& jobs, hipStream_t st) {
    if (jobs.n <= 0) return 0;
    hipLaunchKernelGGL(split_bf16_tiled_multi_kernel, dim3(64, jobs.n), dim3(256), 0, st, jobs);
    return (int)hipGetLastError();
}

size_t split_bf16_image_bytes(int Nout, int K) { return (size_t)((Nout + DBN - 1) / DBN) * DBN * K * sizeof(float); }

int split_bf16_tiled_launch(const float* W, void* img, int Nout, int K, int ldw, hipStream_t st) {
    const size_t total = (size_t)((Nout + DBN - 1) / DBN) * (K / DBK) * 256;
    const unsigned grid = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(split_bf16_tiled_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, W, static_cast<float*>(img),
                       Nout, K, ldw, total);
    return (int)hipGetLastError();
}
