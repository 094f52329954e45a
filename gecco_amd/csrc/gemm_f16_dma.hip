// Fused linear layer, LDS-DMA kernel in fp16 arithmetic (precision 2): same contract as gemm_f32_dma.hip
//
//   C[b, m, n] = residual[b, m, n] + act( sum_k (A[b, m, k] * pa[b, k] + po[b, k]) * W[n, k] + bias[n] )  (+ GroupNorm partials)
//
// with both operands rounded to fp16 (round-to-nearest-even) and ONE v_mfma_f32_32x32x16_f16 per 16 k, fp32
// accumulation.  fp16 keeps 11 significant bits: measured 3.0e-4 from the fp32 reference on the golden networks
// (plain bf16: 2.3e-3, which misses the 1e-3 bar; split-bf16: 1.3e-5 at 3x the matrix work).  Every fp16 operand of
// the path is range-safe by construction: the A operand is an AdaGN output, a GaussianActivation output, a softmax
// average of projected inducer states or a point coordinate scaled by c_in; W are layer weights.
//
// With a third of the matrix work of the split-bf16 kernel the loop is bound by the tile fill, so it differs where
// that matters (measured with tools/probe/dma_rate.hip):
//   * K-steps of 32: the fp32 A tile is [BM][32] — a DMA wave-instruction moves 8 rows x 128 B, whole cache lines —
//     and one barrier / wait / issue round covers 8 MFMAs per wave instead of 4;
//   * W travels as a pre-tiled fp16 image (split_f16_tiled_kernel): one 8 KiB block per (128-column tile, K-step)
//     that IS the LDS image, so its DMA pieces are 1 KiB of consecutive bytes and B costs 2 bytes per element;
//   * k permutation: lane half h owns k = 16h .. 16h+15 of a K-step (64 consecutive bytes of its A row, four
//     ds_read_b128); MFMA c of the step multiplies k = 16h + 8c .. 16h + 8c + 7 — the image stores W in that order.
// Ring of DNS stages with the K-step being multiplied held in registers (two fragment sets), counted vmcnt waits and
// one raw s_barrier per K-step as in the split-bf16 kernel.  Requires K % 32 == 0, Nout % 4 == 0, rows >= 64.
#include "gemm_dma_common.h"

#include <stdlib.h>

namespace {

using dma::DBN;
using dma::DNT;
using dma::D_EPI;
using dma::dma16;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int FBK = 32;                    // k per ring stage
constexpr int FB_TILE = DBN * FBK / 2;     // floats of the fp16 W tile per stage (8 KiB)
constexpr int fa_tile(int bm, bool a16) { return a16 ? bm * FBK / 2 : bm * FBK; }   // floats of the A tile per stage
constexpr int f_stage(int bm, bool a16) { return fa_tile(bm, a16) + FB_TILE; }
constexpr int f_main_floats(int ns, int bm, bool a16) {
    return ns * f_stage(bm, a16) > D_EPI ? ns * f_stage(bm, a16) : D_EPI;
}

__device__ __forceinline__ f16x8 cvt8(const f32x4& x0, const f32x4& x1) {
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = (_Float16)x0[e];
        v[4 + e] = (_Float16)x1[e];
    }
    return v;
}

template <int N>
__device__ __forceinline__ void wait_ahead(int ahead, bool lgkm) {
    // N = DMA wave-instructions per wave per K-step; `ahead` K-steps may stay in flight
    if (lgkm) {
        if (ahead >= 2) dma::wait_vm_lgkm0<2 * N>();
        else if (ahead == 1) dma::wait_vm_lgkm0<N>();
        else dma::wait_vm_lgkm0<0>();
    } else {
        if (ahead >= 2) dma::wait_vm<2 * N>();
        else if (ahead == 1) dma::wait_vm<N>();
        else dma::wait_vm<0>();
    }
}

#ifdef GEMM_STAMPS   // diagnostic build (tools/probe): per-block s_memtime stamps of the kernel's phases
__device__ unsigned long long g_stamps[16384 * 8];
#define STAMP(i)                                                                                     \
    do {                                                                                             \
        if (threadIdx.x == 0 && blockIdx.x < 16384) g_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define STAMP(i)
#endif

// A16: the A operand is already fp16 in memory (an intermediate a previous kernel stored that way): its tile is laid
// out like the W tile (64-byte rows, 16 rows per DMA piece), fragments are read as they are, 16 KiB per stage and
// three blocks per CU.  C16: the output is stored as fp16 (dma::epilogue).
// ACTBWD: the training path's epilogue forms (GemmArgs::mul_u / pre_out, dma::epilogue_t) — their own instantiations, as in
// gemm_f32_dma.hip; the inference kernels are the code they were.
template <int DNS, bool HAS_PRO, int BM, bool A16, bool C16, bool ACTBWD = false>
__global__ __launch_bounds__(DNT, A16 ? 3 : 2) void gemm_f16_kernel(GemmArgs g) {
    static_assert(DNS == 2 || DNS == 3, "ring of 2 or 3 stages");
    static_assert(!(A16 && HAS_PRO), "the AdaGN prologue needs the fp32 operand");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int A_TILE = fa_tile(BM, A16), STAGE = f_stage(BM, A16);
    float* pro_lds = smem + f_main_floats(DNS, BM, A16);   // pa[0..K) | po[0..K)

    STAMP(0);
    const dma::Tile T = dma::tile_of_block<BM>(g);
    const int ct = T.ct, b = T.b, m0 = T.m0;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 128-row tiles: 4 x 1 waves of 32 x 128 (A rows are wave-private); 64-row tiles: 2 x 2 waves of 32 x 64
    constexpr int WMN = BM == 128 ? 4 : 2, WNN = 4 / WMN;
    constexpr int TMW = 1, TNW = 4 / WNN;
    constexpr int NAP = A16 ? BM / 64 : BM / 32;   // A pieces (8 rows x 128 B, or 16 rows x 64 B of fp16) per wave per K-step
    constexpr int NPIECE = NAP + 2;    // + two 1 KiB pieces of the W image
    const int wm = wave / WNN, wn = wave % WNN;
    const int r = lane & 31, h = lane >> 5;

    // ---- DMA sources.  A piece p = rows 8p .. 8p+7, lane l -> row 8p + (l >> 3), LDS chunk l & 7 holding global
    // chunk (l & 7) ^ ((row >> 1) & 7) (the read-side swizzle, applied to the source address).
    // fp16 A: piece p = rows 16p .. 16p+15, lane l -> row 16p + (l >> 2), LDS chunk l & 3 <- global chunk ^ ((row >> 2) & 3).
    const float* asrc[NAP];   // advanced by one K-step = 32 k: 32 floats, or 16 floats' worth of fp16
    constexpr int A_KSTEP = A16 ? FBK / 2 : FBK;
#pragma unroll
    for (int q = 0; q < NAP; ++q) {
        if (A16) {
            const _Float16* Ab = reinterpret_cast<const _Float16*>(g.A) + (size_t)b * g.rows * g.lda;
            const int row = (NAP * wave + q) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((row >> 2) & 3);
            asrc[q] = reinterpret_cast<const float*>(Ab + (size_t)min(m0 + row, g.rows - 1) * g.lda + c * 8);
        } else {
            const float* Ab = g.A + (size_t)b * g.rows * g.lda;
            const int row = (NAP * wave + q) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            asrc[q] = Ab + (size_t)min(m0 + row, g.rows - 1) * g.lda + c * 4;
        }
    }
    const int nk = g.K / FBK;
    const float* bsrc = static_cast<const float*>(g.w_img) + (size_t)b * g.w_img_bstride + (size_t)ct * nk * FB_TILE + (2 * wave) * 256 + lane * 4;
    auto issue = [&](int kt) {
#ifdef GEMM_DIAG_NODMA
        return;
#endif
        float* st = smem + (kt % DNS) * STAGE;
#pragma unroll
        for (int q = 0; q < NAP; ++q) dma16(asrc[q] + kt * A_KSTEP, st + (NAP * wave + q) * 256);
        dma16(bsrc + (size_t)kt * FB_TILE, st + A_TILE + (2 * wave) * 256);
        dma16(bsrc + (size_t)kt * FB_TILE + 256, st + A_TILE + (2 * wave + 1) * 256);
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
    STAMP(1);
#pragma unroll
    for (int p = 0; p < DNS; ++p)
        if (p < nk) issue(p);

    f32x16 acc[TMW][TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;

    // fragment addressing (float offsets inside a stage)
    const int ra = wm * 32 + r;
    int aoff[4], boff[TNW][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        aoff[q] = A16 ? ra * 16 + (((2 * h + (q & 1)) ^ ((ra >> 2) & 3)) << 2)      // q = 0, 1: the two fp16 chunks
                      : ra * FBK + (((4 * h + q) ^ ((ra >> 1) & 7)) << 2);
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int rb = (wn * TNW + j) * 32 + r;   // fp16 row = 64 B = 16 floats; chunk 2h + c ^ ((rb >> 2) & 3)
#pragma unroll
        for (int c = 0; c < 2; ++c) boff[j][c] = A_TILE + rb * 16 + (((2 * h + c) ^ ((rb >> 2) & 3)) << 2);
    }

    f16x8 fa[2][2], fb[2][TNW][2];   // [set][chunk], [set][column tile][chunk]
    auto load_frags = [&](const float* st, int kt, int f) {
        if (A16) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
                fa[f][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(st + aoff[c]));
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    fb[f][j][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(st + boff[j][c]));
            return;
        }
        f32x4 x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const f32x4*>(st + aoff[q]);
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                fb[f][j][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(st + boff[j][c]));
        if (HAS_PRO) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 pa4 = *reinterpret_cast<const f32x4*>(pro_lds + kt * FBK + 16 * h + 4 * q);
                const f32x4 po4 = *reinterpret_cast<const f32x4*>(pro_lds + g.K + kt * FBK + 16 * h + 4 * q);
                x[q] = x[q] * pa4 + po4;
            }
        }
        fa[f][0] = cvt8(x[0], x[1]);
        fa[f][1] = cvt8(x[2], x[3]);
    };

    wait_ahead<NPIECE>(min(nk, DNS) - 1, false);
    __builtin_amdgcn_s_barrier();
    STAMP(2);
    load_frags(smem, 0, 0);   // K-step 0 into set 0
    auto kstep = [&](int kt, int cur) {
        // own reads of stage kt are complete (its slot may be refilled) and own pieces of stage kt + 1 landed
        wait_ahead<NPIECE>(min(nk - 1, kt + DNS - 1) - (kt + 1), true);
        // this step's fragments were read during the previous one and the wait above covered them: an empty asm
        // "redefines" the registers so the compiler's wait-count pass does not park its own lgkmcnt(0) in front of
        // the first MFMA — behind the NEXT step's reads issued below, which would expose their latency every step
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            asm volatile("" : "+v"(fa[cur][c]));
#pragma unroll
            for (int j = 0; j < TNW; ++j) asm volatile("" : "+v"(fb[cur][j][c]));
        }
        __builtin_amdgcn_s_barrier();
        if (kt + DNS < nk) issue(kt + DNS);
        // next K-step's slot; past the end a landed slot is re-read and the values are never used
        const int kn = min(kt + 1, nk - 1);
        load_frags(smem + (kn % DNS) * STAGE, kn, cur ^ 1);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#ifdef GEMM_DIAG_NOMFMA
                acc[0][j][0] += (float)fa[cur][c][0] + (float)fb[cur][j][c][0];
#else
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][c], fb[cur][j][c], acc[0][j], 0, 0, 0);
#endif
    };
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        kstep(kt, 0);
        kstep(kt + 1, 1);
    }
    if (kt < nk) kstep(kt, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // ring is dead: the epilogue reuses it
    STAMP(3);
    dma::epilogue_t<TMW, TNW, WMN, C16, ACTBWD>(g, T, acc, smem, wave, lane, wm, wn);
    STAMP(4);
}

// W (Nout, ldw) fp32 -> the tiled fp16 image: for column tile ct (128 rows of W) and K-step kt (32 k) the 8 KiB block
// at float offset (ct * K/32 + kt) * 2048 holds row rb at rb * 16 floats (64 B = 32 fp16); its 16-byte chunk
// s ^ ((rb >> 2) & 3) holds k = 16 (s >> 1) + 8 (s & 1) .. +7 (lane half s >> 1, MFMA s & 1 of the step).  Rows past
// Nout repeat the last row (masked in the GEMM epilogue).  One thread per (block, row, chunk).
// lo == 1: the image of the LOW part, fp16(W - float(fp16(W))) — the second term of a two-term fp16 weight (mixed mode);
// lo == 4: W is (K, ldw) and the image is of W^T (the training path's dX products)
__device__ __forceinline__ void f16_image_item(const float* __restrict__ W, float* __restrict__ img, int Nout, int K,
                                               int ldw, size_t i, int lo = 0) {
    const int nk = K / FBK;
    const int chp = (int)(i & 3), rb = (int)((i >> 2) & 127);
    const size_t blk = i >> 9;
    const int kt = (int)(blk % nk), ct = (int)(blk / nk);
    const int s = chp ^ ((rb >> 2) & 3);
    f32x4 w0, w1;
    if (lo == 4) {   // the image of W^T from W (K, ldw) itself (the dX product of a linear is linear(dY, W^T)): eight strided reads
        const float* src = W + (size_t)(kt * FBK + 16 * (s >> 1) + 8 * (s & 1)) * ldw + min(ct * DBN + rb, Nout - 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            w0[e] = src[(size_t)e * ldw];
            w1[e] = src[(size_t)(4 + e) * ldw];
        }
    } else {
        const float* src = W + (size_t)min(ct * DBN + rb, Nout - 1) * ldw + kt * FBK + 16 * (s >> 1) + 8 * (s & 1);
        w0 = *reinterpret_cast<const f32x4*>(src);
        w1 = *reinterpret_cast<const f32x4*>(src + 4);
    }
    f16x8 v = cvt8(w0, w1);
    if (lo == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            w0[e] -= (float)v[e];
            w1[e] -= (float)v[4 + e];
        }
        v = cvt8(w0, w1);
    }
    *reinterpret_cast<u32x4*>(img + blk * FB_TILE + rb * 16 + chp * 4) = __builtin_bit_cast(u32x4, v);
}

// The LOW part of a two-term fp16 weight as fp8 (e4m3), scaled by 2^16 (h8_scales.h), for v_mfma_scale_f32_32x32x64_f8f6f4: for column
// tile ct and 64-k step S the 8 KiB block at float offset (ct * K/64 + S) * 2048 holds row rb at rb * 16 floats (64 B =
// 64 fp8); its 16-byte chunk s ^ ((rb >> 2) & 3) holds k = 64 S + 32 (s & 1) + 16 (s >> 1) .. + 15 — lane half s >> 1,
// bytes 16 (s & 1) .. of the 32-byte B operand: the same (lane half, chunk) addressing as the fp16 stages, and the k
// order in which gemm_f16_astat.hip packs its fp16 A fragments into the fp8 A operand.  |w_lo| <= 2^-11 |w|: the scale
// keeps weights up to |w| = 14 inside e4m3's range (448); larger ones saturate (their hi term is unaffected).
constexpr float LO8_SCALE = H8_WL_SCALE;   // 2^16 (h8_scales.h); the kernel's scale_b is its inverse
__device__ __forceinline__ void f8lo_image_item(const float* __restrict__ W, float* __restrict__ img, int Nout, int K,
                                                int ldw, size_t i) {
    const int nk = K / 64;
    const int chp = (int)(i & 3), rb = (int)((i >> 2) & 127);
    const size_t blk = i >> 9;
    const int S = (int)(blk % nk), ct = (int)(blk / nk);
    const int s = chp ^ ((rb >> 2) & 3);
    const float* src = W + (size_t)min(ct * DBN + rb, Nout - 1) * ldw + S * 64 + 32 * (s & 1) + 16 * (s >> 1);
    u32x4 out;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(src + 4 * q);
        float lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) lo[e] = __builtin_fminf(__builtin_fmaxf((w[e] - (float)(_Float16)w[e]) * LO8_SCALE, -448.f), 448.f);
        int pk = 0;
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], pk, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], pk, true);
        out[q] = (unsigned)pk;
    }
    *reinterpret_cast<u32x4*>(img + blk * FB_TILE + rb * 16 + chp * 4) = out;
}

// Sample blockIdx.y's folded image (kernels.h fold_f16_image_launch): the items of f16_image_item with W's k scaled by pa[b, k]; the blocks
// past the image's fold the offsets into the bias, one wave per output column
__global__ __launch_bounds__(256) void fold_f16_image_kernel(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ pa,
                                                             const float* __restrict__ po, float* __restrict__ img, float* __restrict__ bias_out,
                                                             int Nout, int K, int ldw, int item_blocks) {
    const int b = blockIdx.y;
    const int nk = K / FBK, tilesN = (Nout + DBN - 1) / DBN;
    if ((int)blockIdx.x >= item_blocks) {
        const int n = ((int)blockIdx.x - item_blocks) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (n >= Nout) return;
        float acc = 0.f;
        for (int k0 = 0; k0 < K; k0 += 1024) {   // sixteen independent loads per operand in flight, then the sum in a fixed order
            float wv[16], ov[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int k = k0 + 64 * j + lane;
                wv[j] = k < K ? W[(size_t)n * ldw + k] : 0.f;
                ov[j] = k < K ? po[(size_t)b * K + k] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += ov[j] * wv[j];
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) bias_out[(size_t)b * Nout + n] = (bias ? bias[n] : 0.f) + acc;
        return;
    }
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, total = (size_t)tilesN * nk * 512;
    if (i >= total) return;
    const int chp = (int)(i & 3), rb = (int)((i >> 2) & 127);
    const size_t blk = i >> 9;
    const int kt = (int)(blk % nk), ct = (int)(blk / nk);
    const int s = chp ^ ((rb >> 2) & 3);
    const int k0 = kt * FBK + 16 * (s >> 1) + 8 * (s & 1);
    const float* src = W + (size_t)min(ct * DBN + rb, Nout - 1) * ldw + k0;
    const float* sc = pa + (size_t)b * K + k0;
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(src) * *reinterpret_cast<const f32x4*>(sc);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(src + 4) * *reinterpret_cast<const f32x4*>(sc + 4);
    *reinterpret_cast<u32x4*>(img + (size_t)b * tilesN * nk * FB_TILE + blk * FB_TILE + rb * 16 + chp * 4) = __builtin_bit_cast(u32x4, cvt8(w0, w1));
}

__global__ void split_f16_tiled_kernel(const float* __restrict__ W, float* __restrict__ img, int Nout, int K, int ldw,
                                       size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        f16_image_item(W, img, Nout, K, ldw, i);
}

// Two-term fp16 weights for a kernel that walks ONE stream (inducer_chain_f16.hip with TWO): per 128-column tile the NK blocks of
// fp16(W), then the NK blocks of fp16(W - fp16(W)) — block (2 ct + lo) nk + kt of the stream is block ct nk + kt of that image
__device__ __forceinline__ void f16_image_item_hilo(const float* __restrict__ W, float* __restrict__ img, int Nout, int K, int ldw, size_t i) {
    const int nk = K / FBK;
    const size_t blk2 = i >> 9;
    const int idx = (int)(blk2 % (2 * nk)), ct = (int)(blk2 / (2 * nk)), lo = idx >= nk, kt = idx % nk;
    const size_t src_item = (((size_t)ct * nk + kt) << 9) | (i & 511);
    // the plain item writes to img + (ct nk + kt) FB_TILE: shift the base so that it lands on block blk2
    f16_image_item(W, img + ((ptrdiff_t)blk2 - (ptrdiff_t)((size_t)ct * nk + kt)) * FB_TILE, Nout, K, ldw, src_item, lo);
}

__global__ void split_f16_tiled_multi_kernel(SplitJobs jobs) {
    const SplitJob j = jobs.job[blockIdx.y];
    if (j.pad_ == 8) {
        const size_t total2 = (size_t)((j.Nout + DBN - 1) / DBN) * (j.K / FBK) * 1024;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total2; i += (size_t)gridDim.x * blockDim.x)
            f16_image_item_hilo(j.W, j.img, j.Nout, j.K, j.ldw, i);
        return;
    }
    if (j.pad_ == 2) {
        const size_t total8 = (size_t)((j.Nout + DBN - 1) / DBN) * (j.K / 64) * 512;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (size_t)gridDim.x * blockDim.x)
            f8lo_image_item(j.W, j.img, j.Nout, j.K, j.ldw, i);
        return;
    }
    const size_t total = (size_t)((j.Nout + DBN - 1) / DBN) * (j.K / FBK) * 512;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        f16_image_item(j.W, j.img, j.Nout, j.K, j.ldw, i, j.pad_);
}

template <int DNS, int BM, bool A16, bool C16>
int f16_launch_t(const GemmArgs& g, hipStream_t st) {
    const int tilesM = (g.rows + BM - 1) / BM, tilesN = (g.Nout + DBN - 1) / DBN;
    const size_t lds = (size_t)(f_main_floats(DNS, BM, A16) + (g.pro_a ? 2 * g.K : 0)) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        if (!A16)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<DNS, !A16, BM, A16, C16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<DNS, false, BM, A16, C16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    const dim3 grid(g.B * tilesM * tilesN);
    if (g.dot_x && (g.mul_u || g.pre_out || !g.stats || g.C2 || C16 || (g.residual && !A16))) return -9;
    if constexpr (A16 && !C16) {
        if (g.dot_x) {   // (with or without a residual: gemm_dma_common.h res_and_dot)   // the dX product of an MLP's first linear from du stored as halves (autograd.py `_du16_ok`), with the AdaGN backward's partials
            static size_t attr3 = 0;
            if (lds > attr3) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<DNS, false, BM, true, false, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr3 = lds;
            }
            hipLaunchKernelGGL((gemm_f16_kernel<DNS, false, BM, true, false, true>), grid, dim3(DNT), lds, st, g);
            return (int)hipGetLastError();
        }
    }
    if (g.mul_u || g.pre_out || g.dot_x) {   // the training path's epilogue forms: their own kernel instantiations
        // fp32 tensors, except that the KEEP form may store act(u) as fp16 (C16: the hidden layer of an MLP, which only the matrix pipe
        // reads again — as an fp16 operand either way) beside the fp32 pre-activation
        if (A16 || (C16 && g.mul_u) || (g.mul_u && g.pro_a) || g.c_img) return -9;
        if constexpr (!A16) {
            static size_t attr2 = 0;
            if (lds > attr2) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<DNS, true, BM, false, C16, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<DNS, false, BM, false, C16, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr2 = lds;
            }
            if (g.pro_a) hipLaunchKernelGGL((gemm_f16_kernel<DNS, true, BM, false, C16, true>), grid, dim3(DNT), lds, st, g);
            else hipLaunchKernelGGL((gemm_f16_kernel<DNS, false, BM, false, C16, true>), grid, dim3(DNT), lds, st, g);
        }
        return (int)hipGetLastError();
    }
    if (g.pro_a) {
        if (A16) return -9;
        hipLaunchKernelGGL((gemm_f16_kernel<DNS, !A16, BM, A16, C16>), grid, dim3(DNT), lds, st, g);
    } else {
        hipLaunchKernelGGL((gemm_f16_kernel<DNS, false, BM, A16, C16>), grid, dim3(DNT), lds, st, g);
    }
    return (int)hipGetLastError();
}

}  // namespace

bool gemm_f16_dma_supported(const GemmArgs& g) {
    if (g.C2 && ((g.n_split % DBN) || g.stats || g.residual || (g.ldc2 & 3) || g.n_split <= 0 || g.n_split >= g.Nout))
        return false;
    if (g.a_f16 && (g.pro_a || (g.lda & 7) || g.rows < 128)) return false;
    if (g.c_f16 && (g.residual || g.stats || g.rows < 128)) return false;
    // K <= 1024 with the AdaGN prologue (its coefficients are parked in LDS: 8 K bytes); without it up to 2048 (the ConvNeXt
    // conditioner's 4 C -> C linears at C = 384: K = 1536), as gemm_f32_dma.hip
    return g.rows >= 64 && g.K % FBK == 0 && g.K <= (g.pro_a ? 1024 : 2048) && !(g.Nout & 3) && !(g.ldc & 3) && !(g.ldr & 3) &&
           !(g.lda & 3) && !(g.ldw & 3);
}

int gemm_f16_dma_launch(const GemmArgs& g, hipStream_t st) {
    if (!g.w_img) return -9;
    static int ns = 0;
    if (!ns) {
        const char* e = getenv("GECCO_GEMM_F16_STAGES");
        ns = (e && atoi(e) == 2) ? 2 : 3;
    }
    if (g.a_f16 && g.c_f16) return f16_launch_t<3, 128, true, true>(g, st);
    if (g.a_f16) return f16_launch_t<3, 128, true, false>(g, st);
    if (g.c_f16) return f16_launch_t<3, 128, false, true>(g, st);
    if (g.rows < 128) return ns == 2 ? f16_launch_t<2, 64, false, false>(g, st) : f16_launch_t<3, 64, false, false>(g, st);
    return ns == 2 ? f16_launch_t<2, 128, false, false>(g, st) : f16_launch_t<3, 128, false, false>(g, st);
}

size_t split_f16_image_bytes(int Nout, int K) { return (size_t)((Nout + DBN - 1) / DBN) * DBN * K * 2; }

int split_f16_tiled_launch(const float* W, void* img, int Nout, int K, int ldw, hipStream_t st) {
    const size_t total = (size_t)((Nout + DBN - 1) / DBN) * (K / FBK) * 512;
    const unsigned grid = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(split_f16_tiled_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, W, static_cast<float*>(img), Nout,
                       K, ldw, total);
    return (int)hipGetLastError();
}

int fold_f16_image_launch(const float* W, const float* bias, const float* pa, const float* po, void* img, float* bias_out, int B, int Nout, int K,
                          int ldw, hipStream_t st) {
    if (K % FBK || (ldw & 3) || B < 1) return -8;
    const size_t total = (size_t)((Nout + DBN - 1) / DBN) * (K / FBK) * 512;
    const int item_blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(fold_f16_image_kernel, dim3(item_blocks + (Nout + 3) / 4, B), dim3(256), 0, st, W, bias, pa, po, static_cast<float*>(img),
                       bias_out, Nout, K, ldw, item_blocks);
    return (int)hipGetLastError();
}

int split_f16_tiled_multi_launch(const SplitJobs& jobs, hipStream_t st) {
    if (jobs.n <= 0) return 0;
    hipLaunchKernelGGL(split_f16_tiled_multi_kernel, dim3(64, jobs.n), dim3(256), 0, st, jobs);
    return (int)hipGetLastError();
}
