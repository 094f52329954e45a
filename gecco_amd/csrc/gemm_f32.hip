// Fused linear layer on fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   C[b, m, n] = residual[b, m, n] + act( sum_k (A[b, m, k] * pa[b, k] + po[b, k]) * W[n, k] + bias[n] )
//
// and, in the same epilogue, per-(sample, row-tile, column) sum / sum-of-squares of the stored
// values: the GroupNorm statistics of the NEXT AdaGN are emitted by the producer, so the
// (B, N, d) stream is never re-read (or transposed) just to normalise it.
//
// Replaces, per call site, the reference's nn.Linear (+ the AdaGN apply in front of it,
// models/normalization.py:36-44, + GaussianActivation models/activation.py:17-24, + the residual
// add models/set_transformer.py:164,166) — see SURVEY.md section 2.3 rows K1, K2, K5, K6.
//
// Layout: A (B, rows, K) row-major, W (Nout, K) row-major (= nn.Linear.weight), C (B, rows, Nout).
// Tiles never straddle two samples, so the AdaGN coefficients pa/po are per-tile constants in k.
// LDS: two stages of [BM + BN][BK + 4] floats; the +4 pad makes the ds_read_b128 fragment reads
// conflict-free (row stride BK+4 dwords -> 16-B slot index (BK/4+1)*row mod 16 is a bijection).
//
// Persistent: the grid is one block per resident slot (CUs x blocks/CU); each block walks a strided list of
// tiles from its XCD's contiguous chunk, and issues the first global loads of its NEXT tile before the epilogue
// of the current one.  A tile boundary therefore costs neither a block retire/dispatch nor a cold prologue,
// and the epilogue's stores drain behind the next tile's main loop.
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

namespace {

template <int BM, int BN, int WM, int WN, int BK, bool HAS_PRO>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_f32_kernel(GemmArgs g) {
    constexpr int LDP = BK + 4;  // padded LDS row stride (floats)
    constexpr int KC = BK / 4;   // 16-byte k chunks per row of a K-step
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_IT = (BM * KC) / NT, B_IT = (BN * KC) / NT;  // float4 loads per thread per K-step
    static_assert(A_IT * NT == BM * KC && B_IT * NT == BN * KC, "tile/threads mismatch");
    constexpr int STAGE = (BM + BN) * LDP;  // floats per stage: A tile then B tile
    constexpr int ROWS_PER_IT = NT / KC;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tilesM = (g.rows + BM - 1) / BM, tilesN = (g.Nout + BN - 1) / BN;
    const int nblk = g.B * tilesM * tilesN;
    // XCD x (= blockIdx & 7: blocks are dealt round-robin over the 8 XCDs) owns the contiguous chunk
    // [base, base + cnt) of the virtual tile list, so tiles sharing an A panel meet in one L2.
    const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int cnt = q8 + (x < r8 ? 1 : 0), base = x * q8 + min(x, r8);
    int w = slot;
    if (w >= cnt) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int lrow = tid / KC, lk4 = tid % KC;  // this thread's (row, 16-byte k chunk) in a K-step

    // Branch-free staging: out-of-range rows / k are CLAMPED for the load and zeroed when the registers are
    // written to LDS, so the global loads of a K-step issue back to back (the AdaGN affine is applied there too).
    struct Tile {
        int b, rt, m0, n0;
        const float* arow[A_IT];
        const float* brow[B_IT];
        const float *pa, *po;
        unsigned okmask;
    };
    auto setup = [&](int v, Tile& T) {
        const int ct = v % tilesN, panel = v / tilesN;
        T.rt = panel % tilesM;
        T.b = panel / tilesM;
        T.m0 = T.rt * BM;
        T.n0 = ct * BN;
        const float* A = g.A + (size_t)T.b * g.rows * g.lda;
        T.pa = HAS_PRO ? g.pro_a + (size_t)T.b * g.K : nullptr;
        T.po = HAS_PRO ? g.pro_o + (size_t)T.b * g.K : nullptr;
        T.okmask = 0;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int m = T.m0 + lrow + i * ROWS_PER_IT;
            T.arow[i] = A + (size_t)min(m, g.rows - 1) * g.lda;
            T.okmask |= (m < g.rows ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int n = T.n0 + lrow + i * ROWS_PER_IT;
            T.brow[i] = g.W + (size_t)min(n, g.Nout - 1) * g.ldw;
            T.okmask |= (n < g.Nout ? 1u : 0u) << (16 + i);
        }
    };

    // Two register sets (even / odd K-tiles): while K-tile kt is multiplied out of LDS stage kt&1, the registers
    // holding K-tile kt+1 (loaded one step earlier, so no wait) are written to the other stage and the loads of
    // K-tile kt+2 are issued — all inside the MFMA stream, one barrier per K-step.
    struct Regs {
        f32x4 ra[A_IT], rb[B_IT], rpa, rpo;
        bool kok;
    };
    auto load_global = [&](const Tile& T, Regs& R, int kt) {
        const int k = kt * BK + lk4 * 4;
        R.kok = k < g.K;
        const int kc = R.kok ? k : 0;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) R.ra[i] = *reinterpret_cast<const f32x4*>(T.arow[i] + kc);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) R.rb[i] = *reinterpret_cast<const f32x4*>(T.brow[i] + kc);
        if (HAS_PRO) {
            R.rpa = *reinterpret_cast<const f32x4*>(T.pa + kc);
            R.rpo = *reinterpret_cast<const f32x4*>(T.po + kc);
        }
    };
    auto store_lds = [&](const Tile& T, const Regs& R, int s) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            f32x4 xv = R.ra[i];
            if (HAS_PRO) xv = xv * R.rpa + R.rpo;
            if (!(R.kok && ((T.okmask >> i) & 1u))) xv = zero;
            *reinterpret_cast<f32x4*>(smem + s * STAGE + (lrow + i * ROWS_PER_IT) * LDP + lk4 * 4) = xv;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            f32x4 xv = R.rb[i];
            if (!(R.kok && ((T.okmask >> (16 + i)) & 1u))) xv = zero;
            *reinterpret_cast<f32x4*>(smem + s * STAGE + (BM + lrow + i * ROWS_PER_IT) * LDP + lk4 * 4) = xv;
        }
    };

    f32x16 acc[TM][TN];
    const int nk = (g.K + BK - 1) / BK;
    auto mma_kk = [&](int s, int kk) {
        // lane half h holds k = 8*kk + 4*h + e for e = 0..3: the same k permutation on both operands, so the
        // k-sum is complete and each MFMA consumes one register per operand.
        const float* as = smem + s * STAGE + (wm * TM * 32 + r) * LDP + 4 * h;
        const float* bs = smem + s * STAGE + (BM + wn * TN * 32 + r) * LDP + 4 * h;
        f32x4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * LDP + kk * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * LDP + kk * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(fa[i][e], fb[j][e], acc[i][j]);
    };
    // One K-step on stage s: Rs holds K-tile kt+1 (to be staged), Rl receives K-tile kt+2.  (Cutting the staging
    // into pieces placed after every group of 4 MFMAs was measured and is 4 % SLOWER than this block form.)
    auto k_step = [&](const Tile& T, int s, int kt, Regs& Rs, Regs& Rl) {
        mma_kk(s, 0);
#ifndef GEMM_DIAG_NOSTAGE   // (diagnostic builds in tools/probe only)
        if (kt + 1 < nk) store_lds(T, Rs, s ^ 1);       // its loads were issued a whole K-step ago
        if (kt + 2 < nk) load_global(T, Rl, kt + 2);    // a whole K-step to land before they are staged
#endif
#pragma unroll
        for (int kk = 1; kk < BK / 8; ++kk) mma_kk(s, kk);
#ifndef GEMM_DIAG_NOBARRIER
        __syncthreads();
#endif
    };

    const float neg_inv_2a2 = act_is_gauss(g.act) ? -1.0f / (2.0f * g.alpha[0] * g.alpha[0]) : 0.f;
    const bool has_act = g.act != 0;
    const int act_mode = g.act;
    constexpr int WR = TM * 32, WC = TN * 32;       // this wave's output tile
    constexpr int TP = WC + 4;                      // padded row stride of the transpose tile
    constexpr int LPR = WC / 4;                     // lanes per row (16-byte chunks)
    constexpr int RPI = 64 / LPR;                   // rows per wave-instruction
    constexpr int EP_IT = WR / RPI;
    constexpr bool WIDE_FITS = NT / 64 * WR * TP + WM * 2 * BN <= 2 * STAGE;  // transpose tiles fit the staging LDS
    static_assert(WM * 2 * BN <= 2 * STAGE, "stats scratch exceeds the staging LDS");
    float* red = smem + (WIDE_FITS ? (NT / 64) * WR * TP : 0);  // column partials, behind the transpose tiles
    const bool wide = WIDE_FITS && !(g.Nout & 3) && !(g.ldc & 3) && !(g.ldr & 3);

    Tile T;
    Regs R0, R1;
    setup(base + w, T);
    load_global(T, R0, 0);
    if (nk > 1) load_global(T, R1, 1);
    for (;;) {
        store_lds(T, R0, 0);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int kt = 0; kt < nk; kt += 2) {
            k_step(T, 0, kt, R1, R0);
            if (kt + 1 < nk) k_step(T, 1, kt + 1, R0, R1);
        }
        // both register sets are free again: start the NEXT tile's first K-tile before this tile's epilogue
        // (the second one follows the epilogue: holding both sets across it costs too many registers)
        const int wnext = w + nslot;
        const bool more = wnext < cnt;
        Tile Tn;
        if (more) {
            setup(base + wnext, Tn);
            load_global(Tn, R0, 0);
        }
        float* Cb = g.C + (size_t)T.b * g.rows * g.ldc;
        const float* Rb = g.residual ? g.residual + (size_t)T.b * g.rows * g.ldr : nullptr;
        if (wide) {
            // Wide epilogue: the accumulators (column on the lane, rows in registers) go through a wave-private
            // LDS tile and come back row-major, so C is written — and the residual read — as 16-byte pieces of
            // whole 128..256-byte row segments: 4x fewer memory instructions than the register layout allows.
            float* Tt = smem + wave * WR * TP;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = T.n0 + (wn * TN + j) * 32 + r;
                const float bias = g.bias ? g.bias[n < g.Nout ? n : g.Nout - 1] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float val = acc[i][j][e] + bias;
                        if (has_act) val = act_apply(val, neg_inv_2a2, act_mode);
                        Tt[(i * 32 + mfma_row(e, h)) * TP + j * 32 + r] = val;
                    }
                }
            }
            __syncthreads();
            const int lr = lane / LPR, c4 = lane % LPR;
            const int n = T.n0 + wn * WC + c4 * 4;
            const bool nok = n < g.Nout;
            const int nc = nok ? n : 0;
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
            constexpr int CH = EP_IT < 4 ? EP_IT : 4;  // residual rows in flight per chunk (register budget)
#pragma unroll
            for (int it0 = 0; it0 < EP_IT; it0 += CH) {
                f32x4 rres[CH];
                if (Rb) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        const int m = min(T.m0 + wm * WR + (it0 + c) * RPI + lr, g.rows - 1);
                        rres[c] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Rb + (size_t)m * g.ldr + nc));
                    }
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const int it = it0 + c;
                    const int m = T.m0 + wm * WR + it * RPI + lr;
                    f32x4 v4 = *reinterpret_cast<const f32x4*>(Tt + (it * RPI + lr) * TP + c4 * 4);
                    if (Rb) v4 += rres[c];
                    const bool ok = nok && m < g.rows;
                    if (ok) __builtin_nontemporal_store(v4, reinterpret_cast<f32x4*>(Cb + (size_t)m * g.ldc + n));
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 vz = ok ? v4 : z;
                    s1 += vz;
                    s2 += vz * vz;
                }
            }
            if (g.stats) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int o = LPR; o < 64; o <<= 1) {
                        s1[q] += __shfl_xor(s1[q], o, 64);
                        s2[q] += __shfl_xor(s2[q], o, 64);
                    }
                }
                if (lane < LPR) {
                    *reinterpret_cast<f32x4*>(red + (wm * 2 + 0) * BN + wn * WC + c4 * 4) = s1;
                    *reinterpret_cast<f32x4*>(red + (wm * 2 + 1) * BN + wn * WC + c4 * 4) = s2;
                }
            }
        } else {
        float csum[TN], csq[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = T.n0 + (wn * TN + j) * 32 + r;
            const bool nok = n < g.Nout;
            const int nc = nok ? n : g.Nout - 1;
            const float bias = g.bias ? g.bias[nc] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = T.m0 + (wm * TM + i) * 32 + 4 * h;  // row of register e: mb + (e&3) + 8*(e>>2)
                f32x16 val = acc[i][j];
#pragma unroll
                for (int e = 0; e < 16; ++e) val[e] += bias;
                if (has_act) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) val[e] = act_apply(val[e], neg_inv_2a2, act_mode);
                }
                if (Rb) {  // 16 independent loads in flight, then one add pass
                    f32x16 rr;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = min(mb + (e & 3) + 8 * (e >> 2), g.rows - 1);
                        rr[e] = __builtin_nontemporal_load(Rb + (size_t)m * g.ldr + nc);
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) val[e] += rr[e];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mb + (e & 3) + 8 * (e >> 2);
                    const bool ok = nok && m < g.rows;
                    if (ok) __builtin_nontemporal_store(val[e], Cb + (size_t)m * g.ldc + n);
                    const float vz = ok ? val[e] : 0.f;
                    s1 += vz;
                    s2 += vz * vz;
                }
            }
            csum[j] = s1 + xor32(s1);
            csq[j] = s2 + xor32(s2);
        }
            if (g.stats) {
                __syncthreads();
                if (h == 0) {
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int cl = (wn * TN + j) * 32 + r;
                        red[(wm * 2 + 0) * BN + cl] = csum[j];
                        red[(wm * 2 + 1) * BN + cl] = csq[j];
                    }
                }
            }
        }
        if (g.stats) {
            // combine the WM waves that share a column range, then one store per column per tile
            __syncthreads();
            for (int c = tid; c < 2 * BN; c += NT) {
                const int which = c / BN, cl = c % BN, n = T.n0 + cl;
                if (n < g.Nout) {
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) t += red[(w * 2 + which) * BN + cl];
                    g.stats[(((size_t)T.b * tilesM + T.rt) * 2 + which) * g.Nout + n] = t;
                }
            }
        }
        if (!more) break;
        T = Tn;
        w = wnext;
        if (nk > 1) load_global(T, R1, 1);
        __syncthreads();  // the epilogue's LDS scratch is dead before stage 0 is overwritten
    }
}

template <int BM, int BN, int WM, int WN, int BK, bool HAS_PRO>
int launch(const GemmArgs& g, hipStream_t st) {
    constexpr int LDP = BK + 4;
    const int tilesM = (g.rows + BM - 1) / BM, tilesN = (g.Nout + BN - 1) / BN;
    const size_t lds = 2 * (BM + BN) * LDP * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<BM, BN, WM, WN, BK, HAS_PRO>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    // persistent grid: one block per resident slot, a multiple of 8 (one share per XCD)
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int per_cu = (160 * 1024) / (int)lds < 2 ? (160 * 1024) / (int)lds : 2;
        slots = cus * (per_cu > 0 ? per_cu : 1);
    }
    const int nblk = g.B * tilesM * tilesN;
    int grid = nblk < slots ? nblk : slots;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, BK, HAS_PRO>), dim3(grid), dim3(WM * WN * 64), lds, st, g);
    return (int)hipGetLastError();
}

}  // namespace

int gemm_row_tile(int rows) { return rows >= 128 ? 128 : 64; }

int gemm_f32_launch(const GemmArgs& g, hipStream_t st) {
    if (g.K % 4 || g.lda % 4 || g.ldw % 4) return -2;  // 16-byte vector loads
    if (g.precision == 2 && g.w_img && gemm_f16_dma_supported(g)) return gemm_f16_dma_launch(g, st);
    {
        static int use_dma = -1;  // GECCO_GEMM_DMA=0 forces the register-staged kernel (A/B runs)
        if (use_dma < 0) {
            const char* e = getenv("GECCO_GEMM_DMA");
            use_dma = e ? atoi(e) : 1;
        }
        if (use_dma && gemm_f32_dma_supported(g)) return gemm_f32_dma_launch(g, st);
    }
    if (!g.W) return -9;   // a ready weight image without the matrix itself: only the LDS-DMA kernel reads images
    if (g.C2) return -8;   // the two-segment form exists on the LDS-DMA kernel only (callers check dma_supported)
    const bool pro = g.pro_a != nullptr;
    // 128x128x16: 41 KB LDS and <= 256 VGPR -> two persistent blocks per CU (BK = 32 would spill once the next
    // tile's prefetch registers are live across the epilogue)
    if (gemm_row_tile(g.rows) == 128) return pro ? launch<128, 128, 2, 2, 16, true>(g, st) : launch<128, 128, 2, 2, 16, false>(g, st);
    return pro ? launch<64, 64, 2, 2, 32, true>(g, st) : launch<64, 64, 2, 2, 32, false>(g, st);
}
