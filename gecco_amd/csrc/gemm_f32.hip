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
// LDS: two stages of [BM + BN][32 + 4] floats; the +4 pad makes the ds_read_b128 fragment reads
// conflict-free (row stride 36 dwords -> 16-B slot index 9*row mod 16 is a bijection).
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

namespace {

template <int BM, int BN, int WM, int WN, int BK, bool HAS_PRO>
__global__ __launch_bounds__(WM * WN * 64, BK == 16 ? 3 : 1) void gemm_f32_kernel(GemmArgs g) {
    constexpr int LDP = BK + 4;  // padded LDS row stride (floats): 16-B slot index (LDP/4)*row mod 16 is a bijection
    constexpr int KC = BK / 4;   // 16-byte k chunks per row of a K-step
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_IT = (BM * KC) / NT, B_IT = (BN * KC) / NT;  // float4 loads per thread per K-step
    static_assert(A_IT * NT == BM * KC && B_IT * NT == BN * KC, "tile/threads mismatch");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tilesM = (g.rows + BM - 1) / BM, tilesN = (g.Nout + BN - 1) / BN;
    const int nblk = g.B * tilesM * tilesN;
    const int v = xcd_remap(blockIdx.x, nblk);
    const int ct = v % tilesN, panel = v / tilesN;
    const int rt = panel % tilesM, b = panel / tilesM;
    const int m0 = rt * BM, n0 = ct * BN;

#ifdef GEMM_STAMPS
    unsigned long long st0 = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    const float* __restrict__ A = g.A + (size_t)b * g.rows * g.lda;
    const float* __restrict__ W = g.W;
    const float* pa = HAS_PRO ? g.pro_a + (size_t)b * g.K : nullptr;
    const float* po = HAS_PRO ? g.pro_o + (size_t)b * g.K : nullptr;

    constexpr int STAGE = (BM + BN) * LDP;  // floats per stage: A tile then B tile

    const int lrow = tid / KC, lk4 = tid % KC;  // this thread's (row, 16-byte k chunk) in a K-step
    constexpr int ROWS_PER_IT = NT / KC;

    // Branch-free staging: out-of-range rows / k are CLAMPED for the load and zeroed when the
    // registers are written to LDS, so every global load of a K-step is issued back to back and
    // the only wait sits after the MFMA block (the AdaGN affine is applied there too).
    const float* arow[A_IT];
    const float* brow[B_IT];
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + lrow + i * ROWS_PER_IT;
        arow[i] = A + (size_t)min(m, g.rows - 1) * g.lda;
        okmask |= (m < g.rows ? 1u : 0u) << i;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int n = n0 + lrow + i * ROWS_PER_IT;
        brow[i] = W + (size_t)min(n, g.Nout - 1) * g.ldw;
        okmask |= (n < g.Nout ? 1u : 0u) << (16 + i);
    }

    // Two register sets (even / odd K-tiles): while tile kt is multiplied out of LDS stage kt&1, the
    // registers holding tile kt+1 (loaded one iteration earlier, so no wait) are written to the other
    // stage and the loads of tile kt+2 are issued — all in the middle of the MFMA stream.  One barrier
    // per K-step; the two co-resident blocks of a CU no longer idle the matrix pipe in lockstep.
    struct Regs {
        f32x4 ra[A_IT], rb[B_IT], rpa, rpo;
        bool kok;
    };
    Regs R0, R1;
    auto load_global = [&](Regs& R, int kt) {
        const int k = kt * BK + lk4 * 4;
        R.kok = k < g.K;
        const int kc = R.kok ? k : 0;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) R.ra[i] = *reinterpret_cast<const f32x4*>(arow[i] + kc);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) R.rb[i] = *reinterpret_cast<const f32x4*>(brow[i] + kc);
        if (HAS_PRO) {
            R.rpa = *reinterpret_cast<const f32x4*>(pa + kc);
            R.rpo = *reinterpret_cast<const f32x4*>(po + kc);
        }
    };
    auto store_lds = [&](const Regs& R, int s) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            f32x4 x = R.ra[i];
            if (HAS_PRO) x = x * R.rpa + R.rpo;
            if (!(R.kok && ((okmask >> i) & 1u))) x = zero;
            *reinterpret_cast<f32x4*>(smem + s * STAGE + (lrow + i * ROWS_PER_IT) * LDP + lk4 * 4) = x;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            f32x4 x = R.rb[i];
            if (!(R.kok && ((okmask >> (16 + i)) & 1u))) x = zero;
            *reinterpret_cast<f32x4*>(smem + s * STAGE + (BM + lrow + i * ROWS_PER_IT) * LDP + lk4 * 4) = x;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto mma_kk = [&](int s, int kk) {
        // lane half h holds k = 8*kk + 4*h + e for e = 0..3: the same k permutation on both
        // operands, so the k-sum is complete and each MFMA consumes one register per operand.
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
    // one K-step on stage s: Rs holds tile kt+1 (to be staged), Rl receives tile kt+2
    auto k_step = [&](int s, int kt, int nk, Regs& Rs, Regs& Rl) {
        mma_kk(s, 0);
        if (kt + 1 < nk) store_lds(Rs, s ^ 1);       // its loads were issued a whole K-step ago
        if (kt + 2 < nk) load_global(Rl, kt + 2);    // a whole K-step to land before they are staged
#pragma unroll
        for (int kk = 1; kk < BK / 8; ++kk) mma_kk(s, kk);
        __syncthreads();
    };

    const int nk = (g.K + BK - 1) / BK;
    load_global(R0, 0);
    if (nk > 1) load_global(R1, 1);
    store_lds(R0, 0);
    __syncthreads();
#ifdef GEMM_STAMPS
    unsigned long long st1 = __builtin_amdgcn_s_memtime();
#endif
    for (int kt = 0; kt < nk; kt += 2) {
        k_step(0, kt, nk, R1, R0);
        if (kt + 1 < nk) k_step(1, kt + 1, nk, R0, R1);
    }

#ifdef GEMM_STAMPS
    unsigned long long st2 = __builtin_amdgcn_s_memtime();
#endif
    // ---------------------------------------------------------------- epilogue
    const float neg_inv_2a2 = g.act ? -1.0f / (2.0f * g.alpha[0] * g.alpha[0]) : 0.f;
    float* Cb = g.C + (size_t)b * g.rows * g.ldc;
    const float* Rb = g.residual ? g.residual + (size_t)b * g.rows * g.ldr : nullptr;
    const bool has_act = g.act != 0, act_norm = g.act == 1;
    constexpr int WR = TM * 32, WC = TN * 32;       // this wave's output tile
    constexpr int TP = WC + 4;                      // padded row stride of the transpose tile
    constexpr int LPR = WC / 4;                     // lanes per row (16-byte chunks)
    constexpr int RPI = 64 / LPR;                   // rows per wave-instruction
    constexpr int EP_IT = WR / RPI;
    constexpr bool WIDE_FITS = NT / 64 * WR * TP + WM * 2 * BN <= 2 * STAGE;  // transpose tiles fit the staging LDS
    static_assert(WM * 2 * BN <= 2 * STAGE, "stats scratch exceeds the staging LDS");
    float* red = smem + (WIDE_FITS ? (NT / 64) * WR * TP : 0);  // column partials, behind the transpose tiles
    const bool wide = WIDE_FITS && !(g.Nout & 3) && !(g.ldc & 3) && !(g.ldr & 3);
    if (wide) {
        // Wide epilogue: the accumulators (column on the lane, rows in registers) go through a wave-private
        // LDS tile and come back row-major, so C is written — and the residual read — as 16-byte pieces of
        // whole 128..256-byte row segments: 4x fewer memory instructions than the register layout allows.
        float* T = smem + wave * WR * TP;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + r;
            const float bias = g.bias ? g.bias[n < g.Nout ? n : g.Nout - 1] : 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float val = acc[i][j][e] + bias;
                    if (has_act) val = gauss_act(val, neg_inv_2a2, act_norm);
                    T[(i * 32 + mfma_row(e, h)) * TP + j * 32 + r] = val;
                }
            }
        }
        __syncthreads();
        const int lr = lane / LPR, c4 = lane % LPR;
        const int n = n0 + wn * WC + c4 * 4;
        const bool nok = n < g.Nout;
        const int nc = nok ? n : 0;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        f32x4 rres[EP_IT];
        if (Rb) {
#pragma unroll
            for (int it = 0; it < EP_IT; ++it) {
                const int m = min(m0 + wm * WR + it * RPI + lr, g.rows - 1);
                rres[it] = *reinterpret_cast<const f32x4*>(Rb + (size_t)m * g.ldr + nc);
            }
        }
#pragma unroll
        for (int it = 0; it < EP_IT; ++it) {
            const int m = m0 + wm * WR + it * RPI + lr;
            f32x4 v4 = *reinterpret_cast<const f32x4*>(T + (it * RPI + lr) * TP + c4 * 4);
            if (Rb) v4 += rres[it];
            const bool ok = nok && m < g.rows;
            if (ok) *reinterpret_cast<f32x4*>(Cb + (size_t)m * g.ldc + n) = v4;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 vz = ok ? v4 : z;
            s1 += vz;
            s2 += vz * vz;
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
        const int n = n0 + (wn * TN + j) * 32 + r;
        const bool nok = n < g.Nout;
        const int nc = nok ? n : g.Nout - 1;
        const float bias = g.bias ? g.bias[nc] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + (wm * TM + i) * 32 + 4 * h;  // row of register e: mb + (e&3) + 8*(e>>2)
            f32x16 val = acc[i][j];
#pragma unroll
            for (int e = 0; e < 16; ++e) val[e] += bias;
            if (has_act) {
#pragma unroll
                for (int e = 0; e < 16; ++e) val[e] = gauss_act(val[e], neg_inv_2a2, act_norm);
            }
            if (Rb) {  // 16 independent loads in flight, then one add pass
                f32x16 rr;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = min(mb + (e & 3) + 8 * (e >> 2), g.rows - 1);
                    rr[e] = Rb[(size_t)m * g.ldr + nc];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) val[e] += rr[e];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                const bool ok = nok && m < g.rows;
                if (ok) Cb[(size_t)m * g.ldc + n] = val[e];
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
            const int which = c / BN, cl = c % BN, n = n0 + cl;
            if (n < g.Nout) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) t += red[(w * 2 + which) * BN + cl];
                g.stats[(((size_t)b * tilesM + rt) * 2 + which) * g.Nout + n] = t;
            }
        }
    }
#ifdef GEMM_STAMPS
    if (tid == 0 && g_stamps) {
        unsigned long long st3 = __builtin_amdgcn_s_memtime();
        g_stamps[blockIdx.x * 4 + 0] = st0;
        g_stamps[blockIdx.x * 4 + 1] = st1;
        g_stamps[blockIdx.x * 4 + 2] = st2;
        g_stamps[blockIdx.x * 4 + 3] = st3;
    }
#endif
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
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, BK, HAS_PRO>), dim3(g.B * tilesM * tilesN), dim3(WM * WN * 64), lds, st, g);
    return (int)hipGetLastError();
}

}  // namespace

int gemm_row_tile(int rows) { return rows >= 128 ? 128 : 64; }

int gemm_f32_launch(const GemmArgs& g, hipStream_t st) {
    if (g.K % 4 || g.lda % 4 || g.ldw % 4) return -2;  // 16-byte vector loads
    const bool pro = g.pro_a != nullptr;
    static int bk = 0;
    if (!bk) {
        const char* e = getenv("GECCO_GEMM_BK");
        bk = e ? atoi(e) : 16;  // 16: 41 KB LDS + <=168 VGPR -> 3 blocks per CU (measured +2.5 % over BK=32 at 2 blocks)
    }
    if (gemm_row_tile(g.rows) == 128) {
        if (bk == 16) return pro ? launch<128, 128, 2, 2, 16, true>(g, st) : launch<128, 128, 2, 2, 16, false>(g, st);
        return pro ? launch<128, 128, 2, 2, 32, true>(g, st) : launch<128, 128, 2, 2, 32, false>(g, st);
    }
    return pro ? launch<64, 64, 2, 2, 32, true>(g, st) : launch<64, 64, 2, 2, 32, false>(g, st);
}
