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

namespace {

constexpr int BK = 32;
constexpr int LDP = BK + 4;  // padded LDS row stride (floats)

template <int BM, int BN, int WM, int WN, bool HAS_PRO>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f32_kernel(GemmArgs g) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_IT = (BM * 8) / NT, B_IT = (BN * 8) / NT;  // float4 loads per thread per K-step
    static_assert(A_IT * NT == BM * 8 && B_IT * NT == BN * 8, "tile/threads mismatch");
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tilesM = (g.rows + BM - 1) / BM, tilesN = (g.Nout + BN - 1) / BN;
    const int nblk = g.B * tilesM * tilesN;
    const int v = xcd_remap(blockIdx.x, nblk);
    const int ct = v % tilesN, panel = v / tilesN;
    const int rt = panel % tilesM, b = panel / tilesM;
    const int m0 = rt * BM, n0 = ct * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    const float* __restrict__ A = g.A + (size_t)b * g.rows * g.lda;
    const float* __restrict__ W = g.W;
    const float* pa = HAS_PRO ? g.pro_a + (size_t)b * g.K : nullptr;
    const float* po = HAS_PRO ? g.pro_o + (size_t)b * g.K : nullptr;

    constexpr int STAGE = (BM + BN) * LDP;  // floats per stage: A tile then B tile

    const int lrow = tid >> 3, lk4 = tid & 7;  // this thread's (row, 16-byte k chunk) in a K-step
    constexpr int ROWS_PER_IT = NT / 8;

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

    f32x4 ra[A_IT], rb[B_IT], rpa, rpo;
    bool kok_next = true;
    auto load_global = [&](int kt) {
        const int k = kt * BK + lk4 * 4;
        kok_next = k < g.K;
        const int kc = kok_next ? k : 0;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) ra[i] = *reinterpret_cast<const f32x4*>(arow[i] + kc);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) rb[i] = *reinterpret_cast<const f32x4*>(brow[i] + kc);
        if (HAS_PRO) {
            rpa = *reinterpret_cast<const f32x4*>(pa + kc);
            rpo = *reinterpret_cast<const f32x4*>(po + kc);
        }
    };
    auto store_lds = [&](int s) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            f32x4 x = ra[i];
            if (HAS_PRO) x = x * rpa + rpo;
            if (!(kok_next && ((okmask >> i) & 1u))) x = zero;
            *reinterpret_cast<f32x4*>(smem + s * STAGE + (lrow + i * ROWS_PER_IT) * LDP + lk4 * 4) = x;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            f32x4 x = rb[i];
            if (!(kok_next && ((okmask >> (16 + i)) & 1u))) x = zero;
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

    const int nk = (g.K + BK - 1) / BK;
    load_global(0);
    store_lds(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int s = kt & 1;
        if (kt + 1 < nk) load_global(kt + 1);
        const float* as = smem + s * STAGE + (wm * TM * 32 + r) * LDP + 4 * h;
        const float* bs = smem + s * STAGE + (BM + wn * TN * 32 + r) * LDP + 4 * h;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            // lane half h holds k = 8*kk + 4*h + e for e = 0..3: the same k permutation on both
            // operands, so the k-sum is complete and each MFMA consumes one register per operand.
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
        }
        if (kt + 1 < nk) store_lds(s ^ 1);
        __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue
    const float neg_inv_2a2 = g.act ? -1.0f / (2.0f * g.alpha[0] * g.alpha[0]) : 0.f;
    float* Cb = g.C + (size_t)b * g.rows * g.ldc;
    const float* Rb = g.residual ? g.residual + (size_t)b * g.rows * g.ldr : nullptr;
    float csum[TN], csq[TN];
    const bool has_act = g.act != 0, act_norm = g.act == 1;
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
        // combine the WM waves that share a column range, then one store per column per tile
        float* red = smem;  // reuse (all waves are past the last K-step barrier)
        __syncthreads();
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cl = (wn * TN + j) * 32 + r;
                red[(wm * 2 + 0) * BN + cl] = csum[j];
                red[(wm * 2 + 1) * BN + cl] = csq[j];
            }
        }
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
}

template <int BM, int BN, int WM, int WN, bool HAS_PRO>
int launch(const GemmArgs& g, hipStream_t st) {
    const int tilesM = (g.rows + BM - 1) / BM, tilesN = (g.Nout + BN - 1) / BN;
    const size_t lds = 2 * (BM + BN) * LDP * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_kernel<BM, BN, WM, WN, HAS_PRO>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, HAS_PRO>), dim3(g.B * tilesM * tilesN), dim3(WM * WN * 64), lds, st, g);
    return (int)hipGetLastError();
}

}  // namespace

int gemm_row_tile(int rows) { return rows >= 128 ? 128 : 64; }

int gemm_f32_launch(const GemmArgs& g, hipStream_t st) {
    if (g.K % 4 || g.lda % 4 || g.ldw % 4) return -2;  // 16-byte vector loads
    const bool pro = g.pro_a != nullptr;
    if (gemm_row_tile(g.rows) == 128) return pro ? launch<128, 128, 2, 2, true>(g, st) : launch<128, 128, 2, 2, false>(g, st);
    return pro ? launch<64, 64, 2, 2, true>(g, st) : launch<64, 64, 2, 2, false>(g, st);
}
