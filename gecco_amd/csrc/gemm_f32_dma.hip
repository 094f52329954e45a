// Fused linear layer, LDS-DMA variant (the fast path for the N-token GEMMs): same contract as gemm_f32.hip
//
//   C[b, m, n] = residual[b, m, n] + act( sum_k (A[b, m, k] * pa[b, k] + po[b, k]) * W[n, k] + bias[n] )  (+ GroupNorm partials)
//
// but both operand tiles travel global -> LDS by `global_load_lds_dwordx4` (no staging registers, no ds_write, no
// staging VALU), four K-steps deep:
//   * a wave-instruction writes 1 KiB lane-linearly (16 rows x 64 B of a [128][16]-float tile); bank-conflict-free
//     `ds_read_b128` fragment reads come from XOR-swizzling the 16-byte chunk index with (row >> 2) & 3 — applied
//     to the per-lane SOURCE address of the DMA and to the read address (both sides, guide rule 21);
//   * the ring has 4 stages; every K-step: counted `s_waitcnt vmcnt(N)` on the wave's own pieces of the oldest
//     stage, ONE raw `s_barrier`, issue the DMA three K-steps ahead, then 32 MFMAs (v_mfma_f32_32x32x2_f32);
//     `__syncthreads()` is never used while a DMA is in flight (it would drain vmcnt to 0);
//   * the AdaGN affine moves from the staging pass to the A fragment (8 FMAs per 16 MFMAs), its per-(b, k)
//     coefficients parked in LDS once per tile, so no ordinary global load sits in the K loop.
// Rows beyond `rows` are clamped at the source (duplicates of the last row) and masked in the epilogue.
// Requires K % 16 == 0, Nout % 4 == 0, rows >= 128; everything else runs on gemm_f32.hip.
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

namespace {

constexpr int DBM = 128, DBN = 128, DBK = 16, DNT = 256;
constexpr int D_TILE = 128 * DBK;                 // floats per operand tile per stage (8 KiB)
constexpr int D_STAGE = 2 * D_TILE;               // A then B
constexpr int D_TP = 64 + 4;                      // epilogue transpose tile row stride
constexpr int D_EPI = 4 * 32 * D_TP + 2 * 2 * DBN;  // 4 half wave tiles (32 x 64) + column partials = 36 KiB
constexpr int d_main_floats(int ns) { return ns * D_STAGE > D_EPI ? ns * D_STAGE : D_EPI; }

__device__ __forceinline__ void dma16(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int DNS, bool HAS_PRO>
__global__ __launch_bounds__(DNT, DNS == 3 ? 3 : 2) void gemm_f32_dma_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* pro_lds = smem + d_main_floats(DNS);   // pa[0..K) | po[0..K)

    const int tilesM = (g.rows + DBM - 1) / DBM, tilesN = g.Nout / DBN + (g.Nout % DBN ? 1 : 0);
    const int nblk = g.B * tilesM * tilesN;
    const int v = xcd_remap(blockIdx.x, nblk);
    const int ct = v % tilesN, panel = v / tilesN;
    const int rt = panel % tilesM, b = panel / tilesM;
    const int m0 = rt * DBM, n0 = ct * DBN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    // ---- DMA source pointers: wave w moves pieces {2w, 2w+1} of A and of B; a piece = LDS rows 16p .. 16p+15
    const float* __restrict__ Ab = g.A + (size_t)b * g.rows * g.lda;
    const float* asrc[2];
    const float* bsrc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (2 * wave + q) * 16 + (lane >> 2);   // row of the tile this lane's 16 bytes land in
        const int c = (lane & 3) ^ ((row >> 2) & 3);          // global chunk stored at LDS chunk (lane & 3)
        asrc[q] = Ab + (size_t)min(m0 + row, g.rows - 1) * g.lda + c * 4;
        bsrc[q] = g.W + (size_t)min(n0 + row, g.Nout - 1) * g.ldw + c * 4;
    }
    auto issue = [&](int kt) {   // 4 DMA wave-instructions: this wave's share of K-step kt
        float* st = smem + (kt % DNS) * D_STAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            dma16(asrc[q] + kt * DBK, st + (2 * wave + q) * 256);
            dma16(bsrc[q] + kt * DBK, st + D_TILE + (2 * wave + q) * 256);
        }
    };

    const int nk = g.K / DBK;
    if (HAS_PRO) {  // park the AdaGN coefficients of this sample (ordinary loads, drained before the ring starts)
        const float* pa = g.pro_a + (size_t)b * g.K;
        const float* po = g.pro_o + (size_t)b * g.K;
        for (int i = tid; i < g.K; i += DNT) {
            pro_lds[i] = pa[i];
            pro_lds[g.K + i] = po[i];
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int p = 0; p < DNS - 1; ++p)
        if (p < nk) issue(p);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addressing: row R of a tile at R*16 floats; chunk c of that row at LDS chunk c ^ ((R >> 2) & 3)
    int aoff[2][2], boff[2][2];  // [tile][kk]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int ra = (wm * 2 + i) * 32 + r, rb = (wn * 2 + i) * 32 + r;
            aoff[i][kk] = ra * DBK + (((kk * 2 + h) ^ ((ra >> 2) & 3)) << 2);
            boff[i][kk] = D_TILE + rb * DBK + (((kk * 2 + h) ^ ((rb >> 2) & 3)) << 2);
        }

    for (int kt = 0; kt < nk; ++kt) {
        // own pieces of K-step kt have landed once at most the younger K-steps' DMAs are outstanding
        const int ahead = min(nk - 1 - kt, DNS - 2);   // K-steps issued after kt and not yet waited for
        if (DNS >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // everyone's pieces of kt are in; everyone is done reading stage (kt-1) % DNS
        if (kt + DNS - 1 < nk) issue(kt + DNS - 1);
        const float* st = smem + (kt % DNS) * D_STAGE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f32x4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x4*>(st + aoff[i][kk]);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x4*>(st + boff[j][kk]);
            if (HAS_PRO) {
                const f32x4 pa4 = *reinterpret_cast<const f32x4*>(pro_lds + kt * DBK + kk * 8 + 4 * h);
                const f32x4 po4 = *reinterpret_cast<const f32x4*>(pro_lds + g.K + kt * DBK + kk * 8 + 4 * h);
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = fa[i] * pa4 + po4;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(fa[i][e], fb[j][e], acc[i][j]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // ring is dead: the epilogue reuses it

    // ---------------------------------------------------------------- epilogue (wide: through an LDS transpose)
    const bool has_act = g.act != 0, act_norm = g.act == 1;
    const float neg_inv_2a2 = has_act ? -1.0f / (2.0f * g.alpha[0] * g.alpha[0]) : 0.f;
    float* Cb = g.C + (size_t)b * g.rows * g.ldc;
    const float* Rb = g.residual ? g.residual + (size_t)b * g.rows * g.ldr : nullptr;
    float* Tt = smem + wave * 32 * D_TP;
    float* red = smem + 4 * 32 * D_TP;
    const int lr = lane >> 4, c4 = lane & 15;   // 16 lanes per 64-float row, 4 rows per wave-instruction
    const int n = n0 + wn * 64 + c4 * 4;
    const bool nok = n < g.Nout;
    const int nc = nok ? n : 0;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // the wave's two 32-row halves, one after the other through the same LDS tile
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nn = n0 + (wn * 2 + j) * 32 + r;
            const float bias = g.bias ? g.bias[nn < g.Nout ? nn : g.Nout - 1] : 0.f;
            f32x16 val = acc[i][j];
#pragma unroll
            for (int e = 0; e < 16; ++e) val[e] += bias;
            if (has_act) {
#pragma unroll
                for (int e = 0; e < 16; ++e) val[e] = gauss_act(val[e], neg_inv_2a2, act_norm);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) Tt[mfma_row(e, h) * D_TP + j * 32 + r] = val[e];
        }
        __syncthreads();
#pragma unroll
        for (int it0 = 0; it0 < 8; it0 += 4) {
            f32x4 rres[4];
            if (Rb) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int m = min(m0 + wm * 64 + i * 32 + (it0 + c) * 4 + lr, g.rows - 1);
                    rres[c] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Rb + (size_t)m * g.ldr + nc));
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int it = it0 + c;
                const int m = m0 + wm * 64 + i * 32 + it * 4 + lr;
                f32x4 v4 = *reinterpret_cast<const f32x4*>(Tt + (it * 4 + lr) * D_TP + c4 * 4);
                if (Rb) v4 += rres[c];
                const bool ok = nok && m < g.rows;
                if (ok) __builtin_nontemporal_store(v4, reinterpret_cast<f32x4*>(Cb + (size_t)m * g.ldc + n));
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 vz = ok ? v4 : z;
                s1 += vz;
                s2 += vz * vz;
            }
        }
        __syncthreads();
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
            *reinterpret_cast<f32x4*>(red + (wm * 2 + 0) * DBN + wn * 64 + c4 * 4) = s1;
            *reinterpret_cast<f32x4*>(red + (wm * 2 + 1) * DBN + wn * 64 + c4 * 4) = s2;
        }
        __syncthreads();
        for (int c = tid; c < 2 * DBN; c += DNT) {
            const int which = c / DBN, cl = c % DBN, nn = n0 + cl;
            if (nn < g.Nout)
                g.stats[(((size_t)b * tilesM + rt) * 2 + which) * g.Nout + nn] =
                    red[(0 * 2 + which) * DBN + cl] + red[(1 * 2 + which) * DBN + cl];
        }
    }
}

}  // namespace

bool gemm_f32_dma_supported(const GemmArgs& g) {
    return g.rows >= 128 && g.K % DBK == 0 && g.K <= 1024 && !(g.Nout & 3) && !(g.ldc & 3) && !(g.ldr & 3) &&
           !(g.lda & 3) && !(g.ldw & 3);
}

template <int DNS>
static int dma_launch_t(const GemmArgs& g, hipStream_t st) {
    const int tilesM = (g.rows + DBM - 1) / DBM, tilesN = (g.Nout + DBN - 1) / DBN;
    const size_t lds = (size_t)(d_main_floats(DNS) + 2 * g.K) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_dma_kernel<DNS, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_dma_kernel<DNS, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    const dim3 grid(g.B * tilesM * tilesN);
    if (g.pro_a) hipLaunchKernelGGL((gemm_f32_dma_kernel<DNS, true>), grid, dim3(DNT), lds, st, g);
    else hipLaunchKernelGGL((gemm_f32_dma_kernel<DNS, false>), grid, dim3(DNT), lds, st, g);
    return (int)hipGetLastError();
}

int gemm_f32_dma_launch(const GemmArgs& g, hipStream_t st) {
    static int ns = 0;
    if (!ns) {
        const char* e = getenv("GECCO_GEMM_STAGES");
        ns = e ? atoi(e) : 4;
    }
    return ns == 3 ? dma_launch_t<3>(g, st) : dma_launch_t<4>(g, st);
}
