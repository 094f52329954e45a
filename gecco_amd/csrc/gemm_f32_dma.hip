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
//               bf16 planes (split_bf16_kernel); A is split on the fragment, after the AdaGN affine.
// Requires K % 16 == 0, Nout % 4 == 0, rows >= 128; everything else runs on gemm_f32.hip.
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int DBM = 128, DBN = 128, DBK = 16, DNT = 256;
constexpr int D_TILE = 128 * DBK;                 // floats per operand tile per stage (8 KiB)
constexpr int D_STAGE = 2 * D_TILE;               // A then B (fp32 W tile, or bf16 hi | lo planes: same 8 KiB)
constexpr int D_TP = 64 + 4;                      // epilogue transpose tile row stride
constexpr int D_EPI = 4 * 32 * D_TP + 2 * 2 * DBN;  // 4 half wave tiles (32 x 64) + column partials = 36 KiB
constexpr int d_main_floats(int ns) { return ns * D_STAGE > D_EPI ? ns * D_STAGE : D_EPI; }

__device__ __forceinline__ void dma16(const void* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

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

template <int DNS, bool HAS_PRO, bool X3>
__global__ __launch_bounds__(DNT, DNS <= 3 ? 3 : 2) void gemm_dma_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* pro_lds = smem + d_main_floats(DNS);   // pa[0..K) | po[0..K)

    const int tilesM = (g.rows + DBM - 1) / DBM, tilesN = (g.Nout + DBN - 1) / DBN;
    const int nblk = g.B * tilesM * tilesN;
    const int v = xcd_remap(blockIdx.x, nblk);
    const int ct = v % tilesN, panel = v / tilesN;
    const int rt = panel % tilesM, b = panel / tilesM;
    const int m0 = rt * DBM, n0 = ct * DBN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    // ---- DMA source pointers (bytes advance by one K-step = 16 k per iteration)
    // A tile [128][16] fp32: 8 pieces of 16 rows x 64 B; wave w moves pieces {2w, 2w+1}; chunk ^= (row >> 2) & 3.
    // fp32 W tile: the same.  bf16 planes [128][16] bf16: 4 pieces of 32 rows x 32 B each; wave w moves piece w of
    // the hi plane and of the lo plane; chunk ^= (row >> 3) & 1.
    const float* __restrict__ Ab = g.A + (size_t)b * g.rows * g.lda;
    const float* asrc[2];
    const void* bsrc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (2 * wave + q) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        asrc[q] = Ab + (size_t)min(m0 + row, g.rows - 1) * g.lda + c * 4;
        if (!X3) bsrc[q] = g.W + (size_t)min(n0 + row, g.Nout - 1) * g.ldw + c * 4;
    }
    if (X3) {
        const int row = wave * 32 + (lane >> 1);
        const int c = (lane & 1) ^ ((row >> 3) & 1);
        const size_t off = (size_t)min(n0 + row, g.Nout - 1) * g.ldw + c * 8;
        bsrc[0] = g.w_hi + off;
        bsrc[1] = g.w_lo + off;
    }
    auto issue = [&](int kt) {   // 4 DMA wave-instructions: this wave's share of K-step kt
        float* st = smem + (kt % DNS) * D_STAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q) dma16(asrc[q] + kt * DBK, st + (2 * wave + q) * 256);
        if (X3) {
            dma16(static_cast<const unsigned short*>(bsrc[0]) + kt * DBK, st + D_TILE + wave * 256);
            dma16(static_cast<const unsigned short*>(bsrc[1]) + kt * DBK, st + D_TILE + 1024 + wave * 256);
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                dma16(static_cast<const float*>(bsrc[q]) + kt * DBK, st + D_TILE + (2 * wave + q) * 256);
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

    // fragment addressing (float offsets inside a stage).  Every lane reads two 16-byte chunks of an fp32 row per
    // K-step: chunks {h, 2 + h} in fp32 mode (k = 8*kk + 4*h + e for kk = 0, 1), chunks {2h, 2h + 1} in bf16x3 mode
    // (k = 8*h + j); row R's global chunk c sits at LDS chunk c ^ ((R >> 2) & 3).
    int aoff[2][2], boff[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = (wm * 2 + i) * 32 + r, rb = (wn * 2 + i) * 32 + r;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = X3 ? (2 * h + q) : (2 * q + h);
            aoff[i][q] = ra * DBK + ((c ^ ((ra >> 2) & 3)) << 2);
            boff[i][q] = D_TILE + rb * DBK + ((c ^ ((rb >> 2) & 3)) << 2);
        }
        if (X3) {  // bf16 plane row = 32 B = 8 floats; this lane's 8 k-values = chunk h ^ ((row >> 3) & 1)
            const int ch = h ^ ((rb >> 3) & 1);
            boff[i][0] = D_TILE + rb * 8 + ch * 4;          // hi plane
            boff[i][1] = D_TILE + 1024 + rb * 8 + ch * 4;   // lo plane
        }
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
        if (X3) {
            // one 32x32x16 chunk per K-step: lane half h holds k = 8h .. 8h+7 of both operands
            bf16x8 ahi[2], alo[2], bhi[2], blo[2];
            f32x4 pa0, pa1, po0, po1;
            if (HAS_PRO) {
                pa0 = *reinterpret_cast<const f32x4*>(pro_lds + kt * DBK + 8 * h);
                pa1 = *reinterpret_cast<const f32x4*>(pro_lds + kt * DBK + 8 * h + 4);
                po0 = *reinterpret_cast<const f32x4*>(pro_lds + g.K + kt * DBK + 8 * h);
                po1 = *reinterpret_cast<const f32x4*>(pro_lds + g.K + kt * DBK + 8 * h + 4);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 x0 = *reinterpret_cast<const f32x4*>(st + aoff[i][0]);
                f32x4 x1 = *reinterpret_cast<const f32x4*>(st + aoff[i][1]);
                if (HAS_PRO) {
                    x0 = x0 * pa0 + po0;
                    x1 = x1 * pa1 + po1;
                }
                split8(x0, x1, ahi[i], alo[i]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bhi[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(st + boff[j][0]));
                blo[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(st + boff[j][1]));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[i], bhi[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], blo[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], bhi[j], acc[i][j], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // lane half h holds k = 8*kk + 4*h + e: the same k permutation on both operands
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

// W (fp32, n elements) -> hi / lo bf16 planes with the same indexing
__global__ void split_bf16_kernel(const float* __restrict__ W, unsigned short* __restrict__ hi,
                                  unsigned short* __restrict__ lo, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = W[i];
        const unsigned u = __float_as_uint(x);
        hi[i] = (unsigned short)(u >> 16);
        const __bf16 l = (__bf16)(x - __uint_as_float(u & 0xFFFF0000u));
        lo[i] = __builtin_bit_cast(unsigned short, l);
    }
}

template <int DNS, bool X3>
int dma_launch_t(const GemmArgs& g, hipStream_t st) {
    const int tilesM = (g.rows + DBM - 1) / DBM, tilesN = (g.Nout + DBN - 1) / DBN;
    const size_t lds = (size_t)(d_main_floats(DNS) + 2 * g.K) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<DNS, true, X3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<DNS, false, X3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    const dim3 grid(g.B * tilesM * tilesN);
    if (g.pro_a) hipLaunchKernelGGL((gemm_dma_kernel<DNS, true, X3>), grid, dim3(DNT), lds, st, g);
    else hipLaunchKernelGGL((gemm_dma_kernel<DNS, false, X3>), grid, dim3(DNT), lds, st, g);
    return (int)hipGetLastError();
}

}  // namespace

bool gemm_f32_dma_supported(const GemmArgs& g) {
    return g.rows >= 128 && g.K % DBK == 0 && g.K <= 1024 && !(g.Nout & 3) && !(g.ldc & 3) && !(g.ldr & 3) &&
           !(g.lda & 3) && !(g.ldw & 7);
}

int gemm_f32_dma_launch(const GemmArgs& g, hipStream_t st) {
    static int ns3 = -1;
    if (ns3 < 0) {
        const char* e = getenv("GECCO_GEMM_STAGES");   // 3 stages = 51 KB LDS = three blocks per CU (measured best)
        ns3 = (e && atoi(e) == 4) ? 0 : 1;
    }
    if (g.precision == 1 && g.w_hi && g.w_lo) return ns3 ? dma_launch_t<3, true>(g, st) : dma_launch_t<4, true>(g, st);
    return ns3 ? dma_launch_t<3, false>(g, st) : dma_launch_t<4, false>(g, st);
}

int split_bf16_launch(const float* W, unsigned short* hi, unsigned short* lo, size_t n, hipStream_t st) {
    const unsigned grid = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(split_bf16_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, W, hi, lo, n);
    return (int)hipGetLastError();
}
