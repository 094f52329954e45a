// Split-bf16 linear layer whose A operand arrives as a TILED SPLIT IMAGE (GemmArgs::a_img, written by the producing GEMM's
// epilogue) and goes global -> REGISTERS, gfx950.
//
//   C[b, m, n] = residual[b, m, n] + act( sum_k A[b, m, k] * W[n, k] + bias[n] )      (+ GroupNorm partials of C)
//
// Same contract, W image, epilogue and 4 x 1 wave layout as gemm_f32_dma.hip's X3 kernel (every nn.Linear of the reference
// on the point stream; here: mlp.2 of a layer, models/set_transformer.py:166, models/mlp.py).  What differs is where the
// bytes in flight live.  The diagnostics of the LDS-DMA kernel (without its matrix instructions it still
// takes 71 % of its time; neither the barrier, nor the split, nor the fill pattern is the bound) say its K loop waits on
// the global -> LDS fill at ~4 us of latency with all of the LDS (3 stages x 24 KiB x 2 blocks) in flight.  In the 4 x 1
// layout the A rows of a wave are PRIVATE to it — only W is shared — so A does not need the LDS at all: every lane loads
// its own fragments (16 bytes of the hi plane, 16 of the lo plane per 32-row tile and K-step: a wave reads 1 KiB of
// consecutive image bytes per instruction) D K-steps ahead into a register ring, the LDS ring holds W alone (8 KiB per
// stage), and the fragment reads of A disappear from the LDS pipe.
#include "gemm_dma_common.h"

#include <cstdlib>
#include <utility>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

using dma::DBN;
using dma::DNT;
using dma::D_EPI;
using dma::dma16;
constexpr int RBK = 16;            // k per step
constexpr int R_TILE = 128 * RBK;  // floats per W block (bf16 hi | lo planes of [128][16]) and per A image block

template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

constexpr int r_main_floats(int d) { return d * R_TILE > D_EPI ? d * R_TILE : D_EPI; }

// BM = 128 / 256: 4 x 1 waves of 32 / 64 rows x 128 columns.  D (even): K-steps of lookahead = W ring slots = A register sets.
// NK = K / 16 is a template parameter and the K loop is FULLY unrolled: the A registers are written by ordinary loads, and at
// a loop header the compiler's wait-count pass no longer knows how old the load into a given register is — it drains the
// whole queue (s_waitcnt vmcnt(0)) before the first matrix instruction of every iteration; straight-line code lets it see
// that the counted waits below already cover every operand.
template <int BM, int D, int NK>
__global__ __launch_bounds__(DNT, BM == 256 ? 2 : 3) void gemm_x3_areg_kernel(GemmArgs g) {
    static_assert(D % 2 == 0 && D >= 2 && NK % D == 0, "the fragment set of a K-step is static: step % 2");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const dma::Tile T = dma::tile_of_block<BM>(g);
    const int ct = T.ct, b = T.b, m0 = T.m0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int TMW = BM / 128, TNW = 4, WMN = 4;
    constexpr int NPI = 2 + 2 * TMW;   // vector-memory instructions per wave and K-step: 2 W pieces, hi + lo per row tile
    const int r = lane & 31, h = lane >> 5;
    constexpr int nk = NK;

    const float* wimg = static_cast<const float*>(g.w_img) + (size_t)ct * nk * R_TILE + wave * 256 + lane * 4;
    // this lane's 16 bytes of the hi plane of K-step 0, per 32-row tile of the wave (lo plane: + 1024 floats; K-step kt: + kt blocks)
    const u32x4* asrc[TMW];
    {
        const int t128 = (g.rows + 127) >> 7;
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            const int ra = (wave * TMW + i) * 32 + r, rl = ra & 127;
            const int tile = min((m0 >> 7) + (ra >> 7), t128 - 1);
            asrc[i] = reinterpret_cast<const u32x4*>(g.A + ((size_t)b * t128 + tile) * nk * R_TILE + rl * 8 + ((h ^ ((rl >> 3) & 1)) << 2));
        }
    }
    auto issue_w = [&](int kt) {
        float* st = smem + (kt % D) * R_TILE;
        dma16(wimg + (size_t)kt * R_TILE, st + wave * 256);
        dma16(wimg + (size_t)kt * R_TILE + 1024, st + 1024 + wave * 256);
    };
    u32x4 pah[D][TMW], pal[D][TMW];

    // prologue: D K-steps in flight, per step W first, then A (the order of every later step)
    static_for(std::make_integer_sequence<int, D>{}, [&](auto P) {
        constexpr int p = decltype(P)::value;
        issue_w(p);
#pragma unroll
        for (int i = 0; i < TMW; ++i) {
            pah[p][i] = asrc[i][(size_t)p * (R_TILE / 4)];
            pal[p][i] = asrc[i][(size_t)p * (R_TILE / 4) + 256];
        }
    });

    f32x16 acc[TMW][TNW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    int boff[TNW][2];
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int rb = j * 32 + r, ch = h ^ ((rb >> 3) & 1);
        boff[j][0] = rb * 8 + ch * 4;
        boff[j][1] = 1024 + rb * 8 + ch * 4;
    }
    bf16x8 bhi[2][TNW], blo[2][TNW];
    auto load_b = [&](const float* st, int f) {
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            bhi[f][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(st + boff[j][0]));
            blo[f][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(st + boff[j][1]));
        }
    };

    dma::wait_vm<(D - 1) * NPI>();   // K-step 0 landed (own pieces; the barrier makes it everyone's)
    __builtin_amdgcn_s_barrier();
    load_b(smem, 0);

    {
        static_for(std::make_integer_sequence<int, NK>{}, [&](auto KT) {
            constexpr int kt = decltype(KT)::value, S = kt % D, cur = kt & 1;
            // K-steps <= kt + 1 complete: at most the younger steps' instructions outstanding
            constexpr int last_issued = kt + D - 1 < nk - 1 ? kt + D - 1 : nk - 1;
            constexpr int ahead = last_issued - (kt + 1) > 0 ? last_issued - (kt + 1) : 0;
            dma::wait_vm_lgkm0<ahead * NPI>();
#pragma unroll
            for (int j = 0; j < TNW; ++j) {
                asm volatile("" : "+v"(bhi[cur][j]));
                asm volatile("" : "+v"(blo[cur][j]));
            }
            __builtin_amdgcn_s_barrier();
            if (kt + D < nk) issue_w(kt + D);
            constexpr int kn = kt + 1 < nk ? kt + 1 : nk - 1;
            load_b(smem + (kn % D) * R_TILE, cur ^ 1);
#pragma unroll
            for (int j = 0; j < TNW; ++j)
#pragma unroll
                for (int i = 0; i < TMW; ++i) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, pah[S][i]), al = __builtin_bit_cast(bf16x8, pal[S][i]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bhi[cur][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, blo[cur][j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bhi[cur][j], acc[i][j], 0, 0, 0);
                }
            // this step's A registers are free once its matrix instructions are issued: the loads of K-step kt + D
            if (kt + D < nk) {
#pragma unroll
                for (int i = 0; i < TMW; ++i) {
                    pah[S][i] = asrc[i][(size_t)(kt + D) * (R_TILE / 4)];
                    pal[S][i] = asrc[i][(size_t)(kt + D) * (R_TILE / 4) + 256];
                }
            }
        });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // ring is dead: the epilogue reuses it
    dma::epilogue<TMW, TNW, WMN>(g, T, acc, smem, wave, lane, wave, 0);
}

template <int BM, int D, int NK>
int areg_launch_t(const GemmArgs& g, hipStream_t st) {
    const int tilesM = (g.rows + BM - 1) / BM, tilesN = (g.Nout + DBN - 1) / DBN;
    const size_t lds = (size_t)r_main_floats(D) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_areg_kernel<BM, D, NK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL((gemm_x3_areg_kernel<BM, D, NK>), dim3(g.B * tilesM * tilesN), dim3(DNT), lds, st, g);
    return (int)hipGetLastError();
}

}  // namespace

bool gemm_x3_areg_supported(const GemmArgs& g) {
    return g.a_img && g.precision == 1 && g.w_img && !g.pro_a && !g.C2 && !g.c_img && g.rows >= 128 && g.rows % 128 == 0 &&
           (g.K == 128 || g.K == 256 || g.K == 384 || g.K == 512 || g.K == 768 || g.K == 1024) && !(g.Nout & 3) &&
           !(g.ldc & 3) && !(g.ldr & 3);   // out_proj: K = d; mlp.2: K = 2 d
}

int gemm_x3_areg_launch(const GemmArgs& g, hipStream_t st) {
    if (!gemm_x3_areg_supported(g)) return -9;
    const bool tall = g.rows >= 256;
    switch (g.K) {
        case 128: return areg_launch_t<128, 4, 8>(g, st);
        case 256: return areg_launch_t<128, 4, 16>(g, st);
        case 384: {
            // out_proj at d = 384 (K = 384): 64 x 128 wave tiles measured 0.2 % of the evaluation faster than 32 x 128
            // (6.650 vs 6.665 ms, three pairs on one box); GECCO_AREG_TALL384=0 keeps the 128-row tile (A/B runs)
            static const int tall384 = [] { const char* e = getenv("GECCO_AREG_TALL384"); return e ? atoi(e) : 1; }();
            return (tall && tall384) ? areg_launch_t<256, 4, 24>(g, st) : areg_launch_t<128, 4, 24>(g, st);
        }
        case 512: return tall ? areg_launch_t<256, 4, 32>(g, st) : areg_launch_t<128, 4, 32>(g, st);
        case 768: return tall ? areg_launch_t<256, 4, 48>(g, st) : areg_launch_t<128, 4, 48>(g, st);
        case 1024: return tall ? areg_launch_t<256, 4, 64>(g, st) : areg_launch_t<128, 4, 64>(g, st);
        default: return -9;
    }
}
