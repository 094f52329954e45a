// A-stationary fused linear of the fp16 mode: AdaGN apply + fp16 rounding + every column tile of the layer's wide
// projections in ONE pass over x.
//
//   C16[b, m, n] = fp16( act( sum_k fp16(x[b, m, k] * pa[b, k] + po[b, k]) * fp16(W[n, k]) + bias[n] ) )
//
// for the two launches that read an AdaGN-normalised residual stream and write fp16 intermediates: kv_proj | q_proj
// (Nout = 3d, two output tensors) and mlp.0 + GaussianActivation (Nout = 2d).  Same arithmetic, bit for bit, as
// gecco_affine_cast_f16 followed by the fp16 GEMM (gemm_f16_dma.hip) — but:
//   * a block (4 waves) owns 128 rows of x and every wave keeps ITS 32 rows in REGISTERS: it reads the fp32 rows once
//     from HBM, applies the affine, rounds to fp16 and holds them as the MFMA A fragments of all K / 32 K-steps
//     (K = 384: 96 VGPRs).  No fp16 copy of AdaGN(x) in HBM, no cast pass, no A tile re-fetched by every column tile
//     (the tile fill of the streaming kernel was its bound: tools/probe/dma_rate.hip, ~43 B/clk/CU into LDS) and no A
//     fragment re-read from LDS by every MFMA (an LDS-resident panel with 32 x 64 wave tiles needed 1.5 KiB of LDS
//     reads per MFMA — the LDS port, not the matrix pipe, set its pace);
//   * the block walks ALL column tiles with 32 x 128 wave tiles; W streams from its pre-tiled fp16 image through a
//     6-slot LDS-DMA ring (two 1 KiB pieces per wave per K-step), its fragments double-buffered in registers; the
//     K-steps of consecutive column tiles form ONE pipeline, so the next tile's stages are in flight while a tile's
//     epilogue stores drain.  48 KiB of LDS and <= 256 VGPRs: two independent blocks per CU, whose build, MFMA and
//     store phases interleave;
//   * the epilogue needs no LDS: bias and activation in registers, neighbouring lanes exchange one value (DPP) so
//     every lane stores a packed pair of fp16 (rows stay 64-byte contiguous per 32-column tile).
// vmcnt bookkeeping: a wave's epilogue issues exactly 32 stores (full tiles only: rows % 128 == 0, Nout % 128 == 0),
// which sit in the in-order vmcnt queue behind the DMA pieces already in flight; the K-steps after an epilogue whose
// awaited piece was issued before it wait with that count added instead of draining the stores.
#include "gemm_dma_common.h"

#include <stdlib.h>

namespace {

using dma::dma16;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SBK = 32;            // k per stage
constexpr int S_TILE = 2048;       // floats of one [128][32] fp16 W tile (8 KiB)
constexpr int S_NS = 6;            // ring slots
constexpr int S_NT = 256;          // threads
constexpr int S_PW = 2;            // 1 KiB pieces of a W stage per wave
constexpr int S_STORES = 32;       // store instructions a wave's epilogue issues (32 x 128 wave tile)

// s_waitcnt vmcnt(PW * n + STORES * e) lgkmcnt(0): n in [0, S_NS - 2] stages and e in [0, 1] epilogues may stay queued
__device__ __forceinline__ void wait_stages(int n, bool stores) {
    if (stores) {
        switch (n) {
            case 0: dma::wait_vm_lgkm0<S_STORES + 0 * S_PW>(); break;
            case 1: dma::wait_vm_lgkm0<S_STORES + 1 * S_PW>(); break;
            case 2: dma::wait_vm_lgkm0<S_STORES + 2 * S_PW>(); break;
            case 3: dma::wait_vm_lgkm0<S_STORES + 3 * S_PW>(); break;
            default: dma::wait_vm_lgkm0<S_STORES + 4 * S_PW>(); break;
        }
    } else {
        switch (n) {
            case 0: dma::wait_vm_lgkm0<0 * S_PW>(); break;
            case 1: dma::wait_vm_lgkm0<1 * S_PW>(); break;
            case 2: dma::wait_vm_lgkm0<2 * S_PW>(); break;
            case 3: dma::wait_vm_lgkm0<3 * S_PW>(); break;
            default: dma::wait_vm_lgkm0<4 * S_PW>(); break;
        }
    }
}

// value of the neighbouring lane (lane ^ 1): one DPP move (quad_perm [1, 0, 3, 2])
__device__ __forceinline__ float swap_pair(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

#ifdef ASTAT_STAMPS   // diagnostic build (tools/probe): per-block s_memtime stamps
__device__ unsigned long long g_astat_stamps[4096 * 8];
#define ASTAMP(i)                                                                                          \
    do {                                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_astat_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define ASTAMP(i)
#endif

// NK = K / 32 K-steps, compile-time: the A fragments are indexed statically (they are registers)
template <int NK>
__global__ __launch_bounds__(S_NT, 2) void gemm_f16_astat_kernel(GemmArgs g) {
    static_assert(NK % 2 == 0 && NK >= S_NS - 1, "even K-step count, and at most one epilogue per ring depth");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ASTAMP(0);
    float* ring = smem;                                // [S_NS][S_TILE]
    float* bias_lds = ring + S_NS * S_TILE;            // [Nout] (zeros where a segment has no bias)
    float* pro_lds = bias_lds + g.Nout;                // pa[0..K) | po[0..K)
    constexpr int K = NK * SBK;

    const int tilesM = g.rows / 128, tilesN = g.Nout / 128;
    const int b = blockIdx.x / tilesM, rt = blockIdx.x % tilesM, m0 = rt * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // rows 32 wave .. 32 wave + 31
    const int r = lane & 31, h = lane >> 5;

    // ---- W image pieces: wave w moves pieces 2w, 2w+1 (1 KiB each) of every 8 KiB stage
    const float* img = static_cast<const float*>(g.w_img) + (S_PW * wave) * 256 + lane * 4;
    const int nsteps = tilesN * NK;                    // flattened step s = column tile * NK + kt
    auto issue = [&](int s) {
#pragma unroll
        for (int q = 0; q < S_PW; ++q)
            dma16(img + (size_t)s * S_TILE + q * 256, ring + (s % S_NS) * S_TILE + (S_PW * wave + q) * 256);
    };
#pragma unroll
    for (int p = 0; p < S_NS; ++p)
        if (p < nsteps) issue(p);

    for (int n = tid; n < g.Nout; n += S_NT) {
        const bool seg2 = g.C2 != nullptr && n >= g.n_split;
        const float* bp = seg2 ? g.bias2 : g.bias;
        bias_lds[n] = bp ? bp[seg2 ? n - g.n_split : n] : 0.f;
    }
    const bool has_pro = g.pro_a != nullptr;
    if (has_pro) {
        const float* pa = g.pro_a + (size_t)b * K;
        const float* po = g.pro_o + (size_t)b * K;
        for (int i = tid; i < K; i += S_NT) {
            pro_lds[i] = pa[i];
            pro_lds[K + i] = po[i];
        }
    }
    __syncthreads();   // bias and AdaGN coefficients are in LDS (also drains vmcnt: the first S_NS stages have landed)

    // ---- the A operand of this wave: fp16(x * pa + po) for its 32 rows, as the fragments of all NK K-steps.
    // lane (r, h) holds row r, k = 32 kt + 16 h + 8 c .. + 7 in fragment [kt][c]: 64 consecutive bytes of x per K-step
    f16x8 fa[NK][2];
    {
        const float* xr = g.A + ((size_t)b * g.rows + m0 + wave * 32 + r) * g.lda + 16 * h;
        constexpr int KB = 4;   // K-steps per batch: 16 independent 16-byte loads in flight per lane
#pragma unroll
        for (int k0 = 0; k0 < NK; k0 += KB) {
            f32x4 x[KB][4];
#pragma unroll
            for (int u = 0; u < KB; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (k0 + u < NK) x[u][q] = GECCO_NT_LOAD(reinterpret_cast<const f32x4*>(xr + (k0 + u) * SBK + 4 * q));
#pragma unroll
            for (int u = 0; u < KB; ++u) {
                if (k0 + u >= NK) break;
                if (has_pro) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(pro_lds + (k0 + u) * SBK + 16 * h + 4 * q);
                        const f32x4 o4 = *reinterpret_cast<const f32x4*>(pro_lds + K + (k0 + u) * SBK + 16 * h + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[u][q][e] = __builtin_fmaf(x[u][q][e], a4[e], o4[e]);
                    }
                }
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        fa[k0 + u][c][e] = (_Float16)x[u][2 * c][e];
                        fa[k0 + u][c][4 + e] = (_Float16)x[u][2 * c + 1][e];
                    }
            }
        }
    }
    ASTAMP(1);
    const bool has_act = g.act != 0, act_norm = g.act == 1;
    const float neg_inv_2a2 = has_act ? -1.0f / (2.0f * g.alpha[0] * g.alpha[0]) : 0.f;

    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // W fragment addressing (float offsets inside a stage): rows of the four 32-column tiles
    int boff[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rb = j * 32 + r;
#pragma unroll
        for (int c = 0; c < 2; ++c) boff[j][c] = rb * 16 + (((2 * h + c) ^ ((rb >> 2) & 3)) << 2);
    }
    f16x8 fb[2][4][2];
    auto load_b = [&](int slot, int f) {
        const float* st = ring + slot * S_TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c)
                fb[f][j][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(st + boff[j][c]));
    };

    // ---- epilogue of one column tile, from registers: bias, activation, fp16, packed-pair stores
    auto epilogue = [&](int ct) {
        const int n0 = ct * 128;
        const bool seg2 = g.C2 != nullptr && n0 >= g.n_split;
        _Float16* Cb = reinterpret_cast<_Float16*>(seg2 ? g.C2 : g.C);
        const int ldc = seg2 ? g.ldc2 : g.ldc;
        const int nseg0 = seg2 ? n0 - g.n_split : n0;
        _Float16* base = Cb + ((size_t)b * g.rows + m0 + wave * 32) * ldc + nseg0;
        const bool odd = lane & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float bias = bias_lds[n0 + j * 32 + r];
            f32x16 v = acc[j];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                v[e] += bias;
                if (has_act) v[e] = gauss_act(v[e], neg_inv_2a2, act_norm);
                acc[j][e] = 0.f;
            }
#pragma unroll
            for (int e0 = 0; e0 < 16; e0 += 2) {
                // even lane keeps register e0 (its column and the right neighbour's), odd lane register e0 + 1
                const float recv = swap_pair(odd ? v[e0] : v[e0 + 1]);
                f16x2 pk;
                pk[0] = (_Float16)(odd ? recv : v[e0]);
                pk[1] = (_Float16)(odd ? v[e0 + 1] : recv);
                const int row = mfma_row(odd ? e0 + 1 : e0, h);
                *reinterpret_cast<f16x2*>(base + (size_t)row * ldc + j * 32 + (r & ~1)) = pk;
            }
        }
    };

    // the plain loads of the build have returned (their values were consumed above); from here on the wave's vmcnt
    // queue holds only DMA pieces and epilogue stores
    load_b(0, 0);
    int s = 0, slot_next = 1 % S_NS;   // flattened step; ring slot of stage s + 1
    int since_epi = 1000;              // K-steps since this wave's last epilogue stores
    for (int ct = 0; ct < tilesN; ++ct) {
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            const int cur = kt & 1;
            // own pieces of stage s + 1 landed; younger stages may stay in flight, and so may the stores of an epilogue
            // issued after the awaited piece (which was issued at the top of K-step s + 1 - S_NS)
            const int rem = nsteps - 1 - s;
            wait_stages(rem >= S_NS - 1 ? S_NS - 2 : (rem >= 1 ? rem - 1 : 0), since_epi <= S_NS - 1);
            // this step's W fragments were read during the previous one and the wait above covered them: an empty asm
            // "redefines" the registers so the compiler's wait-count pass does not park its own lgkmcnt(0) in front of
            // the first MFMA, behind the NEXT step's reads issued below
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < 2; ++c) asm volatile("" : "+v"(fb[cur][j][c]));
            __builtin_amdgcn_s_barrier();
            if (rem >= S_NS) issue(s + S_NS);
            if (rem >= 1) load_b(slot_next, cur ^ 1);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[kt][c], fb[cur][j][c], acc[j], 0, 0, 0);
            ++since_epi;
            if (kt == NK - 1) {
                epilogue(ct);
                since_epi = 1;
            }
            ++s;
            slot_next = slot_next + 1 == S_NS ? 0 : slot_next + 1;
        }
    }
    ASTAMP(2);
}

template <int NK>
int astat_launch_t(const GemmArgs& g, hipStream_t st) {
    const size_t lds = ((size_t)S_NS * S_TILE + g.Nout + 2 * g.K) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_astat_kernel<NK>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL((gemm_f16_astat_kernel<NK>), dim3(g.B * (g.rows / 128)), dim3(S_NT), lds, st, g);
    return (int)hipGetLastError();
}

}  // namespace

bool gemm_f16_astat_supported(const GemmArgs& g) {
    const int nk = g.K / SBK;
    return g.c_f16 && !g.a_f16 && !g.residual && !g.stats && g.w_img && g.rows >= 128 && !(g.rows % 128) &&
           !(g.Nout % 128) && !(g.K % SBK) && (nk == 6 || nk == 8 || nk == 12) && !(g.lda & 3) && !(g.ldc & 1) &&
           (!g.C2 || (!(g.n_split % 128) && !(g.ldc2 & 1) && g.n_split > 0 && g.n_split < g.Nout)) &&
           ((g.pro_a == nullptr) == (g.pro_o == nullptr));
}

int gemm_f16_astat_launch(const GemmArgs& g, hipStream_t st) {
    switch (g.K / SBK) {
        case 6: return astat_launch_t<6>(g, st);     // d = 192
        case 8: return astat_launch_t<8>(g, st);     // d = 256
        case 12: return astat_launch_t<12>(g, st);   // d = 384 (the shipped configs)
        default: return -9;
    }
}
