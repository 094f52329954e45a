// Linear layer in "fp16 + fp8 cross terms" arithmetic (h8) whose A operand arrives as an h8 ACTIVATION IMAGE written by its
// producer's epilogue and goes global -> REGISTERS, gfx950:
//
//   C[b, m, n] = residual[b, m, n] + sum_k A[b, m, k] * W[n, k] + bias[n]          (+ GroupNorm partials of C)
//
// the second linear of the point MLP and the unpool attention's out_proj in the mixed mode (reference:
// models/set_transformer.py:112,164-166; models/mlp.py:5-39).  Same contract, block tile (128 x 128, 4 x 1 waves of 32 x 128)
// and epilogue as gemm_x3_areg.hip, whose split-bf16 product (3 matrix instructions per 16 k, 4 bytes per operand element) it
// replaces by
//
//       A W = Ah Wh (v_mfma_f32_32x32x16_f16) + fp8(Ah 2^-3) fp8(2^16 Wl) + fp8(2^11 Al) fp8(2^5 W)   (v_mfma_scale_f32_32x32x64_f8f6f4; h8_scales.h)
//
// — 2 matrix-pipe units per product and 3 bytes per element, at split-bf16 accuracy (tools/experiments/fp16_site_sensitivity.py,
// scheme h8: 6.0e-5 on F_x against 6.7e-5).  A = Ah + Al with Ah = fp16(A), Al kept as fp8(2^11 (A - Ah)).
//
// Activation image (GemmArgs::a_img == 2; written by gemm_h8_astat.hip's epilogue and by the unpool attention kernel): per
// (sample, 128-row tile, 64-k group) one 24 KiB block:
//   hi, 16 KiB: [32-row tile rt][sub][c][lane] x 16 bytes = the 8 fp16 of row 32 rt + (lane & 31), k = 32 sub + 16 (lane >> 5) + 8 c + 0 .. 7
//               — the MFMA A fragment of k-step (sub, c), 1 KiB of consecutive bytes per wave-instruction;
//   lo,  8 KiB: [rt][t][lane] x 16 bytes = the 16 fp8 (2^11 lo) of the same row, k = 32 t + 16 (lane >> 5) + 0 .. 15 — one half of the scaled
//               MFMA's 32-byte A operand.
// W image: h8_image_item<128> (gemm_h8_astat.hip): per (128-column tile, 64-k group) a 16 KiB H stage (two [128][32] fp16
// sub-tiles) and a 16 KiB L stage (fp8 Wl | fp8 W), streamed through a ring of NS stages by global_load_lds; the A fragments
// of group g + D are loaded when group g's registers are free.  The K loop is fully unrolled: every wait is a compile-time
// count (young_at below).
#include "gemm_dma_common.h"
#include "h8_scales.h"

#include <stdlib.h>

#include <utility>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

using dma::DBN;
using dma::DNT;
using dma::D_EPI;
using dma::dma16;

constexpr int G_STAGE = 4096;   // floats per 16 KiB W stage: two [128][16 floats] sub-tiles
constexpr int A_BLK = 6144;     // floats per (128-row tile, 64-k group) block of the activation image
constexpr int PW4 = 4;          // 1 KiB W pieces per wave and stage (4 waves; 8 waves: 2)
constexpr int AL = 6;           // A loads per lane and group: 4 hi + 2 lo

template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

// Diagnostic builds (tools/probe/h8areg_probe.hip): -DH8_STAMPS per-block s_memtime stamps; -DH8_DIAG_NOMFMA / _NOEPI
#ifdef H8_STAMPS
__device__ unsigned long long g_h8a_stamps[4096 * 4];
#define ASTAMP(i)                                                                                               \
    do {                                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_h8a_stamps[blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define ASTAMP(i)
#endif
#ifdef H8_DIAG_NOMFMA
__device__ __forceinline__ f32x16 keep16(f16x8 a, f16x8 b, f32x16 c) {
    asm volatile("" ::"v"(a), "v"(b));
    return c;
}
__device__ __forceinline__ f32x16 keep8(i32x8 a, i32x8 b, f32x16 c) {
    asm volatile("" ::"v"(a), "v"(b));
    return c;
}
#define A_MFMA16(a, b, c) keep16(a, b, c)
#define A_MFMA8(a, b, c, sa, sb) keep8(a, b, c)
#else
#define A_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define A_MFMA8(a, b, c, sa, sb) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb)
#endif

// Vector-memory operations issued after the youngest one step t needs, up to (not including) step t's own issues — what
// s_waitcnt vmcnt may leave in flight.  Steps t = 2 g (H stage of group g) and 2 g + 1 (L stage).  Issue order: prologue
// P_0 .. P_{NS-2}, A_0 .. A_{D-1}; step u: P_{u+NS-1} (while it exists), and at the end of an L step A_{g+D} (while it exists).
// Step t needs P_{t+1} (the barrier then makes every wave's pieces of the next stage visible) and, at an H step, A_{t/2}.
constexpr int young_at(int t, int NGK, int NS, int D, int PW = 4) {
    const int nst = 2 * NGK;
    int total = 0, need_end = 0;
    auto mark_p = [&](int s) { if (s == t + 1) need_end = total; };
    auto mark_a = [&](int gq) { if ((t & 1) == 0 && gq == t / 2) need_end = total; };
    for (int s = 0; s < NS - 1 && s < nst; ++s) { total += PW; mark_p(s); }
    for (int gq = 0; gq < D && gq < NGK; ++gq) { total += AL; mark_a(gq); }
    for (int u = 0; u < t; ++u) {
        if (u + NS - 1 < nst) { total += PW; mark_p(u + NS - 1); }
        if ((u & 1) && (u - 1) / 2 + D < NGK) { total += AL; mark_a((u - 1) / 2 + D); }
    }
    return total - need_end;
}

// NW = 8 (DIRECT epilogue only): a 256-row block of 8 waves, one per CU — a W stage brought into the LDS serves 256 rows instead of
// 128 (per 128 rows and 64-k group 16 + 24 KiB enter the CU instead of 32 + 24: the K loop is bound by exactly that,
// profiles/r03k_h8areg_phases.txt)
template <int NGK, int NS, int D, bool DIRECT, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void gemm_h8_areg_kernel(GemmArgs g) {
    static_assert(NS >= 3 && D >= 1 && D <= NGK, "ring / lookahead");
    static_assert(NW == 4 || (NW == 8 && DIRECT), "the LDS-transpose epilogue is written for four waves");
    constexpr int NST = 2 * NGK, PW = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    ASTAMP(0);
    const dma::Tile T = dma::tile_of_block<32 * NW>(g);
    const int ct = T.ct, b = T.b, m0 = T.m0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;

    const float* wimg = static_cast<const float*>(g.w_img) + (size_t)ct * NST * G_STAGE + wave * 256 + lane * 4;
    auto issue_w = [&](int s) {
        float* st = smem + (s % NS) * G_STAGE + wave * 256;
#pragma unroll
        for (int p = 0; p < PW; ++p) dma16(wimg + (size_t)s * G_STAGE + p * NW * 256, st + p * NW * 256);   // piece p NW + wave
    };
    // this wave's 32 rows = 32-row tile `wave` of the 128-row tile: its lane's 16 bytes of fragment (sub, c) of group 0
    const int t128 = g.rows >> 7;
    const size_t ablk = ((size_t)b * t128 + (m0 >> 7) + (wave >> 2)) * NGK * A_BLK;
    const u32x4* asrc = reinterpret_cast<const u32x4*>(g.A + ablk + (wave & 3) * 1024 + lane * 4);
    const u32x4* asrc_lo = reinterpret_cast<const u32x4*>(g.A + ablk + 4096 + (wave & 3) * 512 + lane * 4);
    u32x4 ahi[D][4], alo[D][2];
    auto load_a = [&](int gq, int set) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ahi[set][i] = asrc[(size_t)gq * (A_BLK / 4) + i * 64];
#pragma unroll
        for (int i = 0; i < 2; ++i) alo[set][i] = asrc_lo[(size_t)gq * (A_BLK / 4) + i * 64];
    };

    static_for(std::make_integer_sequence<int, (NS - 1 < NST ? NS - 1 : NST)>{}, [&](auto P) { issue_w(decltype(P)::value); });
    static_for(std::make_integer_sequence<int, D>{}, [&](auto P) { load_a(decltype(P)::value, decltype(P)::value); });

    f32x16 acc[1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;

    // W fragments: rows r, 32 + r, .. of a sub-tile share the swizzle — two lane offsets (c = 0, 1), column block j 2 KiB further
    int boffc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) boffc[c] = r * 16 + (((2 * h + c) ^ ((r >> 2) & 3)) << 2);
    i32x8 fbA[4], fbB[4];
    auto load_f = [&](const float* sub, i32x8(&f)[4]) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(sub + boffc[c] + j * 512);
#pragma unroll
                for (int e = 0; e < 4; ++e) f[j][4 * c + e] = (int)v[e];
            }
    };

    dma::wait_vm<young_at(0, NGK, NS, D, PW)>();
    ASTAMP(1);   // stage 0 (own pieces) and the younger ones' allowance: see young_at
    // step 0 needs P_1 and A_0; the fragments of stage 0 itself need P_0 of every wave: older than both
    __builtin_amdgcn_s_barrier();
    load_f(smem, fbA);
    float one = H8_AH_DIV;   // fp8(Ah / 8): h8_scales.h
    asm volatile("" : "+s"(one));

    static_for(std::make_integer_sequence<int, NST>{}, [&](auto TT) {
        constexpr int t = decltype(TT)::value, gq = t >> 1, set = gq % D;
        constexpr bool lst = (t & 1) != 0;
        constexpr int yv = young_at(t, NGK, NS, D, PW);
        dma::wait_vm_lgkm0<(yv > 63 ? 63 : yv)>();   // 6-bit field; the last step needs nothing: everything may stay in flight
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(fbA[j]));
        __builtin_amdgcn_s_barrier();
        // stage t + NS - 1 reuses the slot of stage t - 1, whose last fragment reads (this step's first set) are complete
        if constexpr (t + NS - 1 < NST) issue_w(t + NS - 1);
        const float* cur = smem + (t % NS) * G_STAGE;
        load_f(cur + 2048, fbB);
        if constexpr (!lst) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const i32x4 wc = c == 0 ? __builtin_shufflevector(fbA[j], fbA[j], 0, 1, 2, 3) : __builtin_shufflevector(fbA[j], fbA[j], 4, 5, 6, 7);
                    acc[0][j] = A_MFMA16(__builtin_bit_cast(f16x8, ahi[set][c]), __builtin_bit_cast(f16x8, wc), acc[0][j]);
                }
        } else {
            // Ah Wl: fp8(Ah) of the group, bytes in the image's k order (16 t + 8 c + e)
            i32x8 a8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f16x8 v = __builtin_bit_cast(f16x8, ahi[set][i]);
                s16x2 p0 = {0, 0}, p1 = {0, 0};
                p0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p0, f16x2{v[0], v[1]}, one, false);
                p0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p0, f16x2{v[2], v[3]}, one, true);
                p1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p1, f16x2{v[4], v[5]}, one, false);
                p1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p1, f16x2{v[6], v[7]}, one, true);
                a8[2 * i] = __builtin_bit_cast(int, p0);
                a8[2 * i + 1] = __builtin_bit_cast(int, p1);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[0][j] = A_MFMA8(a8, fbA[j], acc[0][j], H8_SC_AH, H8_SC_WL);
        }
        if constexpr (t + 1 < NST) load_f(smem + ((t + 1) % NS) * G_STAGE, fbA);
        if constexpr (!lst) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const i32x4 wc = c == 0 ? __builtin_shufflevector(fbB[j], fbB[j], 0, 1, 2, 3) : __builtin_shufflevector(fbB[j], fbB[j], 4, 5, 6, 7);
                    acc[0][j] = A_MFMA16(__builtin_bit_cast(f16x8, ahi[set][2 + c]), __builtin_bit_cast(f16x8, wc), acc[0][j]);
                }
        } else {
            // Al W
            i32x8 al8;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                al8[e] = (int)alo[set][0][e];
                al8[4 + e] = (int)alo[set][1][e];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[0][j] = A_MFMA8(al8, fbB[j], acc[0][j], H8_SC_AL, H8_SC_W8);
            // this group's A registers are free once its matrix instructions are issued: the loads of group gq + D
            if constexpr (gq + D < NGK) load_a(gq + D, set);
        }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // ring is dead: the epilogue reuses it
    ASTAMP(2);
#ifdef H8_DIAG_NOEPI
    if (acc[0][0][0] == 123.456f)
#endif
    if constexpr (DIRECT) {
        // No-transpose epilogue (unpool_outproj_h8.hip): acc[0][j][4 q + e] = (A W^T)[row 8 q + 4 h + e][n0 + 32 j + r] — a register
        // holds 32 consecutive columns of one row across a lane half: the residual comes in as 64 four-byte loads (2 x 128
        // contiguous bytes per instruction) into the registers the fragments and the A sets have left, ALL in flight at once (one
        // exposure of the HBM latency per block instead of one per 32 x 64 sub-tile), the result leaves as 64 stores of the same
        // shape, and a lane's column sums are 16 adds and one lane-half exchange.  Same arithmetic as dma::epilogue:
        // (A W^T + bias) + residual.  Whole 128-column tiles only (Nout % 128 == 0).
        const int n0 = T.n0;
        const float* Rb = g.residual ? g.residual + ((size_t)b * g.rows + m0 + wave * 32) * g.ldr + n0 : nullptr;
        float* Cb = g.C + ((size_t)b * g.rows + m0 + wave * 32) * g.ldc + n0;
        const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Rb ? Rb : Cb), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(Cb, 0, 0x7fffffff, 0x00020000);
        const unsigned vr = (unsigned)((4 * h * g.ldr + r) * 4), vc = (unsigned)((4 * h * g.ldc + r) * 4);
        float res[4][16];
        if (Rb) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int qe = 0; qe < 16; ++qe) {
                    const int row = 8 * (qe >> 2) + (qe & 3);
                    res[j][qe] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, vr, (unsigned)((row * g.ldr + 32 * j) * 4), 0));
                }
        }
        float* red = smem;   // [NW waves][2][128]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float bias = g.bias ? g.bias[n0 + 32 * j + r] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int qe = 0; qe < 16; ++qe) {
                const int row = 8 * (qe >> 2) + (qe & 3);
                float v = acc[0][j][qe] + bias;
                if (Rb) v += res[j][qe];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), crsrc, vc, (unsigned)((row * g.ldc + 32 * j) * 4), 0);
                s1 += v;
                s2 = __builtin_fmaf(v, v, s2);
            }
            if (g.stats) {
                const auto a1 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s1), false, false);
                const auto a2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, s2), __builtin_bit_cast(unsigned, s2), false, false);
                if (lane < 32) {
                    red[(wave * 2 + 0) * 128 + 32 * j + r] = __uint_as_float(a1[0]) + __uint_as_float(a1[1]);
                    red[(wave * 2 + 1) * 128 + 32 * j + r] = __uint_as_float(a2[0]) + __uint_as_float(a2[1]);
                }
            }
        }
        if (g.stats) {
            __syncthreads();
            // per 128-row tile (the consumers count partials per 128 rows): waves 4 half .. 4 half + 3
            const int half = tid >> 8, which = (tid >> 7) & 1, cl = tid & 127;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += red[((4 * half + w) * 2 + which) * 128 + cl];
            g.stats[(((size_t)b * (g.rows >> 7) + (m0 >> 7) + half) * 2 + which) * g.Nout + n0 + cl] = t;
        }
    } else {
        dma::epilogue<1, 4, 4>(g, T, acc, smem, wave, lane, wave, 0);
    }
    ASTAMP(3);
}

template <int NGK>
int h8_areg_launch8_t(const GemmArgs& g, hipStream_t st) {   // 256-row blocks of 8 waves, no-transpose epilogue
    constexpr int NS = 6, D = NGK >= 3 ? 3 : NGK;
    const int tilesM = g.rows / 256, tilesN = g.Nout / DBN;
    constexpr size_t lds = (size_t)NS * G_STAGE * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h8_areg_kernel<NGK, NS, D, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((gemm_h8_areg_kernel<NGK, NS, D, true, 8>), dim3(g.B * tilesM * tilesN), dim3(512), lds, st, g);
    return (int)hipGetLastError();
}

template <int NGK>
int h8_areg_launch_t(const GemmArgs& g, hipStream_t st) {
    static const int w8_env = [] { const char* e = getenv("GECCO_H8AREG_W8"); return e ? atoi(e) : 0; }();
    if (w8_env && g.rows % 256 == 0 && g.Nout % 128 == 0) return h8_areg_launch8_t<NGK>(g, st);
    constexpr int NS = 4, D = NGK >= 3 ? 3 : NGK;
    const int tilesM = g.rows / 128, tilesN = (g.Nout + DBN - 1) / DBN;
    constexpr size_t ring = (size_t)NS * G_STAGE, epi = (size_t)D_EPI;
    const size_t lds = (ring > epi ? ring : epi) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h8_areg_kernel<NGK, NS, D, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_h8_areg_kernel<NGK, NS, D, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    static const int direct_env = [] { const char* e = getenv("GECCO_H8AREG_DIRECT"); return e ? atoi(e) : 0; }();   // measured: epilogue 17.2 K -> 12.7 K ticks per block, prologue + K loop +6.6 K: 197 -> 193 us at K = 768, 125 -> 128 us at K = 384 (profiles/r04d): off
    // whole 128-column tiles: the no-transpose epilogue; a ragged last tile keeps the masked LDS-transpose one
    if (direct_env && g.Nout % 128 == 0) hipLaunchKernelGGL((gemm_h8_areg_kernel<NGK, NS, D, true>), dim3(g.B * tilesM * tilesN), dim3(DNT), lds, st, g);
    else hipLaunchKernelGGL((gemm_h8_areg_kernel<NGK, NS, D, false>), dim3(g.B * tilesM * tilesN), dim3(DNT), lds, st, g);
    return (int)hipGetLastError();
}

}  // namespace

size_t h8_w128_image_bytes(int Nout, int K) { return (size_t)((Nout + 127) / 128) * 128 * K * 4; }

bool gemm_h8_areg_supported(const GemmArgs& g) {
    return g.a_img == 2 && g.w_img && !g.pro_a && !g.C2 && !g.c_img && !g.a_f16 && !g.c_f16 && !g.mul_u && !g.pre_out && g.rows >= 128 &&
           g.rows % 128 == 0 && (g.K == 128 || g.K == 256 || g.K == 384 || g.K == 512 || g.K == 768 || g.K == 1024) && !(g.Nout & 3) &&
           !(g.ldc & 3) && !(g.ldr & 3);
}

int gemm_h8_areg_launch(const GemmArgs& g, hipStream_t st) {
    if (!gemm_h8_areg_supported(g)) return -9;
    switch (g.K) {
        case 128: return h8_areg_launch_t<2>(g, st);
        case 256: return h8_areg_launch_t<4>(g, st);
        case 384: return h8_areg_launch_t<6>(g, st);
        case 512: return h8_areg_launch_t<8>(g, st);
        case 768: return h8_areg_launch_t<12>(g, st);
        case 1024: return h8_areg_launch_t<16>(g, st);
        default: return -9;
    }
}
