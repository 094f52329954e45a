// The point MLP of a layer in ONE launch, mixed mode ("h8" arithmetic), gfx950:
//
//   x += mlp.2( act( mlp.0( x * pa + po ) ) )             (+ per-(sample, 128-row tile, column) sum / sum^2 of the new x)
//
// reference x + mlp(mlp_norm(x)): models/set_transformer.py:164-166, models/mlp.py:5-39, models/activation.py:17-24,
// models/normalization.py:36-44.  Before, two launches: gemm_h8_astat_kernel (AdaGN apply, mlp.0, activation) wrote the 2d-wide
// hidden layer as an h8 activation image (fp16 hi + fp8 lo: 302 MB at C2) and gemm_h8_areg_kernel read it back (mlp.2, residual,
// statistics).  Here the hidden layer never leaves the CU.
//
// A block owns 128 rows from x to x.  Neither product fits a 256-register wave together with the other — the stationary operand
// y = AdaGN(x) of mlp.0 is 144 registers, the 32 x d accumulator of mlp.2 192 — so the block's 8 waves have two ROLES:
//   * waves 0-3 ("P1"), rows 32 w ..: gemm_h8_astat.hip's kernel — y in registers for the whole block (fp16 fragments + fp8 lo),
//     one 64-column tile of the hidden layer at a time, W1 as the MFMA's row operand so that the accumulator holds one point per
//     lane; bias, activation and the hi / lo split happen on the accumulator registers, and the tile leaves as the fp16 / fp8
//     operand fragments of mlp.2, 16 bytes per lane, into an LDS hand-off buffer (double-buffered);
//   * waves 4-7 ("P2"), rows 32 (w - 4) .. — wave w and w + 4 share a SIMD —: the 32 x d accumulator of mlp.2 (standard
//     orientation: a register = 32 consecutive columns of one row), the hidden tile as the stationary-for-one-tile A operand read
//     lane-for-lane from the hand-off buffer, W2's 64-k slice of the tile as 2 d / 64 stages of 64 columns.
// Both consume 8 KiB weight stages from ONE LDS ring filled by all 8 waves (buffer_load ... lds) in consumption order.  One block
// barrier per INTERVAL of two stages (16 KiB, 512 matrix-pipe cycles per SIMD).  A hidden tile costs each role 12 stages, and
// P1 additionally ~2 K cycles of vector work (activation, split) — so a period of 12 intervals is scheduled statically so that
// every interval carries two stages and the vector work sits beside the OTHER role's matrix work:
//       intervals 0-3:   P1: activation / split of its finished tile, a quarter per interval;   P2: two stages each
//       intervals 4-7:   P1: one stage (the next tile),                                         P2: one stage
//       intervals 8-11:  P1: two stages each,                                                   P2: -
// P2 runs a tile and a third behind P1.  The weight stream (mlpf8_image_kernel) holds the stages in exactly that order.
// After the last tile P2's accumulator gets bias + residual (LDS-DMA'd row tiles) and is stored with the no-transpose epilogue
// of unpool_outproj_h8.hip; column sums of the new x for the next GroupNorm.
#include "gemm_dma_common.h"
#include "h8_scales.h"

#include <stdlib.h>

#include <utility>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int F_STAGE = 2048;          // floats per 8 KiB weight stage (two 4 KiB sub-tiles)
constexpr int F_IVAL = 2 * F_STAGE;    // floats per interval (two stages)
constexpr int F_PW = 2;                // 1 KiB pieces per wave and interval (8 waves x 2 = 16 KiB)
constexpr int F_STG = 1536;            // floats of a P1 wave's staging tile during the y build: [32][64] fp16 + [32][64] fp8
constexpr int F_HID = 2048;            // floats of a wave's hand-off buffer: hi 4 KiB | fp8(hi / 8) 2 KiB | fp8(2^11 lo) 2 KiB

__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

// Diagnostic builds (tools/probe/mlpf8_probe.hip): -DMF8_STAMPS per-block s_memtime stamps
#ifdef MF8_STAMPS
__device__ unsigned long long g_mf8_stamps[2048 * 8];
#define FSTAMP(i)                                                                                                  \
    do {                                                                                                           \
        if ((threadIdx.x & 255) == 0 && blockIdx.x < 2048) g_mf8_stamps[blockIdx.x * 8 + (threadIdx.x >> 8) * 4 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define FSTAMP(i)
#endif

#define F_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define F_MFMA8(a, b, c, sa, sb) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb)

__device__ __forceinline__ float clamp448(float v) { return __builtin_amdgcn_fmed3f(v, -448.f, 448.f); }

__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    int pk = 0;
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, pk, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, pk, true);
    return (unsigned)pk;
}

__device__ __forceinline__ float sum_halves(float v) {   // v + (the other lane half's v): v_permlane32_swap, no LDS round trip
    const auto a = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

// ---------------------------------------------------------------------------------------------------------------------
// The interval schedule, shared by the image builder and the kernel.  NST = 2 NG stages per hidden tile and role; a period =
// NST intervals in three thirds (T = NST / 3); NCT = 2 NG hidden tiles; periods p = 0 .. NCT + 1:
//   i in [0, T):     P2: stages T + 2 i, T + 2 i + 1 of tile p - 2
//   i in [T, 2 T):   slot 0: P1 stage i - T of tile p;  slot 1: P2 stage i - T of tile p - 1
//   i in [2 T, 3 T): P1: stages T + 2 (i - 2 T), + 1 of tile p
// EVERY period runs all 3 T intervals and both roles always execute their stages: a stage of a tile that does not exist (P1: tiles
// NCT, NCT + 1; P2: tiles -2, -1 and NCT) is a slot of ZERO weights in the stream — it adds nothing — so that the loop over the
// periods has no predicate around its matrix instructions (predicated, or peeled into range copies, the compiler moved the 192
// accumulator registers of P2 through scratch).  Price: (NCT + 2) / NCT of the stages, 17 % at NCT = 12.
__host__ __device__ constexpr bool f_active(int NCT, int T, int p, int i) { return p >= 0 && p <= NCT + 1 && i >= 0 && i < 3 * T; }
__host__ __device__ constexpr int f_intervals(int NCT, int T) { return (NCT + 2) * 3 * T; }
// what stage sits in slot `sl` of interval (p, i): role (0 P1, 1 P2, -1 padding), tile, stage
__host__ __device__ inline void f_slot(int NCT, int T, int p, int i, int sl, int& role, int& tile, int& st) {
    role = -1; tile = 0; st = 0;
    if (i < T) {
        if (p >= 2) { role = 1; tile = p - 2; st = T + 2 * i + sl; }
    } else if (i < 2 * T) {
        if (sl == 0) { if (p < NCT) { role = 0; tile = p; st = i - T; } }
        else if (p >= 1 && p <= NCT) { role = 1; tile = p - 1; st = i - T; }
    } else if (p < NCT) {
        role = 0; tile = p; st = T + 2 * (i - 2 * T) + sl;
    }
}

// 16-byte chunk `pc` (physical) of row n of sub-tile `sub` of an h8 weight stage (gemm_h8_astat.hip: h8_image_item): kind 0 = fp16
// H stage, 1 = fp8 L stage (sub 0: 2^16 (W - fp16(W)), sub 1: 2^5 W); PERM: the k order of an accumulator-fed operand
template <bool PERM>
__device__ __forceinline__ u32x4 f_chunk(const float* __restrict__ Wrow, int g, int kind, int sub, int n, int pc) {
    const int q = pc ^ ((n >> 2) & 3);
    u32x4 out;
    if (kind == 0) {
        const float* src = Wrow + 64 * g + 32 * sub + (PERM ? 16 * (q & 1) + 4 * (q >> 1) : 8 * q);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(src), w1 = *reinterpret_cast<const f32x4*>(src + (PERM ? 8 : 4));
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (_Float16)w0[e];
            v[4 + e] = (_Float16)w1[e];
        }
        out = __builtin_bit_cast(u32x4, v);
    } else {
        const int h = q >> 1, t = q & 1;
        const float* src = Wrow + 64 * g + 32 * t + (PERM ? 4 * h : 16 * h);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(src + (PERM ? 16 * (c >> 1) + 8 * (c & 1) : 4 * c));
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = clamp448(sub == 0 ? (w[e] - (float)(_Float16)w[e]) * H8_WL_SCALE : w[e] * H8_W8_SCALE);
            out[c] = pack_fp8x4(v[0], v[1], v[2], v[3]);
        }
    }
    return out;
}

// W0 (Wd, C) and W2 (C, Wd) -> the stream of 8 KiB stages in the kernel's consumption order (padding slots are zeros)
__global__ void mlpf8_image_kernel(const float* __restrict__ W0, const float* __restrict__ W2, float* __restrict__ img, int C, int Wd) {
    const int NG = C / 64, NCT = Wd / 64, NST = 2 * NG, T = NST / 3;
    const int p = (int)blockIdx.y / (3 * T), i = (int)blockIdx.y % (3 * T);   // blockIdx.y = interval, 1024 chunks of 16 bytes each
    for (int item = threadIdx.x + blockIdx.x * blockDim.x; item < 1024; item += blockDim.x * gridDim.x) {
        const int pc = item & 3, nr = (item >> 2) & 63, sub = (item >> 8) & 1, sl = item >> 9;
        int role, tile, st;
        f_slot(NCT, T, p, i, sl, role, tile, st);
        u32x4 out;
        if (role < 0) out = u32x4{0u, 0u, 0u, 0u};   // padding: zero weights (read as fp16 or as fp8; the operand beside them is the zero hand-off buffer)
        else if (role == 0) out = f_chunk<false>(W0 + (size_t)(64 * tile + nr) * C, st >> 1, st & 1, sub, nr, pc);
        else out = f_chunk<true>(W2 + (size_t)(64 * (st >> 1) + nr) * Wd, tile, st & 1, sub, nr, pc);
        *reinterpret_cast<u32x4*>(img + (size_t)blockIdx.y * F_IVAL + sl * F_STAGE + sub * 1024 + nr * 16 + pc * 4) = out;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
template <int NG, int NS, int ACT>
__global__ __launch_bounds__(512, 1) void mlp_fused_h8_kernel(MlpH8Args g) {
    constexpr int C = 64 * NG, WD = 2 * C, NCT = WD / 64, NST = 2 * NG, T = NST / 3, NI = f_intervals(NCT, T);
    static_assert(NST % 3 == 0 && T == 4, "the static schedule: four activation chunks = four 16-column k-steps of a hidden tile");
    static_assert(NS >= 4, "lookahead NS - 1 >= 3 intervals");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ring = smem;                               // [NS][F_IVAL]
    float* hid = ring + NS * F_IVAL;                  // [2][4][F_HID] hand-off buffers; first the y build's staging tiles
    float* b0_lds = hid + 2 * 4 * F_HID;              // [WD]
    float* b2_lds = b0_lds + WD;                      // [C]
    float* pro_lds = b2_lds + C;                      // pa[0 .. C) | po[0 .. C)

    const int tilesM = g.rows / 128;
    const int bid = g.rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int b = bid / tilesM, rt = bid % tilesM, m0 = rt * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool p1 = wave < 4;                         // role
    const int wr = wave & 3;                          // row group: rows m0 + 32 wr ..
    const int r = lane & 31, h = lane >> 5;

    FSTAMP(0);
    for (int n = tid; n < WD; n += 512) b0_lds[n] = g.b0 ? g.b0[n] : 0.f;
    for (int n = tid; n < C; n += 512) b2_lds[n] = g.b2 ? g.b2[n] : 0.f;
    {
        const float* pa = g.pro_a + (size_t)b * C;
        const float* po = g.pro_o + (size_t)b * C;
        for (int i = tid; i < C; i += 512) {
            pro_lds[i] = pa[i];
            pro_lds[C + i] = po[i];
        }
    }
    // the hand-off buffers start as zeros: P2's stages of the tiles "-2" and "-1" (periods 0 and 1) multiply them with zero weights —
    // NaN bit patterns must not be there (parity 0 holds the y build's staging tiles first: the P1 waves clear it after the build)
    for (int i = tid; i < 4 * F_HID / 4; i += 512) reinterpret_cast<u32x4*>(hid + 4 * F_HID)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();   // before the first DMA: a block barrier drains the vector-memory queue

    // ---- the weight stream: F_PW 1 KiB pieces per wave and interval; past its end the last interval is fetched again
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.w_img), 0, 0x7fffffff, 0x00020000);
    const unsigned wvoff = (unsigned)(wave * F_PW * 256 + lane * 4) * 4u;
    const unsigned soff_last = (unsigned)(NI - 1) * (F_IVAL * 4u);
    unsigned soff = 0;
    int islot = 0;
    auto issue = [&]() {
#pragma unroll
        for (int p = 0; p < F_PW; ++p) dma16_buf(wrsrc, wvoff + p * 1024u, soff, ring + islot * F_IVAL + (wave * F_PW + p) * 256);
        soff = soff < soff_last ? soff + F_IVAL * 4u : soff_last;
        islot = islot + 1 == NS ? 0 : islot + 1;
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p) issue();

    // ================= P1: the stationary operand y = x pa + po of this wave's 32 rows (gemm_h8_astat.hip)
    f16x8 fa[2 * NG][2];
    i32x8 alo[NG];
    if (p1) {
        const float* xw = g.x + ((size_t)b * g.rows + m0 + wr * 32) * C;
        char* sw = reinterpret_cast<char*>(hid + wr * F_HID);   // the wave's own hand-off buffer of parity 0 (8 KiB >= the 6 KiB tile)
        const int lrow = lane >> 4, c16 = lane & 15;
        f32x4 xs[2][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xs[0][i] = *reinterpret_cast<const f32x4*>(xw + (size_t)(4 * i + lrow) * C + 4 * c16);
        static_for(std::make_integer_sequence<int, NG>{}, [&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (s + 1 < NG) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    xs[(s + 1) & 1][i] = *reinterpret_cast<const f32x4*>(xw + (size_t)(4 * i + lrow) * C + 64 * (s + 1) + 4 * c16);
            }
            const f32x4 pa4 = *reinterpret_cast<const f32x4*>(pro_lds + 64 * s + 4 * c16);
            const f32x4 po4 = *reinterpret_cast<const f32x4*>(pro_lds + C + 64 * s + 4 * c16);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 4 * i + lrow;
                f16x4 hv;
                float lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = h8_clamp(__builtin_fmaf(xs[s & 1][i][e], pa4[e], po4[e]));
                    asm volatile("" : "+v"(y));   // one rounded fp32 value for the hi rounding and the lo difference
                    hv[e] = (_Float16)y;
                    lo[e] = clamp448((y - (float)hv[e]) * H8_AL_SCALE);
                }
                *reinterpret_cast<u32x2*>(sw + row * 128 + (((c16 >> 1) ^ (row & 7)) << 4) + (c16 & 1) * 8) = __builtin_bit_cast(u32x2, hv);
                *reinterpret_cast<unsigned*>(sw + 4096 + row * 64 + (((c16 >> 2) ^ ((row >> 1) & 3)) << 4) + (c16 & 3) * 4) =
                    pack_fp8x4(lo[0], lo[1], lo[2], lo[3]);
            }
            __builtin_amdgcn_wave_barrier();   // a wave's LDS operations execute in order: its reads below see these writes
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int cq = 4 * t + 2 * h + c;
                    fa[2 * s + t][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(sw + r * 128 + ((cq ^ (r & 7)) << 4)));
                }
                const int nc = 2 * t + h;
                const u32x4 v = *reinterpret_cast<const u32x4*>(sw + 4096 + r * 64 + ((nc ^ ((r >> 1) & 3)) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) alo[s][4 * t + e] = (int)v[e];
            }
            __builtin_amdgcn_wave_barrier();
        });
        // P2 reads this buffer in period 0 (the tile "-2", zero weights): it must hold finite fp16 / fp8 patterns — zeros
#pragma unroll
        for (int i = 0; i < F_HID * 4 / 1024; ++i) *reinterpret_cast<u32x4*>(sw + i * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    FSTAMP(1);

    // ---- per-lane addressing of the weight fragments: rows r and 32 + r of a sub-tile share the swizzle
    int boffc[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) boffc[c] = r * 16 + (((2 * h + c) ^ ((r >> 2) & 3)) << 2);
    i32x8 fbA[2], fbB[2];
    auto load_f = [&](const float* sub, i32x8(&f)[2]) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* pc = sub + boffc[c];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(pc + j * 512);
#pragma unroll
                for (int e = 0; e < 4; ++e) f[j][4 * c + e] = (int)v[e];
            }
        }
    };
    // hand-off buffer of row group wr, parity q: hi fragments [4 k-steps][64 lanes] x 16 B | fp8(hi / 8) [2 t][64] x 16 B | fp8 lo [2 t][64] x 16 B
    // (a byte offset, not a pointer: laundered through an asm a pointer loses its LDS address space and is read with flat loads)
    auto hid_of = [&](int q) -> int { return ((q * 4 + wr) * F_HID) * 4 + lane * 16; };
    char* const hidc = reinterpret_cast<char*>(hid);

    f32x16 acc1[2];          // P1: the hidden tile, transposed orientation: acc1[j][4 q + e] = u[row r][32 j + 8 q + 4 h + e]
    f32x16 acc2[NG][2];      // P2: acc2[cb][j][4 q + e] = (h W2^T)[row 8 q + 4 h + e][64 cb + 32 j + r]
    const float c2 = (ACT == 1 || ACT == 2) ? -1.4426950408889634f / (2.0f * g.alpha[0] * g.alpha[0]) : 0.f;
    float eight = H8_AH_DIV;   // the fp8 conversions' scale operand behind an opaque asm (keeps them inside the loops)

    // ---- P1 stage st (compile-time) of the current tile from ring address `cur` (gemm_h8_astat.hip's two sub-steps)
    auto p1_stage = [&](auto ST, const float* cur) {
        constexpr int st = decltype(ST)::value, gq = st >> 1;
        constexpr bool lst = (st & 1) != 0;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        load_f(cur, fbA);
        load_f(cur + 1024, fbB);
        if constexpr (!lst) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const i32x4 wc = c == 0 ? __builtin_shufflevector(fbA[j], fbA[j], 0, 1, 2, 3) : __builtin_shufflevector(fbA[j], fbA[j], 4, 5, 6, 7);
                    acc1[j] = F_MFMA16(__builtin_bit_cast(f16x8, wc), fa[2 * gq][c], (st == 0 && c == 0) ? zero16 : acc1[j]);
                }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const i32x4 wc = c == 0 ? __builtin_shufflevector(fbB[j], fbB[j], 0, 1, 2, 3) : __builtin_shufflevector(fbB[j], fbB[j], 4, 5, 6, 7);
                    acc1[j] = F_MFMA16(__builtin_bit_cast(f16x8, wc), fa[2 * gq + 1][c], acc1[j]);
                }
        } else {
            i32x8 a8;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f16x8 v = fa[2 * gq + t][c];
                    s16x2 q0 = {0, 0}, q1 = {0, 0};
                    q0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q0, f16x2{v[0], v[1]}, eight, false);
                    q0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q0, f16x2{v[2], v[3]}, eight, true);
                    q1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q1, f16x2{v[4], v[5]}, eight, false);
                    q1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q1, f16x2{v[6], v[7]}, eight, true);
                    a8[4 * t + 2 * c] = __builtin_bit_cast(int, q0);
                    a8[4 * t + 2 * c + 1] = __builtin_bit_cast(int, q1);
                }
#pragma unroll
            for (int j = 0; j < 2; ++j) acc1[j] = F_MFMA8(fbA[j], a8, acc1[j], H8_SC_WL, H8_SC_AH);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc1[j] = F_MFMA8(fbB[j], alo[gq], acc1[j], H8_SC_W8, H8_SC_AL);
        }
    };
    // ---- P1: a quarter of the finished tile's epilogue: k-step ks = 2 j + pq of the tile (registers acc1[j][8 pq .. + 7]): bias,
    // activation, clamp, hi / lo split, fp8(hi / 8) -> the hand-off buffer `hb` (this lane's 16-byte slots)
    auto p1_act = [&](auto KS, int tile, int hoff) {
        char* hb = hidc + hoff;
        constexpr int ks = decltype(KS)::value, j = ks >> 1, pq = ks & 1;
        const int n0 = tile * 64 + 32 * j + 16 * pq + 4 * h;
        const f32x4 bA = *reinterpret_cast<const f32x4*>(b0_lds + n0), bB = *reinterpret_cast<const f32x4*>(b0_lds + n0 + 8);
        f16x8 hv;
        float lo[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            f32x2 v = f32x2{acc1[j][8 * pq + e], acc1[j][8 * pq + e + 1]} + (e < 4 ? f32x2{bA[e], bA[e + 1]} : f32x2{bB[e - 4], bB[e - 3]});
            if (ACT == 3) {
                v[0] = fmaxf(v[0], 0.f);
                v[1] = fmaxf(v[1], 0.f);
            }
            if (ACT == 1 || ACT == 2) {
                const f32x2 t = v * v * f32x2{c2, c2};
                v[0] = __builtin_amdgcn_exp2f(t[0]);
                v[1] = __builtin_amdgcn_exp2f(t[1]);
                if (ACT == 1) v = (v - f32x2{0.7f, 0.7f}) * f32x2{1.0f / 0.28f, 1.0f / 0.28f};
            }
            if (ACT == 0 || ACT == 3) {   // unbounded hidden layers: h8_scales.h
                v[0] = h8_clamp(v[0]);
                v[1] = h8_clamp(v[1]);
            }
            asm volatile("" : "+v"(v));   // ONE fp32 value feeds the hi rounding and the lo difference
            hv[e] = (_Float16)v[0];
            hv[e + 1] = (_Float16)v[1];
            const f32x2 d = (v - f32x2{(float)hv[e], (float)hv[e + 1]}) * f32x2{H8_AL_SCALE, H8_AL_SCALE};
            lo[e] = clamp448(d[0]);
            lo[e + 1] = clamp448(d[1]);
        }
        s16x2 q0 = {0, 0}, q1 = {0, 0};
        q0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q0, f16x2{hv[0], hv[1]}, eight, false);
        q0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q0, f16x2{hv[2], hv[3]}, eight, true);
        q1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q1, f16x2{hv[4], hv[5]}, eight, false);
        q1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q1, f16x2{hv[6], hv[7]}, eight, true);
        // operand bytes of the scaled MFMA: byte 16 t + 8 c + e of the lane's 32 = element e of k-step (t, c) = (j, pq)
        *reinterpret_cast<u32x4*>(hb + ks * 1024) = __builtin_bit_cast(u32x4, hv);
        *reinterpret_cast<u32x2*>(hb + 4096 + j * 1024 + pq * 8) = u32x2{(unsigned)__builtin_bit_cast(int, q0), (unsigned)__builtin_bit_cast(int, q1)};
        *reinterpret_cast<u32x2*>(hb + 6144 + j * 1024 + pq * 8) = u32x2{pack_fp8x4(lo[0], lo[1], lo[2], lo[3]), pack_fp8x4(lo[4], lo[5], lo[6], lo[7])};
    };
    // ---- P2 stage st of a tile whose operand sits in hand-off buffer `hb`: column block cb = st / 2, H (even) or L (odd) stage.
    // 192 accumulator registers leave 64 for everything else: one sub-step's weight fragments (16) and operand (8) in use, the
    // next sub-step's in flight — the scheduling fences keep the compiler from reading further ahead (it did, and spilled)
    auto p2_stage = [&](auto ST, const float* cur, int hoff) {
        constexpr int st = decltype(ST)::value, cb = st >> 1;
        constexpr bool lst = (st & 1) != 0;
        // every stage READS its operand again (8 registers at a time): where consecutive stages use the same hand-off buffer the
        // compiler otherwise keeps all 32 operand registers of the tile alive — and spills the accumulators to do so
        asm volatile("" : "+v"(hoff));
        const char* hb = hidc + hoff;
        __builtin_amdgcn_sched_barrier(0);
        load_f(cur, fbA);
        if constexpr (!lst) {
            f16x8 a0 = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(hb));
            f16x8 a1 = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(hb + 1024));
            __builtin_amdgcn_sched_barrier(0);
            load_f(cur + 1024, fbB);
            f16x8 a2 = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(hb + 2048));
            f16x8 a3 = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(hb + 3072));
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[cb][j] = F_MFMA16(a0, __builtin_bit_cast(f16x8, __builtin_shufflevector(fbA[j], fbA[j], 0, 1, 2, 3)), acc2[cb][j]);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[cb][j] = F_MFMA16(a1, __builtin_bit_cast(f16x8, __builtin_shufflevector(fbA[j], fbA[j], 4, 5, 6, 7)), acc2[cb][j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[cb][j] = F_MFMA16(a2, __builtin_bit_cast(f16x8, __builtin_shufflevector(fbB[j], fbB[j], 0, 1, 2, 3)), acc2[cb][j]);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[cb][j] = F_MFMA16(a3, __builtin_bit_cast(f16x8, __builtin_shufflevector(fbB[j], fbB[j], 4, 5, 6, 7)), acc2[cb][j]);
        } else {
            i32x8 a8;
            {
                const u32x4 v0 = *reinterpret_cast<const u32x4*>(hb + 4096), v1 = *reinterpret_cast<const u32x4*>(hb + 4096 + 1024);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a8[e] = (int)v0[e];
                    a8[4 + e] = (int)v1[e];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            load_f(cur + 1024, fbB);
            i32x8 al;
            {
                const u32x4 w0 = *reinterpret_cast<const u32x4*>(hb + 6144), w1 = *reinterpret_cast<const u32x4*>(hb + 6144 + 1024);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    al[e] = (int)w0[e];
                    al[4 + e] = (int)w1[e];
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[cb][j] = F_MFMA8(a8, fbA[j], acc2[cb][j], H8_SC_AH, H8_SC_WL);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[cb][j] = F_MFMA8(al, fbB[j], acc2[cb][j], H8_SC_AL, H8_SC_W8);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // every wave's pieces of the first NS - 1 intervals are in the ring; P1's staging tiles are read
    dma::wait_vm_lgkm0<0>();
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    // top of an interval: its pieces landed (own: the NS - 2 intervals issued after it stay in flight), then everybody's; the next
    // interval's pieces go into the slot of the previous one (every wave has passed the barrier, i.e. is done reading it)
    // The two roles place their DMA issue (~100 - 300 issue cycles per interval and wave) at opposite ends of the interval — P1
    // right after the barrier, P2 behind its matrix instructions — so that one's issue sits beside the other's matrix work
    auto head = [&](bool issue_now) -> const float* {
        dma::wait_vm_lgkm0<(NS - 2) * F_PW>();
        __builtin_amdgcn_s_barrier();
        if (issue_now) issue();
        const float* cur = ring + slot * F_IVAL;
        slot = slot + 1 == NS ? 0 : slot + 1;
        return cur;
    };
    float* red = ring + 4 * 2 * 2048;   // [NG][4][2][64] partial column sums, behind the residual tiles of the final epilogue
    // ================= the schedule.  The two roles are two PROGRAMS (the stationary operand of one and the accumulator of the other
    // must not be live in the same code: 144 + 192 registers), executing the same sequence of barriers and DMA issues
    if (p1) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[j][e] = 0.f;   // period 0 "finishes" tile -1: finite values into the hand-off buffer
        for (int p = 0; p <= NCT + 1; ++p) {
            asm volatile("" : "+s"(eight));
            const int tprev = p - 1 < 0 ? 0 : p - 1 >= NCT ? NCT - 1 : p - 1;   // bias rows of the tile being finished (clamped for the padding tiles)
            static_for(std::make_integer_sequence<int, 3 * T>{}, [&](auto II) {
                constexpr int i = decltype(II)::value;
                const float* cur = head(true);
#ifdef MF8_DIAG_NOP1
                (void)cur;
                return;
#endif
                if constexpr (i < T) {
                    p1_act(std::integral_constant<int, i>{}, tprev, hid_of((p + 1) & 1));   // a quarter of tile p - 1
                } else if constexpr (i < 2 * T) {
                    p1_stage(std::integral_constant<int, i - T>{}, cur);
                } else {
                    p1_stage(std::integral_constant<int, T + 2 * (i - 2 * T)>{}, cur);
                    p1_stage(std::integral_constant<int, T + 2 * (i - 2 * T) + 1>{}, cur + F_STAGE);
                }
            });
        }
        FSTAMP(2);
        dma::wait_vm_lgkm0<0>();           // the re-fetched tail intervals have landed
        __builtin_amdgcn_s_barrier();      // every wave is done with the ring
    } else {
#pragma unroll
        for (int cb = 0; cb < NG; ++cb)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[cb][j][e] = 0.f;
        for (int p = 0; p <= NCT + 1; ++p) {
            static_for(std::make_integer_sequence<int, 3 * T>{}, [&](auto II) {
                constexpr int i = decltype(II)::value;
                const float* cur = head(false);
#ifdef MF8_DIAG_NOP2
                (void)cur;
                issue();
                return;
#endif
                if constexpr (i < T) {
                    const int hb = hid_of(p & 1);   // tile p - 2
                    p2_stage(std::integral_constant<int, T + 2 * i>{}, cur, hb);
                    p2_stage(std::integral_constant<int, T + 2 * i + 1>{}, cur + F_STAGE, hb);
                } else if constexpr (i < 2 * T) {
                    p2_stage(std::integral_constant<int, i - T>{}, cur + F_STAGE, hid_of((p + 1) & 1));   // tile p - 1
                }
                __builtin_amdgcn_sched_barrier(0);
                issue();
            });
        }
        FSTAMP(2);
        // ================= P2: x = x + (acc2 + bias2), column sums.  The ring and the hand-off buffers are dead: residual tiles
        // [32 rows][64 columns] per P2 wave and column block, two in flight, by LDS-DMA (unpool_outproj_h8.hip's epilogue)
        dma::wait_vm_lgkm0<0>();           // the re-fetched tail intervals have landed
        __builtin_amdgcn_s_barrier();      // every wave is done with the ring
        float* xw = g.x + ((size_t)b * g.rows + m0 + wr * 32) * C;
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(xw, 0, 0x7fffffff, 0x00020000);
        float* tt0 = ring + wr * 2 * 2048;   // two tiles of 8 KiB per wave
        const unsigned v0 = (unsigned)(((lane >> 4) * C + 4 * (lane & 15)) * 4);
        auto issue_res = [&](int cb) {
#pragma unroll
            for (int it = 0; it < 8; ++it) dma16_buf(xrsrc, v0 + (unsigned)(it * 4 * C * 4), (unsigned)(cb * 64 * 4), tt0 + (cb & 1) * 2048 + it * 256);
        };
        issue_res(0);
        issue_res(1);
        __builtin_amdgcn_sched_barrier(0);
        const float* tl0 = tt0 + 4 * h * 64 + r;
        const unsigned vo = (unsigned)((4 * h * C + r) * 4);
        static_for(std::make_integer_sequence<int, NG>{}, [&](auto CB) {
            constexpr int cb = decltype(CB)::value;
            // tile cb landed; younger: tile cb + 1 (8 pieces) and, from the second tile on, the 32 stores of the previous one
            if constexpr (cb == 0) dma::wait_vm_lgkm0<8>();
            else if constexpr (cb + 1 < NG) dma::wait_vm_lgkm0<8 + 32>();
            else dma::wait_vm_lgkm0<32>();
            const float* tl = tl0 + (cb & 1) * 2048;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                __builtin_amdgcn_sched_barrier(0);   // one 32-column block at a time: 16 residual reads in flight beside the live accumulators
                const float bc = b2_lds[64 * cb + 32 * j + r];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int qe = 0; qe < 16; ++qe) {
                    const int row = 8 * (qe >> 2) + (qe & 3);   // + 4 h
                    const float v = (acc2[cb][j][qe] + bc) + tl[row * 64 + 32 * j];   // dma::epilogue's order: (A W^T + bias) + residual
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), xrsrc, vo, (unsigned)((row * C + 64 * cb + 32 * j) * 4), 0);
                    s1 += v;
                    s2 = __builtin_fmaf(v, v, s2);
                }
                const float t1 = sum_halves(s1), t2 = sum_halves(s2);
                if (lane < 32) {
                    red[((cb * 4 + wr) * 2 + 0) * 64 + 32 * j + r] = t1;
                    red[((cb * 4 + wr) * 2 + 1) * 64 + 32 * j + r] = t2;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is read before the DMA below overwrites it
            if constexpr (cb + 2 < NG) issue_res(cb + 2);
        });
    }
    if (g.stats) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int i = tid; i < 2 * C; i += 512) {
            const int which = i / C, c = i % C, cb = c >> 6, cl = c & 63;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += red[((cb * 4 + w) * 2 + which) * 64 + cl];
            g.stats[(((size_t)b * tilesM + rt) * 2 + which) * C + c] = t;
        }
    }
    FSTAMP(3);
}

template <int NG, int NS, int ACT>
int mf8_launch_a(const MlpH8Args& g, hipStream_t st) {
    constexpr int C = 64 * NG;
    constexpr size_t lds = ((size_t)NS * F_IVAL + 2 * 4 * F_HID + 2 * C + C + 2 * C) * sizeof(float);
    static_assert(lds <= 160 * 1024, "one block per CU");
    static_assert(NS * F_IVAL >= 4 * 2 * 2048 + NG * 4 * 2 * 64, "the final epilogue's residual tiles and partial column sums fit the dead ring");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_h8_kernel<NG, NS, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((mlp_fused_h8_kernel<NG, NS, ACT>), dim3(g.B * (g.rows / 128)), dim3(512), lds, st, g);
    return (int)hipGetLastError();
}

template <int NG, int NS>
int mf8_launch_t(const MlpH8Args& g, hipStream_t st) {
    switch (g.act) {
        case 0: return mf8_launch_a<NG, NS, 0>(g, st);
        case 1: return mf8_launch_a<NG, NS, 1>(g, st);
        case 2: return mf8_launch_a<NG, NS, 2>(g, st);
        case 3: return mf8_launch_a<NG, NS, 3>(g, st);
        default: return -9;
    }
}

}  // namespace

// feature_dim 384 (2 NG = 12 stages per tile = the schedule's three thirds of four), width = 2 x feature_dim
bool mlp_fused_h8_supported(int C, int Wd, int rows) { return C == 384 && Wd == 2 * C && rows >= 128 && rows % 128 == 0; }

size_t mlp_fused_h8_image_bytes(int C, int Wd) {
    const int NG = C / 64, NCT = Wd / 64, T = 2 * NG / 3;
    return (size_t)f_intervals(NCT, T) * F_IVAL * sizeof(float);
}

int mlp_fused_h8_image_launch(const float* W0, const float* W2, void* img, int C, int Wd, hipStream_t st) {
    if (!mlp_fused_h8_supported(C, Wd, 128)) return -9;
    const int NG = C / 64, NCT = Wd / 64, T = 2 * NG / 3;
    hipLaunchKernelGGL(mlpf8_image_kernel, dim3(1, f_intervals(NCT, T)), dim3(256), 0, st, W0, W2, static_cast<float*>(img), C, Wd);
    return (int)hipGetLastError();
}

int mlp_fused_h8_launch(const MlpH8Args& g0, int C, int Wd, hipStream_t st) {
    if (!mlp_fused_h8_supported(C, Wd, g0.rows) || !g0.pro_a || !g0.pro_o || !g0.w_img) return -9;
    if ((g0.act == 1 || g0.act == 2) && !g0.alpha) return -6;
    static int rev = -1;
    if (rev < 0) {
        const char* e = getenv("GECCO_MF8_REV");
        rev = e ? (atoi(e) != 0) : 0;
    }
    MlpH8Args g = g0;
    g.rev = rev;
    return mf8_launch_t<6, 5>(g, st);
}
