// unpool attention + out_proj + residual + GroupNorm partials in ONE launch, mixed mode ("h8" arithmetic), gfx950.
//
//   x[b, m, :] += softmax(q[b, :, m, :] k[b]^T / sqrt(hd)) v[b] @ W^T + bias          (+ per-(sample, 128-row tile, column) sum / sum^2)
//
// Second half of AttentionPool's round trip (reference models/set_transformer.py:70-75 and :112 — nn.MultiheadAttention with the
// 64 inducer states as keys / values, its out_proj — and the residual of :164).  Before, the mixed mode ran two launches:
// unpool_attn_x3_kernel<hd, fp16, io16> wrote the attention output as an h8 activation image (fp16 hi + fp8 lo: 151 MB at C2) and
// gemm_h8_areg_kernel read it back global -> registers.  The attention output of a row block IS the A operand of out_proj for the
// same rows, so here a block (4 waves, two blocks per CU) owns 128 rows and every wave keeps the attention output of ITS 32 rows
// in registers as the stationary operand of gemm_h8_astat.hip's loop:
//
//   * attention, head by head: k | v of the 64 inducers arrive as a ready fp16 image per (sample, head) — K rows padded, V
//     transposed, key-permuted and zero-padded: the LDS layout of attention_x3.hip, written once per sample by
//     kvh_image_kernel — by buffer_load ... lds into a two-head ring; the q fragments of the wave's 32 rows come straight from
//     the head-major fp16 q; S^T = K q^T, softmax over the 64 keys in registers, O^T = V^T P^T with the probability accumulator
//     as the B operand: the bits unpool_attn_x3_kernel computes.
//   * O^T holds, per lane, ONE query and head-dim indices 8 g + 4 h + e: v = O / l is split as hi = fp16(v), lo = fp8(2^14 (v -
//     hi)) in place — no transpose: the k order of the stationary operand is whatever the accumulator gives (element e of
//     fragment (k-step s, lane half h) is k = 16 s + 8 (e >> 2) + 4 h + (e & 3)), and the W image is written in the same order
//     (h8_image_item<64, true>, SplitJob::pad_ = 16).
//   * out_proj = gemm_h8_astat.hip's main loop (A W = Ah Wh + fp8(Ah 2^-3) fp8(2^16 Wl) + fp8(2^11 Al) fp8(2^5 W), h8_scales.h; 64-column tiles, W
//     streamed through an LDS ring in consumption order).
//   * epilogue per 64-column tile: the residual rows of the tile were fetched by LDS-DMA into a wave-private tile when the tile's
//     K loop started; with the stationary operand as the MFMA's ROW operand an accumulator register holds 32 consecutive columns
//     of one row, so there is no transpose: register by register (A W^T + bias) + residual, a store of 2 x 128 contiguous bytes,
//     and — a lane owning one column — the column sums for the next GroupNorm as 16 adds and one lane-half exchange.
//
// HBM per launch at C2: q 100 MB + x in 201 MB + x out 201 MB (the 151 MB image written + read before never leaves the CU).
#include "gemm_dma_common.h"
#include "h8_scales.h"

#include <stdlib.h>

#include <utility>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

constexpr int U_STAGE = 2048;          // floats per 8 KiB ring stage (two 4 KiB sub-tiles)
constexpr int U_PW = 2;                // 1 KiB W pieces per wave and stage
// Diagnostic builds (tools/probe/uo8_probe.hip): -DUO8_STAMPS per-block s_memtime stamps; -DUO8_DIAG_NOATT / _NOMFMA / _NOEPI /
// _NORES remove one ingredient each (results are then garbage; only the time is of interest)
#if defined(UO8_DIAG_NORES) || defined(UO8_DIAG_NOEPI)
constexpr int U_RES = 0;
#else
constexpr int U_RES = 8;               // residual DMA pieces per wave and column tile
#endif
#ifdef UO8_DIAG_NOEPI
constexpr int U_STORES = 0;
#else
constexpr int U_STORES = 32;           // x stores per wave and column tile (one per accumulator register)
#endif
constexpr int U_TT = 2048;             // floats of a wave's residual tile: [32 rows][64 columns]
constexpr float U_LOG2E = 1.4426950408889634f;
constexpr float U_YL_SCALE = H8_AL_SCALE;   // h8_scales.h

// per (sample, head): K [64][hd + 8] fp16 | V^T [ceil(hd / 32) * 32][72] fp16, padded to whole 4 KiB (one 1 KiB piece per wave)
constexpr int u_kv_bytes(int HD) { return ((64 * (HD + 8) + ((HD + 31) / 32) * 32 * 72) * 2 + 4095) / 4096 * 4096; }
// floats of the region that holds the four waves' residual tiles and, during the attention, two staged heads
constexpr int u_tts_floats(int HD) { return 2 * u_kv_bytes(HD) > 4 * U_TT * 4 ? 2 * u_kv_bytes(HD) / 4 : 4 * U_TT; }

__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}

#ifdef UO8_STAMPS
__device__ unsigned long long g_uo8_stamps[2048 * 4];
__device__ unsigned long long g_uo8_epi[2048 * 4];   // per block: ticks inside the epilogues: transposed phase | row phase | first K step of a tile
#define UTICK() __builtin_amdgcn_s_memtime()
#define USTAMP(i)                                                                                                \
    do {                                                                                                         \
        if (threadIdx.x == 0 && blockIdx.x < 2048) g_uo8_stamps[blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define USTAMP(i)
#endif

#define U_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#ifdef UO8_DIAG_NOMFMA
__device__ __forceinline__ f32x16 u_keep16(f16x8 a, f16x8 b, f32x16 c) {
    asm volatile("" ::"v"(a), "v"(b));
    return c;
}
__device__ __forceinline__ f32x16 u_keep8(i32x8 a, i32x8 b, f32x16 c) {
    asm volatile("" ::"v"(a), "v"(b));
    return c;
}
#define UG_MFMA16(a, b, c) u_keep16(a, b, c)
#define UG_MFMA8(a, b, c, sa, sb) u_keep8(a, b, c)
#else
#define UG_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define UG_MFMA8(a, b, c, sa, sb) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb)
#endif

// the value of the other lane half (lane ^ 32) without the LDS round trip of ds_bpermute: v_permlane32_swap of a register with
// itself leaves {lower half's value, upper half's value} on every lane
__device__ __forceinline__ float max_halves(float v) {
    const auto a = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    return fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
}
__device__ __forceinline__ float sum_halves(float v) {
    const auto a = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    return __uint_as_float(a[0]) + __uint_as_float(a[1]);
}

__device__ __forceinline__ float clamp448(float v) { return __builtin_fminf(__builtin_fmaxf(v, -448.f), 448.f); }

__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
    int pk = 0;
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, pk, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, pk, true);
    return (unsigned)pk;
}

// ---------------------------------------------------------------------------------------------------------------------
// k | v of the inducer states (B, 64, 2C) fp32 -> per (sample, head) the fp16 LDS image of the attention phase:
//   K  [64 keys][hd + 8]: row = key, the head's hd values (pad columns zero);
//   V^T [ceil(hd / 32) * 32][72]: row = head-dim index d (rows >= hd zero), key 16 c + 8 a + 4 g + i at position 16 c + 8 g + 4 a + i
//       (lane half g reads its 8 keys of a 16-key chunk contiguously: attention_x3.hip).
// One block per (sample, head); same roundings (fp32 -> fp16, nearest even) as the staging code of unpool_attn_x3_kernel.  The
// image is assembled in LDS (coalesced 16-byte reads of the key rows, 2-byte transposed writes) and leaves as 16-byte chunks.
__global__ __launch_bounds__(256) void kvh_image_kernel(const float* __restrict__ kvh, u16* __restrict__ img, int C, int H, int HD, int kvb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16* buf = reinterpret_cast<u16*>(smem);
    const int b = blockIdx.x / H, hh = blockIdx.x % H;
    const int KS = HD + 8, nk = 64 * KS, CH = HD / 4;
    for (int i = threadIdx.x; i < kvb / 16; i += 256) reinterpret_cast<u32x4*>(buf)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    const float* src = kvh + (size_t)b * 64 * 2 * C + hh * HD;
    for (int i = threadIdx.x; i < 64 * CH; i += 256) {
        const int key = i / CH, ch = i % CH;
        const f32x4 kf = *reinterpret_cast<const f32x4*>(src + (size_t)key * 2 * C + ch * 4);
        const f32x4 vf = *reinterpret_cast<const f32x4*>(src + (size_t)key * 2 * C + C + ch * 4);
        const int pos = (key & ~15) + 8 * ((key >> 2) & 1) + 4 * ((key >> 3) & 1) + (key & 3);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            buf[key * KS + ch * 4 + e] = __builtin_bit_cast(u16, (_Float16)kf[e]);
            buf[nk + (ch * 4 + e) * 72 + pos] = __builtin_bit_cast(u16, (_Float16)vf[e]);
        }
    }
    __syncthreads();
    u32x4* dst = reinterpret_cast<u32x4*>(img + (size_t)blockIdx.x * (kvb / 2));
    for (int i = threadIdx.x; i < kvb / 16; i += 256) dst[i] = reinterpret_cast<const u32x4*>(buf)[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// NG = C / 64 (64-k groups = 64-column tiles: out_proj is square); HD = head dim; NS ring stages (running slot counter)
// d = 512 (NG = 8, 64-wide heads): 192 registers of stationary operand — one block per CU, a wave per SIMD with the whole register file
template <int NG, int HD, int NS>
__global__ __launch_bounds__(256, NG > 6 ? 1 : 2) void unpool_outproj_h8_kernel(UnpoolH8Args g) {
    constexpr int C = 64 * NG, H = C / HD, NKT = 2 * NG, NC = HD / 16, DT = (HD + 31) / 32;
    constexpr int KS = HD + 8, VS = 72, KVB = u_kv_bytes(HD), PK = KVB / 4096;
    static_assert(C % HD == 0 && HD % 16 == 0 && HD <= 64, "head dims 16 .. 64");
    constexpr int TTSF = u_tts_floats(HD);     // the four residual tiles, or (64-wide heads) the two staged heads that alias them
    static_assert(NS >= 4 && NS <= NKT, "lookahead NS - 1 >= 3 stages; the residual pieces land inside their tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ring = smem;                        // [NS][U_STAGE]
    float* tts = ring + NS * U_STAGE;          // [4][U_TT] residual tiles; during the attention: two staged heads
    float* bias_lds = tts + TTSF;              // [C]
    float* red = bias_lds + C;                 // [4 waves][2][64] column sums of one tile

    const int tilesM = g.rows / 128;
    const int bid = g.rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int b = bid / tilesM, rt = bid % tilesM, m0 = rt * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // rows m0 + 32 wave .. + 31
    const int r = lane & 31, h = lane >> 5;

    USTAMP(0);
    for (int n = tid; n < C; n += 256) bias_lds[n] = g.bias ? g.bias[n] : 0.f;
    // two blocks per CU run the same phases (vector-heavy attention, then the matrix loop): the second of a pair may start late
    if (g.stagger > 0 && (((blockIdx.x >> 3) / g.pair) & 1)) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)g.stagger) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();   // before the first DMA: a block barrier drains the vector-memory queue

    // ---- W stream: U_PW 1 KiB pieces per wave and stage, consumed front to back; past its end the last stage is fetched again
    // (into slots nobody reads any more): every step issues, so every wait below is the same count
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.w_img), 0, 0x7fffffff, 0x00020000);
    const unsigned wvoff = (unsigned)(wave * U_PW * 256 + lane * 4) * 4u;
    const unsigned soff_last = (unsigned)(NG * NKT - 1) * (U_STAGE * 4u);
    unsigned soff = 0;
    int islot = 0;
    auto issue = [&]() {
#pragma unroll
        for (int p = 0; p < U_PW; ++p) dma16_buf(wrsrc, wvoff + p * 1024u, soff, ring + islot * U_STAGE + (wave * U_PW + p) * 256);
        soff = soff < soff_last ? soff + U_STAGE * 4u : soff_last;
        islot = islot + 1 == NS ? 0 : islot + 1;
    };
#pragma unroll
    for (int p = 0; p < NS - 1; ++p) issue();

    // ================= attention: this wave's 32 queries against the 64 inducer keys / values, head by head
    f16x8 fa[2 * NG][2];   // fa[s >> 1][s & 1]: k-step s (16 k) of the stationary operand, element e: k = 16 s + 8 (e >> 2) + 4 h + (e & 3)
    i32x8 alo[NG];         // dword 4 t + 2 c + (e >> 2) of group gq: fp8(2^14 lo) of fa[2 gq + t][c]
#ifdef UO8_DIAG_NOATT
#pragma unroll
    for (int i = 0; i < 2 * NG; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) fa[i][c][e] = (_Float16)(0.01f * (float)((lane + i + c + e) & 63));
#pragma unroll
    for (int i = 0; i < NG; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) alo[i][e] = 0x20202020 + lane + e;
#else
    {
        const __amdgpu_buffer_rsrc_t kvrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(static_cast<const char*>(g.kv_img) + (size_t)b * H * KVB), 0, 0x7fffffff, 0x00020000);
        char* kvbuf = reinterpret_cast<char*>(tts);
        const unsigned kvoff = (unsigned)(wave * PK * 1024 + lane * 16);
        auto kv_issue = [&](int hh) {
#pragma unroll
            for (int p = 0; p < PK; ++p)
                dma16_buf(kvrsrc, kvoff + p * 1024u, (unsigned)hh * KVB, kvbuf + (hh & 1) * KVB + (wave * PK + p) * 1024);
        };
        const _Float16* q16 = reinterpret_cast<const _Float16*>(g.q16);
        u32x4 qf[2][NC];
        auto q_load = [&](int hh, int set) {
            const _Float16* qr = q16 + (((size_t)b * H + hh) * g.rows + m0 + wave * 32 + r) * HD + 8 * h;
#pragma unroll
            for (int c = 0; c < NC; ++c) qf[set][c] = *reinterpret_cast<const u32x4*>(qr + c * 16);
        };
        kv_issue(0);
        q_load(0, 0);
        if (H > 1) {
            kv_issue(1);
            q_load(1, 1);
        }
        const float scale = U_LOG2E * rsqrtf((float)HD);
        static_for(std::make_integer_sequence<int, H>{}, [&](auto HH) {
            constexpr int hh = decltype(HH)::value, set = hh & 1;
            // own pieces of head hh (and its q fragments) landed; younger: head 1's, issued before head 0 was awaited
            if constexpr (hh == 0 && H > 1) dma::wait_vm_lgkm0<PK + NC>();
            else dma::wait_vm_lgkm0<0>();
            __builtin_amdgcn_s_barrier();   // every wave's pieces of head hh are visible; every wave is done with head hh - 1
            if constexpr (hh >= 1 && hh + 1 < H) {
                kv_issue(hh + 1);
                q_load(hh + 1, set ^ 1);
            }
            const u16* Kh = reinterpret_cast<const u16*>(kvbuf + set * KVB);
            const u16* Vt = Kh + 64 * KS;
            f32x16 sc[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) sc[kt][e] = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const u32x4 kf = *reinterpret_cast<const u32x4*>(Kh + (kt * 32 + r) * KS + c * 16 + 8 * h);
                    sc[kt] = U_MFMA16(__builtin_bit_cast(f16x8, kf), __builtin_bit_cast(f16x8, qf[set][c]), sc[kt]);
                }
            // element-wise steps on pairs (v_pk_mul_f32 / v_pk_add_f32: the same IEEE results per element as the scalar forms of
            // unpool_attn_x3_kernel at half the vector instructions); the sum over the keys keeps that kernel's serial order
            const f32x2 scale2 = {scale, scale};
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 t = f32x2{sc[kt][e], sc[kt][e + 1]} * scale2;
                    sc[kt][e] = t[0];
                    sc[kt][e + 1] = t[1];
                    mx = fmaxf(mx, fmaxf(t[0], t[1]));
                }
            mx = max_halves(mx);
            const f32x2 mx2 = {mx, mx};
            float ls = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 t = f32x2{sc[kt][e], sc[kt][e + 1]} - mx2;
                    sc[kt][e] = __builtin_amdgcn_exp2f(t[0]);
                    sc[kt][e + 1] = __builtin_amdgcn_exp2f(t[1]);
                    ls += sc[kt][e];
                    ls += sc[kt][e + 1];
                }
            ls = sum_halves(ls);
            const float inv = 1.0f / ls;
            // the probabilities as the four fp16 B fragments (16-key chunk c16 = 2 kt + sg) before the second product starts: the
            // score accumulators are dead from here on
            f16x8 pf[4];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int sg = 0; sg < 2; ++sg)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[2 * kt + sg][e] = (_Float16)sc[kt][8 * sg + e];
#pragma unroll
            for (int c16 = 0; c16 < 4; ++c16) asm volatile("" : "+v"(pf[c16]));
            f32x16 O[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) O[dt][e] = 0.f;
#pragma unroll
            for (int c16 = 0; c16 < 4; ++c16)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const u32x4 vf = *reinterpret_cast<const u32x4*>(Vt + (dt * 32 + r) * VS + c16 * 16 + 8 * h);
                    O[dt] = U_MFMA16(__builtin_bit_cast(f16x8, vf), pf[c16], O[dt]);
                }
            // O^T: lane (r, h) holds query r, head-dim index d = 32 dt + 8 g4 + 4 h + e in register 4 g4 + e.  Registers of the
            // pair (g4 = 2 p, 2 p + 1) are the 8 values lane half h contributes to k-step (hh hd + 32 dt + 16 p) / 16
            static_for(std::make_integer_sequence<int, 2 * DT>{}, [&](auto PP) {
                constexpr int pp = decltype(PP)::value, dt = pp >> 1, p = pp & 1;
                if constexpr (32 * dt + 16 * p < HD) {
                    constexpr int ks = (hh * HD + 32 * dt + 16 * p) / 16;
                    f16x8 hv;
                    float lo[8];
                    const f32x2 inv2 = {inv, inv}, ysc2 = {U_YL_SCALE, U_YL_SCALE};
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        f32x2 v = f32x2{O[dt][8 * p + e], O[dt][8 * p + e + 1]} * inv2;
                        v[0] = h8_clamp(v[0]);   // h8_scales.h: an h8 operand is finite in every term
                        v[1] = h8_clamp(v[1]);
                        // ONE fp32 value feeds the hi rounding and the lo difference (gemm_h8_astat.hip's epilogue)
                        asm volatile("" : "+v"(v));
                        hv[e] = (_Float16)v[0];
                        hv[e + 1] = (_Float16)v[1];
                        const f32x2 d = (v - f32x2{(float)hv[e], (float)hv[e + 1]}) * ysc2;
                        lo[e] = clamp448(d[0]);
                        lo[e + 1] = clamp448(d[1]);
                    }
                    fa[ks >> 1][ks & 1] = hv;
                    alo[ks >> 2][4 * ((ks >> 1) & 1) + 2 * (ks & 1) + 0] = (int)pack_fp8x4(lo[0], lo[1], lo[2], lo[3]);
                    alo[ks >> 2][4 * ((ks >> 1) & 1) + 2 * (ks & 1) + 1] = (int)pack_fp8x4(lo[4], lo[5], lo[6], lo[7]);
                    // formed here, under the next head's loads: left alone the compiler sinks all 192 conversions behind the last head
                    asm volatile("" : "+v"(fa[ks >> 1][ks & 1]));
                    asm volatile("" : "+v"(alo[ks >> 2][4 * ((ks >> 1) & 1) + 2 * (ks & 1) + 0]), "+v"(alo[ks >> 2][4 * ((ks >> 1) & 1) + 2 * (ks & 1) + 1]));
                }
            });
        });
    }
#endif
    USTAMP(1);

    // ---- per-lane addressing of the W fragments (gemm_kvq_astat_kernel): rows r and 32 + r of a sub-tile share the swizzle
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

    // ---- residual rows of a column tile -> the wave's private tile [32 rows][64 columns], row-major as the DMA writes it
    float* tt = tts + wave * U_TT;
    float* xw = g.x + ((size_t)b * g.rows + m0 + wave * 32) * C;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(xw, 0, 0x7fffffff, 0x00020000);
    // Per-lane addresses of the residual / epilogue phases are formed from `lane` behind an opaque asm where they are used:
    // hoisted out of the column-tile loop (24 registers of loop invariants) they were spilled to scratch, whose reloads wait
    // vmcnt(0) and drain the ring
    auto issue_res = [&](int ct) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const unsigned v0 = (unsigned)(((ln >> 4) * C + 4 * (ln & 15)) * 4);   // DMA instruction `it`: rows 4 it + (lane >> 4), 16 bytes per lane
#pragma unroll
        for (int it = 0; it < 8; ++it) dma16_buf(xrsrc, v0 + (unsigned)(it * 4 * C * 4), (unsigned)(ct * 64 * 4), tt + it * 256);
    };

    f32x16 acc[2];
    // ---- epilogue of one 64-column tile.  The MFMAs take the stationary operand as the ROW operand: acc[j][4 q + e] =
    // (A W^T)[row 8 q + 4 h + e][n0 + 32 j + r] — a register holds 32 consecutive columns of one row across a lane half, so the
    // result needs no transpose: per register one conflict-free 4-byte read of the residual tile and one store of 2 x 128
    // contiguous bytes; a lane owns ONE column per 32-column block: its column sums are 16 adds and one lane-half exchange
#ifdef UO8_STAMPS
    unsigned long long t_e1 = 0, t_e2 = 0;
#endif
    auto epilogue = [&](int ct) {
        const int n0 = ct * 64;
#ifdef UO8_STAMPS
        const unsigned long long te0 = UTICK();
#endif
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int r = ln & 31, h = ln >> 5;
        const float* tl = tt + 4 * h * 64 + r;
        const unsigned vo = (unsigned)((4 * h * C + r) * 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float bc = bias_lds[n0 + 32 * j + r];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int qe = 0; qe < 16; ++qe) {
                const int row = 8 * (qe >> 2) + (qe & 3);   // + 4 h
                const float v = (acc[j][qe] + bc) + tl[row * 64 + 32 * j];   // dma::epilogue's order: (A W^T + bias) + residual
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), xrsrc, vo, (unsigned)((row * C + n0 + 32 * j) * 4), 0);   // default policy: the next kernels re-read x
                s1 += v;
                s2 = __builtin_fmaf(v, v, s2);   // explicit fma, as in dma::epilogue
            }
            if (g.stats) {
                const float t1 = sum_halves(s1), t2 = sum_halves(s2);   // rows 4 h + .. of both lane halves
                if (ln < 32) {
                    red[(wave * 2 + 0) * 64 + 32 * j + r] = t1;
                    red[(wave * 2 + 1) * 64 + 32 * j + r] = t2;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is read (and `red` written) before anything overwrites it
#ifdef UO8_STAMPS
        t_e2 += UTICK() - te0;
#endif
    };
    // column sums of tile ct over the block's 128 rows: after a block barrier that follows every wave's epilogue(ct)
    auto write_stats = [&](int ct) {
        float* dst = g.stats + (((size_t)b * tilesM + rt) * 2) * C + ct * 64 + lane;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += red[(w * 2 + which) * 64 + lane];
            dst[(size_t)which * C] = t;
        }
    };

    // every wave's pieces of the first NS - 1 stages are in the ring; every wave is done with the staged heads
    dma::wait_vm_lgkm0<0>();
    __builtin_amdgcn_s_barrier();
    int slot = 0;                                      // slot of the current stage
    load_f(ring, fbA);
    float one = H8_AH_DIV;   // the fp8 conversions' scale operand (fp8(Ah / 8), h8_scales.h) behind an opaque asm: keeps them inside the column-tile loop
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // one stage: wait / barrier / issue, two sub-steps.  kt even: the H stage of group kt / 2, odd: its L stage
    auto stage = [&](auto KT, int ct) {
        constexpr int kt = decltype(KT)::value, gq = kt >> 1;
        constexpr bool lst = (kt & 1) != 0;
        // own pieces of stage kt + 1 landed.  Younger vector-memory operations that may stay in flight: the pieces of the NS - 3
        // stages after it and, while the awaited pieces are older than them, this tile's residual pieces
        // (issued right after step 0's stage: one step more before a wait covers them) and the stores of the previous tile's epilogue
        constexpr int young = (NS - 3) * U_PW;
        if constexpr (kt == 0) {
            if (ct == 0) dma::wait_vm_lgkm0<young>();
            else dma::wait_vm_lgkm0<young + U_STORES>();
        } else if constexpr (kt <= NS - 3) {
            if (ct == 0) dma::wait_vm_lgkm0<young + U_RES>();
            else dma::wait_vm_lgkm0<young + U_RES + U_STORES>();
        } else if constexpr (kt == NS - 2) {
            dma::wait_vm_lgkm0<young + U_RES>();
        } else {
            dma::wait_vm_lgkm0<young>();
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(fbA[j]));
        __builtin_amdgcn_s_barrier();
        if constexpr (kt == 0) {
            if (g.stats && ct > 0 && wave == ((ct - 1) & 3)) write_stats(ct - 1);
        }
        issue();
#if !defined(UO8_DIAG_NORES) && !defined(UO8_DIAG_NOEPI)
        if constexpr (kt == 0) issue_res(ct);   // the previous tile's epilogue has read the transpose tile: its rows may be overwritten
#endif
        const float* cur = ring + slot * U_STAGE;
        slot = slot + 1 == NS ? 0 : slot + 1;
        const float* nxt = ring + slot * U_STAGE;
        load_f(cur + 1024, fbB);
        if constexpr (!lst) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const i32x4 wc = c == 0 ? __builtin_shufflevector(fbA[j], fbA[j], 0, 1, 2, 3) : __builtin_shufflevector(fbA[j], fbA[j], 4, 5, 6, 7);
                    acc[j] = UG_MFMA16(fa[2 * gq][c], __builtin_bit_cast(f16x8, wc), (kt == 0 && c == 0) ? zero16 : acc[j]);
                }
        } else {
            // Ah Wl: fp8(Ah) of the group's two k-steps, bytes in the image's k order (16 t + 8 c + e)
            i32x8 a8;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f16x8 v = fa[2 * gq + t][c];
                    s16x2 p0 = {0, 0}, p1 = {0, 0};
                    p0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p0, f16x2{v[0], v[1]}, one, false);
                    p0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p0, f16x2{v[2], v[3]}, one, true);
                    p1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p1, f16x2{v[4], v[5]}, one, false);
                    p1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(p1, f16x2{v[6], v[7]}, one, true);
                    a8[4 * t + 2 * c] = __builtin_bit_cast(int, p0);
                    a8[4 * t + 2 * c + 1] = __builtin_bit_cast(int, p1);
                }
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = UG_MFMA8(a8, fbA[j], acc[j], H8_SC_AH, H8_SC_WL);
        }
        load_f(nxt, fbA);
        if constexpr (!lst) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const i32x4 wc = c == 0 ? __builtin_shufflevector(fbB[j], fbB[j], 0, 1, 2, 3) : __builtin_shufflevector(fbB[j], fbB[j], 4, 5, 6, 7);
                    acc[j] = UG_MFMA16(fa[2 * gq + 1][c], __builtin_bit_cast(f16x8, wc), acc[j]);
                }
        } else {
            // Al W
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = UG_MFMA8(alo[gq], fbB[j], acc[j], H8_SC_AL, H8_SC_W8);
        }
    };
    for (int ct = 0; ct < NG; ++ct) {
        asm volatile("" : "+s"(one));
        static_for(std::make_integer_sequence<int, NKT>{}, [&](auto KT) { stage(KT, ct); });
#ifdef UO8_DIAG_NOEPI
        if (acc[0][0] == 123.456f) epilogue(ct);
#else
        epilogue(ct);
#endif
    }
    USTAMP(2);
#ifdef UO8_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 2048) {
        g_uo8_epi[blockIdx.x * 4 + 0] = t_e1;
        g_uo8_epi[blockIdx.x * 4 + 1] = t_e2;
    }
#endif
    if (g.stats) {
        __builtin_amdgcn_s_barrier();   // every wave's partial sums of the last tile are in `red` (lgkmcnt(0) closed its epilogue)
        if (wave == ((NG - 1) & 3)) write_stats(NG - 1);
    }
    dma::wait_vm_lgkm0<0>();   // the re-fetched tail stages still target this block's LDS: land them before it is released
    USTAMP(3);
}

template <int NG, int HD, int NS>
int uo8_launch_t(const UnpoolH8Args& g, hipStream_t st) {
    constexpr int C = 64 * NG;
    constexpr size_t lds = ((size_t)NS * U_STAGE + u_tts_floats(HD) + C + 4 * 2 * 64) * sizeof(float);
    static_assert(lds <= 80 * 1024 || NG > 6, "two blocks per CU");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(unpool_outproj_h8_kernel<NG, HD, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((unpool_outproj_h8_kernel<NG, HD, NS>), dim3(g.B * (g.rows / 128)), dim3(256), lds, st, g);
    return (int)hipGetLastError();
}

}  // namespace

// the shipped shapes with 8 heads: d = 128, 256, 384, and d = 512 at one block per CU
bool unpool_outproj_h8_supported(int C, int H, int rows) {
    if (rows < 128 || rows % 128 || H <= 0 || C % H) return false;
    const int hd = C / H;
    static const int k512 = [] { const char* e = getenv("GECCO_UO8_K512"); return e ? atoi(e) : 1; }();   // 0: d = 512 keeps the two launches
    return (C == 128 && hd == 16) || (C == 256 && hd == 32) || (C == 384 && hd == 48) || (C == 512 && hd == 64 && k512);
}

size_t unpool_outproj_h8_kv_bytes(int B, int C, int H) { return (size_t)B * H * u_kv_bytes(C / H); }

int kvh_image_launch(const float* kvh, void* img, int B, int C, int H, hipStream_t st) {
    const int hd = C / H;
    if (hd % 16 || hd > 64) return -9;
    hipLaunchKernelGGL(kvh_image_kernel, dim3(B * H), dim3(256), (size_t)u_kv_bytes(hd), st, kvh, static_cast<u16*>(img), C, H, hd, u_kv_bytes(hd));
    return (int)hipGetLastError();
}

int unpool_outproj_h8_launch(const UnpoolH8Args& g0, int C, hipStream_t st) {
    if (!unpool_outproj_h8_supported(C, g0.H, g0.rows)) return -9;
    static int rev = -1;   // GECCO_UO8_REV=0: blocks walk the row panels first to last
    if (rev < 0) {
        const char* e = getenv("GECCO_UO8_REV");
        rev = e ? (atoi(e) != 0) : 1;
    }
    static int stagger = -1, pair = 32;   // GECCO_UO8_STAGGER=<ticks>, GECCO_UO8_PAIR=<blocks per XCD between the two blocks of a CU>
    if (stagger < 0) {
        const char* e = getenv("GECCO_UO8_STAGGER");
        stagger = e ? atoi(e) : 0;
        const char* e2 = getenv("GECCO_UO8_PAIR");
        pair = e2 && atoi(e2) > 0 ? atoi(e2) : 32;
    }
    UnpoolH8Args g = g0;
    g.rev = rev;
    g.stagger = g.B * (g.rows / 128) >= 512 ? stagger : 0;
    g.pair = pair;
    switch (C) {
        case 128: return uo8_launch_t<2, 16, 4>(g, st);
        case 256: return uo8_launch_t<4, 32, 5>(g, st);
        case 512: return uo8_launch_t<8, 64, 4>(g, st);
        default: return uo8_launch_t<6, 48, 5>(g, st);
    }
}
