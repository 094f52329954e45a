// Point-stream MLP of a BroadcastingLayer in split-bf16 ("bf16x3") arithmetic, ONE launch (gfx950):
//
//   x <- x + W2 g(W0 (a * x + o) + b0) + b2            (+ GroupNorm partial sums of the new x)
//
// i.e. `x = x + mlp(mlp_norm(x))` (reference models/set_transformer.py:164-166, models/mlp.py, activation.py:17-24,
// normalization.py:36-44) with the AdaGN apply folded to per-(sample, channel) coefficients (a, o).  Same arithmetic as
// the stand-alone split-bf16 linears (every fp32 operand = bf16 hi + bf16 lo, three v_mfma_f32_16x16x32_bf16 per
// product, fp32 accumulation), but nothing of the layer's interior touches HBM or the LDS-fill path:
//
//  * TRANSPOSED products with the points on the MFMA's column (lane) index.  A wave owns 16 P points from x to x:
//    y = a * x + o of its points, split once into hi | lo B-operand fragments, stays in REGISTERS for the whole kernel
//    (C/32 K-steps x P point tiles x 8 registers); the hidden layer is produced 32 units at a time as
//    X^T = W0[units] y^T — an accumulator whose lane already holds, for ITS point, the 8 hidden units a B fragment of
//    the second product needs (the weight image stores W2's k order to match): bias, activation and the hi | lo split
//    happen on the accumulator registers and out^T += W2[:, units] X^T consumes them at once.  The C x 16 P output
//    accumulators live in registers too (C/16 x P tiles).  No activation tile is ever staged in LDS.
//  * The only streamed operand is the weights: one pre-tiled image per layer in consumption order — per hidden chunk
//    of 32 units, C/128 slots of W0 (32 units x 128 k) then C/128 slots of W2 (128 channels x 32 units), each slot
//    16 KiB = exactly its LDS image (hi | lo per 128-byte row, swizzle baked in) — through an 8-slot ring filled by
//    global_load_lds_dwordx4 seven slots ahead (the stream is L2-resident: 4 C^2 * 4 bytes... 2.4 MB at C = 384), with
//    counted vmcnt waits and ONE s_barrier per slot = per 48 MFMAs of a wave.
//  * 4 waves (one per SIMD, up to 512 registers each), 64 P points per block.
//
// Epilogue from the accumulators: + b2 + x (fp32 residual rows re-read), 16-byte stores, per-column sums and sums of
// squares of the new x reduced over the block's points in a fixed order (deterministic).
// Template <NC = C / 128, P>: C = 384 with P = 2 (128 points per block), C = 512 with P = 1.  Hidden width 2 C.
#include "../../../gecco_amd/csrc/common.h"
#include "x3_experimental.h"

#ifdef MX_DIAG_NOMFMA
#define MX_MFMA(a, b, c) ((c) + f32x4{(float)(a)[0] + (float)(b)[0], 0.f, 0.f, 0.f})
#else
#define MX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 0xF) | (0x7 << 4) | ((lgkm & 0xF) << 8) | ((vm >> 4) << 14); }
template <int PER>   // s_waitcnt vmcnt(PER * slots): PER DMA instructions per wave per slot; wave-uniform slots 0..6
__device__ __forceinline__ void wait_vm_slots(int slots) {
    switch (slots) {
        case 0: __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0xF)); break;
        case 1: __builtin_amdgcn_s_waitcnt(waitcnt_imm(PER, 0xF)); break;
        case 2: __builtin_amdgcn_s_waitcnt(waitcnt_imm(2 * PER, 0xF)); break;
        case 3: __builtin_amdgcn_s_waitcnt(waitcnt_imm(3 * PER, 0xF)); break;
        case 4: __builtin_amdgcn_s_waitcnt(waitcnt_imm(4 * PER, 0xF)); break;
        case 5: __builtin_amdgcn_s_waitcnt(waitcnt_imm(5 * PER, 0xF)); break;
        default: __builtin_amdgcn_s_waitcnt(waitcnt_imm(6 * PER, 0xF)); break;
    }
}
// chunk swizzle of a 128-byte LDS row (gemm_x3_planes.hip): conflict-free ds_read_b128 fragment reads
__host__ __device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) ^ (((row >> 2) & 1) * 4) ^ (((row >> 3) & 1) * 6); }

__device__ __forceinline__ void split8p(const float (&x)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned ua = __float_as_uint(x[2 * p]), uc = __float_as_uint(x[2 * p + 1]);
        hi[p] = __builtin_amdgcn_perm(uc, ua, 0x07060302u);
        const float la = x[2 * p] - __uint_as_float(ua & 0xFFFF0000u), lc = x[2 * p + 1] - __uint_as_float(uc & 0xFFFF0000u);
        const __bf16 ba = (__bf16)la, bc = (__bf16)lc;
        lo[p] = (unsigned)__builtin_bit_cast(unsigned short, ba) | ((unsigned)__builtin_bit_cast(unsigned short, bc) << 16);
    }
}

constexpr int SLOT = 16384;

// YL: the lo fragments of y's last YL K-steps live in LDS instead of registers (C = 384, P = 2: y and the output
// accumulators alone are 384 of the 512 registers; the compiler needs ~100 more and would spill y into scratch, whose
// reloads wait vmcnt(0) and drain the weight ring).  RING slots of 16 KiB; AHEAD = RING - 1 are kept in flight.
#ifdef MX_STAMPS
__device__ unsigned long long g_mx_stamps[1024 * 8];
#define MX_STAMP(i) do { if (tid == 0) g_mx_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MX_STAMP(i)
#endif

// NW waves per block (4: one per SIMD, up to 512 registers each; 8: two per SIMD, 256 registers each — the second wave
// fills the issue slots and latencies the first leaves open), each owning 16 P points.
template <int NC, int P, int YL, int RING, int NW>
__global__ __launch_bounds__(64 * NW) void mlp_x3_fused_kernel(MlpX3Args g) {
    constexpr int AHEAD = RING - 1, NT = 64 * NW, DPW = 16 / NW;   // threads; DMA wave-instructions per wave per 16 KiB slot
    constexpr int C = 128 * NC, WD = 2 * C, NK = C / 32, NCT = C / 16, NCH = WD / 32;
    constexpr int S1 = NC, S2 = NC, NS = S1 + S2;    // slots per hidden chunk: W0 pieces of 4 K-steps, W2 pieces of 8 channel tiles
    constexpr int PTS = 16 * P, BM = NW * PTS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* lb0 = reinterpret_cast<float*>(lds + RING * SLOT);   // b0 (WD) | b2 (C)
    float* lb2 = lb0 + WD;
    unsigned char* ylo_lds = reinterpret_cast<unsigned char*>(lb2 + C);   // [4 waves][YL][P][64 lanes] x 16 B
    float* red = reinterpret_cast<float*>(lds);                  // [4 waves][2][C]: reuses the ring once it is dead

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * BM;
    const int b = m0 / g.rows_per_sample;
    const int total = NCH * NS;                      // slots of the whole stream
    const unsigned char* stream = static_cast<const unsigned char*>(g.w_stream);

    MX_STAMP(0);
    // ---- weight ring: each wave copies its quarter (4 KiB = 4 wave-instructions) of every slot, linearly
    auto issue = [&](int sigma) {
        const unsigned char* src = stream + (size_t)sigma * SLOT + wave * (DPW * 1024) + lane * 16;
        unsigned char* dst = lds + (sigma % RING) * SLOT + wave * (DPW * 1024);
#pragma unroll
        for (int q = 0; q < DPW; ++q) dma16(src + q * 1024, dst + q * 1024);
    };
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) issue(s);        // total >= 24 slots

    // ---- biases into LDS (ordinary loads; the wait below also covers the ring's first slots: in-order completion)
    for (int i = tid; i < WD; i += NT) lb0[i] = g.b0 ? g.b0[i] : 0.f;
    for (int i = tid; i < C; i += NT) lb2[i] = g.b2 ? g.b2[i] : 0.f;
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0x3F, 0));   // the LDS writes are done before this wave reaches the first barrier

    // ---- y = a * x + o of this wave's 16 P points, split into B-operand fragments (lane (fr, fq): point fr of the
    // tile, k = 32 kt + 8 fq .. + 7)
    bf16x8 yh[NK][P], yl[NK - YL][P];
    unsigned char* my_ylo = ylo_lds + (size_t)wave * YL * P * 1024 + lane * 16;
    {
        const float* pa = g.pro_a + (size_t)b * C + 8 * fq;
        const float* po = g.pro_o + (size_t)b * C + 8 * fq;
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(pa + 32 * kt), a1 = *reinterpret_cast<const f32x4*>(pa + 32 * kt + 4);
            const f32x4 o0 = *reinterpret_cast<const f32x4*>(po + 32 * kt), o1 = *reinterpret_cast<const f32x4*>(po + 32 * kt + 4);
#pragma unroll
            for (int pt = 0; pt < P; ++pt) {
                const float* xs = g.x + (size_t)(m0 + wave * PTS + 16 * pt + fr) * C + 32 * kt + 8 * fq;
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs), x1 = *reinterpret_cast<const f32x4*>(xs + 4);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = x0[e] * a0[e] + o0[e];
                    v[4 + e] = x1[e] * a1[e] + o1[e];
                }
                u32x4 hi, lo;
                split8p(v, hi, lo);
                yh[kt][pt] = __builtin_bit_cast(bf16x8, hi);
                if (kt < NK - YL) yl[kt < NK - YL ? kt : 0][pt] = __builtin_bit_cast(bf16x8, lo);
                else *reinterpret_cast<u32x4*>(my_ylo + ((kt - (NK - YL)) * P + pt) * 1024) = lo;   // lane-private: no barrier needed
            }
            __builtin_amdgcn_sched_barrier(0);   // one K-step's loads in flight at a time: the registers belong to y
        }
    }

    MX_STAMP(1);
    f32x4 acc[NCT][P];
#pragma unroll
    for (int j = 0; j < NCT; ++j)
#pragma unroll
        for (int pt = 0; pt < P; ++pt) acc[j][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool has_act = g.act != 0;
    const int act_mode = g.act;
    const float neg_inv_2a2 = act_is_gauss(g.act) ? -1.0f / (2.0f * g.alpha[0] * g.alpha[0]) : 0.f;
    // fragment read offsets inside a slot: row (16-row tile base) + fr, chunk fq (hi) / 4 + fq (lo), swizzled by fr
    const int sw = swz(fr);
    const int off_hi = fr * 128 + ((fq ^ sw) * 16), off_lo = fr * 128 + (((4 + fq) ^ sw) * 16);

    // one slot: wait for this wave's pieces, barrier (everyone's pieces are in; everyone is done with the previous
    // slot), refill the slot just freed seven slots ahead
    auto enter_slot = [&](int sigma) -> const unsigned char* {
        const int after = total - 1 - sigma;         // slots issued after sigma and still allowed in flight (<= 6)
#ifndef MX_DIAG_NODMA
        wait_vm_slots<DPW>(after < AHEAD - 1 ? after : AHEAD - 1);
#endif
#ifndef MX_DIAG_NOBARRIER
        __builtin_amdgcn_s_barrier();
#endif
#ifndef MX_DIAG_NODMA
        if (sigma + AHEAD < total) issue(sigma + AHEAD);
#endif
        return lds + (sigma % RING) * SLOT;
    };

    for (int ch = 0; ch < NCH; ++ch) {
        if (ch == 1) MX_STAMP(2);
        if (ch == 2) MX_STAMP(3);
        f32x4 X[2][P];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pt = 0; pt < P; ++pt) X[j][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- X^T[32 units][points] = W0[units] y^T: S1 slots of 4 K-steps; slot rows = ktl * 32 + unit
#pragma unroll
        for (int s1 = 0; s1 < S1; ++s1) {
            const unsigned char* slot = enter_slot(ch * NS + s1);
            // one K-step (4 weight fragments + y's lo fragments when they live in LDS) per sub-step; the NEXT sub-step's
            // reads are issued before this one's MFMAs (two fragment sets; the compiler counts the lgkmcnt waits)
            bf16x8 wh[2][2], wl[2][2], ylk[2][P];
            auto load1 = [&](int kk, int f) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned char* rb = slot + (kk * 32 + 16 * j) * 128;
                    wh[f][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(rb + off_hi));
                    wl[f][j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(rb + off_lo));
                }
                const int kt = 4 * s1 + kk;
                if (kt >= NK - YL) {
#pragma unroll
                    for (int pt = 0; pt < P; ++pt)
                        ylk[f][pt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(my_ylo + ((kt - (NK - YL)) * P + pt) * 1024));
                }
            };
            load1(0, 0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int f = kk & 1, kt = 4 * s1 + kk;
                const bool ylo_now = kt >= NK - YL, ylo_next = kt + 1 >= NK - YL;
                // explicit software pipeline (hipcc would sink the reads to just before their use and wait lgkmcnt(0) every
                // few MFMAs): issue the next sub-step's reads, wait only for THIS sub-step's (LDS returns in order), and
                // "redefine" the fragments so that the compiler's own wait-count pass adds nothing in front of the MFMAs
                if (kk + 1 < 4) {
                    load1(kk + 1, f ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ylo_next) __builtin_amdgcn_s_waitcnt(waitcnt_imm(0x3F, 4 + P));
                    else __builtin_amdgcn_s_waitcnt(waitcnt_imm(0x3F, 4));
                } else {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0x3F, 0));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    asm volatile("" : "+v"(wh[f][j]));
                    asm volatile("" : "+v"(wl[f][j]));
                }
                if (ylo_now) {
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) asm volatile("" : "+v"(ylk[f][pt]));
                }
                // three terms, accumulators interleaved (consecutive MFMAs never share one)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) X[j][pt] = MX_MFMA(wl[f][j], yh[kt][pt], X[j][pt]);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pt = 0; pt < P; ++pt)
                        X[j][pt] = MX_MFMA(wh[f][j], kt < NK - YL ? yl[kt < NK - YL ? kt : 0][pt] : ylk[f][pt], X[j][pt]);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) X[j][pt] = MX_MFMA(wh[f][j], yh[kt][pt], X[j][pt]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- bias, activation, hi | lo split on the accumulator registers: lane (point fr, fq) holds units
        // 32 ch + 4 fq + r (tile 0) and 32 ch + 16 + 4 fq + r (tile 1) = the k set of its B fragment (image order)
        bf16x8 xh[P], xl[P];
        {
            const f32x4 bA = *reinterpret_cast<const f32x4*>(lb0 + 32 * ch + 4 * fq);
            const f32x4 bB = *reinterpret_cast<const f32x4*>(lb0 + 32 * ch + 16 + 4 * fq);
#pragma unroll
            for (int pt = 0; pt < P; ++pt) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = X[0][pt][e] + bA[e];
                    v[4 + e] = X[1][pt][e] + bB[e];
                }
                if (has_act) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = act_apply(v[e], neg_inv_2a2, act_mode);
                }
                u32x4 hi, lo;
                split8p(v, hi, lo);
                xh[pt] = __builtin_bit_cast(bf16x8, hi);
                xl[pt] = __builtin_bit_cast(bf16x8, lo);
            }
        }
        // ---- out^T[C][points] += W2[:, units] X^T: S2 slots of 8 channel tiles; slot rows = 16 ctl + channel row
#pragma unroll
        for (int s2 = 0; s2 < S2; ++s2) {
            const unsigned char* slot = enter_slot(ch * NS + S1 + s2);
            // two channel tiles (4 fragments) per sub-step, next sub-step's reads issued first
            bf16x8 wh[2][2], wl[2][2];
            auto load2 = [&](int qq, int f) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned char* rb = slot + (2 * qq + q) * 16 * 128;
                    wh[f][q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(rb + off_hi));
                    wl[f][q] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(rb + off_lo));
                }
            };
            load2(0, 0);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int f = qq & 1;
                if (qq + 1 < 4) {
                    load2(qq + 1, f ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0x3F, 4));
                } else {
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0x3F, 0));
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    asm volatile("" : "+v"(wh[f][q]));
                    asm volatile("" : "+v"(wl[f][q]));
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) {
                        f32x4& a = acc[8 * s2 + 2 * qq + q][pt];
                        a = MX_MFMA(wl[f][q], xh[pt], a);
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) {
                        f32x4& a = acc[8 * s2 + 2 * qq + q][pt];
                        a = MX_MFMA(wh[f][q], xl[pt], a);
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int pt = 0; pt < P; ++pt) {
                        f32x4& a = acc[8 * s2 + 2 * qq + q][pt];
                        a = MX_MFMA(wh[f][q], xh[pt], a);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    MX_STAMP(4);
    __builtin_amdgcn_s_barrier();   // every wave is done with the ring: its first bytes become the statistics scratch
    // ---- epilogue: x += out + b2.  Channel tile pair (2u, 2u+1): this lane holds channels 32 u + 8 fq .. + 7 of point
    // m0 + wave 16 P + 16 pt + fr (W2's rows are stored in that order)
#pragma unroll
    for (int u = 0; u < NCT / 2; ++u) {
        const int c0 = 32 * u + 8 * fq;
        const f32x4 b0v = *reinterpret_cast<const f32x4*>(lb2 + c0), b1v = *reinterpret_cast<const f32x4*>(lb2 + c0 + 4);
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
        for (int pt = 0; pt < P; ++pt) {
            float* xp = g.x + (size_t)(m0 + wave * PTS + 16 * pt + fr) * C + c0;
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(xp), r1 = *reinterpret_cast<const f32x4*>(xp + 4);
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = r0[e] + (acc[2 * u][pt][e] + b0v[e]);
                v[4 + e] = r1[e] + (acc[2 * u + 1][pt][e] + b1v[e]);
            }
            *reinterpret_cast<f32x4*>(xp) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(xp + 4) = f32x4{v[4], v[5], v[6], v[7]};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1[e] += v[e];
                s2[e] += v[e] * v[e];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (g.stats) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    s1[e] += __shfl_xor(s1[e], o, 64);
                    s2[e] += __shfl_xor(s2[e], o, 64);
                }
            }
            if (fr == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[(wave * 2 + 0) * C + c0 + e] = s1[e];
                    red[(wave * 2 + 1) * C + c0 + e] = s2[e];
                }
            }
        }
    }
    MX_STAMP(5);
    if (g.stats) {
        __syncthreads();
        for (int i = tid; i < 2 * C; i += NT) {
            const int which = i / C, c = i % C;
            float t = red[(0 * 2 + which) * C + c];
#pragma unroll
            for (int wv = 1; wv < NW; ++wv) t += red[(wv * 2 + which) * C + c];   // fixed order: deterministic
            g.stats[((size_t)blockIdx.x * 2 + which) * C + c] = t;
        }
    }
}

// ---- the weight stream image: one thread per (slot, row, 16-byte chunk position) ------------------------------------
// W0 slot (chunk ch, s1): row rho = ktl * 32 + unit: W0[32 ch + unit][32 (4 s1 + ktl) + 0..31] as [hi | lo].
// W2 slot (chunk ch, s2): row rho = 16 ctl + r of channel tile ct = 8 s2 + ctl: image row R = 16 ct + r holds channel
//   32 (R >> 5) + 8 (r >> 2) + 4 ((R >> 4) & 1) + (r & 3); its k position 8 g + e holds unit 32 ch + (e < 4 ? 4 g + e : 16 + 4 g + e - 4).
// Chunk position q of a row holds data chunk q ^ swz(rho) (rho & 15 decides), so the DMA copies slots linearly.
template <int NC>
__global__ void mlp_x3_stream_kernel(const float* __restrict__ W0, const float* __restrict__ W2, unsigned char* __restrict__ img) {
    constexpr int C = 128 * NC, WD = 2 * C, NCH = WD / 32, NS = 2 * NC;
    const size_t total = (size_t)NCH * NS * 128 * 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i & 7), rho = (int)((i >> 3) & 127);
        const int sigma = (int)(i >> 10), ch = sigma / NS, s = sigma % NS;
        const int d = q ^ swz(rho);                  // data chunk: 0..3 hi k 8 d .. , 4..7 lo k 8 (d - 4) ..
        const int kpos = 8 * (d & 3);
        float v[8];
        if (s < NC) {
            const int ktl = rho >> 5, unit = rho & 31;
            const float* src = W0 + (size_t)(32 * ch + unit) * C + 32 * (4 * s + ktl) + kpos;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = src[e];
        } else {
            const int ctl = rho >> 4, r = rho & 15, R = 16 * (8 * (s - NC) + ctl) + r;
            const int chan = 32 * (R >> 5) + 8 * (r >> 2) + 4 * ((R >> 4) & 1) + (r & 3);
            const int gq = d & 3;                    // k positions 8 gq + e
            const float* src = W2 + (size_t)chan * WD + 32 * ch;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = src[e < 4 ? 4 * gq + e : 16 + 4 * gq + e - 4];
        }
        u32x4 hi, lo;
        split8p(v, hi, lo);
        *reinterpret_cast<u32x4*>(img + (size_t)sigma * SLOT + rho * 128 + q * 16) = d < 4 ? hi : lo;
    }
}

template <int NC, int P, int YL, int RING, int NW>
int launch_t(const MlpX3Args& g, hipStream_t st) {
    constexpr int C = 128 * NC;
    constexpr int lds = RING * SLOT + (2 * C + C) * 4 + NW * YL * P * 1024;
    static_assert(lds <= 160 * 1024 && NW * 2 * C * 4 <= RING * SLOT, "LDS budget");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_x3_fused_kernel<NC, P, YL, RING, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
    }
    hipLaunchKernelGGL((mlp_x3_fused_kernel<NC, P, YL, RING, NW>), dim3(g.rows_total / (16 * P * NW)), dim3(64 * NW), lds, st, g);
    return (int)hipGetLastError();
}

}  // namespace

int mlp_x3_fused_row_tile(int C) { return C == 384 || C == 256 || C == 128 ? 128 : (C == 512 ? 64 : 0); }
bool mlp_x3_fused_supported(int C, int Wd, int rows_per_sample) {
    const int bm = mlp_x3_fused_row_tile(C);
    return bm && Wd == 2 * C && rows_per_sample % bm == 0 && (C == 384 || C == 512 || C == 256);
}
size_t mlp_x3_stream_bytes(int C) { return (size_t)4 * C * C * 4; }   // W0 and W2 (2 C x C each), 4 bytes per element

int mlp_x3_stream_launch(const float* W0, const float* W2, void* img, int C, hipStream_t st) {
    unsigned char* p = static_cast<unsigned char*>(img);
    switch (C) {
        case 256: hipLaunchKernelGGL(mlp_x3_stream_kernel<2>, dim3(512), dim3(256), 0, st, W0, W2, p); break;
        case 384: hipLaunchKernelGGL(mlp_x3_stream_kernel<3>, dim3(512), dim3(256), 0, st, W0, W2, p); break;
        case 512: hipLaunchKernelGGL(mlp_x3_stream_kernel<4>, dim3(512), dim3(256), 0, st, W0, W2, p); break;
        default: return -9;
    }
    return (int)hipGetLastError();
}

int mlp_x3_fused_launch(const MlpX3Args& g, int C, hipStream_t st) {
    if (!mlp_x3_fused_supported(C, 2 * C, g.rows_per_sample) || g.rows_total % mlp_x3_fused_row_tile(C)) return -9;
    switch (C) {
        case 256: return launch_t<2, 2, 0, 8, 4>(g, st);
#ifdef MX_P1
        case 384: return launch_t<3, 1, MX_YL, MX_RING, 8>(g, st);
#else
        case 384: return launch_t<3, 2, 8, 5, 4>(g, st);
#endif
        default: return launch_t<4, 1, 2, 8, 4>(g, st);
    }
}
