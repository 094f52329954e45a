// Shared pieces of the LDS-DMA GEMM kernels (gemm_f32_dma.hip: fp32 / split-bf16, gemm_f16_dma.hip: fp16): block ->
// tile mapping with the optional second output segment, and the epilogue (bias, GaussianActivation, residual,
// GroupNorm partials) through a wave-private LDS transpose with 16-byte nontemporal stores.
#pragma once
#include "common.h"
#include "h8_scales.h"
#include "kernels.h"

namespace dma {

constexpr int DBN = 128, DNT = 256;
constexpr int D_TP = 64 + 4;                        // epilogue transpose tile row stride
constexpr int D_EPI = 4 * 32 * D_TP + 4 * 2 * DBN;  // 4 wave sub-tiles (32 x 64) + column partials = 38 KiB

__device__ __forceinline__ void dma16(const void* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Counted waits as the s_waitcnt BUILTIN, not inline asm: the compiler's own wait-count scoreboard understands the
// builtin, so after wait_vm_lgkm0 it knows every earlier LDS read has returned and does not put a second
// s_waitcnt lgkmcnt(0) in front of the first MFMA that uses last step's fragments — which would also wait for the
// fragment reads just issued for the NEXT step and expose their whole latency every K-step (it did, with asm).
// gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14.
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 0xF) | (0x7 << 4) | ((lgkm & 0xF) << 8) | ((vm >> 4) << 14); }
template <int N>
__device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(waitcnt_imm(N, 0xF)); }
template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() { __builtin_amdgcn_s_waitcnt(waitcnt_imm(N, 0)); }

// Which tile this block computes.  Optional second output segment (two linears over the same A in one launch): whole
// column tiles belong to one segment (n_split % 128 == 0); columns below are relative to the segment.
struct Tile {
    int tilesM, ct, rt, b, m0, n0;
    int nseg0, nseg;          // first column of the tile inside its segment; columns of the segment
    const float* Wseg;
    const float* bias_seg;
    float* Cseg;
    int ldc_seg;
};

template <int BM>
__device__ __forceinline__ Tile tile_of_block(const GemmArgs& g) {
    Tile t;
    t.tilesM = (g.rows + BM - 1) / BM;
    const int tilesN = (g.Nout + DBN - 1) / DBN;
    const int nblk = g.B * t.tilesM * tilesN;
    const int v = xcd_remap(blockIdx.x, nblk);
    t.ct = v % tilesN;
    const int panel = v / tilesN;
    t.rt = panel % t.tilesM;
    t.b = panel / t.tilesM;
    t.m0 = t.rt * BM;
    t.n0 = t.ct * DBN;
    const bool seg2 = g.C2 != nullptr && t.n0 >= g.n_split;
    t.nseg0 = seg2 ? t.n0 - g.n_split : t.n0;
    t.nseg = g.C2 ? (seg2 ? g.Nout - g.n_split : g.n_split) : g.Nout;
    t.Wseg = seg2 ? g.W2 : g.W;
    t.bias_seg = seg2 ? g.bias2 : g.bias ? g.bias + (size_t)t.b * g.bias_bstride : nullptr;
    t.Cseg = seg2 ? g.C2 : g.C;
    t.ldc_seg = seg2 ? g.ldc2 : g.ldc;
    return t;
}

// acc[i][j]: the wave's 32 x 32 accumulator tiles (C/D layout), wave (wm, wn) of a WMN x (4 / WMN) wave grid.
// Must be entered by all 256 threads with the operand ring dead (it reuses the LDS from offset 0); the caller's
// barrier before it is the only block-wide one the store phase needs (a second one guards the statistics reduce).
// C16: the output is stored as fp16 (round to nearest even; an intermediate its consumer would round anyway) — no
// residual, no statistics in that form; ldc counts fp16 elements.
template <int TMW, int TNW, int WMN, bool C16, bool ACTBWD>
__device__ __forceinline__ void epilogue_t(const GemmArgs& g, const Tile& t, f32x16 (&acc)[TMW][TNW], float* smem,
                                           int wave, int lane, int wm, int wn) {
    const int tid = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int b = t.b, rt = t.rt, m0 = t.m0, n0 = t.n0, nseg0 = t.nseg0, nseg = t.nseg, tilesM = t.tilesM;
    const float* bias_seg = t.bias_seg;
    float* Cseg = t.Cseg;
    const int ldc_seg = t.ldc_seg;
    const bool has_act = g.act != 0;
    const int act_mode = g.act;
    const bool mulg = ACTBWD && g.mul_u != nullptr && act_is_gauss(g.mul_kind);
    const float alpha0 = (act_is_gauss(g.act) || mulg) ? g.alpha[0] : 1.f;
    const float neg_inv_2a2 = (act_is_gauss(g.act) || mulg) ? -1.0f / (2.0f * alpha0 * alpha0) : 0.f;
    const float inv_a2 = 1.0f / (alpha0 * alpha0);
    const float* Ub = (ACTBWD && g.mul_u) ? g.mul_u + (size_t)b * g.rows * ldc_seg : nullptr;   // the pre-activation, laid out like C
    float ga = 0.f;   // this thread's share of sum (A W^T) d act / d alpha
    float* Cb = Cseg + (size_t)b * g.rows * ldc_seg;
    const float* Rb = g.residual ? g.residual + (size_t)b * g.rows * g.ldr : nullptr;
    const float* Xb = (ACTBWD && g.dot_x) ? g.dot_x + (size_t)b * g.rows * ldc_seg : nullptr;   // second statistic = sum C * dot_x
    float* Tt = smem + wave * 32 * D_TP;
    float* red = smem + 4 * 32 * D_TP;
    const int lr = lane >> 4, c4 = lane & 15;   // 16 lanes per 64-float row, 4 rows per wave-instruction
    constexpr int NJH = TNW / 2;                // 64-column halves of the wave tile
    f32x4 s1[NJH], s2[NJH];
#pragma unroll
    for (int jh = 0; jh < NJH; ++jh) {
        s1[jh] = f32x4{0.f, 0.f, 0.f, 0.f};
        s2[jh] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the wave's 32 x 64 sub-tiles, one after the other through the same wave-private LDS tile
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
#pragma unroll
        for (int jh = 0; jh < NJH; ++jh) {
            const int ncol0 = nseg0 + (wn * TNW + 2 * jh) * 32;
            const int n = ncol0 + c4 * 4;
            const bool nok = n < nseg;
            const int nc = nok ? n : 0;
            const int mrow0 = m0 + (wm * TMW + i) * 32;
            // the residual rows of this sub-tile: all eight 16-byte loads in flight at once, issued before the
            // transpose below so their HBM latency is paid once per sub-tile and partly under it
            // (ACTBWD) residual AND dot_x (round 6: the second dX product of kv_proj | q_proj adds the first and leaves the AdaGN backward's
            // partials): the x rows are loaded into rres once the residual rows are consumed, the statistics run a second pass over the
            // tile's final values, which the first pass writes back into the wave's transpose tile
            const bool res_and_dot = ACTBWD && Rb != nullptr && Xb != nullptr;
            f32x4 rres[8];
            if (Rb) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = min(mrow0 + it * 4 + lr, g.rows - 1);
                    rres[it] = GECCO_NT_LOAD(reinterpret_cast<const f32x4*>(Rb + (size_t)m * g.ldr + nc));
                }
            } else if (ACTBWD && Xb) {   // ... or the rows of the tensor whose AdaGN backward wants {sum dy, sum dy x}
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = min(mrow0 + it * 4 + lr, g.rows - 1);
                    rres[it] = GECCO_NT_LOAD(reinterpret_cast<const f32x4*>(Xb + (size_t)m * ldc_seg + nc));
                }
            } else if (ACTBWD && Ub) {   // the same eight loads in flight serve the pre-activation rows of the backward epilogue
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = min(mrow0 + it * 4 + lr, g.rows - 1);
                    rres[it] = GECCO_NT_LOAD(reinterpret_cast<const f32x4*>(Ub + (size_t)m * ldc_seg + nc));
                }
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int nn = ncol0 + jj * 32 + r;
                const float bias = bias_seg ? bias_seg[nn < nseg ? nn : nseg - 1] : 0.f;
                f32x16 val = acc[i][2 * jh + jj];
#pragma unroll
                for (int e = 0; e < 16; ++e) val[e] += bias;
                if (has_act && !(ACTBWD && g.pre_out)) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) val[e] = act_apply(val[e], neg_inv_2a2, act_mode);
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) Tt[mfma_row(e, h) * D_TP + jj * 32 + r] = val[e];
            }
            // the transpose tile is wave-private: the wave's own LDS operations execute in order, no barrier needed
            if (!C16 && g.c_img) {
                // tiled split image (GemmArgs::c_img): 8 lanes per 64-column row of the sub-tile, 8 rows per instruction; a lane's 8
                // consecutive k are 16 bytes of the hi plane and 16 of the lo plane of one 8 KiB block
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                const int lr8 = lane >> 3, c8 = lane & 7;
                const int n8 = ncol0 + c8 * 8;
                if (g.c_img == 2) {
                    // h8 activation image (gemm_h8_areg.hip): per (sample, 128-row tile, 64-column group) 6144 floats — fp16 hi
                    // fragments [32-row tile][sub][c][lane], then fp8(2^11 lo) halves [32-row tile][t][lane]; a lane's 8
                    // consecutive columns are one 16-byte hi chunk and 8 lo bytes (Nout % 64 == 0)
                    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int m = mrow0 + it * 8 + lr8;
                        f32x4 v0 = *reinterpret_cast<const f32x4*>(Tt + (it * 8 + lr8) * D_TP + c8 * 8);
                        f32x4 v1 = *reinterpret_cast<const f32x4*>(Tt + (it * 8 + lr8) * D_TP + c8 * 8 + 4);
                        f16x8 hv;
                        float lo[8];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v0[e] = h8_clamp(v0[e]);       // h8_scales.h: an h8 operand is finite in every term
                            v1[e] = h8_clamp(v1[e]);
                            hv[e] = (_Float16)v0[e];       // v0 / v1 come from LDS: one fp32 value for the hi rounding and the lo difference
                            hv[4 + e] = (_Float16)v1[e];
                            lo[e] = __builtin_fminf(__builtin_fmaxf((v0[e] - (float)hv[e]) * H8_AL_SCALE, -448.f), 448.f);
                            lo[4 + e] = __builtin_fminf(__builtin_fmaxf((v1[e] - (float)hv[4 + e]) * H8_AL_SCALE, -448.f), 448.f);
                        }
                        int p0 = 0, p1 = 0;
                        p0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], p0, false);
                        p0 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], p0, true);
                        p1 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[4], lo[5], p1, false);
                        p1 = __builtin_amdgcn_cvt_pk_fp8_f32(lo[6], lo[7], p1, true);
                        if (n8 < nseg && m < g.rows) {
                            const int kk = n8 & 63, ml = m & 127, rt = ml >> 5, rr = ml & 31;
                            const int sub = kk >> 5, hh = (kk >> 4) & 1, cc = (kk >> 3) & 1;
                            float* blk = Cseg + (((size_t)b * ((g.rows + 127) >> 7) + (m >> 7)) * (g.Nout >> 6) + (n8 >> 6)) * 6144;
                            GECCO_NT_STORE(__builtin_bit_cast(u32x4, hv), reinterpret_cast<u32x4*>(blk + rt * 1024 + (2 * sub + cc) * 256 + (32 * hh + rr) * 4));
                            GECCO_NT_STORE((u32x2{(unsigned)p0, (unsigned)p1}), reinterpret_cast<u32x2*>(blk + 4096 + rt * 512 + sub * 256 + (32 * hh + rr) * 4 + 2 * cc));
                        }
                    }
                    continue;
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int m = mrow0 + it * 8 + lr8;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(Tt + (it * 8 + lr8) * D_TP + c8 * 8);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(Tt + (it * 8 + lr8) * D_TP + c8 * 8 + 4);
                    u32x4 hi;
                    bf16x8 lo;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float a = p < 2 ? v0[2 * p] : v1[2 * p - 4], c = p < 2 ? v0[2 * p + 1] : v1[2 * p - 3];
                        const unsigned ua = __float_as_uint(a), uc = __float_as_uint(c);
                        hi[p] = __builtin_amdgcn_perm(uc, ua, 0x07060302u);
                        lo[2 * p] = (__bf16)(a - __uint_as_float(ua & 0xFFFF0000u));
                        lo[2 * p + 1] = (__bf16)(c - __uint_as_float(uc & 0xFFFF0000u));
                    }
                    if (n8 < nseg && m < g.rows) {
                        const int kt = n8 >> 4, ml = m & 127;
                        const size_t blk = ((size_t)b * ((g.rows + 127) >> 7) + (m >> 7)) * (g.Nout >> 4) + kt;
                        unsigned short* dst = reinterpret_cast<unsigned short*>(Cseg) + blk * 4096 + ml * 16 +
                                              ((((n8 & 15) >> 3) ^ ((ml >> 3) & 1)) << 3);
                        GECCO_NT_STORE(hi, reinterpret_cast<u32x4*>(dst));
                        GECCO_NT_STORE(__builtin_bit_cast(u32x4, lo), reinterpret_cast<u32x4*>(dst + 2048));
                    }
                }
                continue;
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int m = mrow0 + it * 4 + lr;
                f32x4 v4 = *reinterpret_cast<const f32x4*>(Tt + (it * 4 + lr) * D_TP + c4 * 4);
                const bool ok = nok && m < g.rows;
                if (ACTBWD && g.pre_out) {   // keep u = A W^T + bias, then activate the row piece
                    if (ok) GECCO_NT_STORE(v4, reinterpret_cast<f32x4*>(g.pre_out + ((size_t)b * g.rows + m) * ldc_seg + n));
                    if (has_act) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v4[q] = act_apply(v4[q], neg_inv_2a2, act_mode);
                    }
                }
                if (Rb) {
                    v4 += rres[it];
                    if (res_and_dot) {   // (block-uniform)
                        *reinterpret_cast<f32x4*>(Tt + (it * 4 + lr) * D_TP + c4 * 4) = v4;   // own piece: no other lane touches it
                        rres[it] = GECCO_NT_LOAD(reinterpret_cast<const f32x4*>(Xb + (size_t)min(m, g.rows - 1) * ldc_seg + nc));
                    }
                } else if (ACTBWD && Ub) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float da;
                        const float f = act_prime(rres[it][q], neg_inv_2a2, inv_a2, g.mul_kind, da);
                        if (ok) ga += v4[q] * da;
                        v4[q] *= f;
                    }
                }
                if (C16) {
                    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    f16x4 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[e] = (_Float16)v4[e];
                    if (ok)
                        GECCO_NT_STORE(__builtin_bit_cast(u32x2, hv),
                                                    reinterpret_cast<u32x2*>(reinterpret_cast<_Float16*>(Cseg) +
                                                                             ((size_t)b * g.rows + m) * ldc_seg + n));
                } else if (ok) {
                    GECCO_NT_STORE(v4, reinterpret_cast<f32x4*>(Cb + (size_t)m * ldc_seg + n));
                }
                if (res_and_dot) continue;   // statistics: the pass below
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 vz = ok ? v4 : z;
                s1[jh] += vz;
                // an explicit fma: left to -ffp-contract the compiler fuses this in some instantiations of the template and
                // not in others, and the kernels that share this epilogue stop agreeing to the bit on the statistics
                if (ACTBWD && Xb) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) s2[jh][q] = __builtin_fmaf(vz[q], rres[it][q], s2[jh][q]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) s2[jh][q] = __builtin_fmaf(vz[q], vz[q], s2[jh][q]);
                }
            }
            if (res_and_dot) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int m = mrow0 + it * 4 + lr;
                    const f32x4 v4 = *reinterpret_cast<const f32x4*>(Tt + (it * 4 + lr) * D_TP + c4 * 4);
                    const bool ok = nok && m < g.rows;
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 vz = ok ? v4 : z;
                    s1[jh] += vz;
#pragma unroll
                    for (int q = 0; q < 4; ++q) s2[jh][q] = __builtin_fmaf(vz[q], rres[it][q], s2[jh][q]);
                }
            }
        }
    }
    if (ACTBWD && g.agrad && mulg) {   // (block-uniform) lanes, then the four waves, in a fixed order
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) ga += __shfl_xor(ga, o, 64);
        __syncthreads();                       // every wave is done with its transpose tile: red is free
        if (lane == 0) red[wave] = ga;
        __syncthreads();
        if (tid == 0) {
            // one slot per 64 rows (the smallest row tile: 65 .. 127 rows are TWO 64-row tiles — indexed per 128 rows they shared a
            // slot and the last writer won); 128- / 256-row tiles write every second / fourth slot, the caller zeroes them all
            const int T64 = (g.rows + 63) / 64, tn = (g.Nout + DBN - 1) / DBN;
            g.agrad[((size_t)b * T64 + (m0 >> 6)) * tn + (n0 / DBN)] = (((red[0] + red[1]) + red[2]) + red[3]) / alpha0;
        }
    }
    if (g.stats) {
#pragma unroll
        for (int jh = 0; jh < NJH; ++jh) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s1[jh][q] += __shfl_xor(s1[jh][q], 16, 64);
                s1[jh][q] += __shfl_xor(s1[jh][q], 32, 64);
                s2[jh][q] += __shfl_xor(s2[jh][q], 16, 64);
                s2[jh][q] += __shfl_xor(s2[jh][q], 32, 64);
            }
            if (lane < 16) {
                const int cl = (wn * TNW + 2 * jh) * 32 + c4 * 4;
                *reinterpret_cast<f32x4*>(red + (wm * 2 + 0) * DBN + cl) = s1[jh];
                *reinterpret_cast<f32x4*>(red + (wm * 2 + 1) * DBN + cl) = s2[jh];
            }
        }
        __syncthreads();
        if (TMW * WMN == 8) {
            // 256-row tile: the statistics keep their 128-row granularity (consumers count partials per 128 rows): waves
            // {0, 1} own the first half, {2, 3} the second
            const int T128 = (g.rows + 127) / 128;
            for (int c = tid; c < 4 * DBN; c += DNT) {
                const int half = c / (2 * DBN), which = (c / DBN) & 1, cl = c % DBN, nn = n0 + cl;
                if (nn < g.Nout && 2 * rt + half < T128)
                    g.stats[(((size_t)b * T128 + 2 * rt + half) * 2 + which) * g.Nout + nn] =
                        red[((2 * half) * 2 + which) * DBN + cl] + red[((2 * half + 1) * 2 + which) * DBN + cl];
            }
            return;
        }
        for (int c = tid; c < 2 * DBN; c += DNT) {
            const int which = c / DBN, cl = c % DBN, nn = n0 + cl;
            if (nn < g.Nout) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < WMN; ++w) t += red[(w * 2 + which) * DBN + cl];
                g.stats[(((size_t)b * tilesM + rt) * 2 + which) * g.Nout + nn] = t;
            }
        }
    }
}

// The forward epilogue; the training path's activation-backward products (GemmArgs::mul_u) run a separate KERNEL
// instantiation (gemm_f32_dma.hip: ACTBWD) with the second form, so the inference kernels are the code they were.
template <int TMW, int TNW, int WMN, bool C16 = false>
__device__ __forceinline__ void epilogue(const GemmArgs& g, const Tile& t, f32x16 (&acc)[TMW][TNW], float* smem,
                                         int wave, int lane, int wm, int wn) {
    epilogue_t<TMW, TNW, WMN, C16, false>(g, t, acc, smem, wave, lane, wm, wn);
}

}  // namespace dma
