// HBM-bound pointwise / reduction kernels of the GECCO denoiser (gfx950): GroupNorm statistics,
// AdaGN coefficient finalisation, EDM preconditioning, lift (3 -> d) and lower (d -> 3).
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

namespace {

constexpr int STATS_ROWS = 128;  // row-tile height of the stand-alone / lift statistics producers

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- column statistics of a (B, rows, C) tensor: stats[b][tile][{sum,sumsq}][c]
// Used where no producer epilogue exists (unit-level AdaGN, cached-mode first touch).
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                        int rows, int C, int T) {
    // 16-byte column chunks; the 256 threads cover C/4 chunks x RL row lanes, lane rl sums rows m0 + rl, + RL, ..;
    // the RL partials are combined through LDS in lane order (deterministic)
    __shared__ f32x4 red[2][256];
    const int tile = blockIdx.x % T, b = blockIdx.x / T;
    const int m0 = tile * STATS_ROWS, m1 = min(rows, m0 + STATS_ROWS);
    const float* xb = x + (size_t)b * rows * C;
    if (C % 4 == 0 && C / 4 <= 256) {
        const int c4n = C / 4, RL = 256 / c4n;
        const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        if (rl < RL) {
            for (int m = m0 + rl; m < m1; m += RL) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (size_t)m * C + c4 * 4);
                s1 += v;
                s2 += v * v;
            }
            red[0][threadIdx.x] = s1;
            red[1][threadIdx.x] = s2;
        }
        __syncthreads();
        if (rl == 0) {
            for (int q = 1; q < RL; ++q) {
                s1 += red[0][q * c4n + c4];
                s2 += red[1][q * c4n + c4];
            }
            *reinterpret_cast<f32x4*>(stats + (((size_t)b * T + tile) * 2 + 0) * C + c4 * 4) = s1;
            *reinterpret_cast<f32x4*>(stats + (((size_t)b * T + tile) * 2 + 1) * C + c4 * 4) = s2;
        }
        return;
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s1 = 0.f, s2 = 0.f;
        for (int m = m0; m < m1; ++m) {
            const float v = xb[(size_t)m * C + c];
            s1 += v;
            s2 += v * v;
        }
        stats[(((size_t)b * T + tile) * 2 + 0) * C + c] = s1;
        stats[(((size_t)b * T + tile) * 2 + 1) * C + c] = s2;
    }
}

// ---- AdaGN coefficients (reference models/normalization.py:36-44 folded into y = a*x + o):
//   mean/var per (b, group) over rows * C/G elements (biased), rstd = 1/sqrt(var + eps)
//   s = t . scale_w[c] + scale_b[c],  z = t . bias_w[c] + bias_b[c]
//   a[b,c] = s * rstd,  o[b,c] = z - s * mean * rstd
// With null scale/bias weights: plain GroupNorm(affine=False) (GroupNormBNC, models/ray.py:20-30).
// Partial sums are fp32 per tile, combined in fp64 (E[x^2] - mean^2 cancellation stays < 1e-7).
__global__ __launch_bounds__(1024) void adagn_coeffs_kernel(const float* __restrict__ stats, int T, int rows,
                                                            const float* __restrict__ t, int ctx_dim,
                                                            const float* __restrict__ scale_w,
                                                            const float* __restrict__ scale_b,
                                                            const float* __restrict__ bias_w,
                                                            const float* __restrict__ bias_b, float* __restrict__ a,
                                                            float* __restrict__ o, int C, int G, float eps, int nsl, int Cp) {
    // Cp channels (whole groups) per block: blockIdx.y = the part.  One part per sample (Cp = C) where the batch alone gives the chip
    // enough blocks; a cached upsample (8 samples x 128 tiles) cuts the channels into parts so that 64 blocks with 8 slices each walk
    // the partials instead of 8 blocks with 2 (12.8 -> ~5 us per launch, 12 launches per evaluation)
    extern __shared__ double dsm[];  // [2][Cp] column sums, [2][Gp] mean / rstd, [nsl][2][Cp] slice sums
    const int cpg = C / G, Gp = Cp / cpg;
    double* cs = dsm;
    double* gm = dsm + 2 * Cp;
    double* ps = gm + 2 * Gp;
    const int b = blockIdx.x, c0 = blockIdx.y * Cp;
    // the T tile partials of a column are cut into nsl slices summed by different threads (a cached upsample has 128 tiles per
    // sample and 8 samples: one thread per column walked them in 16 dependent round trips), 16 loads in flight per round
    // trip; slices are combined in slice order: a fixed summation order for a given (T, nsl)
    const int per = (T + nsl - 1) / nsl;
    for (int i = threadIdx.x; i < nsl * Cp; i += blockDim.x) {
        const int sl = i / Cp, cl = i % Cp, c = c0 + cl;
        const int k0 = sl * per, k1 = min(T, k0 + per);
        double s1 = 0.0, s2 = 0.0;
        int k = k0;
        for (; k + 16 <= k1; k += 16) {
            float p1[16], p2[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                p1[u] = stats[(((size_t)b * T + k + u) * 2 + 0) * C + c];
                p2[u] = stats[(((size_t)b * T + k + u) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                s1 += (double)p1[u];
                s2 += (double)p2[u];
            }
        }
        for (; k + 8 <= k1; k += 8) {
            float p1[8], p2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                p1[u] = stats[(((size_t)b * T + k + u) * 2 + 0) * C + c];
                p2[u] = stats[(((size_t)b * T + k + u) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s1 += (double)p1[u];
                s2 += (double)p2[u];
            }
        }
        for (; k < k1; ++k) {
            s1 += (double)stats[(((size_t)b * T + k) * 2 + 0) * C + c];
            s2 += (double)stats[(((size_t)b * T + k) * 2 + 1) * C + c];
        }
        ps[(sl * 2 + 0) * Cp + cl] = s1;
        ps[(sl * 2 + 1) * Cp + cl] = s2;
    }
    __syncthreads();
    for (int cl = threadIdx.x; cl < Cp; cl += blockDim.x) {
        double s1 = 0.0, s2 = 0.0;
        for (int sl = 0; sl < nsl; ++sl) {
            s1 += ps[(sl * 2 + 0) * Cp + cl];
            s2 += ps[(sl * 2 + 1) * Cp + cl];
        }
        cs[cl] = s1;
        cs[Cp + cl] = s2;
    }
    __syncthreads();
    for (int g = threadIdx.x; g < Gp; g += blockDim.x) {
        double s1 = 0.0, s2 = 0.0;
        for (int cl = g * cpg; cl < (g + 1) * cpg; ++cl) {
            s1 += cs[cl];
            s2 += cs[Cp + cl];
        }
        const double n = (double)rows * cpg;
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        gm[g] = mean;
        gm[Gp + g] = 1.0 / sqrt(var + (double)eps);
    }
    __syncthreads();
    for (int cl = threadIdx.x; cl < Cp; cl += blockDim.x) {
        const int g = cl / cpg, c = c0 + cl;
        const float mean = (float)gm[g], rstd = (float)gm[Gp + g];
        float s = 1.f, z = 0.f;
        if (scale_w) {
            s = scale_b[c];
            z = bias_b[c];
            for (int j = 0; j < ctx_dim; ++j) {
                const float tj = t[(size_t)b * ctx_dim + j];
                s += tj * scale_w[(size_t)c * ctx_dim + j];
                z += tj * bias_w[(size_t)c * ctx_dim + j];
            }
        }
        a[(size_t)b * C + c] = s * rstd;
        o[(size_t)b * C + c] = z - s * mean * rstd;
    }
}

// ---- y[b,m,c] = a[b,c] * x[b,m,c] + o[b,c]   (stand-alone AdaGN apply; inducer states h)
__global__ __launch_bounds__(256) void affine_apply_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                           const float* __restrict__ o, float* __restrict__ y,
                                                           size_t total4, int rowsC4, int C4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / rowsC4;
        const int c4 = (int)(i % C4);
        const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 av = reinterpret_cast<const f32x4*>(a)[b * C4 + c4];
        const f32x4 ov = reinterpret_cast<const f32x4*>(o)[b * C4 + c4];
        reinterpret_cast<f32x4*>(y)[i] = xv * av + ov;
    }
}

// ---- y16[b,m,c] = fp16(a[b,c] * x[b,m,c] + o[b,c]): the AdaGN apply of the fp16 mode, one fma and one rounding per
// element — the value the fp16 GEMM's prologue would have formed on the fragment, stored once as its fp16 A operand
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void affine_cast_f16_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                              const float* __restrict__ o, _Float16* __restrict__ y,
                                                              size_t total8, int rowsC8, int C8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total8; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / rowsC8;
        const int c8 = (int)(i % C8);
        const f32x4 x0 = GECCO_NT_LOAD(reinterpret_cast<const f32x4*>(x) + 2 * i);
        const f32x4 x1 = GECCO_NT_LOAD(reinterpret_cast<const f32x4*>(x) + 2 * i + 1);
        const f32x4 a0 = reinterpret_cast<const f32x4*>(a)[(b * C8 + c8) * 2], a1 = reinterpret_cast<const f32x4*>(a)[(b * C8 + c8) * 2 + 1];
        const f32x4 o0 = reinterpret_cast<const f32x4*>(o)[(b * C8 + c8) * 2], o1 = reinterpret_cast<const f32x4*>(o)[(b * C8 + c8) * 2 + 1];
        f16x8_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // two roundings, as everywhere else on the path (fp32 fma, then fp32 -> fp16): the empty asm keeps the
            // compiler from fusing them into v_fma_mixlo_f16, which rounds once and differs on double-rounding ties
            float f0 = __builtin_fmaf(x0[e], a0[e], o0[e]), f1 = __builtin_fmaf(x1[e], a1[e], o1[e]);
            asm volatile("" : "+v"(f0), "+v"(f1));
            v[e] = (_Float16)f0;
            v[4 + e] = (_Float16)f1;
        }
        reinterpret_cast<f16x8_t*>(y)[i] = v;
    }
}

// ---- EDM preconditioning coefficients (reference diffusion.py:46-51):
// coef[4b .. 4b+3] = {c_skip, c_out, c_in, c_noise}; coef[4B + b] = c_noise again, packed (the AdaGN `t`).
__global__ void edm_coeffs_kernel(const float* __restrict__ sigma, float sd, float* __restrict__ coef, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float s = sigma[b];
    const float q = s * s + sd * sd;
    coef[4 * b + 0] = sd * sd / q;
    coef[4 * b + 1] = s * sd / sqrtf(q);
    coef[4 * b + 2] = 1.0f / sqrtf(sd * sd + s * s);
    coef[4 * b + 3] = logf(s) / 4.0f;
    coef[4 * B + b] = logf(s) / 4.0f;
}

// ---- lift: out[b,m,:] = (c_in[b] * x[b,m,:3]) @ W^T + bias   (reference linear_lift.py:44 with
// diffusion.py:54; also RayNetwork.xyz_embed, models/ray.py:99) + column statistics of the result.
// One block = one (sample, 128-row tile); thread = channel (coalesced 4-byte stores along c).
__global__ __launch_bounds__(256) void lift_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                   const float* __restrict__ W, const float* __restrict__ bias,
                                                   float* __restrict__ out, float* __restrict__ stats, int N, int C,
                                                   int T) {
    __shared__ float xs[STATS_ROWS * 3];
    const int tile = blockIdx.x % T, b = blockIdx.x / T;
    const int m0 = tile * STATS_ROWS, m1 = min(N, m0 + STATS_ROWS);
    const float cin = coef ? coef[4 * b + 2] : 1.0f;
    for (int i = threadIdx.x; i < (m1 - m0) * 3; i += blockDim.x) xs[i] = cin * x[((size_t)b * N + m0) * 3 + i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float w0 = W[c * 3 + 0], w1 = W[c * 3 + 1], w2 = W[c * 3 + 2], bb = bias ? bias[c] : 0.f;
        float s1 = 0.f, s2 = 0.f;
        for (int m = 0; m < m1 - m0; ++m) {
            // same association as F.linear's dot product order: ((x0*w0 + x1*w1) + x2*w2) + b
            const float v = xs[m * 3 + 0] * w0 + xs[m * 3 + 1] * w1 + xs[m * 3 + 2] * w2 + bb;
            out[((size_t)b * N + m0 + m) * C + c] = v;
            s1 += v;
            s2 += v * v;
        }
        if (stats) {
            stats[(((size_t)b * T + tile) * 2 + 0) * C + c] = s1;
            stats[(((size_t)b * T + tile) * 2 + 1) * C + c] = s2;
        }
    }
}

// ---- lower + EDM combine: one wave per point.
//   LinearLift: F = Linear(d->3)(LayerNorm_d(feat))                 (reference linear_lift.py:25-29,46)
//   RayNetwork: F = Linear(d->3)(gn_a[b,:] * feat + gn_o[b,:])      (GroupNormBNC(16), models/ray.py:56-59,120)
//   D = c_skip * x + c_out * F                                      (diffusion.py:57)
__global__ __launch_bounds__(256) void lower_edm_kernel(const float* __restrict__ feat, const float* __restrict__ x,
                                                        const float* __restrict__ coef, const float* __restrict__ W,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ gn_a,
                                                        const float* __restrict__ gn_o, float* __restrict__ out,
                                                        float* __restrict__ raw, int B, int N, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (size_t)B * N) return;
    const int b = (int)(row / N);
    const float* f = feat + row * C;
    float mean = 0.f, rstd = 1.f;
    if (!gn_a) {
        float s1 = 0.f;
        for (int c = lane; c < C; c += 64) s1 += f[c];
        mean = wave_sum(s1) / C;
        float s2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float d = f[c] - mean;
            s2 += d * d;
        }
        rstd = rsqrtf(wave_sum(s2) / C + eps);
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = gn_a ? f[c] * gn_a[(size_t)b * C + c] + gn_o[(size_t)b * C + c] : (f[c] - mean) * rstd;
        a0 += v * W[c];
        a1 += v * W[C + c];
        a2 += v * W[2 * C + c];
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if (lane < 3) {
        const float Fv = (lane == 0 ? a0 : lane == 1 ? a1 : a2) + bias[lane];
        if (raw) raw[row * 3 + lane] = Fv;
        if (out) {
            const float cs = coef ? coef[4 * b + 0] : 0.f, co = coef ? coef[4 * b + 1] : 1.f;
            out[row * 3 + lane] = coef ? cs * x[row * 3 + lane] + co * Fv : Fv;
        }
    }
}

// Same contract, 16 lanes per point (4 points per wave): the row sits in registers as CPL f32x4 chunks per lane, so
// it is read once with 16-byte loads and the LayerNorm mean / variance / projection passes never touch memory again.
template <int CPL>
__global__ __launch_bounds__(256) void lower_edm_v4_kernel(const float* __restrict__ feat, const float* __restrict__ x,
                                                           const float* __restrict__ coef, const float* __restrict__ W,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ gn_a,
                                                           const float* __restrict__ gn_o, float* __restrict__ out,
                                                           float* __restrict__ raw, int B, int N, int C, float eps) {
    const int sub = threadIdx.x & 15;
    const size_t rows = (size_t)B * N;
    const size_t row_raw = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = row_raw < rows;
    const size_t row = live ? row_raw : rows - 1;   // idle groups shadow the last row: the shuffles stay full-width
    const int b = (int)(row / N);
    const int nch = C >> 2;
    auto gsum = [](float v) {
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        return v;
    };
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int ch = sub + 16 * i;
        // default-policy loads: the rows were written by the previous kernel and part of them is still in the Infinity
        // Cache (nontemporal loads measured 0.5 % slower end to end)
        v[i] = ch < nch ? *(reinterpret_cast<const f32x4*>(feat + row * C) + ch) : z;
    }
    if (gn_a) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int ch = sub + 16 * i;
            if (ch < nch)
                v[i] = v[i] * reinterpret_cast<const f32x4*>(gn_a + (size_t)b * C)[ch] +
                       reinterpret_cast<const f32x4*>(gn_o + (size_t)b * C)[ch];
        }
    } else {
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) s1 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        const float mean = gsum(s1) / C;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int ch = sub + 16 * i;
            const f32x4 d = ch < nch ? v[i] - mean : z;
            v[i] = d;
            s2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
        const float rstd = rsqrtf(gsum(s2) / C + eps);
#pragma unroll
        for (int i = 0; i < CPL; ++i) v[i] = v[i] * rstd;
    }
    float a[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int ch = sub + 16 * i;
        if (ch < nch) {
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const f32x4 w = reinterpret_cast<const f32x4*>(W + (size_t)o * C)[ch];
                a[o] += (v[i][0] * w[0] + v[i][1] * w[1]) + (v[i][2] * w[2] + v[i][3] * w[3]);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) a[o] = gsum(a[o]);
    if (live && sub < 3) {
        const float Fv = (sub == 0 ? a[0] : sub == 1 ? a[1] : a[2]) + bias[sub];
        if (raw) raw[row * 3 + sub] = Fv;
        if (out) {
            const float cs = coef ? coef[4 * b + 0] : 0.f, co = coef ? coef[4 * b + 1] : 1.f;
            out[row * 3 + sub] = coef ? cs * x[row * 3 + sub] + co * Fv : Fv;
        }
    }
}

// The same arithmetic, RPG rows per 16-lane group: the block stages W (and the block's sample's GroupNorm coefficients) in LDS once
// per 16 * RPG rows and each row costs its CPL 16-byte loads of `feat` only — in the one-row form every row also issued 3 CPL loads
// of W (+ 2 CPL of the coefficients): 24 - 36 vector-memory instructions per thread for 6 that reach HBM, and the texture path, not
// HBM, set the time (72 us for 201 MB at C2).  The next row's chunks are loaded while this one reduces.
template <int CPL, int RPG>
__global__ __launch_bounds__(256) void lower_edm_v5_kernel(const float* __restrict__ feat, const float* __restrict__ x,
                                                           const float* __restrict__ coef, const float* __restrict__ W,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ gn_a,
                                                           const float* __restrict__ gn_o, float* __restrict__ out,
                                                           float* __restrict__ raw, int B, int N, int C, float eps) {
    extern __shared__ __attribute__((aligned(16))) float lw[];   // W [3][C] | gn_a [C] | gn_o [C] of sample b0
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const size_t rows = (size_t)B * N;
    const size_t row0 = (size_t)blockIdx.x * 16 * RPG;
    const int b0 = (int)(row0 / N);
    const int nch = C >> 2;
    for (int i = threadIdx.x; i < 3 * nch; i += 256) reinterpret_cast<f32x4*>(lw)[i] = reinterpret_cast<const f32x4*>(W)[i];
    if (gn_a)
        for (int i = threadIdx.x; i < nch; i += 256) {
            reinterpret_cast<f32x4*>(lw + 3 * C)[i] = reinterpret_cast<const f32x4*>(gn_a + (size_t)b0 * C)[i];
            reinterpret_cast<f32x4*>(lw + 4 * C)[i] = reinterpret_cast<const f32x4*>(gn_o + (size_t)b0 * C)[i];
        }
    __syncthreads();
    auto gsum = [](float v) {
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 2, 64);
        v += __shfl_xor(v, 1, 64);
        return v;
    };
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    // row of (group, step rr): consecutive rows go to consecutive groups, so a wave's four groups read four adjacent rows
    auto row_of = [&](int rr) { return row0 + (size_t)rr * 16 + grp; };
    auto load = [&](f32x4 (&v)[CPL], int rr) {
        const size_t rraw = row_of(rr);
        const size_t row = rraw < rows ? rraw : rows - 1;   // idle groups shadow the last row: the shuffles stay full-width
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int ch = sub + 16 * i;
            v[i] = ch < nch ? *(reinterpret_cast<const f32x4*>(feat + row * C) + ch) : z;
        }
    };
    f32x4 v[CPL], vn[CPL];
    load(v, 0);
#pragma unroll 1
    for (int rr = 0; rr < RPG; ++rr) {
        if (rr + 1 < RPG) load(vn, rr + 1);
        const size_t row_raw = row_of(rr);
        const bool live = row_raw < rows;
        const size_t row = live ? row_raw : rows - 1;
        const int b = (int)(row / N);
        if (gn_a) {
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int ch = sub + 16 * i;
                if (ch < nch) {
                    const f32x4 ga = b == b0 ? reinterpret_cast<const f32x4*>(lw + 3 * C)[ch] : reinterpret_cast<const f32x4*>(gn_a + (size_t)b * C)[ch];
                    const f32x4 go = b == b0 ? reinterpret_cast<const f32x4*>(lw + 4 * C)[ch] : reinterpret_cast<const f32x4*>(gn_o + (size_t)b * C)[ch];
                    v[i] = v[i] * ga + go;
                }
            }
        } else {
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) s1 += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
            const float mean = gsum(s1) / C;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                const int ch = sub + 16 * i;
                const f32x4 d = ch < nch ? v[i] - mean : z;
                v[i] = d;
                s2 += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
            const float rstd = rsqrtf(gsum(s2) / C + eps);
#pragma unroll
            for (int i = 0; i < CPL; ++i) v[i] = v[i] * rstd;
        }
        float a[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int ch = sub + 16 * i;
            if (ch < nch) {
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const f32x4 w = reinterpret_cast<const f32x4*>(lw + (size_t)o * C)[ch];
                    a[o] += (v[i][0] * w[0] + v[i][1] * w[1]) + (v[i][2] * w[2] + v[i][3] * w[3]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) a[o] = gsum(a[o]);
        if (live && sub < 3) {
            const float Fv = (sub == 0 ? a[0] : sub == 1 ? a[1] : a[2]) + bias[sub];
            if (raw) raw[row * 3 + sub] = Fv;
            if (out) {
                const float cs = coef ? coef[4 * b + 0] : 0.f, co = coef ? coef[4 * b + 1] : 1.f;
                out[row * 3 + sub] = coef ? cs * x[row * 3 + sub] + co * Fv : Fv;
            }
        }
#pragma unroll
        for (int i = 0; i < CPL; ++i) v[i] = vn[i];
    }
}

}  // namespace

int stats_row_tile(int rows) { (void)rows; return STATS_ROWS; }

int col_stats_launch(const float* x, float* stats, int B, int rows, int C, hipStream_t st) {
    const int T = (rows + STATS_ROWS - 1) / STATS_ROWS;
    hipLaunchKernelGGL(col_stats_kernel, dim3(B * T), dim3(256), 0, st, x, stats, rows, C, T);
    return (int)hipGetLastError();
}

int adagn_coeffs_launch(const float* stats, int T, int rows, const float* t, int ctx_dim, const float* scale_w,
                        const float* scale_b, const float* bias_w, const float* bias_b, float* a, float* o, int B,
                        int C, int G, float eps, hipStream_t st) {
    if (C % G) return -5;
    // few samples with many tiles each (a cached upsample: 8 x 128): whole groups of channels per block, so that B x parts blocks work
    int parts = 1;
    while (B * parts * 2 <= 64 && T >= 64 && G % (parts * 2) == 0) parts *= 2;
    const int Cp = C / parts;
    // slices of the tile partials per column: as many as 1024 threads give, while a slice keeps >= 8 tiles
    int nsl = 1;
    while (nsl < 8 && (nsl * 2) * Cp <= 1024 && T / (nsl * 2) >= 8) nsl *= 2;
    const int nt = nsl * Cp >= 1024 ? 1024 : (nsl * Cp <= 256 ? 256 : ((nsl * Cp + 63) / 64) * 64);
    const size_t lds = (size_t)(2 * Cp + 2 * (G / parts) + 2 * nsl * Cp) * sizeof(double);
    static size_t attr = 0;
    if (lds > 48 * 1024 && lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(adagn_coeffs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL(adagn_coeffs_kernel, dim3(B, parts), dim3(nt), lds, st, stats, T, rows, t, ctx_dim, scale_w, scale_b,
                       bias_w, bias_b, a, o, C, G, eps, nsl, Cp);
    return (int)hipGetLastError();
}

int affine_apply_launch(const float* x, const float* a, const float* o, float* y, int B, int rows, int C,
                        hipStream_t st) {
    if (C % 4) return -2;
    const size_t total4 = (size_t)B * rows * C / 4;
    const unsigned grid = (unsigned)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(affine_apply_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, x, a, o, y, total4, rows * C / 4,
                       C / 4);
    return (int)hipGetLastError();
}

int affine_cast_f16_launch(const float* x, const float* a, const float* o, void* y16, int B, int rows, int C,
                           hipStream_t st) {
    if (C % 8) return -2;
    const size_t total8 = (size_t)B * rows * C / 8;
    const unsigned grid = (unsigned)((total8 + 255) / 256 < 8192 ? (total8 + 255) / 256 : 8192);
    hipLaunchKernelGGL(affine_cast_f16_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, x, a, o,
                       static_cast<_Float16*>(y16), total8, rows * C / 8, C / 8);
    return (int)hipGetLastError();
}

int edm_coeffs_launch(const float* sigma, float sigma_data, float* coef, int B, hipStream_t st) {
    hipLaunchKernelGGL(edm_coeffs_kernel, dim3((B + 63) / 64), dim3(64), 0, st, sigma, sigma_data, coef, B);
    return (int)hipGetLastError();
}

int lift_launch(const float* x, const float* coef, const float* W, const float* bias, float* out, float* stats, int B,
                int N, int C, hipStream_t st) {
    const int T = (N + STATS_ROWS - 1) / STATS_ROWS;
    hipLaunchKernelGGL(lift_kernel, dim3(B * T), dim3(256), 0, st, x, coef, W, bias, out, stats, N, C, T);
    return (int)hipGetLastError();
}

int lower_edm_launch(const float* feat, const float* x, const float* coef, const float* W, const float* bias,
                     const float* gn_a, const float* gn_o, float* out, float* raw, int B, int N, int C, float eps,
                     hipStream_t st) {
    const size_t rows = (size_t)B * N;
    const int cpl = (C / 4 + 15) / 16;
    const dim3 g16((unsigned)((rows + 15) / 16));
#define LOWER_V4(CPL)                                                                                              \
    hipLaunchKernelGGL((lower_edm_v4_kernel<CPL>), g16, dim3(256), 0, st, feat, x, coef, W, bias, gn_a, gn_o, out, raw, \
                       B, N, C, eps)
    static int v5 = -1;
    if (v5 < 0) {
        const char* e = getenv("GECCO_LOWER_V5");   // 0: one row per 16-lane group (A/B runs)
        v5 = (e && atoi(e) == 0) ? 0 : 1;
    }
    constexpr int RPG = 8;
    if (v5 && C % 4 == 0 && cpl <= 8 && rows >= 16 * RPG * 512) {   // enough 128-row blocks to fill the chip twice
        const dim3 g5((unsigned)((rows + 16 * RPG - 1) / (16 * RPG)));
        const size_t lds = (size_t)5 * C * sizeof(float);
#define LOWER_V5(CPL)                                                                                              \
    hipLaunchKernelGGL((lower_edm_v5_kernel<CPL, RPG>), g5, dim3(256), lds, st, feat, x, coef, W, bias, gn_a, gn_o, out, raw, \
                       B, N, C, eps)
        switch (cpl) {
            case 1: LOWER_V5(1); break;
            case 2: LOWER_V5(2); break;
            case 3: LOWER_V5(3); break;
            case 4: LOWER_V5(4); break;
            case 5: LOWER_V5(5); break;
            case 6: LOWER_V5(6); break;
            case 7: LOWER_V5(7); break;
            default: LOWER_V5(8); break;
        }
#undef LOWER_V5
        return (int)hipGetLastError();
    }
    if (C % 4 == 0 && cpl <= 8 && rows > 0) {
        switch (cpl) {
            case 1: LOWER_V4(1); break;
            case 2: LOWER_V4(2); break;
            case 3: LOWER_V4(3); break;
            case 4: LOWER_V4(4); break;
            case 5: LOWER_V4(5); break;
            case 6: LOWER_V4(6); break;
            case 7: LOWER_V4(7); break;
            default: LOWER_V4(8); break;
        }
    } else {
        hipLaunchKernelGGL(lower_edm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, feat, x, coef, W, bias,
                           gn_a, gn_o, out, raw, B, N, C, eps);
    }
#undef LOWER_V4
    return (int)hipGetLastError();
}
