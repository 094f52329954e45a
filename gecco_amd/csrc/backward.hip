// Backward-pass kernels of the denoiser that are not GEMMs (those are gemm_general_f32.hip): row softmax
// forward/backward for the materialised training attention, GaussianActivation / AdaGN / LayerNorm-lower / lift
// backward.  All reductions that cross workgroups go through per-block partials summed in a fixed order
// (reduce_batch_kernel): gradients are bitwise reproducible.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// P[row, :] = softmax(scale * S[row, :]); one wave per row.
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, float* __restrict__ P,
                                                          size_t rows, int n, float scale) {
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* s = S + row * n;
    float* p = P + row * n;
    float m = -INFINITY;
    for (int i = lane; i < n; i += 64) m = fmaxf(m, s[i] * scale);
    m = wmax(m);
    float l = 0.f;
    for (int i = lane; i < n; i += 64) {
        const float e = __expf(s[i] * scale - m);
        p[i] = e;
        l += e;
    }
    l = wsum(l);
    const float inv = 1.0f / l;
    for (int i = lane; i < n; i += 64) p[i] *= inv;
}

// dS = scale * P * (dP - sum_j P_j dP_j)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dP,
                                                          float* __restrict__ dS, size_t rows, int n, float scale) {
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = P + row * n;
    const float* dp = dP + row * n;
    float d = 0.f;
    for (int i = lane; i < n; i += 64) d += p[i] * dp[i];
    d = wsum(d);
    for (int i = lane; i < n; i += 64) dS[row * n + i] = scale * p[i] * (dp[i] - d);
}

// GaussianActivation backward: du = dy * g'(u); partial[block] = sum dy * dg/dalpha
__global__ __launch_bounds__(256) void gauss_act_bwd_kernel(const float* __restrict__ u, const float* __restrict__ dy,
                                                            const float* __restrict__ alpha, float* __restrict__ du,
                                                            float* __restrict__ partial, size_t n, int normalized) {
    __shared__ float red[4];
    const float a = alpha[0];
    const float k = -1.0f / (2.0f * a * a), nrm = normalized ? 1.0f / 0.28f : 1.0f;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = u[i], g = dy[i];
        const float E = __expf(x * x * k) * nrm;
        du[i] = g * E * (-x / (a * a));
        acc += g * E * (x * x / (a * a * a));
    }
    acc = wsum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// per (b, tile, c): {sum_n dy, sum_n dy * x}
__global__ __launch_bounds__(256) void col_dot_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            float* __restrict__ stats, int rows, int C, int T,
                                                            int tile_rows) {
    __shared__ f32x4 red[2][256];
    const int tile = blockIdx.x % T, b = blockIdx.x / T;
    const int m0 = tile * tile_rows, m1 = min(rows, m0 + tile_rows);
    const size_t base = (size_t)b * rows * C;
    if (C % 4 == 0 && C / 4 <= 256) {
        // 16-byte column chunks x RL row lanes (pointwise.hip's col_stats_kernel); lane partials combined in lane order
        const int c4n = C / 4, RL = 256 / c4n;
        const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n;
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        if (rl < RL) {
            const float* gp = dy + base + c4 * 4;
            const float* xp = x + base + c4 * 4;
            int m = m0 + rl;
            for (; m + 3 * RL < m1; m += 4 * RL) {
                f32x4 g[4], v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    g[u] = *reinterpret_cast<const f32x4*>(gp + (size_t)(m + u * RL) * C);
                    v[u] = *reinterpret_cast<const f32x4*>(xp + (size_t)(m + u * RL) * C);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) { s1 += g[u]; s2 += g[u] * v[u]; }
            }
            for (; m < m1; m += RL) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gp + (size_t)m * C);
                s1 += g;
                s2 += g * *reinterpret_cast<const f32x4*>(xp + (size_t)m * C);
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
            const float g = dy[base + (size_t)m * C + c];
            s1 += g;
            s2 += g * x[base + (size_t)m * C + c];
        }
        stats[(((size_t)b * T + tile) * 2 + 0) * C + c] = s1;
        stats[(((size_t)b * T + tile) * 2 + 1) * C + c] = s2;
    }
}

// AdaGN backward finalisation per sample.  xstats: forward partials {sum x, sum x^2}; gstats: {sum dy, sum dy*x}.
// Emits dx = dy*cA + x*cB + cC coefficients and ds, dz (grads of the per-(b,c) scale / shift).
__global__ __launch_bounds__(512) void adagn_bwd_coeffs_kernel(const float* __restrict__ xstats, int Tx,
                                                               const float* __restrict__ gstats, int Tg, int rows,
                                                               const float* __restrict__ t, int ctx_dim,
                                                               const float* __restrict__ scale_w,
                                                               const float* __restrict__ scale_b,
                                                               float* __restrict__ cA, float* __restrict__ cB,
                                                               float* __restrict__ cC, float* __restrict__ ds,
                                                               float* __restrict__ dz, int C, int G, float eps) {
    extern __shared__ double dsm[];  // [4][C] sums, [4][G] group values
    double* sx = dsm;
    double* sxx = dsm + C;
    double* sg = dsm + 2 * C;
    double* sgx = dsm + 3 * C;
    double* gv = dsm + 4 * C;  // mean, rstd, c1, c2 per group
    const int b = blockIdx.x, cpg = C / G;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        // (unrolled: the loads of eight tiles are in flight together; the additions keep their order)
#pragma unroll 8
        for (int k = 0; k < Tx; ++k) {
            a0 += (double)xstats[(((size_t)b * Tx + k) * 2 + 0) * C + c];
            a1 += (double)xstats[(((size_t)b * Tx + k) * 2 + 1) * C + c];
        }
#pragma unroll 8
        for (int k = 0; k < Tg; ++k) {
            a2 += (double)gstats[(((size_t)b * Tg + k) * 2 + 0) * C + c];
            a3 += (double)gstats[(((size_t)b * Tg + k) * 2 + 1) * C + c];
        }
        sx[c] = a0; sxx[c] = a1; sg[c] = a2; sgx[c] = a3;
    }
    __syncthreads();
    const double nel = (double)rows * cpg;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        double a0 = 0, a1 = 0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a0 += sx[c]; a1 += sxx[c]; }
        const double mean = a0 / nel;
        double var = a1 / nel - mean * mean;
        var = var < 0 ? 0 : var;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        double c1 = 0, c2 = 0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            double s = 1.0;
            if (scale_w) {
                s = scale_b[c];
                for (int j = 0; j < ctx_dim; ++j) s += (double)t[(size_t)b * ctx_dim + j] * scale_w[(size_t)c * ctx_dim + j];
            }
            const double dyxhat = rstd * (sgx[c] - mean * sg[c]);  // sum_n dy * xhat
            c1 += s * sg[c];
            c2 += s * dyxhat;
        }
        gv[g] = mean; gv[G + g] = rstd; gv[2 * G + g] = c1 / nel; gv[3 * G + g] = c2 / nel;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const double mean = gv[g], rstd = gv[G + g], c1 = gv[2 * G + g], c2 = gv[3 * G + g];
        double s = 1.0;
        if (scale_w) {
            s = scale_b[c];
            for (int j = 0; j < ctx_dim; ++j) s += (double)t[(size_t)b * ctx_dim + j] * scale_w[(size_t)c * ctx_dim + j];
        }
        cA[(size_t)b * C + c] = (float)(rstd * s);
        cB[(size_t)b * C + c] = (float)(-rstd * rstd * c2);
        cC[(size_t)b * C + c] = (float)(-rstd * c1 + mean * rstd * rstd * c2);
        if (ds) ds[(size_t)b * C + c] = (float)(rstd * (sgx[c] - mean * sg[c]));
        if (dz) dz[(size_t)b * C + c] = (float)sg[c];
    }
}

// dx = dy * cA[b,c] + x * cB[b,c] + cC[b,c] (+ add: the gradient arriving at x through the residual connection)
__global__ __launch_bounds__(256) void affine2_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ cA, const float* __restrict__ cB,
                                                            const float* __restrict__ cC, const float* __restrict__ add,
                                                            float* __restrict__ dx, size_t total4, int rowsC4, int C4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / rowsC4;
        const int c4 = (int)(i % C4);
        const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i], xv = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 a = reinterpret_cast<const f32x4*>(cA)[b * C4 + c4], bb = reinterpret_cast<const f32x4*>(cB)[b * C4 + c4];
        const f32x4 cc = reinterpret_cast<const f32x4*>(cC)[b * C4 + c4];
        f32x4 r = g * a + xv * bb + cc;
        if (add) r += reinterpret_cast<const f32x4*>(add)[i];
        reinterpret_cast<f32x4*>(dx)[i] = r;
    }
}

// AdaGN parameter gradients from ds, dz (B, C) and t (B, ctx).
// 32 channels x 8 sample lanes per block: lane bl sums samples bl, bl + 8, ..; the lanes are combined in lane order (fixed
// summation order).  One pass per quantity (the biases, then each column of the scale / bias weights).
__global__ __launch_bounds__(256) void adagn_param_grads_kernel(const float* __restrict__ ds, const float* __restrict__ dz,
                                                                const float* __restrict__ t, int B, int C, int ctx_dim,
                                                                float* __restrict__ d_scale_w, float* __restrict__ d_scale_b,
                                                                float* __restrict__ d_bias_w, float* __restrict__ d_bias_b) {
    __shared__ float red[8][32][2];
    const int cl = threadIdx.x & 31, bl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    for (int q = 0; q <= ctx_dim; ++q) {
        float s = 0.f, z = 0.f;
        if (c < C) {
#pragma unroll 4
            for (int b = bl; b < B; b += 8) {
                const float wq = q == 0 ? 1.f : t[(size_t)b * ctx_dim + q - 1];
                s += ds[(size_t)b * C + c] * wq;
                z += dz[(size_t)b * C + c] * wq;
            }
        }
        red[bl][cl][0] = s;
        red[bl][cl][1] = z;
        __syncthreads();
        if (bl == 0 && c < C) {
            float ss = red[0][cl][0], zz = red[0][cl][1];
#pragma unroll
            for (int k = 1; k < 8; ++k) { ss += red[k][cl][0]; zz += red[k][cl][1]; }
            if (q == 0) {
                d_scale_b[c] = ss;
                d_bias_b[c] = zz;
            } else {
                d_scale_w[(size_t)c * ctx_dim + q - 1] = ss;
                d_bias_w[(size_t)c * ctx_dim + q - 1] = zz;
            }
        }
        __syncthreads();
    }
}

// lift backward: partial[b, tile] = {dW[:, 0..2], db} (4, C) from dY (B, N, C) and xin (B, N, 3)
__global__ __launch_bounds__(256) void lift_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ xin,
                                                       float* __restrict__ partial, int N, int C, int T, int tile_rows) {
    extern __shared__ float xs[];
    const int tile = blockIdx.x % T, b = blockIdx.x / T;
    const int m0 = tile * tile_rows, m1 = min(N, m0 + tile_rows);
    for (int i = threadIdx.x; i < (m1 - m0) * 3; i += blockDim.x) xs[i] = xin[((size_t)b * N + m0) * 3 + i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int m = 0; m < m1 - m0; ++m) {
            const float g = dY[((size_t)b * N + m0 + m) * C + c];
            a0 += g * xs[m * 3 + 0];
            a1 += g * xs[m * 3 + 1];
            a2 += g * xs[m * 3 + 2];
            a3 += g;
        }
        float* p = partial + ((size_t)b * T + tile) * 4 * C;
        p[0 * C + c] = a0; p[1 * C + c] = a1; p[2 * C + c] = a2; p[3 * C + c] = a3;
    }
}

// lower backward: F = Linear(C->3)(LN(feat)).  One wave per point: dfeat, and per-block partial {dW (3,C), db (3)}.
__global__ __launch_bounds__(256) void lower_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ dF,
                                                        const float* __restrict__ W, float* __restrict__ dfeat,
                                                        float* __restrict__ partial, size_t rows, int C, float eps,
                                                        int rows_per_block) {
    extern __shared__ float sm[];  // [4 waves][3][C] + [4][3]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* wacc = sm + (size_t)wave * 3 * C;
    for (int c = lane; c < 3 * C; c += 64) wacc[c] = 0.f;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f;
    const size_t r0 = (size_t)blockIdx.x * rows_per_block;
    for (size_t row = r0 + wave; row < r0 + rows_per_block && row < rows; row += 4) {
        const float* f = feat + row * C;
        float s1 = 0.f;
        for (int c = lane; c < C; c += 64) s1 += f[c];
        const float mean = wsum(s1) / C;
        float s2 = 0.f;
        for (int c = lane; c < C; c += 64) { const float d = f[c] - mean; s2 += d * d; }
        const float rstd = rsqrtf(wsum(s2) / C + eps);
        const float g0 = dF[row * 3 + 0], g1 = dF[row * 3 + 1], g2 = dF[row * 3 + 2];
        // dyhat = dF * W ; LN backward needs mean(dyhat) and mean(dyhat * yhat)
        float m1 = 0.f, m2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float yh = (f[c] - mean) * rstd;
            const float dyh = g0 * W[c] + g1 * W[C + c] + g2 * W[2 * C + c];
            m1 += dyh;
            m2 += dyh * yh;
            wacc[c] += g0 * yh;
            wacc[C + c] += g1 * yh;
            wacc[2 * C + c] += g2 * yh;
        }
        m1 = wsum(m1) / C;
        m2 = wsum(m2) / C;
        for (int c = lane; c < C; c += 64) {
            const float yh = (f[c] - mean) * rstd;
            const float dyh = g0 * W[c] + g1 * W[C + c] + g2 * W[2 * C + c];
            dfeat[row * C + c] = rstd * (dyh - m1 - yh * m2);
        }
        b0 += g0; b1 += g1; b2 += g2;
    }
    __syncthreads();
    float* p = partial + (size_t)blockIdx.x * (3 * C + 4);
    for (int c = threadIdx.x; c < 3 * C; c += blockDim.x)
        p[c] = sm[c] + sm[3 * C + c] + sm[6 * C + c] + sm[9 * C + c];
    float* bsm = sm + 12 * C;
    if (lane == 0) { bsm[wave * 3 + 0] = b0; bsm[wave * 3 + 1] = b1; bsm[wave * 3 + 2] = b2; }
    __syncthreads();
    if (threadIdx.x < 3) p[3 * C + threadIdx.x] = bsm[threadIdx.x] + bsm[3 + threadIdx.x] + bsm[6 + threadIdx.x] + bsm[9 + threadIdx.x];
}

// The same for C = 128 CPL (CPL = 1 .. 4): 32 lanes per point, CPL 16-byte channel chunks per lane (chunk q + 32 j), the row,
// the three weight rows and the 3 x C weight-gradient accumulators in registers — one read of feat, one write of dfeat, no LDS
// in the loop (the generic kernel above re-reads every feat element four times and keeps its accumulators in LDS: 317 us
// against ~60 for the 151 + 151 MB of the C2 step).  Same partial layout: (blocks, 3 C + 4).
template <int CPL>
__global__ __launch_bounds__(256) void lower_bwd_v4_kernel(const float* __restrict__ feat, const float* __restrict__ dF,
                                                           const float* __restrict__ W, float* __restrict__ dfeat,
                                                           float* __restrict__ partial, size_t rows, float eps, int rows_per_block) {
    constexpr int C = 128 * CPL;
    __shared__ float red[8][3 * C + 4];
    const int q = threadIdx.x & 31, slot = threadIdx.x >> 5;
    f32x4 w[3][CPL], acc[3][CPL];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            w[k][j] = *reinterpret_cast<const f32x4*>(W + (size_t)k * C + 4 * (q + 32 * j));
            acc[k][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    float b0 = 0.f, b1 = 0.f, b2 = 0.f;
    const size_t r0 = (size_t)blockIdx.x * rows_per_block;
    auto lsum = [](float v) {
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
    for (size_t row = r0 + slot; row < r0 + rows_per_block; row += 8) {
        const bool live = row < rows;
        f32x4 f[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j)
            f[j] = live ? *reinterpret_cast<const f32x4*>(feat + row * C + 4 * (q + 32 * j)) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float g0 = live ? dF[row * 3 + 0] : 0.f, g1 = live ? dF[row * 3 + 1] : 0.f, g2 = live ? dF[row * 3 + 2] : 0.f;
        float s1 = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) s1 += (f[j][0] + f[j][1]) + (f[j][2] + f[j][3]);
        const float mean = lsum(s1) / C;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            f[j] = f[j] - mean;
            const f32x4 d = f[j] * f[j];
            s2 += (d[0] + d[1]) + (d[2] + d[3]);
        }
        const float rstd = rsqrtf(lsum(s2) / C + eps);
        float m1 = 0.f, m2 = 0.f;
        f32x4 dyh[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            f[j] = f[j] * rstd;   // yhat
            dyh[j] = w[0][j] * g0 + w[1][j] * g1 + w[2][j] * g2;
            const f32x4 e = dyh[j] * f[j];
            m1 += (dyh[j][0] + dyh[j][1]) + (dyh[j][2] + dyh[j][3]);
            m2 += (e[0] + e[1]) + (e[2] + e[3]);
            acc[0][j] += f[j] * g0;
            acc[1][j] += f[j] * g1;
            acc[2][j] += f[j] * g2;
        }
        m1 = lsum(m1) / C;
        m2 = lsum(m2) / C;
        if (live) {
#pragma unroll
            for (int j = 0; j < CPL; ++j)
                *reinterpret_cast<f32x4*>(dfeat + row * C + 4 * (q + 32 * j)) = (dyh[j] - m1 - f[j] * m2) * rstd;
        }
        b0 += g0; b1 += g1; b2 += g2;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < CPL; ++j) *reinterpret_cast<f32x4*>(&red[slot][k * C + 4 * (q + 32 * j)]) = acc[k][j];
    if (q == 0) { red[slot][3 * C] = b0; red[slot][3 * C + 1] = b1; red[slot][3 * C + 2] = b2; red[slot][3 * C + 3] = 0.f; }
    __syncthreads();
    float* p = partial + (size_t)blockIdx.x * (3 * C + 4);
    for (int c = threadIdx.x; c < 3 * C + 4; c += 256) {
        float t = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) t += red[s_][c];
        p[c] = t;
    }
}

unsigned grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

int softmax_fwd_launch(const float* S, float* P, size_t rows, int n, float scale, hipStream_t st) {
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, S, P, rows, n, scale);
    return (int)hipGetLastError();
}
int softmax_bwd_launch(const float* P, const float* dP, float* dS, size_t rows, int n, float scale, hipStream_t st) {
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, P, dP, dS, rows, n, scale);
    return (int)hipGetLastError();
}
int gauss_act_bwd_blocks(size_t n) { return (int)grid_for(n); }
int gauss_act_bwd_launch(const float* u, const float* dy, const float* alpha, float* du, float* partial, size_t n,
                         int normalized, hipStream_t st) {
    hipLaunchKernelGGL(gauss_act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, u, dy, alpha, du, partial, n, normalized);
    return (int)hipGetLastError();
}
int col_dot_stats_launch(const float* dy, const float* x, float* stats, int B, int rows, int C, hipStream_t st) {
    const int tr = stats_row_tile(rows), T = (rows + tr - 1) / tr;
    hipLaunchKernelGGL(col_dot_stats_kernel, dim3(B * T), dim3(256), 0, st, dy, x, stats, rows, C, T, tr);
    return (int)hipGetLastError();
}
int adagn_bwd_coeffs_launch(const float* xstats, int Tx, const float* gstats, int Tg, int rows, const float* t,
                            int ctx_dim, const float* scale_w, const float* scale_b, float* cA, float* cB, float* cC,
                            float* ds, float* dz, int B, int C, int G, float eps, hipStream_t st) {
    if (C % G) return -5;
    const size_t lds = (size_t)(4 * C + 4 * G) * sizeof(double);
    const int nt = C >= 512 ? 512 : (C <= 256 ? 256 : (C + 63) / 64 * 64);   // one channel per thread up to 512 channels
    hipLaunchKernelGGL(adagn_bwd_coeffs_kernel, dim3(B), dim3(nt), lds, st, xstats, Tx, gstats, Tg, rows, t, ctx_dim,
                       scale_w, scale_b, cA, cB, cC, ds, dz, C, G, eps);
    return (int)hipGetLastError();
}
int affine2_apply_launch(const float* dy, const float* x, const float* cA, const float* cB, const float* cC, float* dx,
                         int B, int rows, int C, hipStream_t st, const float* add) {
    if (C % 4) return -2;
    const size_t total4 = (size_t)B * rows * C / 4;
    hipLaunchKernelGGL(affine2_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, st, dy, x, cA, cB, cC, add, dx, total4,
                       rows * C / 4, C / 4);
    return (int)hipGetLastError();
}
int adagn_param_grads_launch(const float* ds, const float* dz, const float* t, int B, int C, int ctx_dim,
                             float* d_scale_w, float* d_scale_b, float* d_bias_w, float* d_bias_b, hipStream_t st) {
    hipLaunchKernelGGL(adagn_param_grads_kernel, dim3((C + 31) / 32), dim3(256), 0, st, ds, dz, t, B, C, ctx_dim,
                       d_scale_w, d_scale_b, d_bias_w, d_bias_b);
    return (int)hipGetLastError();
}
int lift_bwd_launch(const float* dY, const float* xin, float* partial, int B, int N, int C, hipStream_t st) {
    const int tr = stats_row_tile(N), T = (N + tr - 1) / tr;
    hipLaunchKernelGGL(lift_bwd_kernel, dim3(B * T), dim3(256), (size_t)tr * 3 * sizeof(float), st, dY, xin, partial, N, C,
                       T, tr);
    return (int)hipGetLastError();
}
int lower_bwd_blocks(size_t rows) { return (int)((rows + 127) / 128); }
int lower_bwd_launch(const float* feat, const float* dF, const float* W, float* dfeat, float* partial, size_t rows,
                     int C, float eps, hipStream_t st) {
    const unsigned nblk = (unsigned)lower_bwd_blocks(rows);
    switch (C) {
        case 128: hipLaunchKernelGGL((lower_bwd_v4_kernel<1>), dim3(nblk), dim3(256), 0, st, feat, dF, W, dfeat, partial, rows, eps, 128); return (int)hipGetLastError();
        case 256: hipLaunchKernelGGL((lower_bwd_v4_kernel<2>), dim3(nblk), dim3(256), 0, st, feat, dF, W, dfeat, partial, rows, eps, 128); return (int)hipGetLastError();
        case 384: hipLaunchKernelGGL((lower_bwd_v4_kernel<3>), dim3(nblk), dim3(256), 0, st, feat, dF, W, dfeat, partial, rows, eps, 128); return (int)hipGetLastError();
        case 512: hipLaunchKernelGGL((lower_bwd_v4_kernel<4>), dim3(nblk), dim3(256), 0, st, feat, dF, W, dfeat, partial, rows, eps, 128); return (int)hipGetLastError();
        default: break;
    }
    const size_t lds = (size_t)(12 * C + 12) * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lower_bwd_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL(lower_bwd_kernel, dim3((unsigned)lower_bwd_blocks(rows)), dim3(256), lds, st, feat, dF, W, dfeat,
                       partial, rows, C, eps, 128);
    return (int)hipGetLastError();
}
