// Projective feature lookup (RayNetwork.extract_image_features, reference models/ray.py:64-87):
// diffusion-space geometry -> data space (reparam.py) -> pinhole projection (kornia
// project_points) -> bilinear taps of F.grid_sample(align_corners=False,
// padding zeros) on every pyramid level -> concatenated (B, N, sum C_l) features, plus the
// GroupNorm(16) partial statistics the following img_feature_proj needs (models/ray.py:52-55).
//
// HBM/L2-gather bound: 4 taps x sum(C_l) floats read + sum(C_l) written per point.  The pyramids
// are consumed channels-last (B, H, W, C): one texel's channels are contiguous, so a wave reads a
// tap as whole 128-byte lines (the reference's NCHW layout would make every channel a separate
// 4-byte gather).  One wave handles one point at a time; lanes run over 16-byte channel chunks.
#include "common.h"

#include <algorithm>
#include "kernels.h"

#pragma clang fp contract(off)  // index math must round like the reference's separate fp32 ops

namespace {

constexpr int LOOKUP_ROWS = 128;  // points per block == row tile of the GN statistics partials

struct Taps {
    int x0, y0;
    float ix, iy;
};

// F.grid_sample's coordinate pipeline in torch's op order:
//   g = uv*2 - 1 (models/ray.py:81);  ix = ((g + 1) * W - 1) / 2;  x0 = floor(ix)
__device__ __forceinline__ Taps bilinear_taps(float u, float v, int Hh, int Ww) {
    const float gx = u * 2.0f - 1.0f, gy = v * 2.0f - 1.0f;
    const float ix = ((gx + 1.0f) * (float)Ww - 1.0f) / 2.0f;
    const float iy = ((gy + 1.0f) * (float)Hh - 1.0f) / 2.0f;
    const float fx = floorf(ix), fy = floorf(iy);
    Taps t;
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    t.ix = ix;
    t.iy = iy;
    return t;
}

// geometry (diffusion space, optionally scaled by c_in) -> normalised image coordinates (u, v)
__device__ __forceinline__ void project_uv(float g0, float g1, float g2, const float* __restrict__ Kb, int kind,
                                           const float* __restrict__ mean, const float* __restrict__ std_,
                                           float logit_scale, float& u, float& v) {
    float X = g0, Y = g1, Z = g2;
    const float fx = Kb[0], fy = Kb[4], cx = Kb[2], cy = Kb[5];
    if (kind == 1) {  // GaussianReparam.diffusion_to_data (reparam.py:61-63)
        X = g0 * std_[0] + mean[0];
        Y = g1 * std_[1] + mean[1];
        Z = g2 * std_[2] + mean[2];
    } else if (kind == 2) {  // UVLReparam.diffusion_to_data (reparam.py:159-177,131-137)
        const float uu = g0 * std_[0] + mean[0], vv = g1 * std_[1] + mean[1], l = g2 * std_[2] + mean[2];
        const float su = (tanhf(uu) * logit_scale + 1.0f) / 2.0f, sv = (tanhf(vv) * logit_scale + 1.0f) / 2.0f;
        const float d = expf(l);
        const float xx = (su - cx) / fx, yy = (sv - cy) / fy;
        float nrm = sqrtf(xx * xx + yy * yy + 1.0f);
        nrm = fmaxf(nrm, 1e-12f);
        X = xx / nrm * d;
        Y = yy / nrm * d;
        Z = 1.0f / nrm * d;
    }
    // kornia project_points: eps-guarded perspective divide, then fx*x + cx
    const float sc = fabsf(Z) > 1e-8f ? 1.0f / (Z + 1e-8f) : 1.0f;
    u = sc * X * fx + cx;
    v = sc * Y * fy + cy;
}

// TEX16: the pyramid levels hold fp16 texels (GeccoPyramid::texel_f16): 8-byte gathers instead of 16-byte ones; converted on arrival, the
// interpolation itself is the fp32 expression below either way
template <bool TEX16>
__device__ __forceinline__ f32x4 texel4(const float* fb, size_t off) {
    if constexpr (TEX16) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        const f16x4 v = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(fb) + off);
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    } else {
        return *reinterpret_cast<const f32x4*>(fb + off);
    }
}

template <bool TEX16, bool OUT16 = false>
__global__ __launch_bounds__(256) void ray_lookup_kernel(const float* __restrict__ geom,
                                                         const float* __restrict__ coef,
                                                         const float* __restrict__ K, LookupArgs a,
                                                         float* __restrict__ out, float* __restrict__ stats, int N,
                                                         int T) {
    __shared__ float red[4][2][1024];  // per-wave column partials (sum C_l <= 1024)
    const int tile = blockIdx.x % T, b = blockIdx.x / T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Ct = a.c_total, C4 = Ct / 4;
    const float cin = coef ? coef[4 * b + 2] : 1.0f;
    const float* Kb = K + (size_t)b * 9;

    constexpr int MAXCH = 4;  // float4 chunks per lane: 64 * 4 * 4 = 1024 channels max
    f32x4 s1[MAXCH], s2[MAXCH];
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        s1[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        s2[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // this lane's chunks: level and channel offset are point-independent
    int lvl[MAXCH], coff[MAXCH];
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = (c * 64 + lane) * 4;
        int l = 0, base = 0;
        while (l + 1 < a.n_levels && ch >= base + a.C[l]) {
            base += a.C[l];
            ++l;
        }
        lvl[c] = l;
        coff[c] = ch - base;
    }

    const int m0 = tile * LOOKUP_ROWS, m1 = min(N, m0 + LOOKUP_ROWS);
    for (int m = m0 + wave; m < m1; m += 4) {
        const float* gp = geom + ((size_t)b * N + m) * 3;
        float u, v;
        project_uv(cin * gp[0], cin * gp[1], cin * gp[2], Kb, a.reparam_kind, a.rp_mean, a.rp_std, a.logit_scale, u, v);
        Taps tp[4];
#pragma unroll
        for (int l = 0; l < 4; ++l)
            if (l < a.n_levels) tp[l] = bilinear_taps(u, v, a.H[l], a.W[l]);
        float* orow = out + ((size_t)b * N + m) * Ct;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            const int c4 = c * 64 + lane;
            if (c4 >= C4) continue;
            const int l = lvl[c];
            Taps t = tp[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (l == q) t = tp[q];
            const int Hh = a.H[l], Ww = a.W[l], Cl = a.C[l];
            const float* fb = a.feat[l];                                      // (TEX16: a pointer to halves; offsets in elements)
            const size_t fo = (size_t)b * Hh * Ww * Cl + coff[c];
            const int x1 = t.x0 + 1, y1 = t.y0 + 1;
            // torch's weights: nw = (x1 - ix)(y1 - iy), ne = (ix - x0)(y1 - iy), sw = (x1 - ix)(iy - y0), se = ...
            const float wx0 = (float)x1 - t.ix, wy0 = (float)y1 - t.iy;
            const float wx1 = t.ix - (float)t.x0, wy1 = t.iy - (float)t.y0;
            const bool bx0 = t.x0 >= 0 && t.x0 <= Ww - 1, bx1 = x1 >= 0 && x1 <= Ww - 1;
            const bool by0 = t.y0 >= 0 && t.y0 <= Hh - 1, by1 = y1 >= 0 && y1 <= Hh - 1;
            const int cx0 = min(max(t.x0, 0), Ww - 1), cx1 = min(max(x1, 0), Ww - 1);
            const int cy0 = min(max(t.y0, 0), Hh - 1), cy1 = min(max(y1, 0), Hh - 1);
            const f32x4 nw = texel4<TEX16>(fb, fo + ((size_t)cy0 * Ww + cx0) * Cl);
            const f32x4 ne = texel4<TEX16>(fb, fo + ((size_t)cy0 * Ww + cx1) * Cl);
            const f32x4 sw = texel4<TEX16>(fb, fo + ((size_t)cy1 * Ww + cx0) * Cl);
            const f32x4 se = texel4<TEX16>(fb, fo + ((size_t)cy1 * Ww + cx1) * Cl);
            const float w_nw = (bx0 && by0) ? wx0 * wy0 : 0.f, w_ne = (bx1 && by0) ? wx1 * wy0 : 0.f;
            const float w_sw = (bx0 && by1) ? wx0 * wy1 : 0.f, w_se = (bx1 && by1) ? wx1 * wy1 : 0.f;
            const f32x4 r = nw * w_nw + ne * w_ne + sw * w_sw + se * w_se;
            if constexpr (OUT16) {   // (B, N, Ct) halves for a matrix kernel that rounds its operand to fp16 anyway; the statistics are the fp32 values
                typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(out) + ((size_t)b * N + m) * Ct)[c4] =
                    f16x4{(_Float16)r[0], (_Float16)r[1], (_Float16)r[2], (_Float16)r[3]};
            } else {
                *reinterpret_cast<f32x4*>(orow + c4 * 4) = r;
            }
            s1[c] += r;
            s2[c] += r * r;
        }
    }
    if (stats) {
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            const int c4 = c * 64 + lane;
            if (c4 < C4) {
                *reinterpret_cast<f32x4*>(&red[wave][0][c4 * 4]) = s1[c];
                *reinterpret_cast<f32x4*>(&red[wave][1][c4 * 4]) = s2[c];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * Ct; i += 256) {
            const int which = i / Ct, ch = i % Ct;
            const float t = red[0][which][ch] + red[1][which][ch] + red[2][which][ch] + red[3][which][ch];
            stats[(((size_t)b * T + tile) * 2 + which) * Ct + ch] = t;
        }
    }
}

// project_uv with its Jacobian: du[k] = d u / d g_k, dv[k] = d v / d g_k (the derivative of exactly the expressions above; floor / the
// eps guard contribute none, as in autograd through the reference's torch ops)
// dku = d u / d (fx, cx), dkv = d v / d (fy, cy): the camera matrix enters the projection and, for the UVL reparametrisation, the
// unprojection in front of it (u depends on fx, cx only and v on fy, cy only up to the normalisation of the ray, which couples them:
// dkuy = d u / d (fy, cy), dkvx = d v / d (fx, cx))
__device__ __forceinline__ void project_uv_jac(float g0, float g1, float g2, const float* __restrict__ Kb, int kind,
                                               const float* __restrict__ mean, const float* __restrict__ std_, float logit_scale,
                                               float (&du)[3], float (&dv)[3], float (&dku)[2], float (&dkv)[2], float (&dkuy)[2],
                                               float (&dkvx)[2]) {
    float X = g0, Y = g1, Z = g2;
    float dX[3] = {1.f, 0.f, 0.f}, dY[3] = {0.f, 1.f, 0.f}, dZ[3] = {0.f, 0.f, 1.f};
    // partials of (X, Y, Z) with respect to (fx, cx) [index 0, 1] and (fy, cy) [index 2, 3]: zero unless the reparametrisation unprojects
    float kX[4] = {0.f, 0.f, 0.f, 0.f}, kY[4] = {0.f, 0.f, 0.f, 0.f}, kZ[4] = {0.f, 0.f, 0.f, 0.f};
    const float fx = Kb[0], fy = Kb[4], cx = Kb[2], cy = Kb[5];
    if (kind == 1) {
        X = g0 * std_[0] + mean[0];
        Y = g1 * std_[1] + mean[1];
        Z = g2 * std_[2] + mean[2];
        dX[0] = std_[0];
        dY[1] = std_[1];
        dZ[2] = std_[2];
    } else if (kind == 2) {
        const float uu = g0 * std_[0] + mean[0], vv = g1 * std_[1] + mean[1], l = g2 * std_[2] + mean[2];
        const float tu = tanhf(uu), tv = tanhf(vv);
        const float su = (tu * logit_scale + 1.0f) / 2.0f, sv = (tv * logit_scale + 1.0f) / 2.0f;
        const float d = expf(l);
        const float xx = (su - cx) / fx, yy = (sv - cy) / fy;
        const float nrm = fmaxf(sqrtf(xx * xx + yy * yy + 1.0f), 1e-12f);
        const float in = 1.0f / nrm, in3 = in * in * in;
        X = xx * in * d;
        Y = yy * in * d;
        Z = in * d;
        const float dxx = (1.0f - tu * tu) * logit_scale * 0.5f * std_[0] / fx;   // d xx / d g0
        const float dyy = (1.0f - tv * tv) * logit_scale * 0.5f * std_[1] / fy;   // d yy / d g1
        const float dd = d * std_[2];                                              // d d / d g2
        dX[0] = d * (in - xx * xx * in3) * dxx;  dX[1] = -d * xx * yy * in3 * dyy;      dX[2] = xx * in * dd;
        dY[0] = -d * xx * yy * in3 * dxx;        dY[1] = d * (in - yy * yy * in3) * dyy; dY[2] = yy * in * dd;
        dZ[0] = -d * xx * in3 * dxx;             dZ[1] = -d * yy * in3 * dyy;            dZ[2] = in * dd;
        // xx = (su - cx) / fx: d xx / d fx = -xx / fx, d xx / d cx = -1 / fx; yy likewise with (fy, cy)
        const float xk[2] = {-xx / fx, -1.0f / fx}, yk[2] = {-yy / fy, -1.0f / fy};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            kX[j] = d * (in - xx * xx * in3) * xk[j];      kX[2 + j] = -d * xx * yy * in3 * yk[j];
            kY[j] = -d * xx * yy * in3 * xk[j];            kY[2 + j] = d * (in - yy * yy * in3) * yk[j];
            kZ[j] = -d * xx * in3 * xk[j];                 kZ[2 + j] = -d * yy * in3 * yk[j];
        }
    }
    const bool live = fabsf(Z) > 1e-8f;
    const float sc = live ? 1.0f / (Z + 1e-8f) : 1.0f;
    const float dsc = live ? -sc * sc : 0.0f;   // d sc / d Z
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        du[k] = fx * (sc * dX[k] + dsc * dZ[k] * X);
        dv[k] = fy * (sc * dY[k] + dsc * dZ[k] * Y);
    }
    // u = sc X fx + cx, v = sc Y fy + cy
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        dku[j] = fx * (sc * kX[j] + dsc * kZ[j] * X) + (j == 0 ? sc * X : 1.0f);
        dkv[j] = fy * (sc * kY[2 + j] + dsc * kZ[2 + j] * Y) + (j == 0 ? sc * Y : 1.0f);
        dkuy[j] = fx * (sc * kX[2 + j] + dsc * kZ[2 + j] * X);
        dkvx[j] = fy * (sc * kY[j] + dsc * kZ[j] * Y);
    }
}

// Gradient of the lookup with respect to the GEOMETRY (autograd of F.grid_sample w.r.t. its grid, then of the projection and the
// reparametrisation: reference models/ray.py:64-87 when a caller differentiates the conditional denoiser with respect to its input
// cloud): dgeom[b, n, k] = sum_l ( W_l sum_c dout_c d out_c / d ix * du/dg_k + H_l sum_c dout_c d out_c / d iy * dv/dg_k ), with
// d out / d ix = wy0 (ne - nw) + wy1 (se - sw) and d out / d iy = wx0 (sw - nw) + wx1 (se - ne) over the in-range taps.  One wave per
// point, lanes over 16-byte channel chunks as in the forward; a wave reduction, three floats out.
// dKpart (optional): (B, T, 4) partials of the gradient with respect to (fx, cx, fy, cy) of the sample's camera matrix, one per block
__global__ __launch_bounds__(256) void ray_lookup_dgeom_kernel(const float* __restrict__ geom, const float* __restrict__ K, LookupArgs a,
                                                               const float* __restrict__ dout, float* __restrict__ dgeom,
                                                               float* __restrict__ dKpart, int N, int T) {
    __shared__ float kred[4][4];
    float kacc[4] = {0.f, 0.f, 0.f, 0.f};   // lane 0 of each wave: sums over the wave's points
    const int tile = blockIdx.x % T, b = blockIdx.x / T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Ct = a.c_total, C4 = Ct / 4;
    const float* Kb = K + (size_t)b * 9;
    constexpr int MAXCH = 4;
    int lvl[MAXCH], coff[MAXCH];
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = (c * 64 + lane) * 4;
        int l = 0, base = 0;
        while (l + 1 < a.n_levels && ch >= base + a.C[l]) {
            base += a.C[l];
            ++l;
        }
        lvl[c] = l;
        coff[c] = ch - base;
    }
    const int m0 = tile * LOOKUP_ROWS, m1 = min(N, m0 + LOOKUP_ROWS);
    for (int m = m0 + wave; m < m1; m += 4) {
        const float* gp = geom + ((size_t)b * N + m) * 3;
        float u, v;
        project_uv(gp[0], gp[1], gp[2], Kb, a.reparam_kind, a.rp_mean, a.rp_std, a.logit_scale, u, v);
        Taps tp[4];
#pragma unroll
        for (int l = 0; l < 4; ++l)
            if (l < a.n_levels) tp[l] = bilinear_taps(u, v, a.H[l], a.W[l]);
        const float* drow = dout + ((size_t)b * N + m) * Ct;
        float gu = 0.f, gv = 0.f;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            const int c4 = c * 64 + lane;
            if (c4 >= C4) continue;
            const int l = lvl[c];
            Taps t = tp[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (l == q) t = tp[q];
            const int Hh = a.H[l], Ww = a.W[l], Cl = a.C[l];
            const float* fb = a.feat[l] + (size_t)b * Hh * Ww * Cl + coff[c];
            const int x1 = t.x0 + 1, y1 = t.y0 + 1;
            const float wx0 = (float)x1 - t.ix, wy0 = (float)y1 - t.iy;
            const float wx1 = t.ix - (float)t.x0, wy1 = t.iy - (float)t.y0;
            const bool bx0 = t.x0 >= 0 && t.x0 <= Ww - 1, bx1 = x1 >= 0 && x1 <= Ww - 1;
            const bool by0 = t.y0 >= 0 && t.y0 <= Hh - 1, by1 = y1 >= 0 && y1 <= Hh - 1;
            const int cx0 = min(max(t.x0, 0), Ww - 1), cx1 = min(max(x1, 0), Ww - 1);
            const int cy0 = min(max(t.y0, 0), Hh - 1), cy1 = min(max(y1, 0), Hh - 1);
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 nw = (bx0 && by0) ? *reinterpret_cast<const f32x4*>(fb + ((size_t)cy0 * Ww + cx0) * Cl) : z4;
            const f32x4 ne = (bx1 && by0) ? *reinterpret_cast<const f32x4*>(fb + ((size_t)cy0 * Ww + cx1) * Cl) : z4;
            const f32x4 sw = (bx0 && by1) ? *reinterpret_cast<const f32x4*>(fb + ((size_t)cy1 * Ww + cx0) * Cl) : z4;
            const f32x4 se = (bx1 && by1) ? *reinterpret_cast<const f32x4*>(fb + ((size_t)cy1 * Ww + cx1) * Cl) : z4;
            const f32x4 dy = *reinterpret_cast<const f32x4*>(drow + c4 * 4);
            const f32x4 dix = (ne - nw) * wy0 + (se - sw) * wy1;
            const f32x4 diy = (sw - nw) * wx0 + (se - ne) * wx1;
            const f32x4 pu = dy * dix, pv = dy * diy;
            gu += (float)Ww * ((pu[0] + pu[1]) + (pu[2] + pu[3]));
            gv += (float)Hh * ((pv[0] + pv[1]) + (pv[2] + pv[3]));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            gu += __shfl_xor(gu, o, 64);
            gv += __shfl_xor(gv, o, 64);
        }
        if (lane == 0) {
            float du[3], dv[3], dku[2], dkv[2], dkuy[2], dkvx[2];
            project_uv_jac(gp[0], gp[1], gp[2], Kb, a.reparam_kind, a.rp_mean, a.rp_std, a.logit_scale, du, dv, dku, dkv, dkuy, dkvx);
            if (dgeom) {
                float* o = dgeom + ((size_t)b * N + m) * 3;
#pragma unroll
                for (int k = 0; k < 3; ++k) o[k] = gu * du[k] + gv * dv[k];
            }
            kacc[0] += gu * dku[0] + gv * dkvx[0];   // fx
            kacc[1] += gu * dku[1] + gv * dkvx[1];   // cx
            kacc[2] += gv * dkv[0] + gu * dkuy[0];   // fy
            kacc[3] += gv * dkv[1] + gu * dkuy[1];   // cy
        }
    }
    if (dKpart) {   // the four waves' sums in a fixed order
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) kred[wave][j] = kacc[j];
        }
        __syncthreads();
        if (threadIdx.x < 4)
            dKpart[((size_t)b * T + tile) * 4 + threadIdx.x] =
                ((kred[0][threadIdx.x] + kred[1][threadIdx.x]) + kred[2][threadIdx.x]) + kred[3][threadIdx.x];
    }
}

// Backward of the lookup with respect to the pyramids (autograd of F.grid_sample w.r.t. its input, reference
// models/ray.py:82-85 under loss.backward()): dfeat[l][b, y, x, c] += w_tap * dout[b, n, c] over the four taps of every
// point.  Same projection and tap arithmetic as the forward; float atomics, like torch's own grid_sampler backward
// (so, unlike every other gradient of the path, the summation order — the last bits — may differ run to run).
struct LookupGrads {
    float* d[4];   // channels-last (B, H, W, C) per level, zero-initialised by the caller
};

__global__ __launch_bounds__(256) void ray_lookup_bwd_kernel(const float* __restrict__ geom,
                                                             const float* __restrict__ coef,
                                                             const float* __restrict__ K, LookupArgs a, LookupGrads gr,
                                                             const float* __restrict__ dout, int N, int T) {
    const int tile = blockIdx.x % T, b = blockIdx.x / T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Ct = a.c_total, C4 = Ct / 4;
    const float cin = coef ? coef[4 * b + 2] : 1.0f;
    const float* Kb = K + (size_t)b * 9;
    constexpr int MAXCH = 4;
    int lvl[MAXCH], coff[MAXCH];
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) {
        const int ch = (c * 64 + lane) * 4;
        int l = 0, base = 0;
        while (l + 1 < a.n_levels && ch >= base + a.C[l]) {
            base += a.C[l];
            ++l;
        }
        lvl[c] = l;
        coff[c] = ch - base;
    }
    const int m0 = tile * LOOKUP_ROWS, m1 = min(N, m0 + LOOKUP_ROWS);
    for (int m = m0 + wave; m < m1; m += 4) {
        const float* gp = geom + ((size_t)b * N + m) * 3;
        float u, v;
        project_uv(cin * gp[0], cin * gp[1], cin * gp[2], Kb, a.reparam_kind, a.rp_mean, a.rp_std, a.logit_scale, u, v);
        Taps tp[4];
#pragma unroll
        for (int l = 0; l < 4; ++l)
            if (l < a.n_levels) tp[l] = bilinear_taps(u, v, a.H[l], a.W[l]);
        const float* grow = dout + ((size_t)b * N + m) * Ct;
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) {
            const int c4 = c * 64 + lane;
            if (c4 >= C4) continue;
            const int l = lvl[c];
            Taps t = tp[0];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (l == q) t = tp[q];
            const int Hh = a.H[l], Ww = a.W[l], Cl = a.C[l];
            float* fb = gr.d[l] + (size_t)b * Hh * Ww * Cl + coff[c];
            const int x1 = t.x0 + 1, y1 = t.y0 + 1;
            const float wx0 = (float)x1 - t.ix, wy0 = (float)y1 - t.iy;
            const float wx1 = t.ix - (float)t.x0, wy1 = t.iy - (float)t.y0;
            const bool bx0 = t.x0 >= 0 && t.x0 <= Ww - 1, bx1 = x1 >= 0 && x1 <= Ww - 1;
            const bool by0 = t.y0 >= 0 && t.y0 <= Hh - 1, by1 = y1 >= 0 && y1 <= Hh - 1;
            const f32x4 g4 = *reinterpret_cast<const f32x4*>(grow + c4 * 4);
            auto add = [&](bool ok, int yy, int xx, float w) {
                if (!ok) return;
                float* p = fb + ((size_t)yy * Ww + xx) * Cl;
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(p + e, g4[e] * w);
            };
            add(bx0 && by0, t.y0, t.x0, wx0 * wy0);
            add(bx1 && by0, t.y0, x1, wx1 * wy0);
            add(bx0 && by1, y1, t.x0, wx0 * wy1);
            add(bx1 && by1, y1, x1, wx1 * wy1);
        }
    }
}

// ---- the same gradient WITHOUT float atomics: sort, then gather.
// The atomic form issues 4 taps x sum(C_l) device-scope float atomics per point (264 M at B = 48, N = 2048, 672 channels:
// 3.3 ms, 80 G atomics/s — the rate of the memory-side atomic units, since the blocks of one image run on all XCDs).  Here
// one block per (image, level) builds the 4 N (texel, point, tap) entries, sorts them by texel in LDS (bitonic, key =
// texel << 14 | entry: ties keep point order, so the summation order is FIXED — this gradient is bit-reproducible too), and
// writes the entry list, the tap weights and the per-texel list starts; a second kernel gives every texel C_l / 4 threads
// that walk the texel's list and gather w * dout rows: every texel is written exactly once (no zero fill, no atomics).
constexpr int SORT_NT = 512;
constexpr unsigned SORT_INVALID = 0xffffffffu;

struct LookupSortWs {
    int* ent;      // (B, L, P) sorted entry ids (4 m + tap); the first start[HW] are valid
    float* wts;    // (B, L, P) tap weight per entry id
    int* start;    // (B, L, SW) list start per texel (HW_l + 1 used)
    int P, SW;
};

__global__ __launch_bounds__(SORT_NT) void lookup_sort_kernel(const float* __restrict__ geom, const float* __restrict__ coef,
                                                               const float* __restrict__ K, LookupArgs a, LookupSortWs ws,
                                                               int N) {
    extern __shared__ unsigned skeys[];   // P
    const int l = blockIdx.x % a.n_levels, b = blockIdx.x / a.n_levels;
    const int P = ws.P, Hh = a.H[l], Ww = a.W[l], HW = Hh * Ww;
    const float cin = coef ? coef[4 * b + 2] : 1.0f;
    const float* Kb = K + (size_t)b * 9;
    float* wts = ws.wts + ((size_t)b * a.n_levels + l) * P;
    int* ent = ws.ent + ((size_t)b * a.n_levels + l) * P;
    int* start = ws.start + ((size_t)b * a.n_levels + l) * ws.SW;
    for (int m = threadIdx.x; m < P / 4; m += SORT_NT) {
        unsigned key[4] = {SORT_INVALID, SORT_INVALID, SORT_INVALID, SORT_INVALID};
        float w[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < N) {
            const float* gp = geom + ((size_t)b * N + m) * 3;
            float u, v;
            project_uv(cin * gp[0], cin * gp[1], cin * gp[2], Kb, a.reparam_kind, a.rp_mean, a.rp_std, a.logit_scale, u, v);
            const Taps t = bilinear_taps(u, v, Hh, Ww);
            const int x1 = t.x0 + 1, y1 = t.y0 + 1;
            const float wx0 = (float)x1 - t.ix, wy0 = (float)y1 - t.iy;
            const float wx1 = t.ix - (float)t.x0, wy1 = t.iy - (float)t.y0;
            const bool bx0 = t.x0 >= 0 && t.x0 <= Ww - 1, bx1 = x1 >= 0 && x1 <= Ww - 1;
            const bool by0 = t.y0 >= 0 && t.y0 <= Hh - 1, by1 = y1 >= 0 && y1 <= Hh - 1;
            const bool ok[4] = {bx0 && by0, bx1 && by0, bx0 && by1, bx1 && by1};
            const int tx[4] = {t.x0, x1, t.x0, x1}, ty[4] = {t.y0, t.y0, y1, y1};
            const float ww[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ok[q]) {
                    key[q] = ((unsigned)(ty[q] * Ww + tx[q]) << 14) | (unsigned)(4 * m + q);
                    w[q] = ww[q];
                }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            skeys[4 * m + q] = key[q];
            wts[4 * m + q] = w[q];
        }
    }
    __syncthreads();
    for (unsigned k = 2; k <= (unsigned)P; k <<= 1)
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            for (unsigned i = threadIdx.x; i < (unsigned)P; i += SORT_NT) {
                const unsigned o = i ^ j;
                if (o > i) {
                    const unsigned x = skeys[i], y = skeys[o];
                    if ((x > y) == ((i & k) == 0)) {
                        skeys[i] = y;
                        skeys[o] = x;
                    }
                }
            }
            __syncthreads();
        }
    // entry list + list starts: position i opens the lists of the texels (t_prev, t_i] (empty ones included)
    for (int i = threadIdx.x; i < P; i += SORT_NT) {
        const unsigned key = skeys[i];
        const int ti = key == SORT_INVALID ? HW : (int)(key >> 14);
        int tp = -1;
        if (i > 0) {
            const unsigned kp = skeys[i - 1];
            tp = kp == SORT_INVALID ? HW : (int)(kp >> 14);
        }
        if (key != SORT_INVALID) ent[i] = (int)(key & 0x3fffu);
        for (int t = tp + 1; t <= ti; ++t) start[t] = i;
        if (i == P - 1 && key != SORT_INVALID)
            for (int t = ti + 1; t <= HW; ++t) start[t] = P;
    }
}

// dfeat[l][b, t, :] = sum over the texel's entries of w * dout[b, m, off_l : off_l + C_l];  C_l / 4 threads per texel
__global__ __launch_bounds__(256) void lookup_gather_bwd_kernel(const float* __restrict__ dout, LookupSortWs ws, float* __restrict__ dfeat,
                                                                int l, int n_levels, int HW, int Cl, int coff, int Ct, int N,
                                                                size_t ntasks) {
    const int tpt = Cl / 4, per = 256 / tpt;
    const int tk = threadIdx.x / tpt, c = (threadIdx.x % tpt) * 4;
    const size_t task = (size_t)blockIdx.x * per + tk;
    if (tk >= per || task >= ntasks) return;
    const int t = (int)(task % HW);
    const size_t b = task / HW;
    const size_t wb = (b * n_levels + l) * ws.P;
    const int* ent = ws.ent + wb;
    const float* wts = ws.wts + wb;
    const int* start = ws.start + (b * n_levels + l) * ws.SW;
    const int s = start[t], e = start[t + 1];
    const float* gb = dout + b * N * (size_t)Ct + coff + c;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int i = s;
    for (; i + 4 <= e; i += 4) {   // four independent gathers in flight, added in list order
        int en[4];
        f32x4 g[4];
        float w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) en[q] = ent[i + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w[q] = wts[en[q]];
            g[q] = *reinterpret_cast<const f32x4*>(gb + (size_t)(en[q] >> 2) * Ct);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += g[q] * w[q];
    }
    for (; i < e; ++i) {
        const int en = ent[i];
        acc += *reinterpret_cast<const f32x4*>(gb + (size_t)(en >> 2) * Ct) * wts[en];
    }
    *reinterpret_cast<f32x4*>(dfeat + task * Cl + c) = acc;
}

// integer tap indices + fractional weights for given uv (the bit-exact part, testable in isolation)
__global__ void bilinear_taps_kernel(const float* __restrict__ uv, int Hh, int Ww, int* __restrict__ x0,
                                     int* __restrict__ y0, float* __restrict__ wx1, float* __restrict__ wy1, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Taps t = bilinear_taps(uv[2 * i], uv[2 * i + 1], Hh, Ww);
    x0[i] = t.x0;
    y0[i] = t.y0;
    wx1[i] = t.ix - (float)t.x0;
    wy1[i] = t.iy - (float)t.y0;
}

// the fused lookup's coordinate chain on its own (diagnostics / parity tests): geometry -> reparametrisation -> projection -> taps, through
// the SAME device functions ray_lookup_kernel calls (project_uv, bilinear_taps), one thread per point.  uv (B, N, 2); per level l:
// x0 / y0 (L, B, N) int32 and the fractional weights wx1 = ix - x0, wy1 = iy - y0 (L, B, N)
__global__ void ray_lookup_taps_kernel(const float* __restrict__ geom, const float* __restrict__ coef, const float* __restrict__ K, LookupArgs a,
                                       float* __restrict__ uv, int* __restrict__ x0, int* __restrict__ y0, float* __restrict__ wx1,
                                       float* __restrict__ wy1, int B, int N) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)B * N;
    if (i >= total) return;
    const int b = (int)(i / N);
    const float cin = coef ? coef[4 * b + 2] : 1.0f;
    const float* gp = geom + i * 3;
    float u, v;
    project_uv(cin * gp[0], cin * gp[1], cin * gp[2], K + (size_t)b * 9, a.reparam_kind, a.rp_mean, a.rp_std, a.logit_scale, u, v);
    uv[2 * i] = u;
    uv[2 * i + 1] = v;
    for (int l = 0; l < a.n_levels; ++l) {
        const Taps t = bilinear_taps(u, v, a.H[l], a.W[l]);
        x0[l * total + i] = t.x0;
        y0[l * total + i] = t.y0;
        wx1[l * total + i] = t.ix - (float)t.x0;
        wy1[l * total + i] = t.iy - (float)t.y0;
    }
}

// (B, C, H, W) -> (B, H, W, C) through a 32x33 LDS tile (both sides coalesced)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, p = p0 + tx;
        tile[j][tx] = (c < C && p < HW) ? src[((size_t)b * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int p = p0 + j, c = c0 + tx;
        if (c < C && p < HW) dst[((size_t)b * HW + p) * C + c] = tile[tx][j];
    }
}

}  // namespace

int lookup_row_tile() { return LOOKUP_ROWS; }

int ray_lookup_launch(const float* geom, const float* coef, const float* K, const LookupArgs& a, float* out,
                      float* stats, int B, int N, hipStream_t st) {
    if (a.n_levels < 1 || a.n_levels > 4 || a.c_total > 1024 || a.c_total % 4) return -8;
    int tot = 0;
    for (int l = 0; l < a.n_levels; ++l) {
        if (a.C[l] % 4) return -8;
        tot += a.C[l];
    }
    if (tot != a.c_total) return -8;
    const int T = (N + LOOKUP_ROWS - 1) / LOOKUP_ROWS;
    if (a.out_f16) {
        if (a.texel_f16) hipLaunchKernelGGL((ray_lookup_kernel<true, true>), dim3(B * T), dim3(256), 0, st, geom, coef, K, a, out, stats, N, T);
        else hipLaunchKernelGGL((ray_lookup_kernel<false, true>), dim3(B * T), dim3(256), 0, st, geom, coef, K, a, out, stats, N, T);
        return (int)hipGetLastError();
    }
    if (a.texel_f16) hipLaunchKernelGGL(ray_lookup_kernel<true>, dim3(B * T), dim3(256), 0, st, geom, coef, K, a, out, stats, N, T);
    else hipLaunchKernelGGL(ray_lookup_kernel<false>, dim3(B * T), dim3(256), 0, st, geom, coef, K, a, out, stats, N, T);
    return (int)hipGetLastError();
}

namespace {
__global__ void cast_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, size_t n4, size_t n) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
        reinterpret_cast<f16x4*>(dst)[i] = f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n & ~size_t(3)) + threadIdx.x] = (_Float16)src[(n & ~size_t(3)) + threadIdx.x];
}
}  // namespace

int cast_f16_launch(const float* src, void* dst, size_t n, hipStream_t st) {
    if (!n) return 0;
    const size_t n4 = n / 4;
    const unsigned blocks = (unsigned)std::min<size_t>((n4 + 255) / 256 + 1, 4096);
    hipLaunchKernelGGL(cast_f16_kernel, dim3(blocks), dim3(256), 0, st, src, static_cast<_Float16*>(dst), n4, n);
    return (int)hipGetLastError();
}

int ray_lookup_dgeom_launch(const float* geom, const float* K, const LookupArgs& a, const float* dout, float* dgeom, float* dKpart, int B,
                            int N, hipStream_t st) {
    if (a.n_levels < 1 || a.n_levels > 4 || a.c_total > 1024 || a.c_total % 4) return -8;
    for (int l = 0; l < a.n_levels; ++l)
        if (a.C[l] % 4) return -8;
    const int T = (N + LOOKUP_ROWS - 1) / LOOKUP_ROWS;
    hipLaunchKernelGGL(ray_lookup_dgeom_kernel, dim3(B * T), dim3(256), 0, st, geom, K, a, dout, dgeom, dKpart, N, T);
    return (int)hipGetLastError();
}

int ray_lookup_bwd_launch(const float* geom, const float* coef, const float* K, const LookupArgs& a,
                          float* const* dfeat, const float* dout, int B, int N, hipStream_t st) {
    if (a.n_levels < 1 || a.n_levels > 4 || a.c_total > 1024 || a.c_total % 4) return -8;
    LookupGrads gr;
    for (int l = 0; l < 4; ++l) gr.d[l] = l < a.n_levels ? dfeat[l] : nullptr;
    const int T = (N + LOOKUP_ROWS - 1) / LOOKUP_ROWS;
    hipLaunchKernelGGL(ray_lookup_bwd_kernel, dim3(B * T), dim3(256), 0, st, geom, coef, K, a, gr, dout, N, T);
    return (int)hipGetLastError();
}

// sorted form: N <= 4096 points, H W <= 2^17 texels per level, C_l % 4 == 0 and C_l <= 1024
static int sort_P(int N) {
    int P = 1024;
    while (P < 4 * N) P <<= 1;
    return P;
}
static int sort_SW(const LookupArgs& a) {
    int m = 0;
    for (int l = 0; l < a.n_levels; ++l) m = std::max(m, a.H[l] * a.W[l]);
    return (m + 1 + 3) & ~3;
}
bool ray_lookup_bwd_sorted_supported(const LookupArgs& a, int N) {
    if (N < 1 || N > 4096 || a.n_levels < 1 || a.n_levels > 4) return false;
    for (int l = 0; l < a.n_levels; ++l)
        if ((size_t)a.H[l] * a.W[l] > (1u << 17) || (a.C[l] & 3) || a.C[l] > 1024) return false;
    return true;
}
size_t ray_lookup_bwd_sorted_ws_bytes(const LookupArgs& a, int B, int N) {
    const size_t per = (size_t)B * a.n_levels;
    return per * sort_P(N) * 8 + per * sort_SW(a) * 4;
}
int ray_lookup_bwd_sorted_launch(const float* geom, const float* coef, const float* K, const LookupArgs& a, float* const* dfeat,
                                 const float* dout, int B, int N, void* wsp, hipStream_t st) {
    if (!ray_lookup_bwd_sorted_supported(a, N)) return -8;
    LookupSortWs ws;
    ws.P = sort_P(N);
    ws.SW = sort_SW(a);
    const size_t per = (size_t)B * a.n_levels;
    ws.ent = static_cast<int*>(wsp);
    ws.wts = reinterpret_cast<float*>(ws.ent + per * ws.P);
    ws.start = reinterpret_cast<int*>(ws.wts + per * ws.P);
    const size_t lds = (size_t)ws.P * sizeof(unsigned);
    static size_t attr = 0;
    if (lds > attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lookup_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL(lookup_sort_kernel, dim3(B * a.n_levels), dim3(SORT_NT), lds, st, geom, coef, K, a, ws, N);
    int coff = 0;
    for (int l = 0; l < a.n_levels; ++l) {
        const int HW = a.H[l] * a.W[l], per_blk = 256 / (a.C[l] / 4);
        const size_t ntasks = (size_t)B * HW;
        hipLaunchKernelGGL(lookup_gather_bwd_kernel, dim3((unsigned)((ntasks + per_blk - 1) / per_blk)), dim3(256), 0, st, dout, ws, dfeat[l],
                           l, a.n_levels, HW, a.C[l], coff, a.c_total, N, ntasks);
        coff += a.C[l];
    }
    return (int)hipGetLastError();
}

int bilinear_taps_launch(const float* uv, int Hh, int Ww, int* x0, int* y0, float* wx1, float* wy1, size_t n,
                         hipStream_t st) {
    hipLaunchKernelGGL(bilinear_taps_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, uv, Hh, Ww, x0, y0,
                       wx1, wy1, n);
    return (int)hipGetLastError();
}

int ray_lookup_taps_launch(const float* geom, const float* coef, const float* K, const LookupArgs& a, float* uv, int* x0, int* y0, float* wx1,
                           float* wy1, int B, int N, hipStream_t st) {
    const size_t total = (size_t)B * N;
    hipLaunchKernelGGL(ray_lookup_taps_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, geom, coef, K, a, uv, x0, y0, wx1, wy1, B, N);
    return (int)hipGetLastError();
}

int nchw_to_nhwc_launch(const float* src, float* dst, int B, int C, int Hh, int Ww, hipStream_t st) {
    const int HW = Hh * Ww;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, st, src, dst, C, HW);
    return (int)hipGetLastError();
}
