// Backward of the channels-last ConvNeXt conditioner (gfx950) — the HBM-bound pieces; the pointwise linears' dX / dW run on
// the GEMMs of the training path (autograd.py: LinearFn).
//
// The reference trains the conditioner end to end with the denoiser: `ConvNeXtExtractor` is an ordinary sub-module of
// `Diffusion` (models/feature_pyramid.py:28-73; diffusion.py:203-222 `training_step` / `configure_optimizers` over
// `self.parameters()`), so the pyramid gradient the projective lookup returns has to reach the stem.  Per CNBlock
//   x -> z = dwconv7(x) + b -> y = LN(z) -> u = W1 y + b1 -> h = GELU(u) -> x + (ls W2) h + ls b2
// this file provides: LayerNorm backward from z (statistics recomputed: two reads of z instead of a stored mean / rstd
// pair per texel), the depthwise convolution's weight gradient (its input gradient is the forward kernel on the reversed
// taps, convnext.hip `ln_w == null`), GELU forward / backward, and the stem's 4 x 4 patch gather (its weight gradient is a
// GEMM over the patch matrix).  Column reductions (LayerNorm weight / bias, conv bias, tap gradients) are per-block partials
// summed in a fixed order by reduce_batch: the gradients are bit-reproducible run to run.
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float lanes_sum(float v, int width) {   // over `width` consecutive lanes (power of two <= 64)
    for (int o = 1; o < width; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- LayerNorm backward over the channels of a texel.  y = (z - mean) rstd g + b:
//   xh = (z - mean) rstd,  gy = dy g,  dz = rstd (gy - mean_c(gy) - xh mean_c(gy xh)),  dg += dy xh,  db += dy,  dzsum += dz
// (dzsum is the bias gradient of the convolution that produced z).  L = C / 12 lanes share a texel (8 / 16 / 32: a power of
// two, sums by lane shuffles, no LDS in the loop); lane q owns the 16-byte chunks q, q + L, q + 2L.  patch2: dy is laid out
// as the 2 x 2 patch matrix of the downsampling convolution, (B, H/2, W/2, (dy, dx, c)) — the transpose of ln_patch2_kernel.
template <int C>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                     const float* __restrict__ ln_w, float* __restrict__ dz,
                                                     float* __restrict__ parts, size_t npix, int H, int W, float eps, int patch2,
                                                     int iters) {
    constexpr int L = C / 12, PIX = 256 / L;
    static_assert(L == 8 || L == 16 || L == 32, "C = 96, 192, 384");
    __shared__ float red[3 * PIX * C];   // 36 KiB for every C
    const int pl = threadIdx.x / L, q = threadIdx.x % L;
    f32x4 g4[3], ag[3], ab[3], az[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        g4[j] = *reinterpret_cast<const f32x4*>(ln_w + 4 * (q + L * j));
        ag[j] = ab[j] = az[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int it = 0; it < iters; ++it) {
        const size_t p = ((size_t)blockIdx.x * iters + it) * PIX + pl;
        const bool live = p < npix;
        f32x4 zv[3], dv[3];
        size_t dyo = p * C;
        if (patch2 && live) {
            const int wx = (int)(p % W), hy = (int)((p / W) % H);
            const size_t b = p / ((size_t)W * H);
            dyo = (((b * (H / 2) + hy / 2) * (W / 2) + wx / 2) * 4 + ((hy & 1) * 2 + (wx & 1))) * C;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = 4 * (q + L * j);
            zv[j] = live ? *reinterpret_cast<const f32x4*>(z + p * C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            dv[j] = live ? *reinterpret_cast<const f32x4*>(dy + dyo + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) s1 += (zv[j][0] + zv[j][1]) + (zv[j][2] + zv[j][3]);
        const float mean = lanes_sum(s1, L) / C;
        float s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            zv[j] = zv[j] - mean;
            const f32x4 d = zv[j] * zv[j];
            s2 += (d[0] + d[1]) + (d[2] + d[3]);
        }
        const float rstd = rsqrtf(lanes_sum(s2, L) / C + eps);
        float sg = 0.f, sgx = 0.f;
        f32x4 gy[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            zv[j] = zv[j] * rstd;   // xh
            gy[j] = dv[j] * g4[j];
            const f32x4 e = gy[j] * zv[j];
            sg += (gy[j][0] + gy[j][1]) + (gy[j][2] + gy[j][3]);
            sgx += (e[0] + e[1]) + (e[2] + e[3]);
        }
        sg = lanes_sum(sg, L) / C;
        sgx = lanes_sum(sgx, L) / C;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const f32x4 o = (gy[j] - sg - zv[j] * sgx) * rstd;
            if (live) *reinterpret_cast<f32x4*>(dz + p * C + 4 * (q + L * j)) = o;
            ag[j] += dv[j] * zv[j];   // dead texels carry zeros
            ab[j] += dv[j];
            az[j] += live ? o : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    // the block's column partials: (3, C) = dg | db | dzsum, summed over its texel slots in slot order
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = 4 * (q + L * j);
        *reinterpret_cast<f32x4*>(red + (0 * PIX + pl) * C + c) = ag[j];
        *reinterpret_cast<f32x4*>(red + (1 * PIX + pl) * C + c) = ab[j];
        *reinterpret_cast<f32x4*>(red + (2 * PIX + pl) * C + c) = az[j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C; i += 256) {
        const int k = i / C, c = i % C;
        float s = 0.f;
        for (int s_ = 0; s_ < PIX; ++s_) s += red[(k * PIX + s_) * C + c];
        parts[(size_t)blockIdx.x * 3 * C + i] = s;
    }
}

// ---- depthwise 7 x 7 weight gradient: dW[tap = (dy, dx)][c] = sum over texels dz[b, y, x, c] x[b, y + dy - 3, x + dx - 3, c].
// Same thread layout as the forward kernel (a 16-byte channel chunk of four texels adjacent in W; a window row is ten texel
// loads), one block per (strip, tap row dy): 7 four-channel accumulators per thread, carried over the block's `iters` batches of groups, then
// summed over the block's groups through LDS.  parts: (gridDim.x, 49, C).
template <int C>
__global__ __launch_bounds__(256) void dwconv7_dw_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                         float* __restrict__ parts, int B, int H, int W, int iters) {
    constexpr int TPP = C / 4, PG = 256 / TPP, TX = 4;
    __shared__ float red[PG * 7 * C];
    const int pg = threadIdx.x / TPP, t = threadIdx.x % TPP, c = 4 * t;
    // the seven tap-row blocks of a texel strip read the same dz and neighbouring rows of x: consecutive virtual ids, one XCD
    const int vb = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int bx = vb / 7, tr = vb % 7;
    const int dy = tr - 3;
    const int GR = (W + TX - 1) / TX;
    const size_t ngroups = (size_t)B * H * GR;
    f32x4 acc[7];
#pragma unroll
    for (int d = 0; d < 7; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const size_t gid = ((size_t)bx * iters + it) * PG + pg;
        if (pg >= PG || gid >= ngroups) continue;
        const int wx0 = (int)(gid % GR) * TX, hy = (int)((gid / GR) % H);
        const size_t b = gid / ((size_t)GR * H);
        const int yy = hy + dy;
        if (yy < 0 || yy >= H) continue;
        const float* dp = dz + ((b * H + hy) * W + wx0) * C + c;
        const float* xr = x + (b * H + yy) * (size_t)W * C + c;
        f32x4 dv[TX], xv[TX + 6];
#pragma unroll
        for (int tx = 0; tx < TX; ++tx)
            dv[tx] = wx0 + tx < W ? *reinterpret_cast<const f32x4*>(dp + (size_t)tx * C) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < TX + 6; ++j) {
            const int xx = wx0 - 3 + j;
            xv[j] = (xx >= 0 && xx < W) ? *reinterpret_cast<const f32x4*>(xr + (size_t)xx * C) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int d = 0; d < 7; ++d)
#pragma unroll
            for (int tx = 0; tx < TX; ++tx) acc[d] += dv[tx] * xv[tx + d];
    }
    if (pg < PG) {
#pragma unroll
        for (int d = 0; d < 7; ++d) *reinterpret_cast<f32x4*>(red + (pg * 7 + d) * C + c) = acc[d];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 7 * C; i += 256) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < PG; ++g) s += red[g * 7 * C + i];
        parts[((size_t)bx * 49 + tr * 7) * C + i] = s;
    }
}

// ---- GELU (erf form, nn.GELU()): the same expression as the GEMM epilogue's act 4 (common.h), and its derivative
//   d/du [u Phi(u)] = Phi(u) + u phi(u)
__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ u, float* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(u)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ u, const float* __restrict__ dy,
                                                       float* __restrict__ du, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(u)[i], g = reinterpret_cast<const f32x4*>(dy)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float cdf = 0.5f * (1.0f + erff(v[e] * 0.70710678118654752f));
            const float pdf = 0.3989422804014327f * __expf(-0.5f * v[e] * v[e]);
            o[e] = g[e] * (cdf + v[e] * pdf);
        }
        reinterpret_cast<f32x4*>(du)[i] = o;
    }
}

// ---- the stem's patch matrix: out[(b, ho, wo), (ci, dy, dx)] = x[b, ci, 4 ho + dy, 4 wo + dx] — the k order of
// conv.weight.reshape(C, 48), so the stem's weight gradient is dz^T @ out.
__global__ __launch_bounds__(256) void im2col4_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int H, int W) {
    const int Ho = H / 4, Wo = W / 4;
    const size_t total = (size_t)B * Ho * Wo * 12;   // one 16-byte piece (ci, dy, dx = 0..3) per index
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int piece = (int)(i % 12), ci = piece >> 2, dy = piece & 3;
        const size_t p = i / 12;
        const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho);
        const size_t b = p / ((size_t)Wo * Ho);
        reinterpret_cast<f32x4*>(out)[i] = *reinterpret_cast<const f32x4*>(x + ((b * 3 + ci) * H + 4 * ho + dy) * W + 4 * wo);
    }
}

int ln_bwd_plan(size_t npix, int C, int* iters) {
    const int pix = 256 / (C / 12);
    const size_t batches = (npix + pix - 1) / pix;
    int it = 1;
    while (batches / it > 1024) it *= 2;
    *iters = it;
    return (int)((batches + it - 1) / it);
}
int dw_plan(int B, int H, int W, int C, int* iters) {
    const int pg = 256 / (C / 4);
    const size_t ngroups = (size_t)B * H * ((W + 3) / 4), batches = (ngroups + pg - 1) / pg;
    int it = 1;
    while (batches / it > 192) it *= 2;   // x 7 tap rows: ~1300 blocks
    *iters = it;
    return (int)((batches + it - 1) / it);
}

}  // namespace

int cnx_ln_bwd_blocks(int B, int H, int W, int C) {
    if (C != 96 && C != 192 && C != 384) return -9;
    int it;
    return ln_bwd_plan((size_t)B * H * W, C, &it);
}
int cnx_ln_bwd_launch(const float* z, const float* dy, const float* ln_w, float* dz, float* parts, int B, int H, int W, int C,
                      float eps, int patch2, hipStream_t st) {
    if ((C != 96 && C != 192 && C != 384) || (patch2 && ((H & 1) || (W & 1)))) return -9;
    const size_t npix = (size_t)B * H * W;
    int iters;
    const int grid = ln_bwd_plan(npix, C, &iters);
    switch (C) {
        case 96: hipLaunchKernelGGL((ln_bwd_kernel<96>), dim3(grid), dim3(256), 0, st, z, dy, ln_w, dz, parts, npix, H, W, eps, patch2, iters); break;
        case 192: hipLaunchKernelGGL((ln_bwd_kernel<192>), dim3(grid), dim3(256), 0, st, z, dy, ln_w, dz, parts, npix, H, W, eps, patch2, iters); break;
        default: hipLaunchKernelGGL((ln_bwd_kernel<384>), dim3(grid), dim3(256), 0, st, z, dy, ln_w, dz, parts, npix, H, W, eps, patch2, iters); break;
    }
    return (int)hipGetLastError();
}
int cnx_dwconv_dw_blocks(int B, int H, int W, int C) {
    if (C != 96 && C != 192 && C != 384) return -9;
    int it;
    return dw_plan(B, H, W, C, &it);
}
int cnx_dwconv_dw_launch(const float* x, const float* dz, float* parts, int B, int H, int W, int C, hipStream_t st) {
    if (C != 96 && C != 192 && C != 384) return -9;
    int iters;
    const int gx = dw_plan(B, H, W, C, &iters);
    const dim3 grid(gx * 7);
    switch (C) {
        case 96: hipLaunchKernelGGL((dwconv7_dw_kernel<96>), grid, dim3(256), 0, st, x, dz, parts, B, H, W, iters); break;
        case 192: hipLaunchKernelGGL((dwconv7_dw_kernel<192>), grid, dim3(256), 0, st, x, dz, parts, B, H, W, iters); break;
        default: hipLaunchKernelGGL((dwconv7_dw_kernel<384>), grid, dim3(256), 0, st, x, dz, parts, B, H, W, iters); break;
    }
    return (int)hipGetLastError();
}
int gelu_launch(const float* u, float* y, size_t n, hipStream_t st) {
    if (n & 3) return -9;
    const size_t n4 = n >> 2;
    const unsigned grid = (unsigned)std::min<size_t>((n4 + 255) / 256, 8192);
    hipLaunchKernelGGL(gelu_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, u, y, n4);
    return (int)hipGetLastError();
}
int gelu_bwd_launch(const float* u, const float* dy, float* du, size_t n, hipStream_t st) {
    if (n & 3) return -9;
    const size_t n4 = n >> 2;
    const unsigned grid = (unsigned)std::min<size_t>((n4 + 255) / 256, 8192);
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, u, dy, du, n4);
    return (int)hipGetLastError();
}
int cnx_im2col4_launch(const float* x, float* out, int B, int H, int W, hipStream_t st) {
    if ((H & 3) || (W & 3)) return -9;
    const size_t total = (size_t)B * (H / 4) * (W / 4) * 12;
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(im2col4_kernel, dim3(grid ? grid : 1), dim3(256), 0, st, x, out, B, H, W);
    return (int)hipGetLastError();
}
