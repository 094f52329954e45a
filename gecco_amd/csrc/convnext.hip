// ConvNeXt conditioner on the device, channels-last end to end (gfx950) — SURVEY.md 8(f) row 2.
//
// Reference: `ConvNeXtExtractor` (models/feature_pyramid.py:28-73) wraps torchvision's ConvNeXt-T/S stages
//   stem      Conv2d(3, 96, k4, s4) + LayerNorm2d(eps 1e-6)
//   CNBlock   x + layer_scale * Linear(4C -> C)(GELU(Linear(C -> 4C)(LayerNorm(dwconv7x7(x)))))      (stochastic depth off)
//   downsample LayerNorm2d + Conv2d(C, 2C, k2, s2)
// and hands NCHW maps to the lookup, which the reference re-runs per `upsample` evaluation (diffusion.py:415-421).  Here
// every activation lives as (B, H, W, C) fp32 — one texel's channels contiguous, what ray_lookup_kernel gathers — so the
// pyramid needs no NCHW -> NHWC transpose; the two pointwise linears of a block (96 % of its FLOPs) are the fused GEMM
// of gemm_f32*.hip on rows = B H W (bias + GELU, or bias + residual with layer_scale folded into the weights), and
// this file holds the HBM-bound rest: patchify stem, depthwise 7x7 + LayerNorm, LayerNorm + 2x2 patch gather, the
// layer_scale fold.  LayerNorm statistics are per texel over its channels (fp32 sums over <= 384 values).
#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float group_sum(float v, int width) {   // sum over `width` consecutive lanes (power of two <= 64)
    for (int o = 1; o < width; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- stem: out[b, h, w, :] = LN(W x_patch + bias), x NCHW (B, 3, H, W), patch 4 x 4 stride 4, weight (C, 3, 4, 4).
// 8 lanes per output texel, lane q computes channels q, q + 8, ...; the 48 patch values sit in LDS.
template <int C>
__global__ __launch_bounds__(256) void stem_conv_ln_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ ln_w,
                                                           const float* __restrict__ ln_b, float* __restrict__ out,
                                                           float* __restrict__ zout, int B, int H, int W, float eps) {
    constexpr int CPL = C / 8;                 // channels per lane
    __shared__ float sw[48 * C];               // [k][c]: lanes of a texel read consecutive c
    __shared__ float sp[32][48];
    const int Ho = H / 4, Wo = W / 4;
    for (int i = threadIdx.x; i < 48 * C; i += 256) sw[(i % 48) * C + i / 48] = w[i];   // w[c][k] -> sw[k][c]
    const size_t npix = (size_t)B * Ho * Wo;
    const size_t p0 = (size_t)blockIdx.x * 32;
    for (int i = threadIdx.x; i < 32 * 48; i += 256) {
        const int pl = i / 48, k = i % 48;
        const size_t p = p0 + pl;
        float v = 0.f;
        if (p < npix) {
            const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), b = (int)(p / ((size_t)Wo * Ho));
            const int ci = k / 16, dy = (k / 4) & 3, dx = k & 3;
            v = x[(((size_t)b * 3 + ci) * H + 4 * ho + dy) * W + 4 * wo + dx];
        }
        sp[pl][k] = v;
    }
    __syncthreads();
    const int pl = threadIdx.x >> 3, q = threadIdx.x & 7;
    const size_t p = p0 + pl;
    float acc[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) acc[j] = bias[q + 8 * j];
    for (int k = 0; k < 48; ++k) {
        const float v = sp[pl][k];
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j] += v * sw[k * C + q + 8 * j];
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) s1 += acc[j];
    s1 = group_sum(s1, 8);
    const float mean = s1 / C;
#pragma unroll
    for (int j = 0; j < CPL; ++j) s2 += (acc[j] - mean) * (acc[j] - mean);
    s2 = group_sum(s2, 8);
    const float rstd = rsqrtf(s2 / C + eps);
    if (p < npix) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const int c = q + 8 * j;
            out[p * C + c] = (acc[j] - mean) * rstd * ln_w[c] + ln_b[c];
            if (zout) zout[p * C + c] = acc[j];   // training: the LayerNorm's input, what its backward starts from
        }
    }
}

// ---- depthwise 7 x 7 (padding 3) + bias + LayerNorm over channels, channels-last.
// A thread owns one 16-byte channel chunk of a GROUP of TX = 4 texels adjacent in W: a row of the window needs 10 texel
// loads for the four outputs instead of 28, and one weight read per tap serves four texels.  The weights (C, 1, 7, 7) are
// given TAP-MAJOR, (49, C) = weight.reshape(C, 49).T, and staged once per block in LDS (a tap's 4 channels are one 16-byte
// read, the same address for every group of the block — a broadcast); a block walks `iters` batches of 256 / (C / 4) groups
// to amortise that.  (The first form read w[c][tap] — four strided scalar loads per tap and texel — and ran 40x off the HBM
// floor of the layer.)
template <int C>
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ ln_w,
                                                         const float* __restrict__ ln_b, float* __restrict__ out,
                                                         float* __restrict__ zout, int B, int H, int W, float eps, int iters,
                                                         const float* __restrict__ addp, int flip) {
    // raw mode extras: flip — the taps are staged reversed (w[48 - tap]: the convolution's input gradient from dz without a
    // flipped copy of the weights); addp — a tensor added to the result (the skip connection's gradient of a CNBlock)
    // zout: also the convolution's own output (training: the LayerNorm's input).  ln_w == null: no LayerNorm — out is the
    // convolution (bias may be null too): with the taps reversed this is the convolution's input gradient.
    constexpr int TPP = C / 4, PG = 256 / TPP, TX = 4, NPART = 4;
    static_assert(TPP % NPART == 0, "the LayerNorm partial sums split a group's threads in four");
    extern __shared__ __attribute__((aligned(16))) float cs[];
    float* wl = cs;                       // [49][C]
    float* red = wl + 49 * C;             // [TX][256] per-thread partials
    float* part = red + TX * 256;         // [PG][TX][NPART]
    const int pg = threadIdx.x / TPP, t = threadIdx.x % TPP, c = 4 * t;
    for (int i = threadIdx.x; i < 49 * C / 4; i += 256) {   // w arrives tap-major (49, C): a coalesced copy
        const int tap = i / (C / 4), c4 = i % (C / 4);
        reinterpret_cast<f32x4*>(wl)[i] = reinterpret_cast<const f32x4*>(w)[flip ? (48 - tap) * (C / 4) + c4 : i];
    }
    const f32x4 bias4 = pg < PG && bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 g4 = pg < PG && ln_w ? *reinterpret_cast<const f32x4*>(ln_w + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 b4 = pg < PG && ln_w ? *reinterpret_cast<const f32x4*>(ln_b + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int GR = (W + TX - 1) / TX;     // groups per image row
    const size_t ngroups = (size_t)B * H * GR;
    // neighbouring strips share their halo rows (a window is 7 rows high): consecutive strips on one XCD, its L2 serves them
    const int vbx = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        const size_t gid = ((size_t)vbx * iters + it) * PG + pg;
        const bool live = pg < PG && gid < ngroups;
        f32x4 acc[TX];
#pragma unroll
        for (int tx = 0; tx < TX; ++tx) acc[tx] = bias4;
        int wx0 = 0, hy = 0, b = 0;
        if (live) {
            wx0 = (int)(gid % GR) * TX;
            hy = (int)((gid / GR) % H);
            b = (int)(gid / ((size_t)GR * H));
            const float* xb = x + (size_t)b * H * W * C + c;
#pragma unroll 1
            for (int dy = -3; dy <= 3; ++dy) {
                const int yy = hy + dy;
                if (yy < 0 || yy >= H) continue;
                const float* xr = xb + (size_t)yy * W * C;
                f32x4 xv[TX + 6];
#pragma unroll
                for (int j = 0; j < TX + 6; ++j) {
                    const int xx = wx0 - 3 + j;
                    xv[j] = (xx >= 0 && xx < W) ? *reinterpret_cast<const f32x4*>(xr + (size_t)xx * C) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                const float* wr = wl + (dy + 3) * 7 * C + c;
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + dx * C);
#pragma unroll
                    for (int tx = 0; tx < TX; ++tx) acc[tx] += xv[tx + dx] * wv;
                }
            }
        }
        if (!ln_w) {   // grid-uniform
            if (live) {
                float* op = out + (((size_t)b * H + hy) * W + wx0) * C + c;
#pragma unroll
                for (int tx = 0; tx < TX; ++tx)
                    if (wx0 + tx < W) {
                        f32x4 r = acc[tx];
                        if (addp) r += *reinterpret_cast<const f32x4*>(addp + (op - out) + (size_t)tx * C);
                        *reinterpret_cast<f32x4*>(op + (size_t)tx * C) = r;
                    }
            }
            continue;
        }
        if (zout && live) {
            float* zp = zout + (((size_t)b * H + hy) * W + wx0) * C + c;
#pragma unroll
            for (int tx = 0; tx < TX; ++tx)
                if (wx0 + tx < W) *reinterpret_cast<f32x4*>(zp + (size_t)tx * C) = acc[tx];
        }
        // LayerNorm over each texel's C channels, two passes (mean, centred variance): per-thread partials in LDS, four
        // threads per (group, texel) add a quarter of them each, everyone combines the four quarters
        float mean[TX], rstd[TX];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int tx = 0; tx < TX; ++tx) {
                f32x4 d = acc[tx];
                if (pass == 1) { d = d - mean[tx]; d = d * d; }
                red[tx * 256 + threadIdx.x] = d[0] + d[1] + d[2] + d[3];
            }
            __syncthreads();
            if (pg < PG && t < TX * NPART) {
                const int tx = t % TX, pt = t / TX;
                float s = 0.f;
                const float* rp = red + tx * 256 + pg * TPP + pt * (TPP / NPART);
#pragma unroll 4
                for (int i = 0; i < TPP / NPART; ++i) s += rp[i];
                part[(pg * TX + tx) * NPART + pt] = s;
            }
            __syncthreads();
            if (pg < PG) {
#pragma unroll
                for (int tx = 0; tx < TX; ++tx) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(part + (pg * TX + tx) * NPART);
                    const float s = ((q[0] + q[1]) + q[2]) + q[3];
                    if (pass == 0) mean[tx] = s / C;
                    else rstd[tx] = rsqrtf(s / C + eps);
                }
            }
        }
        if (live) {
            float* op = out + (((size_t)b * H + hy) * W + wx0) * C + c;
#pragma unroll
            for (int tx = 0; tx < TX; ++tx)
                if (wx0 + tx < W) *reinterpret_cast<f32x4*>(op + (size_t)tx * C) = (acc[tx] - mean[tx]) * rstd[tx] * g4 + b4;
        }
    }
}

// ---- downsample front half: LayerNorm over channels of every input texel, written where the 2 x 2 stride-2 conv's
// GEMM reads it: out[b, h/2, w/2, (dy, dx, c)] (K = 4 C contiguous per output texel).
template <int C>
__global__ __launch_bounds__(256) void ln_patch2_kernel(const float* __restrict__ x, const float* __restrict__ ln_w,
                                                        const float* __restrict__ ln_b, float* __restrict__ out, int B, int H,
                                                        int W, float eps) {
    constexpr int TPP = C / 4, PIX = 256 / TPP;
    __shared__ float red[2][256];
    const int pl = threadIdx.x / TPP, t = threadIdx.x % TPP;
    const size_t npix = (size_t)B * H * W;
    const size_t p = (size_t)blockIdx.x * PIX + pl;
    const bool live = pl < PIX && p < npix;
    const int c = 4 * t;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (live) v = *reinterpret_cast<const f32x4*>(x + p * C + c);
    red[0][threadIdx.x] = v[0] + v[1] + v[2] + v[3];
    __syncthreads();
    float s1 = 0.f;
    if (pl < PIX)
        for (int i = 0; i < TPP; ++i) s1 += red[0][pl * TPP + i];
    const float mean = s1 / C;
    const f32x4 d = v - mean;
    red[1][threadIdx.x] = d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
    __syncthreads();
    float s2 = 0.f;
    if (pl < PIX)
        for (int i = 0; i < TPP; ++i) s2 += red[1][pl * TPP + i];
    const float rstd = rsqrtf(s2 / C + eps);
    if (live) {
        const int wx = (int)(p % W), hy = (int)((p / W) % H), b = (int)(p / ((size_t)W * H));
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(ln_w + c), b4 = *reinterpret_cast<const f32x4*>(ln_b + c);
        const size_t op = ((size_t)b * (H / 2) + hy / 2) * (W / 2) + wx / 2;
        *reinterpret_cast<f32x4*>(out + op * 4 * C + ((hy & 1) * 2 + (wx & 1)) * C + c) = d * rstd * g4 + b4;
    }
}

// W'[n, k] = s[n] W[n, k], b'[n] = s[n] b[n]   (layer_scale folded into pwconv2: x + ls * (W h + b) = x + W' h + b')
__global__ void fold_scale_kernel(const float* __restrict__ Wm, const float* __restrict__ b, const float* __restrict__ s,
                                  float* __restrict__ Wo, float* __restrict__ bo, int N, int K) {
    const size_t total = (size_t)N * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) Wo[i] = Wm[i] * s[i / K];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)N; i += (size_t)gridDim.x * blockDim.x) bo[i] = b[i] * s[i];
}

// backward of the layer_scale fold: given dW' (N, K), db' (N) of the folded weights W' = s W, b' = s b:
//   dW[n, k] = s[n] dW'[n, k],  db[n] = s[n] db'[n],  ds[n] = sum_k dW'[n, k] W[n, k] + db'[n] b[n]     (one block per row n)
__global__ __launch_bounds__(256) void fold_scale_bwd_kernel(const float* __restrict__ dWp, const float* __restrict__ dbp,
                                                             const float* __restrict__ Wm, const float* __restrict__ b,
                                                             const float* __restrict__ s, float* __restrict__ dW,
                                                             float* __restrict__ db, float* __restrict__ ds, int K) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    const float sn = s[n];
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float g = dWp[(size_t)n * K + k];
        dW[(size_t)n * K + k] = sn * g;
        acc += g * Wm[(size_t)n * K + k];
    }
    for (int o = 1; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        ds[n] = ((red[0] + red[1]) + red[2]) + red[3] + dbp[n] * b[n];
        db[n] = sn * dbp[n];
    }
}

}  // namespace

#define CNX_DISPATCH(KERNEL, C, grid, ...)                                                                      \
    switch (C) {                                                                                                \
        case 96: hipLaunchKernelGGL((KERNEL<96>), grid, dim3(256), 0, st, __VA_ARGS__); break;                  \
        case 192: hipLaunchKernelGGL((KERNEL<192>), grid, dim3(256), 0, st, __VA_ARGS__); break;                \
        case 384: hipLaunchKernelGGL((KERNEL<384>), grid, dim3(256), 0, st, __VA_ARGS__); break;                \
        case 768: return -9; /* the fourth stage is not part of the pyramid (n_stages <= 3 in every config) */  \
        default: return -9;                                                                                     \
    }

int cnx_stem_launch(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b, float* out, float* zout,
                    int B, int H, int W, int C, float eps, hipStream_t st) {
    if (C != 96 || (H & 3) || (W & 3)) return -9;
    const size_t npix = (size_t)B * (H / 4) * (W / 4);
    hipLaunchKernelGGL((stem_conv_ln_kernel<96>), dim3((unsigned)((npix + 31) / 32)), dim3(256), 0, st, x, w, bias, ln_w, ln_b, out, zout, B,
                       H, W, eps);
    return (int)hipGetLastError();
}
int cnx_dwconv_ln_launch(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b, float* out,
                         float* zout, int B, int H, int W, int C, float eps, hipStream_t st, const float* addp, int flip) {
    if (C != 96 && C != 192 && C != 384) return -9;
    const int pg = 256 / (C / 4);
    const size_t ngroups = (size_t)B * H * ((W + 3) / 4), batches = (ngroups + pg - 1) / pg;
    // a block stages 49 C weights: enough batches per block to amortise that, enough blocks to fill the chip
    int iters = 1;
    while (iters < 16 && batches / (iters * 2) >= 512) iters *= 2;
    const unsigned grid = (unsigned)((batches + iters - 1) / iters);
    const size_t lds = (size_t)(49 * C + 4 * 256 + pg * 4 * 4) * sizeof(float);
#define CNX_DW(CV)                                                                                                          \
    case CV: {                                                                                                              \
        static bool attr = false;                                                                                           \
        if (!attr) {                                                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv7_ln_kernel<CV>),                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
            attr = true;                                                                                                    \
        }                                                                                                                   \
        hipLaunchKernelGGL((dwconv7_ln_kernel<CV>), dim3(grid), dim3(256), lds, st, x, w, bias, ln_w, ln_b, out, zout, B, H, W, eps, iters, addp, flip); \
        break;                                                                                                              \
    }
    switch (C) {
        CNX_DW(96)
        CNX_DW(192)
        CNX_DW(384)
    }
#undef CNX_DW
    return (int)hipGetLastError();
}
int cnx_ln_patch2_launch(const float* x, const float* ln_w, const float* ln_b, float* out, int B, int H, int W, int C, float eps,
                         hipStream_t st) {
    if ((H & 1) || (W & 1)) return -9;
    const int pix = 256 / (C / 4);
    const dim3 grid((unsigned)(((size_t)B * H * W + pix - 1) / pix));
    CNX_DISPATCH(ln_patch2_kernel, C, grid, x, ln_w, ln_b, out, B, H, W, eps);
    return (int)hipGetLastError();
}
int cnx_fold_scale_launch(const float* Wm, const float* b, const float* s, float* Wo, float* bo, int N, int K, hipStream_t st) {
    hipLaunchKernelGGL(fold_scale_kernel, dim3(512), dim3(256), 0, st, Wm, b, s, Wo, bo, N, K);
    return (int)hipGetLastError();
}
int cnx_fold_scale_bwd_launch(const float* dWp, const float* dbp, const float* Wm, const float* b, const float* s, float* dW, float* db,
                              float* ds, int N, int K, hipStream_t st) {
    hipLaunchKernelGGL(fold_scale_bwd_kernel, dim3(N), dim3(256), 0, st, dWp, dbp, Wm, b, s, dW, db, ds, K);
    return (int)hipGetLastError();
}
