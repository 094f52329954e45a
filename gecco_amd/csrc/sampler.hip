// fp64 sampler-state kernels (EDM stochastic Heun sampler, reference diffusion.py:271-352 and the
// upsampler diffusion.py:354-470), reparameterisations (reparam.py) and the stand-alone
// GaussianActivation — all HBM-bound pointwise work on (B, N, 3) / (B, N, C) tensors.
//
// Every per-step scalar (t_cur, t_hat, t_next, churn and redo coefficients) is read from a DEVICE
// schedule table indexed by a DEVICE step counter, so one captured hipGraph of a sampler step can
// be replayed for every step: no scalar is baked into a kernel node and the host never compares a
// device value (the reference syncs on `S_min <= t_cur <= S_max` each step, diffusion.py:318-322).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int SCHED_COLS = 8;  // {t_cur, t_hat, t_next, churn, redo, -, -, -}

__device__ __forceinline__ size_t gid() { return (size_t)blockIdx.x * blockDim.x + threadIdx.x; }

// x_out = x_cur + (double)((float)sched[s][col] * noise)   — the product is fp32 in the reference: a 0-dim fp64
// tensor times an fp32 tensor stays fp32 (diffusion.py:325,465); the sum is fp64.
__global__ void add_noise_f64_kernel(const double* __restrict__ x_cur, const float* __restrict__ noise,
                                     size_t noise_step_stride, const double* __restrict__ sched,
                                     const int* __restrict__ step, int col, int sigma_col, double* __restrict__ x_out,
                                     float* __restrict__ x_in, float* __restrict__ sigma, size_t n, int B) {
    const int s = *step;
    const float c = (float)sched[(size_t)s * SCHED_COLS + col];
    const float* nz = noise + (size_t)s * noise_step_stride;
    for (size_t i = gid(); i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double v = x_cur[i] + (double)(c * nz[i]);
        x_out[i] = v;
        if (x_in) x_in[i] = (float)v;
    }
    if (sigma && gid() < (size_t)B) sigma[gid()] = (float)sched[(size_t)s * SCHED_COLS + sigma_col];
}

// Inpainting sampler (gecco-jax models/stochastic.py:136-143): the KNOWN points of the state are re-drawn at the current
// noise level every sub-step: x[b, m + j] = known[b, j] + noise[b, j] * sigma_cur.  x is (B, m + n_known, 3) fp64.
__global__ void refresh_known_f64_kernel(double* __restrict__ x, const float* __restrict__ known, const float* __restrict__ noise,
                                         const double* __restrict__ sched, const int* __restrict__ step, int col, int m,
                                         int n_known, int B) {
    const float c = (float)sched[(size_t)(*step) * SCHED_COLS + col];
    const size_t per = (size_t)n_known * 3, total = (size_t)B * per;
    for (size_t i = gid(); i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / per, r = i % per;
        x[(b * (size_t)(m + n_known) + m) * 3 + r] = (double)known[i] + (double)(c * noise[i]);
    }
}

// data_ctx = data + noise * (float)t_cur, all fp32 (diffusion.py:430)
__global__ void add_noise_f32_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                     size_t noise_step_stride, const double* __restrict__ sched,
                                     const int* __restrict__ step, int col, float* __restrict__ out,
                                     float* __restrict__ sigma, size_t n, int B) {
    const int s = *step;
    const float c = (float)sched[(size_t)s * SCHED_COLS + col];
    const float* nz = noise + (size_t)s * noise_step_stride;
    for (size_t i = gid(); i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = x[i] + nz[i] * c;
    if (sigma && gid() < (size_t)B) sigma[gid()] = c;
}

// d_cur = (x_hat - den)/t_hat ; x_next = x_hat + (t_next - t_hat) * d_cur   (diffusion.py:335-336)
__global__ void euler_kernel(const double* __restrict__ x_hat, const float* __restrict__ den,
                             const double* __restrict__ sched, const int* __restrict__ step,
                             double* __restrict__ d_cur, double* __restrict__ x_next, float* __restrict__ x_in,
                             float* __restrict__ sigma, size_t n, int B) {
    const int s = *step;
    const double t_hat = sched[(size_t)s * SCHED_COLS + 1], t_next = sched[(size_t)s * SCHED_COLS + 2];
    for (size_t i = gid(); i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double xh = x_hat[i];
        const double d = (xh - (double)den[i]) / t_hat;
        const double xn = xh + (t_next - t_hat) * d;
        d_cur[i] = d;
        x_next[i] = xn;
        if (x_in) x_in[i] = (float)xn;
    }
    if (sigma && gid() < (size_t)B) sigma[gid()] = (float)t_next;
}

// d' = (x_next - den)/t_next ; x = x_hat + (t_next - t_hat) * (0.5 d_cur + 0.5 d')   (diffusion.py:346-347)
__global__ void heun_kernel(const double* __restrict__ x_hat, const double* __restrict__ x_next,
                            const float* __restrict__ den, const double* __restrict__ d_cur,
                            const double* __restrict__ sched, const int* __restrict__ step,
                            double* __restrict__ x_out, size_t n) {
    const int s = *step;
    const double t_hat = sched[(size_t)s * SCHED_COLS + 1], t_next = sched[(size_t)s * SCHED_COLS + 2];
    for (size_t i = gid(); i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double dp = (x_next[i] - (double)den[i]) / t_next;
        x_out[i] = x_hat[i] + (t_next - t_hat) * (0.5 * d_cur[i] + 0.5 * dp);
    }
}

__global__ void advance_kernel(int* step, int delta) { *step += delta; }

// x = (double)latents * t  (diffusion.py:308)
__global__ void scale_f64_kernel(const float* __restrict__ latents, double t, double* __restrict__ x, size_t n) {
    for (size_t i = gid(); i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = (double)latents[i] * t;
}

// ------------------------------------------------------------------------------ reparam.py
template <typename T>
__global__ void gaussian_reparam_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                        const float* __restrict__ sigma, T* __restrict__ y, size_t n, int dim,
                                        int inverse) {
    for (size_t i = gid(); i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % dim);
        // data_to_diffusion (data - mean)/sigma (reparam.py:57-59); diffusion_to_data diff*sigma + mean (:61-63)
        y[i] = inverse ? x[i] * (T)sigma[c] + (T)mean[c] : (x[i] - (T)mean[c]) / (T)sigma[c];
    }
}

template <typename T> __device__ __forceinline__ T t_tanh(T v);
template <> __device__ __forceinline__ float t_tanh(float v) { return tanhf(v); }
template <> __device__ __forceinline__ double t_tanh(double v) { return tanh(v); }
template <typename T> __device__ __forceinline__ T t_atanh(T v);
template <> __device__ __forceinline__ float t_atanh(float v) { return atanhf(v); }
template <> __device__ __forceinline__ double t_atanh(double v) { return atanh(v); }
template <typename T> __device__ __forceinline__ T t_exp(T v);
template <> __device__ __forceinline__ float t_exp(float v) { return expf(v); }
template <> __device__ __forceinline__ double t_exp(double v) { return exp(v); }
template <typename T> __device__ __forceinline__ T t_log(T v);
template <> __device__ __forceinline__ float t_log(float v) { return logf(v); }
template <> __device__ __forceinline__ double t_log(double v) { return log(v); }
template <typename T> __device__ __forceinline__ T t_sqrt(T v);
template <> __device__ __forceinline__ float t_sqrt(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double t_sqrt(double v) { return sqrt(v); }

// UVLReparam (reparam.py:69-201) with the kornia pinhole model of SURVEY.md Appendix A.5.
// K (B, 3, 3) fp32; one thread per point.
template <typename T>
__global__ void uvl_reparam_kernel(const T* __restrict__ x, const float* __restrict__ K,
                                   const float* __restrict__ mean, const float* __restrict__ std_, double logit_scale,
                                   T* __restrict__ y, int B, int N, int inverse) {
    const size_t p = gid();
    if (p >= (size_t)B * N) return;
    const int b = (int)(p / N);
    const T fx = (T)K[b * 9 + 0], fy = (T)K[b * 9 + 4], cx = (T)K[b * 9 + 2], cy = (T)K[b * 9 + 5];
    const T a0 = x[p * 3 + 0], a1 = x[p * 3 + 1], a2 = x[p * 3 + 2];
    const T ls = (T)logit_scale;
    if (inverse) {  // diffusion -> data: uvl_to_hwd (:159-177), hwd_to_xyz (:131-137)
        const T u = a0 * (T)std_[0] + (T)mean[0], v = a1 * (T)std_[1] + (T)mean[1], l = a2 * (T)std_[2] + (T)mean[2];
        const T su = (t_tanh(u) * ls + (T)1) / (T)2, sv = (t_tanh(v) * ls + (T)1) / (T)2, d = t_exp(l);
        const T xx = (su - cx) / fx, yy = (sv - cy) / fy;
        T nrm = t_sqrt(xx * xx + yy * yy + (T)1);
        nrm = nrm < (T)1e-12 ? (T)1e-12 : nrm;
        y[p * 3 + 0] = xx / nrm * d;
        y[p * 3 + 1] = yy / nrm * d;
        y[p * 3 + 2] = (T)1 / nrm * d;
    } else {  // data -> diffusion: xyz_to_hwd (:112-129), hwd_to_uvl (:139-157)
        const T z = a2;
        const T az = z < 0 ? -z : z;
        const T sc = az > (T)1e-8 ? (T)1 / (z + (T)1e-8) : (T)1;
        const T uu = sc * a0 * fx + cx, vv = sc * a1 * fy + cy;
        const T d = t_sqrt(a0 * a0 + a1 * a1 + a2 * a2);
        const T r0 = t_atanh(((T)2 * uu - (T)1) / ls), r1 = t_atanh(((T)2 * vv - (T)1) / ls), r2 = t_log(d);
        y[p * 3 + 0] = (r0 - (T)mean[0]) / (T)std_[0];
        y[p * 3 + 1] = (r1 - (T)mean[1]) / (T)std_[1];
        y[p * 3 + 2] = (r2 - (T)mean[2]) / (T)std_[2];
    }
}

// stand-alone GaussianActivation.forward (models/activation.py:17-24)
__global__ void gaussian_act_kernel(const float* __restrict__ x, const float* __restrict__ alpha, float* __restrict__ y,
                                    size_t n, int normalized) {
    const float a = alpha[0];
    const float k = -1.0f / (2.0f * a * a);
    for (size_t i = gid(); i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = gauss_act(x[i], k, normalized != 0);
}

unsigned grid_for(size_t n) {
    size_t g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

int sampler_add_noise_f64_launch(const double* x_cur, const float* noise, size_t noise_step_stride, const double* sched,
                                 const int* step, int col, int sigma_col, double* x_out, float* x_in, float* sigma,
                                 size_t n, int B, hipStream_t st) {
    hipLaunchKernelGGL(add_noise_f64_kernel, dim3(grid_for(n)), dim3(256), 0, st, x_cur, noise, noise_step_stride,
                       sched, step, col, sigma_col, x_out, x_in, sigma, n, B);
    return (int)hipGetLastError();
}
int sampler_add_noise_f32_launch(const float* x, const float* noise, size_t noise_step_stride, const double* sched,
                                 const int* step, int col, float* out, float* sigma, size_t n, int B, hipStream_t st) {
    hipLaunchKernelGGL(add_noise_f32_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, noise, noise_step_stride, sched,
                       step, col, out, sigma, n, B);
    return (int)hipGetLastError();
}
int sampler_euler_launch(const double* x_hat, const float* den, const double* sched, const int* step, double* d_cur,
                         double* x_next, float* x_in, float* sigma, size_t n, int B, hipStream_t st) {
    hipLaunchKernelGGL(euler_kernel, dim3(grid_for(n)), dim3(256), 0, st, x_hat, den, sched, step, d_cur, x_next, x_in,
                       sigma, n, B);
    return (int)hipGetLastError();
}
int sampler_heun_launch(const double* x_hat, const double* x_next, const float* den, const double* d_cur,
                        const double* sched, const int* step, double* x_out, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(heun_kernel, dim3(grid_for(n)), dim3(256), 0, st, x_hat, x_next, den, d_cur, sched, step, x_out,
                       n);
    return (int)hipGetLastError();
}
int sampler_advance_launch(int* step, int delta, hipStream_t st) {
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, st, step, delta);
    return (int)hipGetLastError();
}
int sampler_scale_launch(const float* latents, double t, double* x, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(scale_f64_kernel, dim3(grid_for(n)), dim3(256), 0, st, latents, t, x, n);
    return (int)hipGetLastError();
}
int sampler_refresh_known_launch(double* x, const float* known, const float* noise, const double* sched, const int* step, int col,
                                 int m, int n_known, int B, hipStream_t st) {
    hipLaunchKernelGGL(refresh_known_f64_kernel, dim3(grid_for((size_t)B * n_known * 3)), dim3(256), 0, st, x, known, noise, sched, step,
                       col, m, n_known, B);
    return (int)hipGetLastError();
}

int gaussian_reparam_launch(const void* x, const float* mean, const float* sigma, void* y, size_t n, int dim,
                            int inverse, int is_f64, hipStream_t st) {
    if (is_f64)
        hipLaunchKernelGGL(gaussian_reparam_kernel<double>, dim3(grid_for(n)), dim3(256), 0, st, (const double*)x, mean,
                           sigma, (double*)y, n, dim, inverse);
    else
        hipLaunchKernelGGL(gaussian_reparam_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)x, mean,
                           sigma, (float*)y, n, dim, inverse);
    return (int)hipGetLastError();
}
int uvl_reparam_launch(const void* x, const float* K, const float* mean, const float* std_, double logit_scale, void* y,
                       int B, int N, int inverse, int is_f64, hipStream_t st) {
    const size_t n = (size_t)B * N;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (is_f64)
        hipLaunchKernelGGL(uvl_reparam_kernel<double>, dim3(grid), dim3(256), 0, st, (const double*)x, K, mean, std_,
                           logit_scale, (double*)y, B, N, inverse);
    else
        hipLaunchKernelGGL(uvl_reparam_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, K, mean, std_,
                           logit_scale, (float*)y, B, N, inverse);
    return (int)hipGetLastError();
}
namespace {
__global__ void relu_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = gid(); i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = fmaxf(x[i], 0.f);
}
__global__ void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ du, size_t n) {
    for (size_t i = gid(); i < n; i += (size_t)gridDim.x * blockDim.x) du[i] = y[i] > 0.f ? dy[i] : 0.f;
}
}  // namespace
int relu_launch(const float* x, float* y, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(relu_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, y, n);
    return (int)hipGetLastError();
}
int relu_bwd_launch(const float* y, const float* dy, float* du, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, st, y, dy, du, n);
    return (int)hipGetLastError();
}
int gaussian_act_launch(const float* x, const float* alpha, float* y, size_t n, int normalized, hipStream_t st) {
    hipLaunchKernelGGL(gaussian_act_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, alpha, y, n, normalized);
    return (int)hipGetLastError();
}
