// Optimizer step of the training path (gfx950): Adam + the EMA shadow weights in ONE pass over flat fp32 buffers.
//
// Reference: `Diffusion.configure_optimizers` = torch.optim.Adam(lr=1e-4) (diffusion.py:210-211) wrapped by
// `EMAOptimizer` (ema.py:200-325), whose `update()` runs `ema_update` (ema.py:187-194: ema = ema * decay +
// (1 - decay) * param) after every optimizer step — three full passes over the parameters and their three state
// tensors per step there (Adam's foreach kernels, the parameter copy, mul_ + add_).  Here every element is read and
// written once: p, g, m, v, ema in; p, m, v, ema out = 36 B per parameter, HBM-bound (13.5 M parameters = 485 MB:
// ~80 us at 6 TB/s against ~0.5 ms for the unfused sequence).
//
// Arithmetic follows torch.optim.Adam's single-tensor path op for op, in fp32:
//   g' = g * grad_scale (+ weight_decay * p)          (grad_scale: 1 / world size when the all-reduce summed)
//   m  = m + (1 - beta1) * (g' - m)                   (exp_avg.lerp_)
//   v  = beta2 * v + (1 - beta2) * g' * g'            (mul_ + addcmul_)
//   p  = p - (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
//   ema = ema * decay + (1 - decay) * p               (when do_ema)
// The flat buffers are 16-byte aligned and the element count is padded to a multiple of 4 by the host side.
#include "common.h"
#include "kernels.h"

#pragma clang fp contract(off)   // keep torch's op order: no fma contraction across its separate kernels

namespace {

__global__ __launch_bounds__(256) void adam_ema_kernel(AdamEmaArgs a) {
    const size_t n4 = a.n / 4;
    f32x4* __restrict__ p4 = reinterpret_cast<f32x4*>(a.p);
    const f32x4* __restrict__ g4 = reinterpret_cast<const f32x4*>(a.g);
    f32x4* __restrict__ m4 = reinterpret_cast<f32x4*>(a.m);
    f32x4* __restrict__ v4 = reinterpret_cast<f32x4*>(a.v);
    f32x4* __restrict__ e4 = reinterpret_cast<f32x4*>(a.ema);
    const float w1 = a.w1, w2 = a.w2, we = a.ema_w;
    float gscale = a.grad_scale, step_size = a.step_size, bc2_sqrt = a.bc2_sqrt;
    if (a.found_inf) {   // (launch-uniform) GradScaler protocol: decided on the device, the host never waits for the gradients
        if (*a.found_inf != 0.f) {   // inf / nan in the gradients: torch skips optimizer.step() (and with it the EMA update)
            if (a.skipped && blockIdx.x == 0 && threadIdx.x == 0) *a.skipped += 1;
            return;
        }
        if (a.amp_scale) gscale = gscale / *a.amp_scale;   // scales are powers of two: exact
        const int sk = a.skipped ? *a.skipped : 0;
        if (sk > 0) {   // Adam's step count did not advance on the skipped steps: bias corrections of the steps actually taken
            const double st = (double)(a.step - sk);
            step_size = (float)(a.lr / (1.0 - pow(a.beta1, st)));
            bc2_sqrt = (float)sqrt(1.0 - pow(a.beta2d, st));
        }
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 p = p4[i], g = GECCO_NT_LOAD(g4 + i), m = m4[i], v = v4[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float ge = g[e] * gscale;
            if (a.weight_decay != 0.f) ge = ge + a.weight_decay * p[e];
            m[e] = __builtin_fmaf(w1, ge - m[e], m[e]);   // lerp_ (ATen: fma(weight, end - start, start))
            v[e] = v[e] * a.beta2 + w2 * (ge * ge);
            const float denom = sqrtf(v[e]) / bc2_sqrt + a.eps;
            p[e] = p[e] - step_size * (m[e] / denom);
        }
        p4[i] = p;
        m4[i] = m;
        v4[i] = v;
        if (a.do_ema) {
            f32x4 s = e4[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = s[e] * a.ema_decay + we * p[e];
            e4[i] = s;
        }
    }
}

__global__ __launch_bounds__(256) void ema_only_kernel(const float* __restrict__ p, float* __restrict__ ema, size_t n,
                                                       float decay, float we) {
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 x = reinterpret_cast<const f32x4*>(p)[i];
        f32x4 s = reinterpret_cast<f32x4*>(ema)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = s[e] * decay + we * x[e];
        reinterpret_cast<f32x4*>(ema)[i] = s;
    }
}

unsigned grid_for4(size_t n) {
    const size_t blocks = (n / 4 + 255) / 256;
    return (unsigned)(blocks < 1 ? 1 : (blocks > 256 * 16 ? 256 * 16 : blocks));
}

}  // namespace

int adam_ema_launch(const AdamEmaArgs& a, hipStream_t st) {
    if (a.n % 4) return -2;
    if (a.n == 0) return 0;
    hipLaunchKernelGGL(adam_ema_kernel, dim3(grid_for4(a.n)), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

int ema_update_launch(const float* p, float* ema, size_t n, double decay, hipStream_t st) {
    if (n % 4) return -2;
    if (n == 0) return 0;
    hipLaunchKernelGGL(ema_only_kernel, dim3(grid_for4(n)), dim3(256), 0, st, p, ema, n, (float)decay, (float)(1.0 - decay));
    return (int)hipGetLastError();
}
