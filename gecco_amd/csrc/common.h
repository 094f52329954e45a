// Shared device helpers for the gfx950 (MI355X) GECCO kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GECCO_WAVE 64

// Streaming hints on the big once-per-kernel streams.  GECCO_PLAIN_MEMOPS builds the same kernels with default-policy
// accesses (A/B of what the next kernel finds in the Infinity Cache; tools/README in DESIGN.md section 6).
#ifdef GECCO_PLAIN_MEMOPS
#define GECCO_NT_LOAD(p) (*(p))
#define GECCO_NT_STORE(v, p) (*(p) = (v))
#else
#define GECCO_NT_LOAD(p) __builtin_nontemporal_load(p)
#define GECCO_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif

// Row of accumulator register `reg` (0..15) for lane-half `h` of a 32x32 MFMA tile
// (C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).
__device__ __forceinline__ int mfma_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// D = A(32xK=2) * B(2x32) + C, exact fp32 (v_mfma_f32_32x32x2_f32).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous chunk of the virtual
// grid so neighbouring tiles (which share an operand panel) hit the same L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// GaussianActivation (reference models/activation.py:17-24): (exp(-u^2/(2 a^2)) - 0.7)/0.28
__device__ __forceinline__ float gauss_act(float u, float neg_inv_2a2, bool normalized) {
    float y = __expf(u * u * neg_inv_2a2);
    // (y - 0.7) / 0.28 as a multiply by the rounded reciprocal: <= 1 ulp from the IEEE division, ~9 VALU fewer
    return normalized ? (y - 0.7f) * (1.0f / 0.28f) : y;
}
// Epilogue activation by code (GemmArgs::act): 1 / 2 = GaussianActivation normalized / raw, 3 = ReLU (the reference's
// default `activation=nn.ReLU`, models/mlp.py:12, set_transformer.py:81,133), 4 = GELU (erf form; the conditioner's
// CNBlocks, models/feature_pyramid.py:28-73 through torchvision).  The code is wave-uniform.
__device__ __forceinline__ float act_apply(float u, float neg_inv_2a2, int act) {
    if (act == 3) return fmaxf(u, 0.f);
    if (act == 4) return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));   // nn.GELU() (exact erf form): ConvNeXt blocks
    return gauss_act(u, neg_inv_2a2, act == 1);
}
__device__ __forceinline__ bool act_is_gauss(int act) { return act == 1 || act == 2; }
// d act / d u, and for GaussianActivation d act / d alpha (gauss_act_bwd_kernel's expressions, backward.hip)
__device__ __forceinline__ float act_prime(float u, float neg_inv_2a2, float inv_a2, int kind, float& dalpha) {
    dalpha = 0.f;
    if (kind == 3) return u > 0.f ? 1.f : 0.f;
    if (kind == 4) {
        const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * u * u);
        return cdf + u * pdf;
    }
    const float E = __expf(u * u * neg_inv_2a2) * (kind == 1 ? 1.0f / 0.28f : 1.0f);
    dalpha = E * (u * u * inv_a2);   // x (1 / alpha) by the caller: E u^2 / alpha^3
    return E * (-u * inv_a2);
}
