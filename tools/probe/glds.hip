// Probe: __builtin_amdgcn_global_load_lds semantics on gfx950 (per-lane source, wave-uniform LDS base + lane*16).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each wave copies 64 x 16 B: lane l fetches chunk perm(l) of the wave's 1 KiB source block
    const int perm = lane ^ 5;
    const float* g = src + wave * 256 + perm * 4;
    float* dst = lds + wave * 256;  // wave-uniform base
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256 * 4; i += blockDim.x) out[i] = lds[i];
}
int main() {
    float h[1024], o[1024]; for (int i = 0; i < 1024; ++i) h[i] = i;
    float *d, *e; hipMalloc(&d, 4096); hipMalloc(&e, 4096); hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, d, e); hipMemcpy(o, e, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        const float exp = w * 256 + (l ^ 5) * 4 + j;  // LDS slot of lane l holds source chunk perm(l)
        if (o[w * 256 + l * 4 + j] != exp) ++bad;
    }
    printf("glds probe: %d mismatches (0 = LDS[base + lane*16] <- per-lane source)\n", bad);
    return 0;
}
