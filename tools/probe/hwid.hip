// Probe: which HW_ID bits tell two co-resident workgroups of a CU apart (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned long long* t) {
  extern __shared__ float s[];
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
    out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    t[blockIdx.x] = __builtin_amdgcn_s_memtime();
  }
  s[threadIdx.x] = threadIdx.x;
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(100);
}
int main() {
  const int n = 1024; unsigned* d; unsigned long long* dt;
  hipMalloc(&d, n * 8); hipMalloc(&dt, n * 8);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 73728);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 73728, 0, d, dt);
  unsigned h[2 * n]; unsigned long long ht[n];
  hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost); hipMemcpy(ht, dt, n * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ht[0]; for (int i = 0; i < n; ++i) if (ht[i] < t0) t0 = ht[i];
  for (int i = 0; i < n; i += (i < 48 ? 1 : 37)) {
    unsigned hw = h[2 * i];
    printf("blk %4d xcc %u hw %08x wave %u simd %u pipe %u cu %u sh %u se %u tg %u  t+%llu\n", i, h[2 * i + 1] & 15, hw, hw & 15, (hw >> 4) & 3,
           (hw >> 6) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 16) & 15, (ht[i] - t0) / 100);
  }
  return 0;
}
