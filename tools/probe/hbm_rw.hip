// Diagnostic: achievable HBM bandwidth for write-only, read-only and copy streams with 16-byte accesses (plain and
// nontemporal), 600 MB each — the ceiling the GEMM epilogues (C written: 2/3 of a launch's bytes) run against.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n, float* sink) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (MODE == 0) {   // write
            const f32x4 v = {1.f, 2.f, 3.f, (float)i};
            if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
        } else if (MODE == 1) {   // read
            acc += NT ? __builtin_nontemporal_load(src + i) : src[i];
        } else {   // copy
            const f32x4 v = NT ? __builtin_nontemporal_load(src + i) : src[i];
            if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
        }
    }
    if (MODE == 1 && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = 1.f;
}

template <int MODE, bool NT>
static void run(const char* name, const f32x4* src, f32x4* dst, size_t n, float* sink, int blocks) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) k<MODE, NT><<<blocks, 256>>>(src, dst, n, sink);
    (void)hipEventRecord(a, 0);
    const int it = 10;
    for (int i = 0; i < it; ++i) k<MODE, NT><<<blocks, 256>>>(src, dst, n, sink);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= it;
    const double bytes = (double)n * 16 * (MODE == 2 ? 2 : 1);
    printf("%-28s blocks=%5d  %.3f ms  %.2f TB/s\n", name, blocks, ms, bytes / ms / 1e9);
}

int main() {
    const size_t n = (size_t)600e6 / 16;
    f32x4 *src, *dst; float* sink;
    (void)hipMalloc(&src, n * 16); (void)hipMalloc(&dst, n * 16); (void)hipMalloc(&sink, 4);
    (void)hipMemset(src, 1, n * 16);
    for (int blocks : {2048, 8192, 32768}) {
        run<0, false>("write", src, dst, n, sink, blocks);
        run<0, true>("write nontemporal", src, dst, n, sink, blocks);
        run<1, false>("read", src, dst, n, sink, blocks);
        run<1, true>("read nontemporal", src, dst, n, sink, blocks);
        run<2, false>("copy", src, dst, n, sink, blocks);
        run<2, true>("copy nontemporal", src, dst, n, sink, blocks);
    }
    return 0;
}
