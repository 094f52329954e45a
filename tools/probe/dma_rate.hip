// Diagnostic: what does the global -> LDS DMA path deliver per CU for the GEMM's tile-fill access patterns?
// No compute, no LDS reads: every block walks the K dimension of its (row tile, column tile) pair like the GEMM
// does, 4 waves, ring of 3 stages with counted vmcnt waits, and only the shape of one wave-instruction differs:
//   A_MODE 0: 16 rows x 64 B  (BK = 16 fp32)        B_MODE 0: 32 rows x 32 B per bf16 plane (BK = 16)
//   A_MODE 1:  8 rows x 128 B (BK = 32 fp32)        B_MODE 1: one contiguous 1 KiB (pre-tiled W image)
//   A_MODE 2: no A traffic                          B_MODE 2: no B traffic
// Prints ms and the L2 -> LDS rate in TB/s and B/clk/CU (at 2.4 GHz nominal).
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ void dma16(const void* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// the same fill with buffer_load ... lds: per-lane 32-bit offsets computed once, the K-step advance in the scalar offset
template <int NS>
__global__ __launch_bounds__(256) void fill_buf_kernel(const float* A, const float* Wtiled, int rows, int K, int Nout, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tilesN = Nout / 128, nblk = (rows / 128) * tilesN;
    const int v = xcd_remap(blockIdx.x, nblk);
    const int ct = v % tilesN, rt = v / tilesN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int BK = 32, STAGE = 128 * BK * 2;
    const int nk = K / BK;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)rt * 128 * K), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(Wtiled + (size_t)ct * nk * 4096), 0, 0x7fffffff, 0x00020000);
    unsigned aoff[4], boff[4];
    for (int q = 0; q < 4; ++q) {
        const int row = (4 * wave + q) * 8 + (lane >> 3);
        aoff[q] = (unsigned)(row * K + (lane & 7) * 4) * 4u;
        boff[q] = (unsigned)((4 * wave + q) * 256 + lane * 4) * 4u;
    }
    auto issue = [&](int kt) {
        float* st = smem + (kt % NS) * STAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16_buf(ra, aoff[q], (unsigned)kt * BK * 4u, st + (4 * wave + q) * 256);
#pragma unroll
        for (int q = 0; q < 4; ++q) dma16_buf(rb, boff[q], (unsigned)kt * 4096u * 4u, st + 4096 + (4 * wave + q) * 256);
    };
    for (int p = 0; p < NS - 1; ++p)
        if (p < nk) issue(p);
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(nk - 1 - kt, NS - 2);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nk) issue(kt + NS - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink && smem[threadIdx.x] == 123.456f) sink[0] = 1.f;
}

template <int NS>
static void run_buf(const char* name, const float* A, const float* Wt, int rows, int K, int Nout, float* sink) {
    const int nblk = (rows / 128) * (Nout / 128);
    const size_t lds = (size_t)NS * 128 * 32 * 2 * 4;
    (void)hipFuncSetAttribute((const void*)fill_buf_kernel<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) fill_buf_kernel<NS><<<nblk, 256, lds, 0>>>(A, Wt, rows, K, Nout, sink);
    (void)hipEventRecord(a, 0);
    const int it = 8;
    for (int i = 0; i < it; ++i) fill_buf_kernel<NS><<<nblk, 256, lds, 0>>>(A, Wt, rows, K, Nout, sink);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= it;
    const double bytes = (double)nblk * K * 128 * 8;
    printf("%-34s K=%d Nout=%d stages=%d blocks/CU=%d: %.3f ms  %.2f TB/s into LDS  %.1f B/clk/CU\n", name, K, Nout, NS,
           (int)(160 * 1024 / lds), ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

template <int A_MODE, int B_MODE, int NS>
__global__ __launch_bounds__(256) void fill_kernel(const float* A, const unsigned short* Whi, const unsigned short* Wlo,
                                                   const float* Wtiled, int rows, int K, int Nout, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tilesN = Nout / 128, nblk = (rows / 128) * tilesN;
    const int v = xcd_remap(blockIdx.x, nblk);
    const int ct = v % tilesN, rt = v / tilesN;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int BK = 32;                       // k per ring step (A_MODE 0 issues two 16-k halves per step)
    constexpr int STAGE = 128 * BK * 2;          // floats: A 16 KiB + B 16 KiB
    const int nk = K / BK;
    const float* abase = A + (size_t)rt * 128 * K;
    auto issue = [&](int kt) {
        float* st = smem + (kt % NS) * STAGE;
        if (A_MODE == 0) {
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int row = (2 * wave + q) * 16 + (lane >> 2);
                    dma16(abase + (size_t)row * K + kt * BK + hlf * 16 + (lane & 3) * 4, st + hlf * 2048 + (2 * wave + q) * 256);
                }
        } else if (A_MODE == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = (4 * wave + q) * 8 + (lane >> 3);
                dma16(abase + (size_t)row * K + kt * BK + (lane & 7) * 4, st + (4 * wave + q) * 256);
            }
        }
        if (B_MODE == 0) {
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const int row = ct * 128 + wave * 32 + (lane >> 1);
                const size_t off = (size_t)row * K + kt * BK + hlf * 16 + (lane & 1) * 8;
                dma16(Whi + off, st + 4096 + hlf * 2048 + wave * 256);
                dma16(Wlo + off, st + 4096 + hlf * 2048 + 1024 + wave * 256);
            }
        } else if (B_MODE == 1) {
            const float* src = Wtiled + ((size_t)ct * nk + kt) * 4096;   // one 16 KiB image per (column tile, step)
#pragma unroll
            for (int q = 0; q < 4; ++q) dma16(src + (4 * wave + q) * 256 + lane * 4, st + 4096 + (4 * wave + q) * 256);
        }
    };
    constexpr int PER = (A_MODE == 2 ? 0 : 4) + (B_MODE == 2 ? 0 : 4);
    for (int p = 0; p < NS - 1; ++p)
        if (p < nk) issue(p);
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(nk - 1 - kt, NS - 2);
        if (ahead >= 2) { if (PER == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else if (ahead == 1) { if (PER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + NS - 1 < nk) issue(kt + NS - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink && smem[threadIdx.x] == 123.456f) sink[0] = 1.f;   // keep the LDS image observable
}

template <int A_MODE, int B_MODE, int NS>
static void run(const char* name, const float* A, const unsigned short* hi, const unsigned short* lo, const float* Wt,
                int rows, int K, int Nout, float* sink) {
    const int nblk = (rows / 128) * (Nout / 128);
    const size_t lds = (size_t)NS * 128 * 32 * 2 * 4;
    (void)hipFuncSetAttribute((const void*)fill_kernel<A_MODE, B_MODE, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) fill_kernel<A_MODE, B_MODE, NS><<<nblk, 256, lds, 0>>>(A, hi, lo, Wt, rows, K, Nout, sink);
    (void)hipEventRecord(a, 0);
    const int it = 8;
    for (int i = 0; i < it; ++i) fill_kernel<A_MODE, B_MODE, NS><<<nblk, 256, lds, 0>>>(A, hi, lo, Wt, rows, K, Nout, sink);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= it;
    const double bytes = (double)nblk * K * 128 * ((A_MODE == 2 ? 0 : 4) + (B_MODE == 2 ? 0 : 4));
    printf("%-34s K=%d Nout=%d stages=%d blocks/CU=%d: %.3f ms  %.2f TB/s into LDS  %.1f B/clk/CU   (HBM-unique %.0f MB)\n", name, K,
           Nout, NS, (int)(160 * 1024 / lds), ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9,
           (A_MODE == 2 ? 0.0 : (double)rows * K * 4 / 1e6));
}

int main() {
    const int rows = 64 * 2048, K = 384;
    float *A, *Wt, *sink; unsigned short *hi, *lo;
    (void)hipMalloc(&A, (size_t)rows * 768 * 4); (void)hipMalloc(&Wt, 768 * 768 * 4 * 2); (void)hipMalloc(&sink, 4);
    (void)hipMalloc(&hi, 768 * 768 * 2); (void)hipMalloc(&lo, 768 * 768 * 2);
    (void)hipMemset(A, 0x11, (size_t)rows * 768 * 4); (void)hipMemset(Wt, 0x11, 768 * 768 * 8);
    (void)hipMemset(hi, 0x11, 768 * 768 * 2); (void)hipMemset(lo, 0x11, 768 * 768 * 2);
    for (int Nout : {768, 384}) {
        run<0, 0, 2>("A 64B rows + B 32B rows (today)", A, hi, lo, Wt, rows, K, Nout, sink);
        run<1, 0, 2>("A 128B rows + B 32B rows", A, hi, lo, Wt, rows, K, Nout, sink);
        run<0, 1, 2>("A 64B rows + B contiguous", A, hi, lo, Wt, rows, K, Nout, sink);
        run<1, 1, 2>("A 128B rows + B contiguous", A, hi, lo, Wt, rows, K, Nout, sink);
        run<1, 1, 3>("A 128B rows + B contiguous", A, hi, lo, Wt, rows, K, Nout, sink);
        run_buf<2>("same, buffer_load lds", A, Wt, rows, K, Nout, sink);
        run_buf<3>("same, buffer_load lds", A, Wt, rows, K, Nout, sink);
        run<1, 1, 4>("A 128B rows + B contiguous", A, hi, lo, Wt, rows, K, Nout, sink);
        run<0, 2, 2>("A 64B rows only", A, hi, lo, Wt, rows, K, Nout, sink);
        run<1, 2, 2>("A 128B rows only", A, hi, lo, Wt, rows, K, Nout, sink);
        run<2, 0, 2>("B 32B rows only", A, hi, lo, Wt, rows, K, Nout, sink);
        run<2, 1, 2>("B contiguous only", A, hi, lo, Wt, rows, K, Nout, sink);
    }
    return 0;
}
