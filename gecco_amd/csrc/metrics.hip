// Point-cloud evaluation metrics on the device (gfx950): pairwise distances, Chamfer distance, Sinkhorn EMD.
//
// Reference: gecco-jax/src/gecco_jax/metrics.py:92-156 (`chamfer_distance`, `scipy_emd`, `sinkhorn_emd`) over
// gecco-jax/src/gecco_jax/geometry.py:8-24 (`distance_matrix`: |a|^2 + |b|^2 - 2 a.b clamped at 0, optional sqrt).
// The README of gecco-torch lists these as absent from the torch package (gecco-torch/README.md:49-52); they consume
// the sampler's output clouds, so they sit right behind the hot path.  All clouds are (B, N, 3) fp32; every kernel is
// batched over B (the JAX code vmaps single clouds).  HBM/LDS-bound integer-free arithmetic: no MFMA reshaping — the
// inner dimension is 3.
#include "common.h"
#include "kernels.h"

namespace {

// d(a, b) exactly as the reference forms it: aa + bb - 2 ab, clamped at 0 (the clamp hides the cancellation noise of
// that form for near-identical points), sqrt unless `squared`
__device__ __forceinline__ float pair_dist(float ax, float ay, float az, float aa, float bx, float by, float bz, float bb,
                                           bool squared) {
    const float ab = ax * bx + ay * by + az * bz;
    const float d2 = fmaxf(aa + bb - 2.f * ab, 0.f);
    return squared ? d2 : sqrtf(d2);
}

// D[b, i, j] = dist(a[b, i], b[b, j]): a 256 x 256 tile per block, the thread owns COLUMN j (its b point in registers) and walks
// the tile's a points in LDS (a broadcast read): every store instruction writes 256 consecutive floats of a row
__global__ __launch_bounds__(256) void dist_matrix_kernel(const float* __restrict__ A, const float* __restrict__ Bp,
                                                          float* __restrict__ D, int N, int M, int squared) {
    __shared__ float sa[256 * 4];
    const int b = blockIdx.z, i0 = blockIdx.y * 256, j = blockIdx.x * 256 + threadIdx.x;
    {
        const int i = i0 + threadIdx.x;
        const float* a = A + ((size_t)b * N + min(i, N - 1)) * 3;
        const float ax = a[0], ay = a[1], az = a[2];
        sa[threadIdx.x * 4 + 0] = ax; sa[threadIdx.x * 4 + 1] = ay; sa[threadIdx.x * 4 + 2] = az;
        sa[threadIdx.x * 4 + 3] = ax * ax + ay * ay + az * az;
    }
    const float* q = Bp + ((size_t)b * M + min(j, M - 1)) * 3;
    const float bx = q[0], by = q[1], bz = q[2], bb = bx * bx + by * by + bz * bz;
    __syncthreads();
    if (j >= M) return;
    float* dcol = D + ((size_t)b * N + i0) * M + j;
    const int in = min(256, N - i0);
    for (int i = 0; i < in; ++i)
        dcol[(size_t)i * M] = pair_dist(sa[i * 4], sa[i * 4 + 1], sa[i * 4 + 2], sa[i * 4 + 3], bx, by, bz, bb, squared != 0);
}

// mins[b, i] = min_j dist(a[b, i], b[b, j]) — one thread per a-point, the b cloud streamed through LDS in tiles
__global__ __launch_bounds__(256) void nearest_dist_kernel(const float* __restrict__ A, const float* __restrict__ Bp,
                                                           float* __restrict__ mins, int N, int M, int squared) {
    __shared__ float sb[256 * 4];
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const float* a = A + ((size_t)b * N + min(i, N - 1)) * 3;
    const float ax = a[0], ay = a[1], az = a[2], aa = ax * ax + ay * ay + az * az;
    float best = 3.0e38f;
    for (int j0 = 0; j0 < M; j0 += 256) {
        __syncthreads();
        const int j = j0 + threadIdx.x;
        const float* q = Bp + ((size_t)b * M + min(j, M - 1)) * 3;
        const float bx = q[0], by = q[1], bz = q[2];
        sb[threadIdx.x * 4 + 0] = bx; sb[threadIdx.x * 4 + 1] = by; sb[threadIdx.x * 4 + 2] = bz;
        sb[threadIdx.x * 4 + 3] = bx * bx + by * by + bz * bz;
        __syncthreads();
        const int jn = min(256, M - j0);
        for (int jj = 0; jj < jn; ++jj) {
            // the min commutes with the monotone sqrt: compare squared distances, root once at the end
            const float ab = ax * sb[jj * 4] + ay * sb[jj * 4 + 1] + az * sb[jj * 4 + 2];
            best = fminf(best, fmaxf(aa + sb[jj * 4 + 3] - 2.f * ab, 0.f));
        }
    }
    if (i < N) mins[(size_t)b * N + i] = squared ? best : sqrtf(best);
}

// out[b] = mean_i v[b, i] (fixed-order tree in one block per sample: deterministic)
__global__ __launch_bounds__(256) void row_mean_kernel(const float* __restrict__ v, float* __restrict__ out, int n, float scale, int accumulate) {
    __shared__ double red[256];
    const int b = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)v[(size_t)b * n + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float r = (float)(red[0] / n) * scale;
        out[b] = accumulate ? out[b] + r : r;
    }
}

// ---- Sinkhorn (log domain, uniform marginals 1/N, 1/M) on a cost matrix C (B, N, M):
//   f_i = -eps * LSE_j((g_j - C_ij) / eps + log(1/M)),   g_j = -eps * LSE_i((f_i - C_ij) / eps + log(1/N))
// rows: one wave per row i (coalesced over j); cols: one thread per column j walking the rows (coalesced across threads)
__global__ __launch_bounds__(256) void sinkhorn_rows_kernel(const float* __restrict__ C, const float* __restrict__ g, float* __restrict__ f,
                                                            int N, int M, float eps, float logw) {
    const int b = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    const float* c = C + ((size_t)b * N + i) * M;
    const float* gb = g + (size_t)b * M;
    const float inv = 1.f / eps;
    // one pass over the row: per-lane running (max, sum), merged across the wave in a fixed butterfly
    float mx = -3.0e38f, s = 0.f;
    for (int j = lane; j < M; j += 64) {
        const float v = (gb[j] - c[j]) * inv, mn = fmaxf(mx, v);
        s = s * __expf(mx - mn) + __expf(v - mn);
        mx = mn;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o, 64), os = __shfl_xor(s, o, 64), mn = fmaxf(mx, om);
        s = s * __expf(mx - mn) + os * __expf(om - mn);
        mx = mn;
    }
    if (lane == 0) f[(size_t)b * N + i] = -eps * (mx + __logf(s) + logw);
}
__global__ __launch_bounds__(256) void sinkhorn_cols_kernel(const float* __restrict__ C, const float* __restrict__ f, float* __restrict__ g,
                                                            int N, int M, float eps, float logw) {
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const float* c = C + (size_t)b * N * M + j;
    const float* fb = f + (size_t)b * N;
    const float inv = 1.f / eps;
    // one pass down the column, four independent (max, sum) chains so that loads of consecutive rows are in flight together
    float mx[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f}, s[4] = {0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + 4 <= N; i += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (fb[i + u] - c[(size_t)(i + u) * M]) * inv;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float mn = fmaxf(mx[u], v[u]);
            s[u] = s[u] * __expf(mx[u] - mn) + __expf(v[u] - mn);
            mx[u] = mn;
        }
    }
    for (; i < N; ++i) {
        const float v = (fb[i] - c[(size_t)i * M]) * inv, mn = fmaxf(mx[0], v);
        s[0] = s[0] * __expf(mx[0] - mn) + __expf(v - mn);
        mx[0] = mn;
    }
    float M4 = fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3])), S4 = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) S4 += s[u] * __expf(mx[u] - M4);
    g[(size_t)b * M + j] = -eps * (M4 + __logf(S4) + logw);
}
// rowcost[b, i] = sum_j P_ij C_ij with P_ij = exp((f_i + g_j - C_ij) / eps) / (N M)
__global__ __launch_bounds__(256) void sinkhorn_cost_kernel(const float* __restrict__ C, const float* __restrict__ f, const float* __restrict__ g,
                                                            float* __restrict__ rowcost, int N, int M, float eps, float logw) {
    const int b = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N) return;
    const float* c = C + ((size_t)b * N + i) * M;
    const float* gb = g + (size_t)b * M;
    const float fi = f[(size_t)b * N + i], inv = 1.f / eps;
    float s = 0.f;
    for (int j = lane; j < M; j += 64) s += __expf((fi + gb[j] - c[j]) * inv + logw) * c[j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) rowcost[(size_t)b * N + i] = s;
}
__global__ __launch_bounds__(256) void row_sum_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
    __shared__ double red[256];
    const int b = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)v[(size_t)b * n + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[b] = (float)red[0];
}

// ---- set-vs-set distances (gecco-jax/src/gecco_jax/benchmark.py:21-39 `batched_pairwise_distance` with `chamfer_distance` /
// `chamfer_distance_squared`): out[s, t] = Chamfer(a_s, b_t) for EVERY pair of a set of S clouds with a set of T clouds — S T N M point
// pairs (2.7e11 per direction at S = T = 256, N = M = 2048): vector-ALU work, the inner dimension is 3.  One direction per launch:
//     half[s, t] = mean_i min_j d(a[s, i], b[t, j])
// A block owns ONE a cloud chunk (8 points per thread in registers) and walks a group of b clouds, each staged in LDS as (x, y, z, |b|^2):
// a thread reads a b point once (a broadcast read) and updates its 8 running minima —
//     min_j (|a|^2 + |b_j|^2 - 2 a.b_j) = |a|^2 + min_j (|b_j|^2 - 2 a.b_j):   3 FMAs + 1 min per point pair
// (the reference adds |a|^2 before the minimum: the same value up to the rounding of that one addition), clamp at 0 and the root are
// monotone, so they follow the minimum; no N x M matrix per pair exists anywhere.  The mean is a fixed-order block tree.
constexpr int SC_KP = 8;                    // a points per thread
constexpr int SC_TILE = 2048;               // b points per LDS tile (32 KiB)
__global__ __launch_bounds__(256, 2) void set_nearest_mean_kernel(const float* __restrict__ A, const float* __restrict__ Bp, float* __restrict__ out, int N,
                                                                  int M, int T, int tgroup, int squared, int ld_s, int ld_t, float scale, int accumulate) {
    __shared__ __attribute__((aligned(16))) float sb[SC_TILE * 4];
    __shared__ double red[256];
    const int s = blockIdx.y, t0 = blockIdx.x * tgroup, tid = threadIdx.x;
    const int nchunk = (N + 256 * SC_KP - 1) / (256 * SC_KP);
    for (int t = t0; t < min(t0 + tgroup, T); ++t) {
        double total = 0.0;
        for (int ch = 0; ch < nchunk; ++ch) {
            float ax[SC_KP], ay[SC_KP], az[SC_KP], aa[SC_KP], best[SC_KP];
#pragma unroll
            for (int k = 0; k < SC_KP; ++k) {
                const int i = min((ch * SC_KP + k) * 256 + tid, N - 1);
                const float* a = A + ((size_t)s * N + i) * 3;
                ax[k] = -2.f * a[0]; ay[k] = -2.f * a[1]; az[k] = -2.f * a[2];
                aa[k] = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
                best[k] = 3.0e38f;
            }
            for (int j0 = 0; j0 < M; j0 += SC_TILE) {
                __syncthreads();
                const int jn = min(SC_TILE, M - j0);
                for (int j = tid; j < jn; j += 256) {
                    const float* q = Bp + ((size_t)t * M + j0 + j) * 3;
                    const float bx = q[0], by = q[1], bz = q[2];
                    *reinterpret_cast<f32x4*>(sb + 4 * j) = f32x4{bx, by, bz, bx * bx + by * by + bz * bz};
                }
                __syncthreads();
#pragma unroll 4
                for (int j = 0; j < jn; ++j) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(sb + 4 * j);
#pragma unroll
                    for (int k = 0; k < SC_KP; ++k)
                        best[k] = fminf(best[k], __builtin_fmaf(ax[k], b[0], __builtin_fmaf(ay[k], b[1], __builtin_fmaf(az[k], b[2], b[3]))));
                }
            }
#pragma unroll
            for (int k = 0; k < SC_KP; ++k) {
                const int i = (ch * SC_KP + k) * 256 + tid;
                const float d2 = fmaxf(aa[k] + best[k], 0.f);
                if (i < N) total += (double)(squared ? d2 : sqrtf(d2));
            }
        }
        __syncthreads();
        red[tid] = total;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) {
            const float r = (float)(red[0] / N) * scale;
            float* dst = out + (size_t)s * ld_s + (size_t)t * ld_t;
            *dst = accumulate ? *dst + r : r;
        }
    }
}

// ---- 1-NN accuracy, MMD, coverage on the distance matrices of a generated set against a reference set (benchmark.py:128-156:
// `_assemble_dist_m`, `_one_nn_acc`, `_mmd`, `_cov`), n clouds each.  ss (n, n) sample-sample, sd (n, n) sample (row) - data (column), dd
// (n, n) data-data.  One block; out[0] = 1-NNA, out[1] = MMD, out[2] = COV.  Reference semantics kept to the letter: the block matrix
// [[ss, sd], [sd^T, dd]] with an infinite diagonal, the nearest neighbour of every COLUMN (numpy's argmin: the first of equal minima), a
// sample counted correct when that index is <= n (sic: index n, the first data cloud, counts for the samples), a data cloud when it is > n.
__global__ __launch_bounds__(256) void set_metrics_kernel(const float* __restrict__ ss, const float* __restrict__ sd, const float* __restrict__ dd, int n,
                                                          float* __restrict__ out, int* __restrict__ flags) {
    __shared__ int red_i[256];
    __shared__ float red_f[256];
    const int tid = threadIdx.x;
    int correct = 0;
    for (int c = tid; c < 2 * n; c += 256) {
        float best = __builtin_inff();
        int arg = 0;
        for (int r = 0; r < 2 * n; ++r) {
            float v;
            if (r == c) v = __builtin_inff();
            else if (c < n) v = r < n ? ss[(size_t)r * n + c] : sd[(size_t)c * n + (r - n)];          // lower-left block = sd^T
            else v = r < n ? sd[(size_t)r * n + (c - n)] : dd[(size_t)(r - n) * n + (c - n)];
            if (v < best) { best = v; arg = r; }
        }
        correct += c < n ? (arg <= n) : (arg > n);
    }
    float mn = __builtin_inff();
    for (size_t i = tid; i < (size_t)n * n; i += 256) mn = fminf(mn, sd[i]);
    for (int c = tid; c < n; c += 256) flags[c] = 0;
    __syncthreads();
    for (int r = tid; r < n; r += 256) {   // the data cloud nearest to sample r
        float best = __builtin_inff();
        int arg = 0;
        for (int c = 0; c < n; ++c) {
            const float v = sd[(size_t)r * n + c];
            if (v < best) { best = v; arg = c; }
        }
        flags[arg] = 1;
    }
    __syncthreads();
    int covered = 0;
    for (int c = tid; c < n; c += 256) covered += flags[c];
    red_i[tid] = correct; red_f[tid] = mn;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red_i[tid] += red_i[tid + o]; red_f[tid] = fminf(red_f[tid], red_f[tid + o]); }
        __syncthreads();
    }
    const int tot_correct = red_i[0];
    const float tot_min = red_f[0];
    __syncthreads();
    red_i[tid] = covered;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red_i[tid] += red_i[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        out[0] = (float)((double)tot_correct / (2.0 * n));
        out[1] = tot_min;
        out[2] = (float)((double)red_i[0] / n);
    }
}

}  // namespace

int dist_matrix_launch(const float* A, const float* Bp, float* D, int B, int N, int M, int squared, hipStream_t st) {
    if (B <= 0 || N <= 0 || M <= 0) return -2;
    hipLaunchKernelGGL(dist_matrix_kernel, dim3((M + 255) / 256, (N + 255) / 256, B), dim3(256), 0, st, A, Bp, D, N, M, squared);
    return (int)hipGetLastError();
}
int nearest_dist_launch(const float* A, const float* Bp, float* mins, int B, int N, int M, int squared, hipStream_t st) {
    if (B <= 0 || N <= 0 || M <= 0) return -2;
    hipLaunchKernelGGL(nearest_dist_kernel, dim3((N + 255) / 256, B), dim3(256), 0, st, A, Bp, mins, N, M, squared);
    return (int)hipGetLastError();
}
int row_mean_launch(const float* v, float* out, int B, int n, float scale, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(row_mean_kernel, dim3(B), dim3(256), 0, st, v, out, n, scale, accumulate);
    return (int)hipGetLastError();
}
int sinkhorn_step_launch(const float* C, float* f, float* g, int B, int N, int M, float eps, hipStream_t st) {
    hipLaunchKernelGGL(sinkhorn_rows_kernel, dim3((N + 3) / 4, B), dim3(256), 0, st, C, g, f, N, M, eps, -logf((float)M));
    hipLaunchKernelGGL(sinkhorn_cols_kernel, dim3((M + 255) / 256, B), dim3(256), 0, st, C, f, g, N, M, eps, -logf((float)N));
    return (int)hipGetLastError();
}
int sinkhorn_cost_launch(const float* C, const float* f, const float* g, float* rowcost, float* out, int B, int N, int M, float eps,
                         hipStream_t st) {
    hipLaunchKernelGGL(sinkhorn_cost_kernel, dim3((N + 3) / 4, B), dim3(256), 0, st, C, f, g, rowcost, N, M, eps,
                       -logf((float)N) - logf((float)M));
    hipLaunchKernelGGL(row_sum_kernel, dim3(B), dim3(256), 0, st, rowcost, out, N);
    return (int)hipGetLastError();
}

// out (S, T): accumulate == 0 writes scale * mean_i min_j d(a[s, i], b[t, j]) to out[s * ld_s + t * ld_t], 1 adds it
int set_nearest_mean_launch(const float* A, const float* Bp, float* out, int S, int T, int N, int M, int squared, int ld_s, int ld_t, float scale,
                            int accumulate, hipStream_t st) {
    if (S <= 0 || T <= 0 || N <= 0 || M <= 0) return -2;
    const int tgroup = T >= 64 ? 8 : 1;     // b clouds per block: the a points are loaded once per group
    hipLaunchKernelGGL(set_nearest_mean_kernel, dim3((T + tgroup - 1) / tgroup, S), dim3(256), 0, st, A, Bp, out, N, M, T, tgroup, squared, ld_s, ld_t,
                       scale, accumulate);
    return (int)hipGetLastError();
}
int set_metrics_launch(const float* ss, const float* sd, const float* dd, int n, float* out, int* flags, hipStream_t st) {
    if (n <= 0) return -2;
    hipLaunchKernelGGL(set_metrics_kernel, dim3(1), dim3(256), 0, st, ss, sd, dd, n, out, flags);
    return (int)hipGetLastError();
}
