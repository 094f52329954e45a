// Inducing-point attention on fp32 MFMA, gfx950.  There is no N x N attention in GECCO
// (SURVEY.md 0.2): every layer has
//   pool   : I = 64 learned queries  <- N point keys/values  (AttentionPool, reference
//            models/set_transformer.py:47-65): flash-style, keys streamed, split over N;
//   unpool : N point queries <- I = 64 inducer keys/values (nn.MultiheadAttention at
//            models/set_transformer.py:90,112): all 64 keys resident in LDS, one pass.
//
// Both compute S^T = K Q^T (keys on the MFMA row index, queries on the lane) so that
//   * the softmax reduction over keys is in-lane (16 registers per 32x32 tile) + one xor-32
//     exchange between the two lane halves, and
//   * the probability tile is already the B operand of O^T = V^T P^T (accumulator-as-operand:
//     register e of lane half h is key mfma_row(e, h); the V fragment is read in that order).
// Softmax runs in the log2 domain (queries pre-scaled by log2(e)/sqrt(hd)).
#include "common.h"
#include "kernels.h"

#include <stdlib.h>

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// K/V/Q staging tiles are wave-private: a wave's own LDS writes are ordered before its later reads once they have
// completed, so a counter wait replaces the workgroup barrier and the four waves run decoupled.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ------------------------------------------------------------------------------------- pool
template <int HD>
__global__ __launch_bounds__(256, 2) void pool_attn_kernel(const float* __restrict__ KV,
                                                        const float* __restrict__ Qind,
                                                        float* __restrict__ part_o, float* __restrict__ part_ml,
                                                        int B, int N, int C, int H, int nsplit) {
    constexpr int KP = HD + 4;  // padded row stride of K / Q tiles (conflict-free ds_read_b128)
    constexpr int DT = (HD + 31) / 32;
    constexpr int CH = HD / 4;
    constexpr int LD_IT = (32 * CH + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
    const int b = bh / H, hh = bh % H;

    float* Qs = smem;
    float* Kt = smem + 64 * KP + wave * (32 * KP + 32 * HD);
    float* Vt = Kt + 32 * KP;

    const int ks = (((N + nsplit - 1) / nsplit) + 31) / 32 * 32;
    const int k_begin = split * ks;
    const int k_end = min(N, k_begin + ks);
    const int ntiles = k_end > k_begin ? (k_end - k_begin + 31) / 32 : 0;
    const int nit = (ntiles + 3) / 4;

    const float scale = LOG2E * rsqrtf((float)HD);
    for (int f = tid; f < 64 * CH; f += 256) {
        const int row = f / CH, ch = f % CH;
        f32x4 v = *reinterpret_cast<const f32x4*>(Qind + ((size_t)hh * 64 + row) * HD + ch * 4);
        *reinterpret_cast<f32x4*>(Qs + row * KP + ch * 4) = v * scale;
    }

    const size_t ldkv = 2 * (size_t)C;
    const float* Kg = KV + (size_t)b * N * ldkv + hh * HD;
    const float* Vg = Kg + C;

    f32x4 rk[LD_IT], rv[LD_IT];
    auto load_tile = [&](int tile) {
        const int base = k_begin + tile * 32;
#pragma unroll
        for (int it = 0; it < LD_IT; ++it) {
            const int f = it * 64 + lane, row = f / CH, ch = f % CH, key = base + row;
            f32x4 zk = {0.f, 0.f, 0.f, 0.f}, zv = {0.f, 0.f, 0.f, 0.f};
            if (f < 32 * CH && tile < ntiles && key < k_end) {
                zk = *reinterpret_cast<const f32x4*>(Kg + key * ldkv + ch * 4);
                zv = *reinterpret_cast<const f32x4*>(Vg + key * ldkv + ch * 4);
            }
            rk[it] = zk;
            rv[it] = zv;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < LD_IT; ++it) {
            const int f = it * 64 + lane, row = f / CH, ch = f % CH;
            if (f < 32 * CH) {
                *reinterpret_cast<f32x4*>(Kt + row * KP + ch * 4) = rk[it];
                *reinterpret_cast<f32x4*>(Vt + row * HD + ch * 4) = rv[it];
            }
        }
    };

    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    f32x16 O[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) O[dt][j][e] = 0.f;

    load_tile(wave);
    __syncthreads();  // the shared query tile Qs is complete
    for (int it = 0; it < nit; ++it) {
        const int tile = wave + 4 * it;
        store_tile();
        wave_lds_sync();
        load_tile(tile + 4);
        if (tile < ntiles) {
            f32x16 s[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s[j][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < HD / 8; ++kk) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(Kt + r * KP + kk * 8 + 4 * h);
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(Qs + r * KP + kk * 8 + 4 * h);
                const f32x4 q1 = *reinterpret_cast<const f32x4*>(Qs + (32 + r) * KP + kk * 8 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[0] = mfma32(a[e], q0[e], s[0]);
                    s[1] = mfma32(a[e], q1[e], s[1]);
                }
            }
            const int kbase = k_begin + tile * 32;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float mx = -INFINITY;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    if (kbase + mfma_row(e, h) >= k_end) s[j][e] = -INFINITY;
                    mx = fmaxf(mx, s[j][e]);
                }
                mx = fmaxf(mx, xor32(mx));
                const float mn = fmaxf(m[j], mx);  // finite: the tile holds >= 1 valid key
                const float alpha = exp2f(m[j] - mn);
                float ps = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    s[j][e] = exp2f(s[j][e] - mn);
                    ps += s[j][e];
                }
                l[j] = l[j] * alpha + ps;
                m[j] = mn;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) O[dt][j][e] *= alpha;
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int dcol = min(dt * 32 + r, HD - 1);  // padded rows duplicate a valid column
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float av = Vt[mfma_row(e, h) * HD + dcol];
                    O[dt][0] = mfma32(av, s[0][e], O[dt][0]);
                    O[dt][1] = mfma32(av, s[1][e], O[dt][1]);
                }
            }
        }
        wave_lds_sync();  // this wave's reads of Kt / Vt are done before it overwrites them
    }
    __syncthreads();      // every wave is done with its staging area: the combine below reuses the LDS

    // ---- combine the four waves' (m, l, O) and emit one partial per (b, head, split)
    float* Ow = smem;                 // [4][HD][64]
    float* Mw = smem + 4 * HD * 64;   // [4][64]
    float* Lw = Mw + 256;             // [4][64]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float lt = l[j] + xor32(l[j]);
        if (h == 0) {
            Mw[wave * 64 + j * 32 + r] = m[j];
            Lw[wave * 64 + j * 32 + r] = lt;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int d = dt * 32 + mfma_row(e, h);
                if (d < HD) Ow[(wave * HD + d) * 64 + j * 32 + r] = O[dt][j][e];
            }
    }
    __syncthreads();
    {
        const int q = tid & 63, part = tid >> 6;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, Mw[w * 64 + q]);
        float f[4], L = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = Mw[w * 64 + q];
            f[w] = (mw == -INFINITY) ? 0.f : exp2f(mw - M);
            L += f[w] * Lw[w * 64 + q];
        }
        const size_t pbase = ((size_t)bh * nsplit + split) * 64 + q;
        for (int d = part; d < HD; d += 4) {
            float o = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) o += f[w] * Ow[(w * HD + d) * 64 + q];
            part_o[pbase * HD + d] = o;
        }
        if (part == 0) {
            part_ml[pbase * 2 + 0] = M;
            part_ml[pbase * 2 + 1] = L;
        }
    }
}

// merged[b, i, h*HD + d] = sum_s f_s O_s / sum_s f_s l_s  ("b h i d -> b i (h d)")
__global__ void pool_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                  float* __restrict__ merged, int B, int C, int H, int nsplit) {
    const int HD = C / H;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * 64 * C) return;
    const int c = idx % C, i = (idx / C) % 64, b = idx / ((size_t)C * 64);
    const int hh = c / HD, d = c % HD;
    const size_t base = ((size_t)(b * H + hh) * nsplit) * 64 + i;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part_ml[(base + (size_t)s * 64) * 2]);
    float num = 0.f, den = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const size_t p = base + (size_t)s * 64;
        const float ms = part_ml[p * 2];
        const float f = (ms == -INFINITY) ? 0.f : exp2f(ms - M);
        num += f * part_o[p * HD + d];
        den += f * part_ml[p * 2 + 1];
    }
    merged[idx] = num / den;
}

// ----------------------------------------------------------------------------------- unpool
template <int HD>
__global__ __launch_bounds__(256) void unpool_attn_kernel(const float* __restrict__ q, const float* __restrict__ kvh,
                                                          float* __restrict__ out, int B, int N, int C, int H,
                                                          int tiles_per_wave, int nchunk) {
    constexpr int KP = HD + 4;
    constexpr int DT = (HD + 31) / 32;
    constexpr int CH = HD / 4;
    constexpr int LD_IT = (32 * CH + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int chunk = blockIdx.x % nchunk, bh = blockIdx.x / nchunk;
    const int b = bh / H, hh = bh % H;

    float* Ks = smem;               // [64][KP]
    float* Vs = smem + 64 * KP;     // [64][HD]
    float* Qt = Vs + 64 * HD + wave * 32 * KP;  // [32][KP] per wave

    for (int f = tid; f < 64 * CH; f += 256) {
        const int row = f / CH, ch = f % CH;
        const float* src = kvh + ((size_t)b * 64 + row) * 2 * C + hh * HD + ch * 4;
        *reinterpret_cast<f32x4*>(Ks + row * KP + ch * 4) = *reinterpret_cast<const f32x4*>(src);
        *reinterpret_cast<f32x4*>(Vs + row * HD + ch * 4) = *reinterpret_cast<const f32x4*>(src + C);
    }
    const float scale = LOG2E * rsqrtf((float)HD);
    const float* qb = q + (size_t)b * N * C + hh * HD;
    float* ob = out + (size_t)b * N * C + hh * HD;

    // the next tile's queries travel global -> registers while the current tile is in the matrix pipe
    f32x4 rq[LD_IT];
    auto load_q = [&](int it) {
        const int q0 = (chunk * tiles_per_wave + it) * 128 + wave * 32;
#pragma unroll
        for (int ld = 0; ld < LD_IT; ++ld) {
            const int f = ld * 64 + lane, row = f / CH, ch = f % CH, n = q0 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (f < 32 * CH && it < tiles_per_wave && n < N) v = *reinterpret_cast<const f32x4*>(qb + (size_t)n * C + ch * 4);
            rq[ld] = v;
        }
    };
    load_q(0);
    for (int it = 0; it < tiles_per_wave; ++it) {
        const int q0 = (chunk * tiles_per_wave + it) * 128 + wave * 32;
#pragma unroll
        for (int ld = 0; ld < LD_IT; ++ld) {
            const int f = ld * 64 + lane, row = f / CH, ch = f % CH;
            if (f < 32 * CH) *reinterpret_cast<f32x4*>(Qt + row * KP + ch * 4) = rq[ld] * scale;
        }
        load_q(it + 1);
        if (it == 0) __syncthreads();  // the shared key / value tiles Ks, Vs are complete
        wave_lds_sync();
        f32x16 s[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) s[rt][e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < HD / 8; ++kk) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(Qt + r * KP + kk * 8 + 4 * h);
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(Ks + r * KP + kk * 8 + 4 * h);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(Ks + (32 + r) * KP + kk * 8 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[0] = mfma32(a0[e], bq[e], s[0]);
                s[1] = mfma32(a1[e], bq[e], s[1]);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[rt][e]);
        mx = fmaxf(mx, xor32(mx));
        float ls = 0.f;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[rt][e] = exp2f(s[rt][e] - mx);
                ls += s[rt][e];
            }
        ls += xor32(ls);
        const float inv = 1.0f / ls;
        f32x16 O[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) O[dt][e] = 0.f;
            const int dcol = min(dt * 32 + r, HD - 1);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    O[dt] = mfma32(Vs[(rt * 32 + mfma_row(e, h)) * HD + dcol], s[rt][e], O[dt]);
        }
        wave_lds_sync();
        // transpose O^T (query on the lane) back to rows through the wave's Q tile
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = dt * 32 + 8 * g4 + 4 * h;
                if (d < HD) {
                    f32x4 v = {O[dt][4 * g4], O[dt][4 * g4 + 1], O[dt][4 * g4 + 2], O[dt][4 * g4 + 3]};
                    *reinterpret_cast<f32x4*>(Qt + r * KP + d) = v * inv;
                }
            }
        wave_lds_sync();
#pragma unroll
        for (int ld = 0; ld < LD_IT; ++ld) {
            const int f = ld * 64 + lane, row = f / CH, ch = f % CH, n = q0 + row;
            if (f < 32 * CH && n < N)
                *reinterpret_cast<f32x4*>(ob + (size_t)n * C + ch * 4) =
                    *reinterpret_cast<const f32x4*>(Qt + row * KP + ch * 4);
        }
        wave_lds_sync();
    }
}

template <int HD>
int pool_launch_t(const float* KV, const float* ind, float* po, float* pml, int B, int N, int C, int H, int nsplit,
                  hipStream_t st) {
    constexpr int KP = HD + 4;
    const size_t a = (size_t)64 * KP + 4 * (32 * KP + 32 * HD), c = (size_t)4 * HD * 64 + 512;
    const size_t lds = (a > c ? a : c) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pool_attn_kernel<HD>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((pool_attn_kernel<HD>), dim3(B * H * nsplit), dim3(256), lds, st, KV, ind, po, pml, B, N, C, H,
                       nsplit);
    return (int)hipGetLastError();
}

template <int HD>
int unpool_launch_t(const float* q, const float* kvh, float* out, int B, int N, int C, int H, hipStream_t st) {
    constexpr int KP = HD + 4;
    const size_t lds = ((size_t)64 * KP + 64 * HD + 4 * 32 * KP) * sizeof(float);
    const int tiles = (N + 127) / 128;
    int tpw = 1;
    while (tpw < 4 && (long)B * H * ((tiles + tpw * 2 - 1) / (tpw * 2)) >= 2048) tpw *= 2;
    const int nchunk = (tiles + tpw - 1) / tpw;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(unpool_attn_kernel<HD>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((unpool_attn_kernel<HD>), dim3(B * H * nchunk), dim3(256), lds, st, q, kvh, out, B, N, C, H, tpw,
                       nchunk);
    return (int)hipGetLastError();
}

}  // namespace

int pool_attn_nsplit(int B, int N, int H) {
    // A function of N only: the key-split (hence the summation order) must not depend on the batch
    // size, so a sample's result is bit-identical whatever batch (or GPU shard) it is evaluated in.
    (void)B; (void)H;
    static int keys = 0;
    if (!keys) {
        // keys per split below which a cloud is not split further.  2048 since the inducer chain merges the partials in every block of
        // its cluster (three times per sample at d = 384): one split at N = 2048 instead of two — C2 4.873 / 4.882 -> 4.842 / 4.844 ms,
        // the other shapes and the training step unchanged (1024 before: profiles/r04d_negative_results.txt had it neutral then)
        const char* e = getenv("GECCO_POOL_SPLIT_KEYS");
        keys = e && atoi(e) >= 32 ? atoi(e) : 2048;
    }
    int ns = 1;
    while (ns < 8 && N / (ns * 2) >= keys) ns *= 2;
    return ns;
}

int pool_attn_launch(const float* KV, const float* inducers, float* part_o, float* part_ml, float* merged, int B,
                     int N, int C, int H, int I, int nsplit, hipStream_t st, int precision, int io16, int hm) {
    if (I != 64 || C % H) return -3;
    if (io16 && !(precision == 2 && attn_x3_supported(C / H))) return -9;
    if (hm && !io16) return -9;
    const int HD = C / H;
    int rc;
    if (precision >= 1 && attn_x3_supported(HD)) rc = pool_attn_x3_partials_launch(KV, inducers, part_o, part_ml, B, N, C, H, nsplit, st, precision, io16, hm);
    else switch (HD) {
        case 8: rc = pool_launch_t<8>(KV, inducers, part_o, part_ml, B, N, C, H, nsplit, st); break;
        case 16: rc = pool_launch_t<16>(KV, inducers, part_o, part_ml, B, N, C, H, nsplit, st); break;
        case 24: rc = pool_launch_t<24>(KV, inducers, part_o, part_ml, B, N, C, H, nsplit, st); break;
        case 32: rc = pool_launch_t<32>(KV, inducers, part_o, part_ml, B, N, C, H, nsplit, st); break;
        case 40: rc = pool_launch_t<40>(KV, inducers, part_o, part_ml, B, N, C, H, nsplit, st); break;
        case 48: rc = pool_launch_t<48>(KV, inducers, part_o, part_ml, B, N, C, H, nsplit, st); break;
        case 56: rc = pool_launch_t<56>(KV, inducers, part_o, part_ml, B, N, C, H, nsplit, st); break;
        case 64: rc = pool_launch_t<64>(KV, inducers, part_o, part_ml, B, N, C, H, nsplit, st); break;
        default: return -4;
    }
    if (rc) return rc;
    if (!merged) return 0;   // the caller merges the partials itself (inducer_chain_f16.hip)
    const size_t total = (size_t)B * 64 * C;
    hipLaunchKernelGGL(pool_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, part_o, part_ml,
                       merged, B, C, H, nsplit);
    return (int)hipGetLastError();
}

int unpool_attn_launch(const float* q, const float* kvh, float* out, int B, int N, int C, int H, int I,
                       hipStream_t st, int precision, int io16, int hm, int out_img) {
    if (I != 64 || C % H) return -3;
    if (io16 && !(precision == 2 && attn_x3_supported(C / H))) return -9;
    if (hm && !io16) return -9;
    if (out_img && !(precision >= 1 && attn_x3_supported(C / H))) return -9;
    if (precision >= 1 && attn_x3_supported(C / H)) return unpool_attn_x3_launch(q, kvh, out, B, N, C, H, st, precision, io16, hm, out_img);
    switch (C / H) {
        case 8: return unpool_launch_t<8>(q, kvh, out, B, N, C, H, st);
        case 16: return unpool_launch_t<16>(q, kvh, out, B, N, C, H, st);
        case 24: return unpool_launch_t<24>(q, kvh, out, B, N, C, H, st);
        case 32: return unpool_launch_t<32>(q, kvh, out, B, N, C, H, st);
        case 40: return unpool_launch_t<40>(q, kvh, out, B, N, C, H, st);
        case 48: return unpool_launch_t<48>(q, kvh, out, B, N, C, H, st);
        case 56: return unpool_launch_t<56>(q, kvh, out, B, N, C, H, st);
        case 64: return unpool_launch_t<64>(q, kvh, out, B, N, C, H, st);
        default: return -4;
    }
}
