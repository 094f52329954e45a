// Backward of the inducing-point attention (training path), fused, on fp32 MFMA — gfx950.
//
// Reference: autograd through `F.scaled_dot_product_attention` in AttentionPool (models/set_transformer.py:55-63) and
// through nn.MultiheadAttention (models/set_transformer.py:112) under `loss.backward()` (diffusion.py:213-222).  The
// unfused form (five strided-batched GEMMs + softmax kernels per attention, scores and probabilities of shape
// (B, H, N, 64) materialised in HBM three times) was 36 % of the training step; here the probabilities are recomputed
// tile by tile from the saved operands and never leave the CU:
//
//   pool   (64 inducer queries <- N keys):   P = exp2(S log2e/sqrt(hd) - LSE)      LSE saved by the forward (per b, h, i)
//   unpool (N queries <- 64 inducer keys):   the whole softmax row is recomputed (64 keys are resident)
//   D_i = sum_d dO O (pool; from the saved output)   /   D_n = sum_i P dP (unpool; in registers)
//   dS = P (dP - D) / sqrt(hd);   dV = P^T dO;   dK = dS^T Q;   dQ = dS K
//
// Layout idiom of attention_f32.hip: scores are produced TRANSPOSED (keys on the MFMA row index, queries on the lane),
// so the softmax statistics are per-lane scalars, and a product that contracts over the accumulator's ROW index takes
// the accumulator registers as its B operand directly (dQ^T = K^T dS^T for pool, dq^T = k^T dS^T for unpool).  The two
// products that contract over the LANE index (pool: dV, dK over the 64 queries; unpool: dv, dk over the tile's 32
// queries) read the probability / dS tile back from a wave-private LDS tile as the A operand.
// Deterministic: per-block partials of the reductions over N (dQ of the pool, dk | dv of the unpool) are summed in a
// fixed order by reduce_batch_kernel — no float atomics.
#include "common.h"
#include "kernels.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// lse2[b, h, i] = log2 sum_n exp2(s2[i, n]) from the forward's per-split (max, sum) partials (log2 domain, scaled scores)
__global__ void pool_lse_kernel(const float* __restrict__ part_ml, float* __restrict__ lse, int total, int nsplit) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int bh = idx / 64, i = idx % 64;
    const size_t base = (size_t)bh * nsplit * 64 + i;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part_ml[(base + (size_t)s * 64) * 2]);
    float L = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float ms = part_ml[(base + (size_t)s * 64) * 2];
        if (ms != -INFINITY) L += exp2f(ms - M) * part_ml[(base + (size_t)s * 64) * 2 + 1];
    }
    lse[idx] = M + log2f(L);
}

// ------------------------------------------------------------------------------------- pool
// grid: (b, head, split of the keys); 4 waves, each walks 32-key tiles (wave-private K / V / P tiles).
template <int HD>
__global__ __launch_bounds__(256) void pool_attn_bwd_kernel(const float* __restrict__ KV, const float* __restrict__ Qind,
                                                            const float* __restrict__ Omerged, const float* __restrict__ lse,
                                                            const float* __restrict__ dO, float* __restrict__ dKV,
                                                            float* __restrict__ dQpart, int B, int N, int C, int H, int nsplit) {
    constexpr int KP = HD + 4, DT = (HD + 31) / 32, CH = HD / 4, LD_IT = (32 * CH + 63) / 64, PP = 68;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x / nsplit, split = blockIdx.x % nsplit;
    const int b = bh / H, hh = bh % H;

    float* Qs = smem;                    // [64][KP] inducer queries of this head (raw)
    float* Gs = Qs + 64 * KP;            // [64][KP] dO rows of (b, head)
    float* Ls = Gs + 64 * KP;            // [64] lse2, [64] D
    float* Kt = Ls + 128 + wave * (2 * 32 * KP + 32 * PP);
    float* Vt = Kt + 32 * KP;
    float* Pt = Vt + 32 * KP;            // [32 keys][PP]: P^T, then dS^T, as the A operand of dV / dK

    const int ks = (((N + nsplit - 1) / nsplit) + 31) / 32 * 32;
    const int k_begin = split * ks, k_end = min(N, k_begin + ks);
    const int ntiles = k_end > k_begin ? (k_end - k_begin + 31) / 32 : 0;
    const int nit = (ntiles + 3) / 4;

    for (int f = tid; f < 64 * CH; f += 256) {
        const int row = f / CH, ch = f % CH;
        *reinterpret_cast<f32x4*>(Qs + row * KP + ch * 4) = *reinterpret_cast<const f32x4*>(Qind + ((size_t)hh * 64 + row) * HD + ch * 4);
        *reinterpret_cast<f32x4*>(Gs + row * KP + ch * 4) = *reinterpret_cast<const f32x4*>(dO + ((size_t)b * 64 + row) * C + hh * HD + ch * 4);
    }
    if (tid < 64) {
        const float* o = Omerged + ((size_t)b * 64 + tid) * C + hh * HD;
        const float* g = dO + ((size_t)b * 64 + tid) * C + hh * HD;
        float d = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < CH; ++c4) {
            const f32x4 ov = *reinterpret_cast<const f32x4*>(o + c4 * 4), gv = *reinterpret_cast<const f32x4*>(g + c4 * 4);
            d += ov[0] * gv[0] + ov[1] * gv[1] + ov[2] * gv[2] + ov[3] * gv[3];
        }
        Ls[64 + tid] = d;
        Ls[tid] = lse[(size_t)bh * 64 + tid];
    }

    const size_t ldkv = 2 * (size_t)C;
    const float* Kg = KV + (size_t)b * N * ldkv + hh * HD;
    const float* Vg = Kg + C;
    float* dKg = dKV + (size_t)b * N * ldkv + hh * HD;
    float* dVg = dKg + C;

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
                *reinterpret_cast<f32x4*>(Vt + row * KP + ch * 4) = rv[it];
            }
        }
    };

    f32x16 dQ[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) dQ[dt][j][e] = 0.f;

    const float sc = rsqrtf((float)HD), scale2 = LOG2E * sc;
    load_tile(wave);
    __syncthreads();   // Qs, Gs, Ls complete
    float lsej[2], Dj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        lsej[j] = Ls[32 * j + r];
        Dj[j] = Ls[64 + 32 * j + r];
    }
    int dcol[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dcol[dt] = min(dt * 32 + r, HD - 1);   // padded columns duplicate a valid one (never stored)

    for (int it = 0; it < nit; ++it) {
        const int tile = wave + 4 * it;
        store_tile();
        wave_lds_sync();
        load_tile(tile + 4);
        if (tile < ntiles) {
            const int kbase = k_begin + tile * 32;
            f32x16 p[2], dp[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) { p[j][e] = 0.f; dp[j][e] = 0.f; }
            // S^T = K Q^T and dP^T = V dO^T (keys on the row index, queries on the lane)
#pragma unroll
            for (int kk = 0; kk < HD / 8; ++kk) {
                const f32x4 ak = *reinterpret_cast<const f32x4*>(Kt + r * KP + kk * 8 + 4 * h);
                const f32x4 av = *reinterpret_cast<const f32x4*>(Vt + r * KP + kk * 8 + 4 * h);
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(Qs + r * KP + kk * 8 + 4 * h);
                const f32x4 q1 = *reinterpret_cast<const f32x4*>(Qs + (32 + r) * KP + kk * 8 + 4 * h);
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(Gs + r * KP + kk * 8 + 4 * h);
                const f32x4 g1 = *reinterpret_cast<const f32x4*>(Gs + (32 + r) * KP + kk * 8 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[0] = mfma32(ak[e], q0[e], p[0]);
                    p[1] = mfma32(ak[e], q1[e], p[1]);
                    dp[0] = mfma32(av[e], g0[e], dp[0]);
                    dp[1] = mfma32(av[e], g1[e], dp[1]);
                }
            }
            // P^T, then dS^T (in dp)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const bool valid = kbase + mfma_row(e, h) < k_end;
                    const float pv = valid ? exp2f(p[j][e] * scale2 - lsej[j]) : 0.f;
                    p[j][e] = pv;
                    dp[j][e] = pv * (dp[j][e] - Dj[j]) * sc;
                }
            // dQ^T[d, i] += sum_key K[key, d] dS^T[key, i]: the dS^T registers are the B operand as they are
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float a = Kt[mfma_row(e, h) * KP + dcol[dt]];
                    dQ[dt][0] = mfma32(a, dp[0][e], dQ[dt][0]);
                    dQ[dt][1] = mfma32(a, dp[1][e], dQ[dt][1]);
                }
            // dV[key, d] = sum_i P^T[key, i] dO[i, d];  dK[key, d] = sum_i dS^T[key, i] Q[i, d]
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) Pt[mfma_row(e, h) * PP + 32 * j + r] = pass == 0 ? p[j][e] : dp[j][e];
                wave_lds_sync();
                const float* Bs = pass == 0 ? Gs : Qs;
                f32x16 acc[DT];
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[dt][e] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(Pt + r * PP + kk * 8 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = kk * 8 + 4 * h + e;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) acc[dt] = mfma32(a[e], Bs[i * KP + dcol[dt]], acc[dt]);
                    }
                }
                float* dst = pass == 0 ? dVg : dKg;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kbase + mfma_row(e, h), d = dt * 32 + r;
                        if (key < k_end && d < HD) dst[key * ldkv + d] = acc[dt][e];
                    }
                wave_lds_sync();   // the tile's reads are done before the next pass / tile overwrites it
            }
        }
        wave_lds_sync();
    }
    __syncthreads();   // every wave is done with its staging area: the combine below reuses the LDS

    // ---- sum the four waves' dQ^T in wave order and emit the partial of this (b, head, split)
    float* Dw = smem;   // [4][HD][64]
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int d = dt * 32 + mfma_row(e, h);
                if (d < HD) Dw[(wave * HD + d) * 64 + 32 * j + r] = dQ[dt][j][e];
            }
    __syncthreads();
    float* out = dQpart + (((size_t)b * nsplit + split) * H + hh) * 64 * HD;
    for (int f = tid; f < 64 * HD; f += 256) {
        const int i = f / HD, d = f % HD;
        out[f] = ((Dw[(0 * HD + d) * 64 + i] + Dw[(1 * HD + d) * 64 + i]) + Dw[(2 * HD + d) * 64 + i]) + Dw[(3 * HD + d) * 64 + i];
    }
}

// ----------------------------------------------------------------------------------- unpool
// grid: (b, head, chunk of the queries); 4 waves, each walks 32-query tiles; the 64 inducer keys / values are resident.
template <int HD>
__global__ __launch_bounds__(256) void unpool_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ kvh,
                                                              const float* __restrict__ dO, float* __restrict__ dq,
                                                              float* __restrict__ dkv_part, int B, int N, int C, int H,
                                                              int tiles_per_wave, int nchunk) {
    constexpr int KP = HD + 4, DT = (HD + 31) / 32, CH = HD / 4, LD_IT = (32 * CH + 63) / 64, PP = 36;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int chunk = blockIdx.x % nchunk, bh = blockIdx.x / nchunk;
    const int b = bh / H, hh = bh % H;

    float* Ks = smem;               // [64][KP]
    float* Vs = Ks + 64 * KP;       // [64][KP]
    float* Qt = Vs + 64 * KP + wave * (2 * 32 * KP + 64 * PP);   // [32][KP] raw queries of the tile
    float* Gt = Qt + 32 * KP;       // [32][KP] dO rows of the tile
    float* Pt = Gt + 32 * KP;       // [64 keys][PP]: P^T, then dS^T

    for (int f = tid; f < 64 * CH; f += 256) {
        const int row = f / CH, ch = f % CH;
        const float* src = kvh + ((size_t)b * 64 + row) * 2 * C + hh * HD + ch * 4;
        *reinterpret_cast<f32x4*>(Ks + row * KP + ch * 4) = *reinterpret_cast<const f32x4*>(src);
        *reinterpret_cast<f32x4*>(Vs + row * KP + ch * 4) = *reinterpret_cast<const f32x4*>(src + C);
    }
    const float sc = rsqrtf((float)HD), scale2 = LOG2E * sc;
    const float* qb = q + (size_t)b * N * C + hh * HD;
    const float* gb = dO + (size_t)b * N * C + hh * HD;
    float* dqb = dq + (size_t)b * N * C + hh * HD;

    f32x4 rq[LD_IT], rg[LD_IT];
    auto load_q = [&](int it) {
        const int q0 = (chunk * tiles_per_wave + it) * 128 + wave * 32;
#pragma unroll
        for (int ld = 0; ld < LD_IT; ++ld) {
            const int f = ld * 64 + lane, row = f / CH, ch = f % CH, n = q0 + row;
            f32x4 v = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
            if (f < 32 * CH && it < tiles_per_wave && n < N) {
                v = *reinterpret_cast<const f32x4*>(qb + (size_t)n * C + ch * 4);
                g = *reinterpret_cast<const f32x4*>(gb + (size_t)n * C + ch * 4);
            }
            rq[ld] = v;
            rg[ld] = g;
        }
    };

    f32x16 dk[2][DT], dv[2][DT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { dk[rt][dt][e] = 0.f; dv[rt][dt][e] = 0.f; }
    int dcol[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dcol[dt] = min(dt * 32 + r, HD - 1);

    load_q(0);
    for (int it = 0; it < tiles_per_wave; ++it) {
        const int q0 = (chunk * tiles_per_wave + it) * 128 + wave * 32;
#pragma unroll
        for (int ld = 0; ld < LD_IT; ++ld) {
            const int f = ld * 64 + lane, row = f / CH, ch = f % CH;
            if (f < 32 * CH) {
                *reinterpret_cast<f32x4*>(Qt + row * KP + ch * 4) = rq[ld];
                *reinterpret_cast<f32x4*>(Gt + row * KP + ch * 4) = rg[ld];
            }
        }
        load_q(it + 1);
        if (it == 0) __syncthreads();   // Ks, Vs complete
        wave_lds_sync();
        if (q0 < N) {   // wave-uniform: tiles past the end carry nothing (zero rows contribute zero anyway)
            f32x16 p[2], dp[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { p[rt][e] = 0.f; dp[rt][e] = 0.f; }
            // S^T = K q^T, dP^T = V dO^T (inducer keys on the row index, the tile's queries on the lane)
#pragma unroll
            for (int kk = 0; kk < HD / 8; ++kk) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(Qt + r * KP + kk * 8 + 4 * h);
                const f32x4 bg = *reinterpret_cast<const f32x4*>(Gt + r * KP + kk * 8 + 4 * h);
                const f32x4 k0 = *reinterpret_cast<const f32x4*>(Ks + r * KP + kk * 8 + 4 * h);
                const f32x4 k1 = *reinterpret_cast<const f32x4*>(Ks + (32 + r) * KP + kk * 8 + 4 * h);
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(Vs + r * KP + kk * 8 + 4 * h);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(Vs + (32 + r) * KP + kk * 8 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[0] = mfma32(k0[e], bq[e], p[0]);
                    p[1] = mfma32(k1[e], bq[e], p[1]);
                    dp[0] = mfma32(v0[e], bg[e], dp[0]);
                    dp[1] = mfma32(v1[e], bg[e], dp[1]);
                }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { p[rt][e] *= scale2; mx = fmaxf(mx, p[rt][e]); }
            mx = fmaxf(mx, xor32(mx));
            float ls = 0.f;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { p[rt][e] = exp2f(p[rt][e] - mx); ls += p[rt][e]; }
            ls += xor32(ls);
            const float inv = 1.0f / ls;
            float Dn = 0.f;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) { p[rt][e] *= inv; Dn += p[rt][e] * dp[rt][e]; }
            Dn += xor32(Dn);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int e = 0; e < 16; ++e) dp[rt][e] = p[rt][e] * (dp[rt][e] - Dn) * sc;   // dS^T
            // dq^T[d, n] = sum_i k[i, d] dS^T[i, n]: accumulator registers as the B operand
            f32x16 O[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) O[dt][e] = 0.f;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        O[dt] = mfma32(Ks[(rt * 32 + mfma_row(e, h)) * KP + dcol[dt]], dp[rt][e], O[dt]);
            }
            // dv[i, d] += sum_n P^T[i, n] dO[n, d];  dk[i, d] += sum_n dS^T[i, n] q[n, d]
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) Pt[(rt * 32 + mfma_row(e, h)) * PP + r] = pass == 0 ? p[rt][e] : dp[rt][e];
                wave_lds_sync();
                const float* Bs = pass == 0 ? Gt : Qt;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(Pt + r * PP + kk * 8 + 4 * h);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(Pt + (32 + r) * PP + kk * 8 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = kk * 8 + 4 * h + e;
#pragma unroll
                        for (int dt = 0; dt < DT; ++dt) {
                            const float bv = Bs[n * KP + dcol[dt]];
                            if (pass == 0) {
                                dv[0][dt] = mfma32(a0[e], bv, dv[0][dt]);
                                dv[1][dt] = mfma32(a1[e], bv, dv[1][dt]);
                            } else {
                                dk[0][dt] = mfma32(a0[e], bv, dk[0][dt]);
                                dk[1][dt] = mfma32(a1[e], bv, dk[1][dt]);
                            }
                        }
                    }
                }
                wave_lds_sync();
            }
            // dq rows: transpose dq^T (query on the lane) through the wave's dO tile, then coalesced row stores
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d = dt * 32 + 8 * g4 + 4 * h;
                    if (d < HD) {
                        f32x4 v = {O[dt][4 * g4], O[dt][4 * g4 + 1], O[dt][4 * g4 + 2], O[dt][4 * g4 + 3]};
                        *reinterpret_cast<f32x4*>(Gt + r * KP + d) = v;
                    }
                }
            wave_lds_sync();
#pragma unroll
            for (int ld = 0; ld < LD_IT; ++ld) {
                const int f = ld * 64 + lane, row = f / CH, ch = f % CH, n = q0 + row;
                if (f < 32 * CH && n < N)
                    *reinterpret_cast<f32x4*>(dqb + (size_t)n * C + ch * 4) = *reinterpret_cast<const f32x4*>(Gt + row * KP + ch * 4);
            }
        }
        wave_lds_sync();
    }
    __syncthreads();

    // ---- sum the four waves' dk | dv in wave order: partial of this (chunk, b, head)
    float* Dw = smem;   // [4 waves][2 (k, v)][64][HD]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = rt * 32 + mfma_row(e, h), d = dt * 32 + r;
                if (d < HD) {
                    Dw[((wave * 2 + 0) * 64 + i) * HD + d] = dk[rt][dt][e];
                    Dw[((wave * 2 + 1) * 64 + i) * HD + d] = dv[rt][dt][e];
                }
            }
    __syncthreads();
    float* out = dkv_part + ((size_t)chunk * B + b) * 64 * 2 * C + hh * HD;
    for (int f = tid; f < 2 * 64 * HD; f += 256) {
        const int kv = f / (64 * HD), i = (f / HD) % 64, d = f % HD;
        const int o = (kv * 64 + i) * HD + d;
        out[(size_t)i * 2 * C + kv * C + d] = ((Dw[o] + Dw[o + 2 * 64 * HD]) + Dw[o + 4 * 64 * HD]) + Dw[o + 6 * 64 * HD];
    }
}

template <int HD>
int pool_bwd_t(const float* KV, const float* ind, const float* O, const float* lse, const float* dO, float* dKV, float* dQp,
               int B, int N, int C, int H, int nsplit, hipStream_t st) {
    constexpr int KP = HD + 4;
    const size_t a = (size_t)2 * 64 * KP + 128 + 4 * (2 * 32 * KP + 32 * 68), c = (size_t)4 * HD * 64;
    const size_t lds = (a > c ? a : c) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pool_attn_bwd_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((pool_attn_bwd_kernel<HD>), dim3(B * H * nsplit), dim3(256), lds, st, KV, ind, O, lse, dO, dKV, dQp, B, N, C, H, nsplit);
    return (int)hipGetLastError();
}

template <int HD>
int unpool_bwd_t(const float* q, const float* kvh, const float* dO, float* dq, float* part, int B, int N, int C, int H, int tpw,
                 int nchunk, hipStream_t st) {
    constexpr int KP = HD + 4;
    const size_t a = (size_t)2 * 64 * KP + 4 * (2 * 32 * KP + 64 * 36), c = (size_t)8 * 64 * HD;
    const size_t lds = (a > c ? a : c) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(unpool_attn_bwd_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((unpool_attn_bwd_kernel<HD>), dim3(B * H * nchunk), dim3(256), lds, st, q, kvh, dO, dq, part, B, N, C, H, tpw, nchunk);
    return (int)hipGetLastError();
}

}  // namespace

int pool_attn_lse_launch(const float* part_ml, float* lse, int B, int H, int nsplit, hipStream_t st) {
    const int total = B * H * 64;
    hipLaunchKernelGGL(pool_lse_kernel, dim3((total + 255) / 256), dim3(256), 0, st, part_ml, lse, total, nsplit);
    return (int)hipGetLastError();
}

// key split of the pool backward: enough blocks to fill the chip, at least 8 key tiles per block
int pool_attn_bwd_nsplit(int B, int N, int H) {
    int ns = 1;
    while ((long)B * H * ns < 1024 && N / (ns * 2) >= 256) ns *= 2;
    return ns;
}

// query tiles (of 128) per block of the unpool backward, and the number of chunks (= partials of dk | dv) it leaves
int unpool_attn_bwd_chunks(int B, int N, int H, int* tiles_per_wave) {
    const int tiles = (N + 127) / 128;
    int tpw = 1;
    while (tpw < 16 && (long)B * H * ((tiles + tpw * 2 - 1) / (tpw * 2)) >= 1024) tpw *= 2;
    if (tiles_per_wave) *tiles_per_wave = tpw;
    return (tiles + tpw - 1) / tpw;
}

int pool_attn_bwd_launch(const float* KV, const float* inducers, const float* merged, const float* lse, const float* dO,
                         float* dKV, float* dQpart, int B, int N, int C, int H, int I, int nsplit, hipStream_t st, int precision) {
    if (I != 64 || C % H) return -3;
    if (precision == 3 && !(attn_bwd_x3_supported(C / H) && C % 4 == 0)) return -4;   // fp16 tensors: the x3 kernels only
    if (precision >= 1 && attn_bwd_x3_supported(C / H))
        return pool_attn_bwd_x3_launch(KV, inducers, merged, lse, dO, dKV, dQpart, B, N, C, H, nsplit, st, precision == 3 ? 2 : precision == 2);
    switch (C / H) {
        case 8: return pool_bwd_t<8>(KV, inducers, merged, lse, dO, dKV, dQpart, B, N, C, H, nsplit, st);
        case 16: return pool_bwd_t<16>(KV, inducers, merged, lse, dO, dKV, dQpart, B, N, C, H, nsplit, st);
        case 24: return pool_bwd_t<24>(KV, inducers, merged, lse, dO, dKV, dQpart, B, N, C, H, nsplit, st);
        case 32: return pool_bwd_t<32>(KV, inducers, merged, lse, dO, dKV, dQpart, B, N, C, H, nsplit, st);
        case 40: return pool_bwd_t<40>(KV, inducers, merged, lse, dO, dKV, dQpart, B, N, C, H, nsplit, st);
        case 48: return pool_bwd_t<48>(KV, inducers, merged, lse, dO, dKV, dQpart, B, N, C, H, nsplit, st);
        case 56: return pool_bwd_t<56>(KV, inducers, merged, lse, dO, dKV, dQpart, B, N, C, H, nsplit, st);
        case 64: return pool_bwd_t<64>(KV, inducers, merged, lse, dO, dKV, dQpart, B, N, C, H, nsplit, st);
        default: return -4;
    }
}

int unpool_attn_bwd_launch(const float* q, const float* kvh, const float* dO, float* dq, float* dkv_part, int B, int N, int C,
                           int H, int I, hipStream_t st, int precision) {
    if (I != 64 || C % H) return -3;
    int tpw;
    const int nchunk = unpool_attn_bwd_chunks(B, N, H, &tpw);
    if (precision == 3 && !(attn_bwd_x3_supported(C / H) && C % 4 == 0)) return -4;
    if (precision >= 1 && attn_bwd_x3_supported(C / H))
        return unpool_attn_bwd_x3_launch(q, kvh, dO, dq, dkv_part, B, N, C, H, tpw, nchunk, st, precision == 3 ? 2 : precision == 2);
    switch (C / H) {
        case 8: return unpool_bwd_t<8>(q, kvh, dO, dq, dkv_part, B, N, C, H, tpw, nchunk, st);
        case 16: return unpool_bwd_t<16>(q, kvh, dO, dq, dkv_part, B, N, C, H, tpw, nchunk, st);
        case 24: return unpool_bwd_t<24>(q, kvh, dO, dq, dkv_part, B, N, C, H, tpw, nchunk, st);
        case 32: return unpool_bwd_t<32>(q, kvh, dO, dq, dkv_part, B, N, C, H, tpw, nchunk, st);
        case 40: return unpool_bwd_t<40>(q, kvh, dO, dq, dkv_part, B, N, C, H, tpw, nchunk, st);
        case 48: return unpool_bwd_t<48>(q, kvh, dO, dq, dkv_part, B, N, C, H, tpw, nchunk, st);
        case 56: return unpool_bwd_t<56>(q, kvh, dO, dq, dkv_part, B, N, C, H, tpw, nchunk, st);
        case 64: return unpool_bwd_t<64>(q, kvh, dO, dq, dkv_part, B, N, C, H, tpw, nchunk, st);
        default: return -4;
    }
}
