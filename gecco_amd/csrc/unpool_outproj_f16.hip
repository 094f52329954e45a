// unpool attention + out_proj + residual + GroupNorm partials in ONE launch, fp16 mode (precision 2), gfx950.
//
// Second half of AttentionPool's round trip (reference models/set_transformer.py:70-75 and :112, nn.MultiheadAttention
// with the 64 inducer states as keys / values, then its out_proj and the residual of :164): the attention output of a
// row block is exactly the A operand of out_proj for the same rows.  Before, unpool_attn_x3_kernel wrote it to HBM as
// fp16 (100 MB at C2, in 96-byte row pieces) and the fp16-A GEMM read it back through its LDS-DMA ring, two launches.
// Here a block owns 128 rows:
//   * per pair of heads, k | v of the 64 inducers (fp32, L2) are staged as fp16 in the layouts of attention_x3.hip
//     (K rows padded, V transposed and key-permuted); wave (row group, head of the pair) computes S^T = K q^T with the
//     q fragments of its 32 rows loaded straight from the head-major q (contiguous 32 x hd slab), softmax over the 64
//     keys in registers, O^T = V^T P^T with the probability accumulator as the B operand, and writes fp16(O / l) into
//     the block's A buffer (128 x C fp16, XOR-swizzled 16-byte chunks) — the bits unpool_attn_x3_kernel stored;
//   * out_proj runs from that buffer with the step machine, weight ring and 16-byte transposed epilogue of
//     mlp_fused_f16.hip (8 waves = 4 row groups x 2 column halves, 128 x C fp32 accumulator in registers): bias,
//     residual x, store, per-(sample, row tile, column) GroupNorm partials.
// Bit-identical to unpool_attn_x3_kernel<hd, fp16, io16> followed by gemm_f16_kernel (fp16 A, residual, stats).
#include "gemm_dma_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

constexpr int MF_NT = 512;             // threads per block
constexpr int MF_TILE = 2048;          // floats per 8 KiB weight block
constexpr int UO_LDS_BYTES = 160 * 1024;
constexpr float UO_LOG2E = 1.4426950408889634f;

constexpr int uo_kv_u16(int HD) { return 64 * (HD + 8) + ((HD + 31) / 32) * 32 * 72; }   // per head: K rows | V^T rows
constexpr int uo_fixed_bytes(int C, int HD) { return 128 * 2 * C + 2 * uo_kv_u16(HD) * 2; }
constexpr int uo_ns(int C, int HD) {
    const int n = (UO_LDS_BYTES - uo_fixed_bytes(C, HD)) / (MF_TILE * 4);
    const int blocks = (C / 128) * (C / 32);   // the whole stream
    const int m = n > 12 ? 12 : n;
    return m > blocks ? blocks : m;
}
constexpr size_t uo_scratch_bytes(int C) { return (size_t)(8 * 32 * (64 + 4) + 4 * 2 * C) * 4; }   // epilogue: 8 transpose tiles + partial sums
constexpr size_t uo_lds_bytes(int C, int HD) {
    const size_t a = (size_t)uo_ns(C, HD) * MF_TILE * 4 + uo_fixed_bytes(C, HD), b = uo_scratch_bytes(C);
    return a > b ? a : b;
}

__device__ __forceinline__ void mf_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, float* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void mf_lds_barrier() {   // this wave's LDS writes landed, then the block barrier (no vmcnt wait)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ u32x4 uo_frag(const u16* p) { return *reinterpret_cast<const u32x4*>(p); }

template <int NT1, int HD>
__global__ __launch_bounds__(MF_NT, 1) void unpool_outproj_f16_kernel(UnpoolProjArgs g) {
    constexpr int C = 128 * NT1, H = C / HD;
    constexpr int NK = C / 32;                       // weight blocks per output tile
    constexpr int S_TOTAL = NT1 * NK;
    constexpr int NS = uo_ns(C, HD);
    constexpr int AHEAD = NS - 2;
    constexpr int RSY = 2 * C;                       // bytes per row of the A buffer (swizzled, no padding)
    constexpr int KS = HD + 8, VS = 64 + 8, DT = (HD + 31) / 32, NC = HD / 16, CH = HD / 4;
    constexpr int KVH = uo_kv_u16(HD);               // u16 per staged head
    static_assert(C % HD == 0 && H % 2 == 0 && HD % 16 == 0 && HD <= 64, "heads come in pairs; head dim 16 .. 64");
    static_assert(NS >= 4 && S_TOTAL >= NS && AHEAD <= 63, "ring");
    static_assert(CH % 4 == 0, "staging items per thread");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ring = smem;
    char* abuf = reinterpret_cast<char*>(smem + NS * MF_TILE);   // [128][C] fp16, chunk ci of row r at ci ^ (r & 15)
    u16* kvs = reinterpret_cast<u16*>(abuf + 128 * RSY);          // [2 heads]{K [64][KS] | V^T [DT * 32][VS]}

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int tilesM = g.rows / 128;
    const int bid = gridDim.x - 1 - blockIdx.x;   // newest q / x first (see pool_attn_x3_kernel)
    const int b = bid / tilesM, rt = bid % tilesM, m0 = rt * 128;

    // start-up stagger (mlp_fused_f16.hip): spread the HBM-bound phases of the rounds of blocks over 8 offsets
    if (g.stagger > 0 && blockIdx.x < 256) {
        const long long until = (long long)__builtin_amdgcn_s_memtime() + (long long)((blockIdx.x >> 3) & 7) * g.stagger;
        while ((long long)__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(8);
    }

    // ---- the weight stream: the fp16 image of out_proj, consecutive 8 KiB blocks; wave w moves piece w of every block
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w_stream), 0, 0x7fffffff, 0x00020000);
    const unsigned voff = (unsigned)(wave * 256 + lane * 4) * 4u;
    unsigned soff = 0;
    int islot = 0, issued = 0;
    auto issue = [&]() {
        mf_dma16(wrsrc, voff, soff, ring + islot * MF_TILE + wave * 256);
        soff += MF_TILE * 4u;
        islot = islot + 1 == NS ? 0 : islot + 1;
        ++issued;
    };
#pragma unroll 1
    for (int p = 0; p < NS - 1; ++p) issue();

    // ================= attention: 64 inducer keys / values, this block's 128 queries, two heads at a time
    {
        for (int i = tid; i < 2 * KVH / 2; i += MF_NT) reinterpret_cast<unsigned*>(kvs)[i] = 0u;   // padded V^T rows stay finite
        constexpr int SI = CH / 4;   // staging items (key, 4-float chunk) per thread and per K / V of a head pair
        const float* kvb = g.kvh + (size_t)b * 64 * 2 * C;
        f32x4 sk[SI], sv[SI];
        auto stage_load = [&](int hp) {
#pragma unroll
            for (int u = 0; u < SI; ++u) {
                const int i = tid + u * MF_NT, hh = 2 * hp + i / (64 * CH), rem = i % (64 * CH), key = rem / CH, ch = rem % CH;
                const float* src = kvb + (size_t)key * 2 * C + hh * HD + ch * 4;
                sk[u] = *reinterpret_cast<const f32x4*>(src);
                sv[u] = *reinterpret_cast<const f32x4*>(src + C);
            }
        };
        auto stage_store = [&]() {
#pragma unroll
            for (int u = 0; u < SI; ++u) {
                const int i = tid + u * MF_NT, hl = i / (64 * CH), rem = i % (64 * CH), key = rem / CH, ch = rem % CH;
                u16* Kh = kvs + hl * KVH;
                u16* Vt = Kh + 64 * KS;
                f16x4 kk, vv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    kk[e] = (_Float16)sk[u][e];
                    vv[e] = (_Float16)sv[u][e];
                }
                *reinterpret_cast<u32x2*>(Kh + key * KS + ch * 4) = __builtin_bit_cast(u32x2, kk);
                // key 16c + 8a + 4g + i sits at position 16c + 8g + 4a + i: lane half g reads its 8 keys contiguously
                const int pos = (key & ~15) + 8 * ((key >> 2) & 1) + 4 * ((key >> 3) & 1) + (key & 3);
                const u32x2 vb = __builtin_bit_cast(u32x2, vv);
#pragma unroll
                for (int e = 0; e < 4; ++e) Vt[(ch * 4 + e) * VS + pos] = (u16)(vb[e >> 1] >> (16 * (e & 1)));
            }
        };
        const float scale = UO_LOG2E * rsqrtf((float)HD);
        const _Float16* q16 = reinterpret_cast<const _Float16*>(g.q16);
        u32x4 qf[2][NC];   // q fragments of this wave's head, current pair and the next
        auto q_load = [&](int hp, int set) {
            const _Float16* qr = q16 + (((size_t)b * H + 2 * hp + wn) * g.rows + m0 + wm * 32 + r) * HD + 8 * h;
#pragma unroll
            for (int c = 0; c < NC; ++c) qf[set][c] = *reinterpret_cast<const u32x4*>(qr + c * 16);
        };
        stage_load(0);
        q_load(0, 0);
        mf_lds_barrier();   // zero fill done
#pragma unroll
        for (int hp = 0; hp < H / 2; ++hp) {
            const int set = hp & 1;
            if (hp) mf_lds_barrier();   // every wave is done with the previous pair's keys / values
            stage_store();
            if (hp + 1 < H / 2) {
                stage_load(hp + 1);
                q_load(hp + 1, set ^ 1);
            }
            mf_lds_barrier();
            const u16* Kh = kvs + wn * KVH;
            const u16* Vt = Kh + 64 * KS;
            f32x16 sc[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) sc[kt][e] = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const u32x4 kf = uo_frag(Kh + (kt * 32 + r) * KS + c * 16 + 8 * h);
                    sc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kf),
                                                                    __builtin_bit_cast(f16x8, qf[set][c]), sc[kt], 0, 0, 0);
                }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    sc[kt][e] *= scale;
                    mx = fmaxf(mx, sc[kt][e]);
                }
            mx = fmaxf(mx, xor32(mx));
            float ls = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    sc[kt][e] = __builtin_amdgcn_exp2f(sc[kt][e] - mx);
                    ls += sc[kt][e];
                }
            ls += xor32(ls);
            const float inv = 1.0f / ls;
            f32x16 O[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) O[dt][e] = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int sg = 0; sg < 2; ++sg) {
                    f16x8 pf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[e] = (_Float16)sc[kt][8 * sg + e];
                    const int c16 = 2 * kt + sg;   // 16-key chunk of the 64 inducers
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const u32x4 vf = uo_frag(Vt + (dt * 32 + r) * VS + c16 * 16 + 8 * h);
                        O[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, vf), pf, O[dt], 0, 0, 0);
                    }
                }
            // O^T (query on the lane, head-dim index in the registers) -> fp16(O / l) into the A buffer: row = query,
            // k = head * hd + d; registers 4 g4 .. + 3 are d = 32 dt + 8 g4 + 4 h .. + 3, half a 16-byte chunk
            const int row = wm * 32 + r;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int d = dt * 32 + 8 * g4 + 4 * h;
                    if (d < HD) {
                        f16x4 o4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = O[dt][4 * g4 + e] * inv;
                            asm volatile("" : "+v"(t));   // two roundings (fp32, then fp16) as in unpool_attn_x3_kernel:
                            o4[e] = (_Float16)t;          // no v_fma_mixlo_f16
                        }
                        const int k = (2 * hp + wn) * HD + d;
                        *reinterpret_cast<u32x2*>(abuf + row * RSY + (((k >> 3) ^ (row & 15)) << 4) + (k & 7) * 2) =
                            __builtin_bit_cast(u32x2, o4);
                    }
                }
        }
    }
    mf_lds_barrier();   // the A operand of out_proj is complete

    // ---- the step machine (inducer_chain_f16.hip): one weight block per step.  D = issued - s is NS at a primed
    // step: the wait for block s + 1 leaves the NS - 2 younger pieces in flight, the barrier frees block s's slot and
    // one block is issued into it; where the A operand changes the pipeline drains (D = NS - 1) and is primed again:
    // same wait, the barrier frees block s - 1's slot, one issue.  Tiles that reach the end of the stream (TAIL) stop
    // issuing and wait for everything.
    int s = 0, rslot = 0;
    const int row = wm * 32 + r;
    const char* yrow = abuf + row * RSY;
    const int ysw = row & 15;
    int boff[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int rb = (wn * 2 + j) * 32 + r;
#pragma unroll
        for (int c = 0; c < 2; ++c) boff[j][c] = rb * 16 + (((2 * h + c) ^ ((rb >> 2) & 3)) << 2);
    }
    f16x8 fa[2][2], fb[2][2][2];   // [set][chunk], [set][n-block][chunk]
    auto a_addr = [&](bool, int kt, int c) -> const char* { return yrow + (((kt * 4 + 2 * h + c) ^ ysw) << 4); };
    auto load_frags = [&](auto set_tag, bool gemm_b, int kt) {
        constexpr int set = decltype(set_tag)::value;
        const float* st = ring + rslot * MF_TILE;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            fa[set][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(a_addr(gemm_b, kt, c)));
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fb[set][j][c] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(st + boff[j][c]));
        }
        rslot = rslot + 1 == NS ? 0 : rslot + 1;
    };
    auto wait_block = [&](auto tail_tag) {
        if (decltype(tail_tag)::value && issued >= S_TOTAL) dma::wait_vm_lgkm0<0>();
        else dma::wait_vm_lgkm0<AHEAD>();
    };
    auto kstep = [&](auto cur_tag, auto tail_tag, bool has_next, bool gemm_b, int kt_next, f32x16& a0, f32x16& a1) {
        constexpr int cur = decltype(cur_tag)::value;
        if (has_next) {
            wait_block(tail_tag);
            // this step's fragments were read during the previous one and the wait above covered them: "redefine" them
            // so the compiler's wait-count pass does not put an lgkmcnt(0) in front of the first MFMA
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                asm volatile("" : "+v"(fa[cur][c]));
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(fb[cur][j][c]));
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!decltype(tail_tag)::value || issued < S_TOTAL) issue();
            load_frags(std::integral_constant<int, cur ^ 1>{}, gemm_b, kt_next);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][c], fb[cur][0][c], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[cur][c], fb[cur][1][c], a1, 0, 0, 0);
        }
    };
    // n (even) blocks of one 128-column tile; first: prime the pipeline; more: another tile over the same A follows
    auto tile_steps = [&](auto tail_tag, int n, bool gemm_b, bool first, bool more, f32x16& a0, f32x16& a1) {
        constexpr std::integral_constant<int, 0> set0{};
        constexpr std::integral_constant<int, 1> set1{};
        if (first) {
            wait_block(tail_tag);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (!decltype(tail_tag)::value || issued < S_TOTAL) issue();
            load_frags(set0, gemm_b, 0);
        }
#pragma unroll 1
        for (int kt = 0; kt < n - 2; kt += 2) {
            kstep(set0, tail_tag, true, gemm_b, kt + 1, a0, a1);
            kstep(set1, tail_tag, true, gemm_b, kt + 2, a0, a1);
        }
        kstep(set0, tail_tag, true, gemm_b, n - 1, a0, a1);
        kstep(set1, tail_tag, more, gemm_b, 0, a0, a1);
        s += n;
    };
    auto run_tile = [&](int n, bool gemm_b, bool first, bool more, f32x16& a0, f32x16& a1) {
        if (s + n + NS > S_TOTAL) tile_steps(std::true_type{}, n, gemm_b, first, more, a0, a1);
        else tile_steps(std::false_type{}, n, gemm_b, first, more, a0, a1);
    };
    auto zero = [](f32x16& a) {
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = 0.f;
    };


    f32x16 acc2[NT1][2];   // the 128 x C result: this wave's 32 rows x (64 columns of every output tile)
    float bias_r[NT1][2];
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
        zero(acc2[t][0]);
        zero(acc2[t][1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) bias_r[t][j] = g.bias ? g.bias[t * 128 + wn * 64 + j * 32 + r] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < NT1; ++t) run_tile(NK, false, t == 0, t + 1 < NT1, acc2[t][0], acc2[t][1]);

    // ---- epilogue (the store phase of dma::epilogue for a 4 x 2 wave grid): per output tile, the wave's 32 x 64
    // sub-tile goes through its private LDS tile and comes back as rows: lane (lr, c4) owns rows it * 4 + lr, columns
    // c4 * 4 .. + 3.  The residual rows of the next tile are fetched while this one is added and stored.
    {
        constexpr int TP = 64 + 4;   // transpose tile row stride (floats)
        // the whole LDS allocation is dead: 8 tiles of 8.5 KiB + the partial sums
        static_assert(uo_scratch_bytes(C) <= uo_lds_bytes(C, HD), "epilogue scratch: ring | A buffer | staging, all dead");
        float* Tt = smem + wave * 32 * TP;
        float* red2 = smem + 8 * 32 * TP;   // [4 row groups][2][C]
        const int lr = lane >> 4, c4 = lane & 15;
        float* xw = g.x + ((size_t)b * g.rows + m0 + wm * 32) * C + wn * 64 + c4 * 4;
        f32x4 rres[2][8];
        auto fetch = [&](int t, int set) {
#pragma unroll
            for (int it = 0; it < 8; ++it)
                rres[set][it] = *reinterpret_cast<const f32x4*>(xw + (size_t)(it * 4 + lr) * C + t * 128);
        };
        fetch(0, 0);
        mf_lds_barrier();   // every wave is done with the ring and the attention-output buffer
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            const int set = t & 1;
            if (t + 1 < NT1) fetch(t + 1, set ^ 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float bias = bias_r[t][j];
#pragma unroll
                for (int e = 0; e < 16; ++e) Tt[mfma_row(e, h) * TP + j * 32 + r] = acc2[t][j][e] + bias;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tile: the wave's own LDS operations are in order
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                f32x4 v4 = *reinterpret_cast<const f32x4*>(Tt + (it * 4 + lr) * TP + c4 * 4);
                v4 += rres[set][it];
                *reinterpret_cast<f32x4*>(xw + (size_t)(it * 4 + lr) * C + t * 128) = v4;   // default policy: the next kernel re-reads x (201 MB: Infinity Cache)
                s1 += v4;
                // explicit fma, as in dma::epilogue: the kernels sharing this store phase must agree to the bit on the statistics
#pragma unroll
                for (int q = 0; q < 4; ++q) s2[q] = __builtin_fmaf(v4[q], v4[q], s2[q]);
            }
            if (g.stats) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s1[q] += __shfl_xor(s1[q], 16, 64);
                    s1[q] += __shfl_xor(s1[q], 32, 64);
                    s2[q] += __shfl_xor(s2[q], 16, 64);
                    s2[q] += __shfl_xor(s2[q], 32, 64);
                }
                if (lane < 16) {
                    const int cl = t * 128 + wn * 64 + c4 * 4;
                    *reinterpret_cast<f32x4*>(red2 + (wm * 2 + 0) * C + cl) = s1;
                    *reinterpret_cast<f32x4*>(red2 + (wm * 2 + 1) * C + cl) = s2;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is read before the next one overwrites it
        }
        if (g.stats) {
            mf_lds_barrier();
            for (int i = tid; i < 2 * C; i += MF_NT) {
                const int which = i / C, c = i % C;
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) t += red2[(w * 2 + which) * C + c];
                g.stats[(((size_t)b * tilesM + rt) * 2 + which) * C + c] = t;
            }
        }
    }
}

template <int NT1, int HD>
int uo_launch_t(const UnpoolProjArgs& g, hipStream_t st) {
    constexpr int C = 128 * NT1;
    constexpr size_t lds = uo_lds_bytes(C, HD);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(unpool_outproj_f16_kernel<NT1, HD>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((unpool_outproj_f16_kernel<NT1, HD>), dim3(g.B * (g.rows / 128)), dim3(MF_NT), lds, st, g);
    return (int)hipGetLastError();
}

}  // namespace

// the shipped shapes: d = 128 / 384 with 8 heads, d = 256 with 8 heads
bool unpool_outproj_f16_supported(int C, int H, int rows) {
    if (rows < 128 || rows % 128 || H <= 0 || C % H) return false;
    const int hd = C / H;
    return (C == 128 && hd == 16) || (C == 256 && hd == 32) || (C == 384 && hd == 48);
}

int unpool_outproj_f16_launch(const UnpoolProjArgs& g0, int C, hipStream_t st) {
    if (!unpool_outproj_f16_supported(C, g0.H, g0.rows)) return -9;
    static int stagger = -1;   // GECCO_UNPOOL_STAGGER=<cycles> (0: off)
    if (stagger < 0) {
        const char* e = getenv("GECCO_UNPOOL_STAGGER");
        stagger = e ? atoi(e) : 4000;
    }
    UnpoolProjArgs g = g0;
    g.stagger = g.B * (g.rows / 128) >= 512 ? stagger : 0;
    switch (C) {
        case 128: return uo_launch_t<1, 16>(g, st);
        case 256: return uo_launch_t<2, 32>(g, st);
        default: return uo_launch_t<3, 48>(g, st);
    }
}
