// C ABI of libgecco_hip.so (declared in include/gecco_hip.h) and the host-side orchestration of a
// SetTransformer evaluation: which kernel runs when, which buffer feeds which.  Nothing here
// allocates, synchronises or reads device memory, so a caller may capture any entry point in a
// hipGraph.
#include "../../include/gecco_hip.h"
#include "kernels.h"

#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

namespace {

thread_local char g_err[512] = "";

int fail(int rc, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return rc;
}
int check(int rc, const char* what) {
    if (rc == 0) return 0;
    if (rc > 0) return fail(rc, "%s: HIP error %d (%s)", what, rc, hipGetErrorString((hipError_t)rc));
    return fail(rc, "%s: unsupported arguments (code %d)", what, rc);
}
#define TRY(expr, what)                  \
    do {                                 \
        int rc_ = check((expr), (what)); \
        if (rc_) return rc_;             \
    } while (0)

// Bump allocator over the caller's workspace (256-byte aligned carves).  With base == nullptr it
// only measures, so *_workspace_bytes() and the forward use the same code path.
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base(static_cast<char*>(b)) {}
    float* f32(size_t n) {
        off = (off + 255) & ~size_t(255);
        float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
        off += n * sizeof(float);
        return p;
    }
};

struct STWorkspace {
    float *big, *q, *attn;             // (B,N,2C), (B,N,C), (B,N,C)
    float *stats_x, *stats_s;          // (B,T,2,C) stream partials; (B,1,2,2C) inducer partials
    float *a1, *o1, *a2, *o2, *as, *os;  // AdaGN coefficients (B,C)
    float *part_o, *part_ml;           // pool partials
    float *merged, *h0, *u, *h2, *h, *kvh;  // inducer chain (B,I,*)
    float* wsplit;                     // tiled bf16 hi | lo image of the weight in use (unit calls, split-bf16 mode)
    float* wimg;                       // images of every layer's N-token weights, built once per forward
    size_t wimg_layer, o_q, o_out, o_w0, o_w2;   // floats per layer and the per-weight offsets inside (kv at 0)
    size_t o_pout, o_b0, o_b2, o_ukv;            // the 64-inducer chain: pool.out_proj, broadcast mlp, unpool k|v
    size_t o_mf;                                 // fused point MLP: W0 tile j | W2 K-slice j (two halves), j = 0 .. width/128
    size_t bytes;
};

int max_i(int a, int b) { return a > b ? a : b; }
int row_tiles_gemm(int rows) { const int bm = gemm_row_tile(rows); return (rows + bm - 1) / bm; }
int row_tiles_stats(int rows) { const int bm = stats_row_tile(rows); return (rows + bm - 1) / bm; }

STWorkspace carve_st(const GeccoSetTransformer* st, int B, int N, void* base) {
    Carver c(base);
    STWorkspace w;
    const size_t C = st->C, I = st->I, W = st->width;
    const size_t big = (size_t)B * N * (2 * C > W ? 2 * C : W);
    w.big = c.f32(big);
    w.q = c.f32((size_t)B * N * C);
    w.attn = c.f32((size_t)B * N * C);
    const int T = max_i(row_tiles_gemm(N), row_tiles_stats(N));
    w.stats_x = c.f32((size_t)B * T * 2 * C);
    w.stats_s = c.f32((size_t)B * 2 * 2 * (2 * C > W ? 2 * C : W));
    w.a1 = c.f32(B * C); w.o1 = c.f32(B * C);
    w.a2 = c.f32(B * C); w.o2 = c.f32(B * C);
    w.as = c.f32(B * C); w.os = c.f32(B * C);
    const int ns = pool_attn_nsplit(B, N, st->H);
    const size_t HD = C / st->H;
    w.part_o = c.f32((size_t)B * st->H * ns * 64 * HD);
    w.part_ml = c.f32((size_t)B * st->H * ns * 64 * 2);
    w.merged = c.f32(B * I * C);
    w.h0 = c.f32(B * I * C);
    w.u = c.f32(B * I * W);
    w.h2 = c.f32(B * I * C);
    w.h = c.f32(B * I * C);
    w.kvh = c.f32(B * I * 2 * C);
    {   // tiled bf16 hi | lo image of the weight in use: output rows padded to the 128-column GEMM tile
        const size_t wmax = 3 * C + 128 > W ? 3 * C + 128 : W;   // kv_proj | q_proj share one image
        w.wsplit = c.f32(((wmax + 127) / 128 * 128) * (size_t)(W > C ? W : C));
    }
    {   // per layer: kv_proj | q_proj (contiguous: the fused pair streams them as one image), out_proj, mlp.0, mlp.2
        // (floats: 4 bytes per weight element in split-bf16 mode, 2 in fp16 mode)
        const int prec = st->precision == 4 ? 3 : st->precision;   // 4 ("w2") = the mixed mode with the one-launch point MLP
        const size_t half = prec == 2 ? 2 : 1;
        auto pad = [half](size_t n) { return (n + 127) / 128 * 128 / half; };
        w.o_q = pad(2 * C) * C;
        w.o_out = w.o_q + pad(C) * C;
        w.o_w0 = w.o_out + pad(C) * C;
        w.o_w2 = w.o_w0 + pad(W) * C;
        w.o_pout = w.o_w2 + pad(C) * W;
        w.o_b0 = w.o_pout + pad(C) * C;
        w.o_b2 = w.o_b0 + pad(W) * C;
        w.o_ukv = w.o_b2 + pad(C) * W;
        w.o_mf = w.o_ukv + pad(2 * C) * C;
        w.wimg_layer = w.o_mf + (prec == 2 ? pad(W) * C + pad(C) * W : 0);
        // "w2" mode: the weight stream of the one-launch point MLP (mlp_fused_w.hip) at o_mf
        if (st->precision == 4 && mlp_fused_w_supported((int)C, (int)W, 128)) w.wimg_layer += mlp_fused_w_image_bytes((int)C, (int)W) / sizeof(float);
        w.wimg = prec >= 1 ? c.f32(w.wimg_layer * st->n_layers) : nullptr;
    }
    w.bytes = (c.off + 255) & ~size_t(255);
    return w;
}

// precision 1 (split-bf16) applies to the N-token GEMMs the LDS-DMA kernel takes; `wsplit` receives the tiled bf16
// hi | lo image of W first (a ~3 us pass over <= 1.2 MB: weights may change between calls, nothing is cached).
int linear(const float* A, const float* W, const float* bias, const float* pa, const float* po, const float* alpha,
           const float* res, float* C, float* stats, int B, int rows, int K, int Nout, int act, hipStream_t s,
           int precision = 0, float* wsplit = nullptr, const float* img_ready = nullptr, int a_f16 = 0, int c_f16 = 0,
           int a_img = 0, int c_img = 0, const float* mul_u = nullptr, int mul_kind = 0, float* agrad = nullptr,
           float* pre_out = nullptr, const float* dot_x = nullptr) {
    GemmArgs g{};
    g.pre_out = pre_out;
    g.dot_x = dot_x;
    g.a_img = a_img; g.c_img = c_img;   // activation handed over as a tiled split image (kernels.h); callers check act_image_ok
    g.mul_u = mul_u; g.mul_kind = mul_kind; g.agrad = agrad;   // activation backward as the epilogue (LDS-DMA kernels only)
    g.A = A; g.W = W; g.bias = bias; g.pro_a = pa; g.pro_o = po; g.alpha = alpha; g.residual = res; g.C = C;
    g.stats = stats; g.B = B; g.rows = rows; g.K = K; g.Nout = Nout;
    g.lda = K; g.ldw = K; g.ldc = Nout; g.ldr = Nout; g.act = act;
    g.precision = 0; g.w_img = nullptr;
    if (act < 0 || act > 4) return -6;
    if ((act == 1 || act == 2) && !alpha) return -6;
    g.a_f16 = a_f16; g.c_f16 = c_f16;   // fp16 tensors exist only between the fp16 kernels (st_forward checks support)
    const bool fast = precision == 1 ? gemm_f32_dma_supported(g, 1) : precision == 2 ? gemm_f16_dma_supported(g) : false;
    if ((a_f16 || c_f16) && !(fast && precision == 2 && (wsplit || img_ready))) return -9;
    if ((a_img || c_img) && !(fast && precision == 1 && (wsplit || img_ready))) return -9;
    if ((mul_u || pre_out || dot_x) && !(precision == 0 ? gemm_f32_dma_supported(g, 0) : (fast && (wsplit || img_ready)))) return -9;
    if (!W && !(fast && img_ready)) return -9;
    if (fast && (wsplit || img_ready)) {
        if (img_ready) {
            g.w_img = img_ready;   // already converted this forward
        } else {
            int rc = precision == 1 ? split_bf16_tiled_launch(W, wsplit, Nout, K, g.ldw, s)
                                    : split_f16_tiled_launch(W, wsplit, Nout, K, g.ldw, s);
            if (rc) return rc;
            g.w_img = wsplit;
        }
        g.precision = precision;
    }
    return gemm_f32_launch(g, s);
}

// Two linears over the same (AdaGN-modulated) A in one launch: C1 = A' W1^T + b1 (Nout1 columns), C2 = A' W2^T + b2.
// Returns 1 when the fused form does not apply (caller issues the two linears), 0 on success, <0 on error.
int linear_pair(const float* A, const float* W1, const float* b1, int Nout1, float* C1, const float* W2,
                const float* b2, int Nout2, float* C2, const float* pa, const float* po, int B, int rows, int K,
                hipStream_t s, int precision, float* wsplit, const float* img_ready = nullptr, int c_f16 = 0, int a_f16 = 0) {
    GemmArgs g{};
    g.c_f16 = c_f16;
    g.a_f16 = a_f16;
    if ((c_f16 || a_f16) && precision != 2) return -9;
    g.A = A; g.W = W1; g.bias = b1; g.pro_a = pa; g.pro_o = po; g.C = C1;
    g.B = B; g.rows = rows; g.K = K; g.Nout = Nout1 + Nout2;
    g.lda = K; g.ldw = K; g.ldc = Nout1; g.ldr = Nout1;
    g.C2 = C2; g.W2 = W2; g.bias2 = b2; g.n_split = Nout1; g.ldc2 = Nout2;
    if (precision == 2 ? !gemm_f16_dma_supported(g) : !gemm_f32_dma_supported(g)) return 1;
    if (precision == 1 || precision == 2) {
        if (img_ready) {
            g.w_img = img_ready;
        } else {
            if (!wsplit) return 1;
            int rc = precision == 1 ? split_bf16_tiled_launch(W1, wsplit, Nout1, K, K, s)
                                    : split_f16_tiled_launch(W1, wsplit, Nout1, K, K, s);
            if (rc) return rc;
            float* img2 = wsplit + (precision == 1 ? split_bf16_image_bytes(Nout1, K) : split_f16_image_bytes(Nout1, K)) / sizeof(float);
            rc = precision == 1 ? split_bf16_tiled_launch(W2, img2, Nout2, K, K, s)
                                : split_f16_tiled_launch(W2, img2, Nout2, K, K, s);
            if (rc) return rc;
            g.w_img = wsplit;
        }
        g.precision = precision;
    }
    if (precision == 2) return gemm_f16_dma_launch(g, s);
    return gemm_f32_dma_launch(g, s);
}

// Path switches for A/B runs and tests: gecco_set_option, or the environment (GECCO_ASTAT, GECCO_CHAIN) on first use.
enum { OPT_ASTAT = 0, OPT_CHAIN = 1, OPT_HEADMAJOR = 2, OPT_MLPFUSED = 3, OPT_UNPOOLFUSED = 4, OPT_LO8 = 5, OPT_ACTIMG = 6, OPT_H8 = 7, OPT_KVQ64 = 8, OPT_H8AREG = 9, OPT_CHAIN2 = 10, OPT_UNPOOLH8 = 11, OPT_MLPW = 12, OPT_CHAINCL = 13, OPT_H6 = 14, OPT_KVFOLD = 15, OPT_MLPWSHARE = 16, OPT_KVQPERM = 17, OPT_IMGPROJ16 = 18, OPT_COUNT = 19 };
int g_options[OPT_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
const char* const g_option_names[OPT_COUNT] = {"astat", "chain", "headmajor", "mlpfused", "unpoolfused", "lo8", "actimg", "h8", "kvq64", "h8areg", "chain2", "unpoolh8", "mlpw", "chaincl", "h6", "kvfold", "mlpwshare", "kvqperm", "imgproj16"};
const char* const g_option_env[OPT_COUNT] = {"GECCO_ASTAT", "GECCO_CHAIN", "GECCO_HEADMAJOR", "GECCO_MLPFUSED", "GECCO_UNPOOLFUSED", "GECCO_LO8",
                                             "GECCO_ACTIMG", "GECCO_H8", "GECCO_KVQ64", "GECCO_H8AREG", "GECCO_CHAIN2", "GECCO_UNPOOLH8", "GECCO_MLPW", "GECCO_CHAINCL", "GECCO_H6", "GECCO_KVFOLD", "GECCO_MLPWSHARE",
                                             "GECCO_KVQPERM", "GECCO_IMGPROJ16"};
// A plan's own switches (GeccoSetTransformer.opt_mask / opt_vals: gecco_option_index(name) is the bit) win over the process-wide ones
// while that plan's forward runs on this thread: two plans, or two host threads, never see each other's settings.
thread_local const GeccoSetTransformer* t_plan = nullptr;
struct PlanScope {
    const GeccoSetTransformer* prev;
    explicit PlanScope(const GeccoSetTransformer* p) : prev(t_plan) { t_plan = p; }
    ~PlanScope() { t_plan = prev; }
};
int option(int which) {
    if (t_plan && ((t_plan->opt_mask >> which) & 1u)) return (int)((t_plan->opt_vals >> which) & 1u);
    if (g_options[which] < 0) {
        const char* e = getenv(g_option_env[which]);
        // "mlpwshare" (the one-launch MLP leaves CUs to a second stream's kernels) is off unless the caller runs two streams: hip_ops.py
        // sets it around a two-stream evaluation
        // "imgproj16" (img_feature_proj on one-term fp16 operands) is opt-in: 2 % of a C3 evaluation for a fifth of the w2 mode's error budget
        g_options[which] = e ? (atoi(e) != 0) : (which != OPT_MLPWSHARE && which != OPT_IMGPROJ16);
    }
    return g_options[which];
}

// fp16 mode: C16 (| C2_16) = fp16(act(fp16(x * pa + po) W^T + bias)) in one pass over x (gemm_f16_astat.hip).
// Returns 1 when the shape is outside that kernel's reach (caller: cast pass + streaming GEMM), 0 on success.
int astat_linear(const float* x, const float* pa, const float* po, const float* img, const float* bias1, int Nout1,
                 float* C1, const float* bias2, int Nout2, float* C2, const float* alpha, int act, int B, int rows,
                 int K, hipStream_t s, int hm_hd = 0, const float* img_lo = nullptr, int use64 = 0) {
    if (use64) {
        // mixed mode on the 64-column-tile kernel (gemm_h8_astat.hip: gemm_kvq_astat_kernel): `img` is the kvq stream — Nout1's
        // tiles, the V half of a K | V pair with its fp8 second weight term (use64 == 2), then Nout2's
        GemmArgs g{};
        g.A = x; g.pro_a = pa; g.pro_o = po; g.bias = bias1; g.C = C1;
        g.B = B; g.rows = rows; g.K = K; g.Nout = Nout1 + Nout2; g.lda = K; g.ldw = K; g.ldc = Nout1; g.ldr = Nout1;
        g.precision = 2; g.w_img = img; g.c_f16 = 1; g.hm_hd = hm_hd;
        if (use64 == 2) { g.lo_begin = Nout1 / 128; g.lo_tiles = Nout1 / 64; }
        if (C2) { g.C2 = C2; g.bias2 = bias2; g.n_split = Nout1; g.ldc2 = Nout2; }
        g.kvq_perm = option(OPT_KVQPERM) && kvq_perm48_ok(hm_hd, K, Nout1, Nout2);   // (the stream's builder decides by the same rule)
        if (!img || act || !gemm_kvq_astat_supported(g)) return 1;
        return gemm_kvq_astat_launch(g, s);
    }
    GemmArgs g{};
    g.A = x; g.pro_a = pa; g.pro_o = po; g.bias = bias1; g.alpha = alpha; g.act = act; g.C = C1;
    g.B = B; g.rows = rows; g.K = K; g.Nout = Nout1 + Nout2; g.lda = K; g.ldw = K; g.ldc = Nout1; g.ldr = Nout1;
    g.precision = 2; g.w_img = img; g.c_f16 = 1; g.hm_hd = hm_hd; g.w_img2 = img_lo;   // img_lo: two-term weights (mixed mode)
    if (img_lo && C2) {   // kv_proj | q_proj: two-term weights for the V half only; the K half and the q segment stay one-term
        g.lo_begin = Nout1 / 256;
        g.lo_tiles = Nout1 / 128;
    }
    g.lo_fp8 = img_lo && option(OPT_LO8) && gemm_f16_astat_lo8_supported(K);   // the caller built the lo image by the same rule
    if (C2) { g.C2 = C2; g.bias2 = bias2; g.n_split = Nout1; g.ldc2 = Nout2; }
    // option "astat" = 0 falls back to the cast pass + streaming GEMM (A/B runs; same bits)
    if (!option(OPT_ASTAT) || !img || !gemm_f16_astat_supported(g)) return 1;
    return gemm_f16_astat_launch(g, s);
}

// mixed mode: C = residual + A W^T + bias (+ statistics) with A an h8 activation image and W the 128-column-tile h8 stream
// (gemm_h8_areg.hip): mlp.2 and out_proj
int h8_linear(const float* a_img, const float* w_img, const float* bias, const float* res, float* C, float* stats, int B, int rows,
              int K, int Nout, hipStream_t s) {
    GemmArgs g{};
    g.A = a_img; g.bias = bias; g.residual = res; g.C = C; g.stats = stats; g.B = B; g.rows = rows; g.K = K; g.Nout = Nout;
    g.lda = K; g.ldw = K; g.ldc = Nout; g.ldr = Nout; g.a_img = 2; g.w_img = w_img; g.precision = 1;
    return gemm_h8_areg_launch(g, s);
}

int coeffs(const float* stats, int T, int rows, const float* t, int ctx, const GeccoAdaGN* p, float* a, float* o,
           int B, int C, int G, hipStream_t s) {
    return adagn_coeffs_launch(stats, T, rows, t, ctx, p ? p->scale_w : nullptr, p ? p->scale_b : nullptr,
                               p ? p->bias_w : nullptr, p ? p->bias_b : nullptr, a, o, B, C, G, 1e-5f, s);
}

int st_forward(const GeccoSetTransformer* st, float* x, const float* t, const float* stats_x, int stats_T,
               const float* const* h_in, float* const* h_out, float* stats_out, int B, int N, void* ws,
               size_t ws_bytes, hipStream_t s) {
    if (!st || !x || !t) return fail(-1, "set_transformer: null argument");
    if (st->I != 64) return fail(-3, "set_transformer: num_inducers must be 64 (got %d)", st->I);
    if (st->C % st->H || st->C % st->G || st->C % 4) return fail(-3, "set_transformer: bad feature_dim %d", st->C);
    if (st->act < 0 || st->act > 3) return fail(-3, "set_transformer: act must be 0 (identity), 1 / 2 (GaussianActivation normalized / raw) or 3 (ReLU)");
    PlanScope plan_scope(st);
    STWorkspace w = carve_st(st, B, N, ws);
    if (ws_bytes < w.bytes) return fail(-7, "set_transformer: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
    const int C = st->C, I = st->I, H = st->H, G = st->G, Wd = st->width, ctx = st->ctx_dim, act = st->act;
    // precision 3 ("mixed"): kv_proj | q_proj with fp16 activations and TWO-TERM fp16 weights (A-stationary kernel), fp16
    // K | V / q and fp16 attention products; everything that feeds the residual stream or the shared inducer states
    // (pool.out_proj .. unpool k|v on the 64 inducers, unpool.out_proj, the point MLP) in split-bf16 arithmetic.
    // tools/experiments/fp16_site_sensitivity.py: those are the products whose operand rounding reaches the output.
    // Shapes the A-stationary kv_proj | q_proj kernel does not take (rows not a multiple of 128, C outside 128 .. 512 in steps
    // of 128, head dims the fp16 attention kernels do not have) run the whole evaluation in split-bf16 — at least as
    // accurate, slower — instead of failing: a drop-in caller's N need not be a multiple of 128.
    const int prec = st->precision == 4 ? 3 : st->precision;   // 4 ("w2"): the mixed mode with the point MLP as one launch (below)
    const bool mixed = prec == 3 && N >= 128 && N % 128 == 0 && C % 128 == 0 && C <= 512 && !(C % H) &&
                       attn_x3_supported(C / H) && st->I == 64 && option(OPT_ASTAT);
    const int pr = mixed ? 1 : (prec == 3 ? 1 : prec);       // arithmetic of the generic linears
    const int Tn = row_tiles_gemm(N), Ti = row_tiles_gemm(I);
    const int ns = pool_attn_nsplit(B, N, H);

    const float* sx = stats_x;
    int sT = stats_T;
    if (!sx) {
        TRY(col_stats_launch(x, w.stats_x, B, N, C, s), "col_stats");
        sx = w.stats_x;
        sT = row_tiles_stats(N);
    }
    const int kmod = (pr == 2 || mixed) ? 32 : 16;   // K granularity of the fast kernels
    // fp16 mode: everything on the 64 inducers between the two attentions is one launch (inducer_chain_f16.hip)
    // fp16 mode: the point-stream MLP of a layer (AdaGN, mlp.0, activation, mlp.2, residual, statistics) is one launch
    const bool mlpf_on = pr == 2 && w.wimg && option(OPT_MLPFUSED) && mlp_fused_f16_supported(C, Wd, N);
    // mixed mode: mlp.0 as fp16 main product + two fp8 cross terms, A-stationary over 256-row blocks (gemm_h8_astat.hip); its
    // output is the tiled split image mlp.2 loads into registers (option "actimg")
    bool h8_on = false;
    if (mixed && w.wimg && option(OPT_H8) && option(OPT_ACTIMG) && (act == 0 || (act >= 1 && act <= 3))) {
        GemmArgs hg{};
        hg.c_img = 1; hg.w_img = w.wimg; hg.rows = N; hg.Nout = Wd; hg.K = C; hg.lda = C; hg.act = act;
        h8_on = gemm_h8_astat_supported(hg) && Wd % 16 == 0 && C % 16 == 0;
    }
    // mixed mode: kv_proj | q_proj on the 64-column-tile A-stationary kernel (option "kvq64"; two 128-row blocks per CU, W bytes
    // shared by 128 rows, 16-byte head-major stores).  The weight images are then built in the kvq format: a shape the kernel rejects
    // is an error (-3) of the mixed mode, not a fallback (option "kvq64" = 0 selects the 128-column-tile kernel and its images)
    bool kvq_on = false;
    if (mixed && w.wimg && option(OPT_KVQ64) && option(OPT_HEADMAJOR) && (C == 128 || C == 256 || C == 384 || C == 512) && !((C / H) & 7)) kvq_on = true;
    // fp16 mode at feature_dim 512: the 128-column-tile A-stationary kernel needs 128 fragment registers + 128 accumulator
    // registers there and spills (gemm_f16_astat_kernel<16, 4, *>: 24 - 54 VGPRs to scratch); the 64-column-tile kernel takes the
    // same one-term product without scratch (gemm_kvq_astat_kernel<8, 6>, no L stages)
    const bool kvq16_on = !mixed && pr == 2 && w.wimg && option(OPT_KVQ64) && option(OPT_HEADMAJOR) && option(OPT_ASTAT) && C == 512 &&
                          !((C / H) & 7) && N >= 128 && N % 128 == 0;
    // ... and mlp.2 / out_proj as h8 products fed from h8 activation images (gemm_h8_areg.hip; option "h8areg")
    bool h8x = false, h8o = false;
    if (mixed && w.wimg && option(OPT_H8AREG) && option(OPT_ACTIMG) && Wd % 64 == 0 && N % 128 == 0) {
        GemmArgs hg{};
        hg.a_img = 2; hg.w_img = w.wimg; hg.rows = N; hg.Nout = C; hg.K = Wd; hg.lda = Wd; hg.ldc = C; hg.ldr = C;
        h8x = gemm_h8_areg_supported(hg);
    }
    if (mixed && w.wimg && option(OPT_H8AREG) && option(OPT_ACTIMG) && I == 64 && C % 64 == 0 && attn_x3_supported(C / H)) {
        GemmArgs hg{};
        hg.a_img = 2; hg.w_img = w.wimg; hg.rows = N; hg.Nout = C; hg.K = C; hg.lda = C; hg.ldc = C; hg.ldr = C;
        h8o = gemm_h8_areg_supported(hg);
    }
    // mixed mode, option "h6": mlp.0's two cross terms as fp6 x fp6 with per-block scales (half the matrix cycles of the fp8 form; its
    // weight stream is built in the h6 form) — where mlp.0 writes the h8 activation image (the kernel's only F6 instantiations)
    const bool h6_on = h8_on && h8x && option(OPT_H6);
    // mixed mode: unpool attention + out_proj (h8) + residual + statistics in ONE launch (unpool_outproj_h8.hip; option "unpoolh8"):
    // the attention output of a row block is the stationary operand of out_proj and never leaves the CU.  Needs the head-major
    // fp16 q of the kvq kernel; the k | v image of the inducers lives in the (then idle) attention-output buffer
    const bool uo8_on = h8o && kvq_on && option(OPT_UNPOOLH8) && I == 64 && unpool_outproj_h8_supported(C, H, N) &&
                        unpool_outproj_h8_kv_bytes(B, C, H) <= (size_t)B * N * C * sizeof(float);
    // "w2" mode (option "mlpw" = 0 runs it as the mixed mode): the point MLP as ONE launch, the hidden layer kept as register fragments,
    // its second term dropped (mlp_fused_w.hip); the weight stream (1.9 MB at d = 384) has its own workspace slot (o_mf).  Shapes the kernel
    // does not take (feature_dim off 128 .. 512 in steps of 128, point counts off 128) run the mixed mode's two launches — at least as accurate
    const bool mfw_on = mixed && st->precision == 4 && option(OPT_MLPW) && w.wimg && mlp_fused_w_supported(C, Wd, N) &&
                        (size_t)B * N * C * sizeof(float) < ((size_t)1 << 31);
    const bool chain_on = pr == 2 && w.wimg && option(OPT_CHAIN) && inducer_chain_f16_supported(C, Wd, H, G, I) &&
                          (ns == 1 || ns == 2 || ns == 4 || ns == 8);
    // mixed mode: the same one-launch chain with TWO-TERM fp16 weights (option "chain2") instead of five 64-row split-bf16 GEMMs
    // and their coefficient launches: the chain's activation rounding does not reach the output, its weight rounding does
    // (tools/experiments/precision_search.py: chain = x2a keeps F_x at 1.0e-4 .. 1.3e-4)
    // feature_dim <= 384: at 512 one block per sample streams 7 MB of weights through one CU and loses to the five launches
    // (C4, B = 32: 10.25 vs 10.03 ms per evaluation); the cluster form (option "chaincl": 4 blocks per sample) wins there too (9.94 vs 10.12)
    const bool cl_ok = option(OPT_CHAINCL) && C >= 256 && (size_t)st->n_layers * 8 <= (size_t)I * C;
    const bool chain2_on = mixed && w.wimg && option(OPT_CHAIN2) && (C <= 384 || cl_ok) && inducer_chain_f16_supported(C, Wd, H, G, I) &&
                           (ns == 1 || ns == 2 || ns == 4 || ns == 8) && (act >= 0 && act <= 3);
    if (pr >= 1 && w.wimg && !(C % kmod) && !(Wd % kmod) && !st->images_ready) {
        // split-bf16 mode: every N-token weight of every layer becomes its tiled hi | lo image in ONE launch per
        // 6 layers (weights may change between calls: nothing is cached across forwards unless the caller vouches for the
        // workspace's images — GeccoSetTransformer.images_ready)
        SplitJobs jobs, jobs16, jobs8;   // jobs16: the fp16 images of the mixed mode (kv_proj | q_proj, hi and lo); jobs8: mlp.0's h8 image
        jobs.n = 0;
        jobs16.n = 0;
        jobs8.n = 0;
        constexpr int kMjCap = 16;
        MlpWImageJob mjobs[kMjCap];
        int nmj = 0;
        // every insertion goes through here: the table is flushed BEFORE a write that would not fit
        constexpr int kJobCap = (int)(sizeof(jobs.job) / sizeof(jobs.job[0]));
        auto push_ld = [&](const float* Wp, float* img, int Nout, int K, int ldw) -> int {
            if (jobs.n >= kJobCap) {
                int rc = pr == 2 ? split_f16_tiled_multi_launch(jobs, s) : split_bf16_tiled_multi_launch(jobs, s);
                jobs.n = 0;
                if (rc) return rc;
            }
            jobs.job[jobs.n++] = SplitJob{Wp, img, Nout, K, ldw, 0};
            return 0;
        };
        auto push = [&](const float* Wp, float* img, int Nout, int K) -> int { return push_ld(Wp, img, Nout, K, K); };
        // head dim 48: the kvq stream in the head-aligned column order (option "kvqperm"; astat_linear decides by the same rule)
        const int kvq_p48 = (option(OPT_KVQPERM) && option(OPT_HEADMAJOR) && kvq_perm48_ok(C / H, C, 2 * C, C)) ? 64 : 0;
        for (int li = 0; li < st->n_layers; ++li) {
            const GeccoLayer& L = st->layers[li];
            float* base = w.wimg + (size_t)li * w.wimg_layer;
            if (mixed && kvq_on) {
                // the kvq stream: K | V tiles (the V half with L stages), then the q tiles
                if (!(h_in && h_in[li])) {
                    if (jobs8.n >= kJobCap) { TRY(h8_image_multi_launch(jobs8, s), "split(kv_proj, kvq)"); jobs8.n = 0; }
                    jobs8.job[jobs8.n++] = SplitJob{L.kv_proj_w, base, 2 * C, C, C, 1 | kvq_p48 | ((C / 64) << 8) | ((2 * C / 64) << 20)};
                }
                if (jobs8.n >= kJobCap) { TRY(h8_image_multi_launch(jobs8, s), "split(q_proj, kvq)"); jobs8.n = 0; }
                jobs8.job[jobs8.n++] = SplitJob{L.in_proj_w, base + kvq_image_bytes(2 * C, C, C) / sizeof(float), C, C, C, 1 | kvq_p48};
            } else if (mixed) {
                // fp16 hi images of kv_proj | q_proj back to back (one stream for the A-stationary kernel), then their lo images
                const size_t hkv = (size_t)(2 * C + 127) / 128 * 128 * C / 2, hq = (size_t)(C + 127) / 128 * 128 * C / 2;
                for (int lo = 0; lo < 2; ++lo) {
                    float* dst = base + lo * (hkv + hq);
                    if (!(h_in && h_in[li])) {
                        if (jobs16.n >= kJobCap) { TRY(split_f16_tiled_multi_launch(jobs16, s), "split(weights)"); jobs16.n = 0; }
                        // lo image: fp8 x 2^19 in 64-k blocks where the A-stationary kernel has that form (astat_linear decides alike)
                        const int lo_kind = lo && option(OPT_LO8) && gemm_f16_astat_lo8_supported(C) ? 2 : lo;
                        jobs16.job[jobs16.n++] = SplitJob{L.kv_proj_w, dst, 2 * C, C, C, lo_kind};
                    }
                    if (lo) continue;   // q_proj is one-term: no lo image
                    if (jobs16.n >= kJobCap) { TRY(split_f16_tiled_multi_launch(jobs16, s), "split(weights)"); jobs16.n = 0; }
                    jobs16.job[jobs16.n++] = SplitJob{L.in_proj_w, dst + hkv, C, C, C, 0};
                }
            }
            const bool chain2_here = chain2_on && !(h_in && h_in[li]);
            if (chain2_here) {
                // one stream of fp16 blocks, hi | lo per column tile, in the order the chain consumes them: pool.out_proj, broadcast.mlp.0,
                // broadcast.mlp.2 K-half by K-half, unpool k|v (o_pout, o_b0, o_b2, o_ukv are consecutive and, at 4 bytes per weight
                // element, exactly as large as the two-term images)
                auto push2 = [&](const float* Wp, float* img, int Nout, int K, int ldw) -> int {
                    if (jobs16.n >= kJobCap) { int rc = split_f16_tiled_multi_launch(jobs16, s); jobs16.n = 0; if (rc) return rc; }
                    jobs16.job[jobs16.n++] = SplitJob{Wp, img, Nout, K, ldw, 8};
                    return 0;
                };
                TRY(push2(L.pool_out_w, base + w.o_pout, C, C, C), "split(pool.out_proj, two-term)");
                TRY(push2(L.bmlp.w0, base + w.o_b0, Wd, C, C), "split(broadcast.mlp.0, two-term)");
                for (int hf = 0; hf < Wd / C; ++hf)
                    TRY(push2(L.bmlp.w2 + (size_t)hf * C, base + w.o_b2 + (size_t)hf * C * C, C, C, Wd), "split(broadcast.mlp.2 K-half, two-term)");
                TRY(push2(L.in_proj_w + (size_t)C * C, base + w.o_ukv, 2 * C, C, C), "split(unpool.in_proj kv, two-term)");
            }
            if (kvq16_on) {   // one-term kvq stream: K | V tiles, then the q tiles (o_q = the end of the kv image: 2 bytes per weight)
                if (!(h_in && h_in[li])) {
                    if (jobs8.n >= kJobCap) { TRY(h8_image_multi_launch(jobs8, s), "split(kv_proj, kvq)"); jobs8.n = 0; }
                    jobs8.job[jobs8.n++] = SplitJob{L.kv_proj_w, base, 2 * C, C, C, 1};
                }
                if (jobs8.n >= kJobCap) { TRY(h8_image_multi_launch(jobs8, s), "split(q_proj, kvq)"); jobs8.n = 0; }
                jobs8.job[jobs8.n++] = SplitJob{L.in_proj_w, base + w.o_q, C, C, C, 1};
            }
            if (!(h_in && h_in[li]) && !chain2_here) {
                if (!mixed && !kvq16_on) TRY(push(L.kv_proj_w, base, 2 * C, C), "split(kv_proj)");
                TRY(push(L.pool_out_w, base + w.o_pout, C, C), "split(pool.out_proj)");
                TRY(push(L.bmlp.w0, base + w.o_b0, Wd, C), "split(broadcast.mlp.0)");
                if (chain_on) {   // the one-launch chain walks mlp.2 K-half by K-half: one (C x C) image per half
                    for (int hf = 0; hf < Wd / C; ++hf)
                        TRY(push_ld(L.bmlp.w2 + (size_t)hf * C, base + w.o_b2 + (size_t)hf * C * C / 2, C, C, Wd),
                            "split(broadcast.mlp.2 K-half)");
                } else {
                    TRY(push(L.bmlp.w2, base + w.o_b2, C, Wd), "split(broadcast.mlp.2)");
                }
            }
            if (!chain2_here) TRY(push(L.in_proj_w + (size_t)C * C, base + w.o_ukv, 2 * C, C), "split(unpool.in_proj kv)");
            if (!mixed && !kvq16_on) TRY(push(L.in_proj_w, base + w.o_q, C, C), "split(q_proj)");
            if (h8o) {
                if (jobs8.n >= kJobCap) { TRY(h8_image_multi_launch(jobs8, s), "split(out_proj, h8)"); jobs8.n = 0; }
                jobs8.job[jobs8.n++] = SplitJob{L.unpool_out_w, base + w.o_out, C, C, C, uo8_on ? 16 : 2};   // 16: 64-column tiles, attention k order
            } else {
                TRY(push(L.unpool_out_w, base + w.o_out, C, C), "split(out_proj)");
            }
            if (mlpf_on) {   // one stream in consumption order: per hidden chunk j, W0 tile j, then W2[:, chunk j] in two K-halves
                const int nkb = C / 32;
                for (int jc = 0; jc < Wd / 128; ++jc) {
                    float* cb = base + w.o_mf + (size_t)jc * 2 * nkb * 2048;
                    TRY(push_ld(L.mlp.w0 + (size_t)jc * 128 * C, cb, 128, C, C), "split(mlp.0 tile)");
                    for (int hf = 0; hf < 2; ++hf)
                        TRY(push_ld(L.mlp.w2 + (size_t)jc * 128 + hf * 64, cb + (size_t)(nkb + hf * (nkb / 2)) * 2048, C, 64, Wd),
                            "split(mlp.2 K-slice)");
                }
            } else if (mfw_on) {   // the one-launch point MLP's streams: all layers in one launch, behind the loop
                if (nmj == kMjCap) { TRY(mlp_fused_w_images_launch(mjobs, nmj, C, Wd, act, s), "split(mlp, w2 streams)"); nmj = 0; }
                mjobs[nmj++] = MlpWImageJob{L.mlp.w0, L.mlp.b0, L.mlp.w2, L.mlp.b2, base + w.o_mf, L.mlp.alpha};
            } else {
                if (h8_on) {   // same bytes as the split-bf16 image it replaces: fp16 hi + fp8 lo + fp8 W per element
                    if (jobs8.n >= kJobCap) { TRY(h8_image_multi_launch(jobs8, s), "split(mlp.0, h8)"); jobs8.n = 0; }
                    jobs8.job[jobs8.n++] = SplitJob{L.mlp.w0, base + w.o_w0, Wd, C, C, h6_on ? 32 : 0};
                } else {
                    TRY(push(L.mlp.w0, base + w.o_w0, Wd, C), "split(mlp.0)");
                }
                if (h8x) {
                    if (jobs8.n >= kJobCap) { TRY(h8_image_multi_launch(jobs8, s), "split(mlp.2, h8)"); jobs8.n = 0; }
                    jobs8.job[jobs8.n++] = SplitJob{L.mlp.w2, base + w.o_w2, C, Wd, Wd, 2};
                } else {
                    TRY(push(L.mlp.w2, base + w.o_w2, C, Wd), "split(mlp.2)");
                }
            }
        }
        TRY(pr == 2 ? split_f16_tiled_multi_launch(jobs, s) : split_bf16_tiled_multi_launch(jobs, s), "split(weights)");
        if (mixed) TRY(split_f16_tiled_multi_launch(jobs16, s), "split(weights, fp16)");
        TRY(h8_image_multi_launch(jobs8, s), "split(mlp.0, h8)");
        if (nmj) TRY(mlp_fused_w_images_launch(mjobs, nmj, C, Wd, act, s), "split(mlp, w2 streams)");
    }
    const bool imgs = pr >= 1 && w.wimg && !(C % kmod) && !(Wd % kmod);
    // fp16 mode: the point-stream intermediates every consumer rounds to fp16 anyway (K|V, q, the attention output,
    // the MLP hidden layer) are STORED as fp16 — the same bits reach the matrix pipe, a third of the layer's HBM
    // bytes never move.  x (the residual stream) and everything on the 64 inducers stay fp32.
    const bool io16 = (pr == 2 || mixed) && imgs && N >= 128 && attn_x3_supported(C / H) && !(C % 8) && !(Wd % 8);
    if (mixed && !(io16 && !(N % 128) && !(C % 128)))
        return fail(-3, "set_transformer: the mixed mode needs rows %% 128 == 0, feature_dim %% 128 == 0 and a head dim of 16 / 32 / 48 / 64");
    const int apr = mixed ? 2 : pr;                  // arithmetic of the attention products
    const size_t kvq_lo = ((size_t)(2 * C + 127) / 128 * 128 + (size_t)(C + 127) / 128 * 128) * C / 2;   // floats: mixed mode's lo images
    // the cluster form of the one-launch chain (option "chaincl"): its per-(layer, sample) counters live in `merged` (unused by the chain,
    // 64 * C floats per sample) and are zeroed here, once per forward
    const bool chain_cl = (chain_on || chain2_on) && imgs && cl_ok;
    if (chain_cl) TRY((int)hipMemsetAsync(w.merged, 0, (size_t)st->n_layers * B * 8 * sizeof(unsigned), s), "inducer chain counters");
    // option "kvfold": the chain's last epilogue writes the fused unpool kernel's k | v image itself.  Only when EVERY layer runs the chain
    // (no cached inducer states: their layers bring the fp16 cast of x into the same buffer) — the image's pad positions are zeroed here,
    // once per forward, and nothing else touches the buffer in between
    bool kvfold = uo8_on && (chain_on || chain2_on) && imgs && option(OPT_KVFOLD);
    for (int li = 0; kvfold && li < st->n_layers; ++li)
        if (h_in && h_in[li]) kvfold = false;
    if (kvfold) TRY((int)hipMemsetAsync(w.attn, 0, unpool_outproj_h8_kv_bytes(B, C, H), s), "k | v image pads");
    for (int li = 0; li < st->n_layers; ++li) {
        const GeccoLayer& L = st->layers[li];
        const float* im = imgs ? w.wimg + (size_t)li * w.wimg_layer : nullptr;
        // y = AdaGN(x) is never materialised: (a1, o1) ride in the prologue of the two GEMMs that read x
        TRY(coeffs(sx, sT, N, t, ctx, &L.broadcast_norm, w.a1, w.o1, B, C, G, s), "adagn_coeffs(broadcast_norm)");
        const float* h = h_in ? h_in[li] : nullptr;
        bool q_done = false, kvh_done = false, kv_img_done = false;
        int hm = 0;   // K | V and q of this layer are head-major
        if (!h) {
            // pool: KV projection, 64 inducer queries over the N points, out_proj
            // kv_proj and the unpool's q projection read the same AdaGN(x): one launch, x read once
            // fp16 mode: AdaGN(x) is formed once as the fp16 operand both projections read (in the attention-output
            // buffer, idle until the unpool) instead of on every column tile's fragments
            float* y16 = w.attn;
            // K | V and q leave the A-stationary kernel head-major: one contiguous (N, hd) slab per (sample, head),
            // which is what a pool / unpool block streams (row-major: hd-wide pieces of rows shared by all heads)
            const int hd_try = option(OPT_HEADMAJOR) ? C / H : 0;
            const float* im_lo = mixed && im ? im + kvq_lo : nullptr;
            int fused = io16 ? astat_linear(x, w.a1, w.o1, (2 * C) % 128 == 0 ? im : nullptr, nullptr, 2 * C, w.big,
                                            L.in_proj_b, C, w.q, nullptr, 0, B, N, C, s, hd_try, im_lo, kvq_on ? 2 : kvq16_on ? 1 : 0)
                             : 1;
            if (io16 && fused == 1 && hd_try)   // shape outside the head-major form: row-major
                fused = astat_linear(x, w.a1, w.o1, (2 * C) % 128 == 0 ? im : nullptr, nullptr, 2 * C, w.big, L.in_proj_b,
                                     C, w.q, nullptr, 0, B, N, C, s, 0, im_lo, kvq_on ? 2 : kvq16_on ? 1 : 0);
            if ((mixed || kvq16_on) && fused != 0) return fail(fused < 0 ? fused : -3, "set_transformer: kv_proj | q_proj outside the A-stationary kernel's reach (its weight images are in that kernel's format)");
            else if (fused == 0 && hd_try)
                hm = 1;
            if (fused < 0) TRY(fused, "kv_proj|q_proj (A-stationary)");
            if (fused == 0) {
                q_done = true;
            } else {
            if (io16) TRY(affine_cast_f16_launch(x, w.a1, w.o1, y16, B, N, C, s), "broadcast_norm -> fp16");
            fused = linear_pair(io16 ? y16 : x, L.kv_proj_w, nullptr, 2 * C, w.big, L.in_proj_w, L.in_proj_b, C, w.q,
                                    io16 ? nullptr : w.a1, io16 ? nullptr : w.o1, B, N, C, s, pr, w.wsplit,
                                    (2 * C) % 128 == 0 ? im : nullptr, io16, io16);
            if (fused < 0) TRY(fused, "kv_proj|q_proj");
            q_done = fused == 0;
            if (!q_done)
                TRY(linear(io16 ? y16 : x, L.kv_proj_w, nullptr, io16 ? nullptr : w.a1, io16 ? nullptr : w.o1, nullptr, nullptr,
                           w.big, nullptr, B, N, C, 2 * C, 0, s, pr, w.wsplit, im, io16, io16), "kv_proj");
            }
            const bool chain = (chain_on || chain2_on) && im;
            TRY(pool_attn_launch(w.big, L.inducers, w.part_o, w.part_ml, chain ? nullptr : w.merged, B, N, C, H, I, ns, s,
                                 apr, io16, hm), "pool_attn");
            if (chain) {
                ChainArgs ca{};
                ca.part_o = w.part_o; ca.part_ml = w.part_ml; ca.nsplit = ns; ca.H = H;
                ca.w_stream = im + w.o_pout;   // o_pout, o_b0, o_b2, o_ukv are consecutive (carve_st)
                ca.b0 = L.bmlp.b0; ca.b2 = L.bmlp.b2; ca.bkv = L.in_proj_b + C; ca.alpha = L.bmlp.alpha; ca.act = act;
                ca.n1_scale_w = L.norm_1.scale_w; ca.n1_scale_b = L.norm_1.scale_b;
                ca.n1_bias_w = L.norm_1.bias_w; ca.n1_bias_b = L.norm_1.bias_b;
                ca.n2_scale_w = L.norm_2.scale_w; ca.n2_scale_b = L.norm_2.scale_b;
                ca.n2_bias_w = L.norm_2.bias_w; ca.n2_bias_b = L.norm_2.bias_b;
                ca.t = t; ca.ctx_dim = ctx; ca.G = G; ca.eps = 1e-5f;
                float* hdst = (h_out && h_out[li]) ? h_out[li] : w.h;
                ca.h_out = hdst; ca.kvh = w.kvh; ca.B = B;
                ca.two_term = chain2_on ? 1 : 0;
                if (chain_cl) {   // C / 128 blocks per sample; the stand-alone chain's buffers carry what they hand each other
                    ca.cluster = 1;
                    ca.x1 = w.h0; ca.x3 = w.h2; ca.xu = reinterpret_cast<unsigned*>(w.u);
                    ca.flags = reinterpret_cast<unsigned*>(w.merged) + (size_t)li * B * 8;
                }
                if (kvfold) {   // k | v leave the chain as the fused unpool kernel's fp16 image (pads zeroed above): no fp32 kvh, no reformat pass
                    ca.kv_img = reinterpret_cast<unsigned short*>(w.attn);
                    ca.kv_img_bytes = (int)(unpool_outproj_h8_kv_bytes(1, C, H) / H);
                    kv_img_done = true;
                }
                if ((act == 1 || act == 2) && !L.bmlp.alpha) return fail(-6, "inducer chain: GaussianActivation needs alpha");
                TRY(inducer_chain_f16_launch(ca, C, Wd, s), "inducer chain");
                h = hdst;
                kvh_done = true;
            } else {
            TRY(linear(w.merged, L.pool_out_w, nullptr, nullptr, nullptr, nullptr, nullptr, w.h0, w.stats_s, B, I, C, C,
                       0, s, pr, w.wsplit, im ? im + w.o_pout : nullptr), "pool.out_proj");
            // h = norm_2(mlp(norm_1(h0)))
            TRY(coeffs(w.stats_s, Ti, I, t, ctx, &L.norm_1, w.as, w.os, B, C, G, s), "adagn_coeffs(norm_1)");
            TRY(linear(w.h0, L.bmlp.w0, L.bmlp.b0, w.as, w.os, L.bmlp.alpha, nullptr, w.u, nullptr, B, I, C, Wd, act, s,
                       pr, w.wsplit, im ? im + w.o_b0 : nullptr), "broadcast.mlp.0");
            TRY(linear(w.u, L.bmlp.w2, L.bmlp.b2, nullptr, nullptr, nullptr, nullptr, w.h2, w.stats_s, B, I, Wd, C, 0, s,
                       pr, w.wsplit, im ? im + w.o_b2 : nullptr), "broadcast.mlp.2");
            TRY(coeffs(w.stats_s, Ti, I, t, ctx, &L.norm_2, w.as, w.os, B, C, G, s), "adagn_coeffs(norm_2)");
            float* hdst = (h_out && h_out[li]) ? h_out[li] : w.h;
            TRY(affine_apply_launch(w.h2, w.as, w.os, hdst, B, I, C, s), "norm_2 apply");
            h = hdst;
            }
        }
        // unpool: k|v of the 64 inducer states, q of the N points, attention, out_proj + residual
        if (!kvh_done)
            TRY(linear(h, L.in_proj_w + (size_t)C * C, L.in_proj_b + C, nullptr, nullptr, nullptr, nullptr, w.kvh, nullptr,
                       B, I, C, 2 * C, 0, s, pr, w.wsplit, im ? im + w.o_ukv : nullptr), "unpool.in_proj(kv)");
        if (!q_done && io16 && h_in && h_in[li]) {
            // cached inducer states (upsampling): the layer only needs q — the same one-pass kernel, one segment
            const int hd_try = option(OPT_HEADMAJOR) ? C / H : 0;
            // q's image: after kv_proj's (fp16 mode: o_q floats in; mixed mode: fp16 images inside the 4-byte layout)
            const float* qim = !im ? nullptr : kvq_on ? im + kvq_image_bytes(2 * C, C, C) / sizeof(float)
                                                      : mixed ? im + (size_t)(2 * C + 127) / 128 * 128 * C / 2 : im + w.o_q;
            const float* qim_lo = nullptr;   // q_proj's weights stay one-term (their rounding does not reach the output)
            int one = astat_linear(x, w.a1, w.o1, qim, L.in_proj_b, C, w.q, nullptr, 0, nullptr,
                                   nullptr, 0, B, N, C, s, hd_try, qim_lo, (kvq_on || kvq16_on) ? 1 : 0);
            if (one == 1 && hd_try)
                one = astat_linear(x, w.a1, w.o1, qim, L.in_proj_b, C, w.q, nullptr, 0, nullptr, nullptr,
                                   0, B, N, C, s, 0, qim_lo, (kvq_on || kvq16_on) ? 1 : 0);
            if ((mixed || kvq16_on) && one != 0) return fail(one < 0 ? one : -3, "set_transformer: q projection outside the A-stationary kernel's reach (its weight image is in that kernel's format)");
            else if (one == 0 && hd_try)
                hm = 1;
            if (one < 0) TRY(one, "unpool.in_proj(q) (A-stationary)");
            q_done = one == 0;
        }
        if (!q_done) {
            if (io16 && h_in && h_in[li]) TRY(affine_cast_f16_launch(x, w.a1, w.o1, w.attn, B, N, C, s), "broadcast_norm -> fp16");
            TRY(linear(io16 ? w.attn : x, L.in_proj_w, L.in_proj_b, io16 ? nullptr : w.a1, io16 ? nullptr : w.o1, nullptr,
                       nullptr, w.q, nullptr, B, N, C, C, 0, s, pr, w.wsplit, im ? im + w.o_q : nullptr, io16, io16),
                "unpool.in_proj(q)");
        }
        const bool a16 = io16 && !mixed;   // fp16-stored operands of the generic linears (fp16 mode only)
        if (uo8_on) {
            if (!hm || !im) return fail(-3, "set_transformer: unpool + out_proj (h8) needs the head-major fp16 q");
            if (!kv_img_done) TRY(kvh_image_launch(w.kvh, w.attn, B, C, H, s), "unpool k | v image");
            UnpoolH8Args ua{};
            ua.x = x; ua.q16 = w.q; ua.kv_img = w.attn; ua.w_img = im + w.o_out; ua.bias = L.unpool_out_b; ua.stats = w.stats_x;
            ua.B = B; ua.rows = N; ua.H = H;
            TRY(unpool_outproj_h8_launch(ua, C, s), "unpool attention + out_proj (h8)");
        } else if (!mixed && hm && im && I == 64 && option(OPT_UNPOOLFUSED) && unpool_outproj_f16_supported(C, H, N)) {
            // fp16 mode, head-major q: attention, out_proj, residual and statistics in one launch (the attention output
            // of a row block is the A operand of out_proj for the same rows and never leaves the CU)
            UnpoolProjArgs ua{};
            ua.x = x; ua.q16 = w.q; ua.kvh = w.kvh; ua.w_stream = im + w.o_out; ua.bias = L.unpool_out_b; ua.stats = w.stats_x;
            ua.B = B; ua.rows = N; ua.H = H;
            TRY(unpool_outproj_f16_launch(ua, C, s), "unpool attention + out_proj");
        } else {
            // mixed mode: fp16 q in, fp32 attention output (io16 = 2) = the operand of the split-bf16 out_proj — handed over as
            // a tiled split image where out_proj can load it straight into registers (gemm_x3_areg.hip)
            const int aimg = pr == 1 && !a16 && im && option(OPT_ACTIMG) && I == 64 && attn_x3_supported(C / H) && N >= 128 &&
                             N % 128 == 0 && C % 64 == 0;
            const bool o8 = h8o && aimg && im && N % 128 == 0;
            TRY(unpool_attn_launch(w.q, w.kvh, w.attn, B, N, C, H, I, s, apr, mixed ? 2 : (int)io16, hm, o8 ? 2 : aimg), "unpool_attn");
            if (o8)
                TRY(h8_linear(w.attn, im + w.o_out, L.unpool_out_b, x, x, w.stats_x, B, N, C, C, s), "unpool.out_proj+residual (h8)");
            else
            TRY(linear(w.attn, L.unpool_out_w, L.unpool_out_b, nullptr, nullptr, nullptr, x, x, w.stats_x, B, N, C, C, 0, s, pr,
                       w.wsplit, im ? im + w.o_out : nullptr, a16, 0, aimg, 0), "unpool.out_proj+residual");
        }
        // x += mlp(AdaGN(x))
        TRY(coeffs(w.stats_x, Tn, N, t, ctx, &L.mlp_norm, w.a2, w.o2, B, C, G, s), "adagn_coeffs(mlp_norm)");
        float* so = (li + 1 < st->n_layers) ? w.stats_x : stats_out;
        if (mlpf_on && im) {
            MlpArgs ma{};
            ma.x = x; ma.pro_a = w.a2; ma.pro_o = w.o2; ma.w_stream = im + w.o_mf;
            ma.b0 = L.mlp.b0; ma.b2 = L.mlp.b2; ma.alpha = L.mlp.alpha; ma.act = act; ma.stats = so; ma.B = B; ma.rows = N;
            if ((act == 1 || act == 2) && !L.mlp.alpha) return fail(-6, "mlp: GaussianActivation needs alpha");
            TRY(mlp_fused_f16_launch(ma, C, Wd, s), "mlp (fused)");
            sx = w.stats_x;
            sT = Tn;
            continue;
        }
        int m0_done = a16 ? astat_linear(x, w.a2, w.o2, im ? im + w.o_w0 : nullptr, L.mlp.b0, Wd, w.big, nullptr, 0,
                                         nullptr, L.mlp.alpha, act, B, N, C, s)
                          : 1;
        if (m0_done < 0) TRY(m0_done, "mlp.0 (A-stationary)");
        // split-bf16 products: the hidden layer goes from mlp.0 to mlp.2 as a tiled split image (same bytes as the fp32 tensor
        // it replaces, in the same buffer): contiguous DMA pieces and no hi / lo split in mlp.2's K loop
        const int himg = pr == 1 && !a16 && im && option(OPT_ACTIMG) && N >= 128 && N % 128 == 0 && Wd % 16 == 0 && C % 16 == 0;
        if (m0_done == 1 && mfw_on && im) {
            MlpWArgs ma{};
            ma.x = x; ma.out = x; ma.pro_a = w.a2; ma.pro_o = w.o2; ma.w_img = im + w.o_mf; ma.alpha = L.mlp.alpha;
            ma.act = act; ma.stats = so; ma.B = B; ma.rows = N; ma.share = option(OPT_MLPWSHARE);
            if ((act == 1 || act == 2) && !L.mlp.alpha) return fail(-6, "mlp: GaussianActivation needs alpha");
            TRY(mlp_fused_w_launch(ma, C, Wd, s), "mlp (one launch, w2)");
            sx = w.stats_x;
            sT = Tn;
            continue;
        }
        if (m0_done == 1 && h8_on && himg && im) {
            GemmArgs hg{};
            hg.A = x; hg.pro_a = w.a2; hg.pro_o = w.o2; hg.bias = L.mlp.b0; hg.alpha = L.mlp.alpha; hg.act = act; hg.C = w.big;
            hg.B = B; hg.rows = N; hg.K = C; hg.Nout = Wd; hg.lda = C; hg.ldw = C; hg.ldc = Wd; hg.c_img = h8x ? 2 : 1; hg.w_img = im + w.o_w0; hg.h6 = h6_on ? 1 : 0;
            if ((act == 1 || act == 2) && !L.mlp.alpha) return fail(-6, "mlp.0: GaussianActivation needs alpha");
            TRY(gemm_h8_astat_launch(hg, s), "mlp.0 (h8)");
            m0_done = 0;
            if (h8x) {
                TRY(h8_linear(w.big, im + w.o_w2, L.mlp.b2, x, x, so, B, N, Wd, C, s), "mlp.2+residual (h8)");
                sx = w.stats_x;
                sT = Tn;
                continue;
            }
        }
        if (m0_done == 1) {
        if (a16) TRY(affine_cast_f16_launch(x, w.a2, w.o2, w.attn, B, N, C, s), "mlp_norm -> fp16");
        TRY(linear(a16 ? w.attn : x, L.mlp.w0, L.mlp.b0, a16 ? nullptr : w.a2, a16 ? nullptr : w.o2, L.mlp.alpha, nullptr,
                   w.big, nullptr, B, N, C, Wd, act, s, pr, w.wsplit, im ? im + w.o_w0 : nullptr, a16, a16, 0, (himg && h8x) ? 2 : himg), "mlp.0");
        if (himg && h8x && im) {   // feature_dim 512: mlp.0 on the split-bf16 kernel writing the h8 activation image, mlp.2 as the h8 product
            TRY(h8_linear(w.big, im + w.o_w2, L.mlp.b2, x, x, so, B, N, Wd, C, s), "mlp.2+residual (h8)");
            sx = w.stats_x;
            sT = Tn;
            continue;
        }
        }
        TRY(linear(w.big, L.mlp.w2, L.mlp.b2, nullptr, nullptr, nullptr, x, x, so, B, N, Wd, C, 0, s, pr, w.wsplit,
                   im ? im + w.o_w2 : nullptr, a16, 0, himg, 0), "mlp.2+residual");
        sx = w.stats_x;
        sT = Tn;
    }
    return 0;
}

struct LLWorkspace {
    float *feat, *coef, *stats;
    void* st_ws;
    size_t st_bytes, bytes;
};

LLWorkspace carve_ll(const GeccoLinearLift* m, int B, int N, void* base) {
    Carver c(base);
    LLWorkspace w;
    w.feat = c.f32((size_t)B * N * m->inner.C);
    w.coef = c.f32((size_t)B * 5);
    w.stats = c.f32((size_t)B * row_tiles_stats(N) * 2 * m->inner.C);
    c.off = (c.off + 255) & ~size_t(255);
    w.st_bytes = carve_st(&m->inner, B, N, nullptr).bytes;
    w.st_ws = base ? static_cast<char*>(base) + c.off : nullptr;
    w.bytes = c.off + w.st_bytes;
    return w;
}

}  // namespace

extern "C" {

int gecco_abi_version(void) { return GECCO_ABI_VERSION; }
const char* gecco_build_arch(void) { return "gfx950"; }
const char* gecco_last_error(void) { return g_err; }

int gecco_option_index(const char* name) {
    if (name)
        for (int i = 0; i < OPT_COUNT; ++i)
            if (!strcmp(name, g_option_names[i])) return i;
    return -1;
}

int gecco_set_option(const char* name, int value) {
    if (!name) return fail(-1, "set_option: null name");
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(name, g_option_names[i])) {
            g_options[i] = value < 0 ? -1 : (value != 0);   // < 0: back to the environment / default
            return 0;
        }
    return fail(-2, "set_option: unknown option '%s' (astat, chain, headmajor, mlpfused, unpoolfused, lo8, actimg, h8, kvq64, h8areg, chain2, unpoolh8, mlpw, chaincl, h6, kvfold, mlpwshare, kvqperm, imgproj16)", name);
}

int gecco_linear_row_tiles(int rows) { return row_tiles_gemm(rows); }
int gecco_stats_row_tiles(int rows) { return row_tiles_stats(rows); }

int gecco_linear_f32(const float* A, const float* W, const float* bias, const float* pro_a, const float* pro_o,
                     const float* alpha, const float* residual, float* C, float* stats, int B, int rows, int K,
                     int Nout, int act, void* stream) {
    if (!A || !W || !C) return fail(-1, "linear: null argument");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "linear: pro_a/pro_o must both be set");
    TRY(linear(A, W, bias, pro_a, pro_o, alpha, residual, C, stats, B, rows, K, Nout, act, (hipStream_t)stream),
        "linear");
    return 0;
}

int gecco_linear_ex_f32(const float* A, const float* W, const float* bias, const float* pro_a, const float* pro_o,
                        const float* alpha, const float* residual, float* C, float* stats, int B, int rows, int K,
                        int Nout, int act, int precision, void* wsplit, void* stream) {
    if (!A || !C) return fail(-1, "linear: null argument");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "linear: pro_a/pro_o must both be set");
    if (precision < 0 || precision > 2) return fail(-2, "linear: precision must be 0 (fp32), 1 (split-bf16) or 2 (fp16)");
    if (precision >= 1 && !wsplit) return fail(-1, "linear: precision 1 / 2 need the wsplit scratch");
    if (!W) {   // wsplit already holds the image of W (gecco_split_bf16_images_f32 / gecco_split_f16_images_f32): kernel launch only
        if (precision == 0 || !(precision == 1 ? gecco_linear_image_ok(rows, K, Nout, pro_a != nullptr)
                                               : gecco_linear_image_ok_f16(rows, K, Nout, pro_a != nullptr)))
            return fail(-2, "linear: W == NULL (image ready) needs precision 1 / 2 and a shape gecco_linear_image_ok[_f16] accepts");
        TRY(linear(A, nullptr, bias, pro_a, pro_o, alpha, residual, C, stats, B, rows, K, Nout, act, (hipStream_t)stream, precision, nullptr,
                   static_cast<const float*>(wsplit)), "linear");
        return 0;
    }
    TRY(linear(A, W, bias, pro_a, pro_o, alpha, residual, C, stats, B, rows, K, Nout, act, (hipStream_t)stream, precision,
               static_cast<float*>(wsplit)), "linear");
    return 0;
}

int gecco_linear_pair_f32(const float* A, const float* W1, const float* bias1, int Nout1, float* C1, const float* W2,
                          const float* bias2, int Nout2, float* C2, const float* pro_a, const float* pro_o, int B,
                          int rows, int K, int precision, void* wsplit, void* stream) {
    if (!A || !C1 || !C2 || (!W1) != (!W2)) return fail(-1, "linear_pair: null argument");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "linear_pair: pro_a/pro_o must both be set");
    if (precision < 0 || precision > 2) return fail(-2, "linear_pair: precision must be 0 (fp32), 1 (split-bf16) or 2 (fp16)");
    if (precision >= 1 && !wsplit) return fail(-1, "linear_pair: precision 1 / 2 need the wsplit scratch");
    hipStream_t s = (hipStream_t)stream;
    float* ws = static_cast<float*>(wsplit);
    if (!W1) {   // wsplit holds the images of W1 and, from the next 128-column tile boundary, W2
        if (precision == 0) return fail(-2, "linear_pair: W == NULL (images ready) needs precision 1 / 2");
        int rc = linear_pair(A, nullptr, bias1, Nout1, C1, nullptr, bias2, Nout2, C2, pro_a, pro_o, B, rows, K, s, precision, nullptr, ws);
        if (rc == 1) return fail(-2, "linear_pair: images ready, but the shape is outside the fused kernel's reach");
        TRY(rc, "linear_pair");
        return 0;
    }
    int rc = linear_pair(A, W1, bias1, Nout1, C1, W2, bias2, Nout2, C2, pro_a, pro_o, B, rows, K, s, precision, ws);
    if (rc < 0) TRY(rc, "linear_pair");
    if (rc == 1) {   // shape outside the fused kernel's reach: the two linears, same results
        TRY(linear(A, W1, bias1, pro_a, pro_o, nullptr, nullptr, C1, nullptr, B, rows, K, Nout1, 0, s, precision, ws), "linear_pair[0]");
        TRY(linear(A, W2, bias2, pro_a, pro_o, nullptr, nullptr, C2, nullptr, B, rows, K, Nout2, 0, s, precision, ws), "linear_pair[1]");
    }
    return 0;
}

int gecco_linear_image_ok(int rows, int K, int Nout, int with_prologue) {
    GemmArgs g{};
    float dummy = 0.f;
    g.A = &dummy; g.W = &dummy; g.C = &dummy; g.pro_a = with_prologue ? &dummy : nullptr; g.pro_o = g.pro_a;
    g.B = 1; g.rows = rows; g.K = K; g.Nout = Nout; g.lda = K; g.ldw = K; g.ldc = Nout; g.ldr = Nout;
    return gemm_f32_dma_supported(g, 1) ? 1 : 0;
}
int gecco_linear_image_ok_f16(int rows, int K, int Nout, int with_prologue) {
    GemmArgs g{};
    float dummy = 0.f;
    g.A = &dummy; g.W = &dummy; g.C = &dummy; g.pro_a = with_prologue ? &dummy : nullptr; g.pro_o = g.pro_a;
    g.B = 1; g.rows = rows; g.K = K; g.Nout = Nout; g.lda = K; g.ldw = K; g.ldc = Nout; g.ldr = Nout;
    return gemm_f16_dma_supported(g) ? 1 : 0;
}
size_t gecco_split_f16_image_bytes(int Nout, int K) { return split_f16_image_bytes(Nout, K); }
int gecco_split_f16_images_f32(const GeccoSplitJob* jobs, int n, void* stream) {
    if (n < 0 || (n > 0 && !jobs)) return fail(-1, "split_f16_images: null argument");
    SplitJobs sj;
    sj.n = 0;
    for (int i = 0; i < n; ++i) {
        const GeccoSplitJob& j = jobs[i];
        if (!j.W || !j.img || j.Nout <= 0 || j.K <= 0 || (j.K % 32) || (!j.transposed && (j.ldw & 3)))
            return fail(-2, "split_f16_images: job %d needs K %% 32 == 0 (and ldw %% 4 == 0 unless transposed)", i);
        sj.job[sj.n++] = SplitJob{j.W, static_cast<float*>(j.img), j.Nout, j.K, j.ldw, j.transposed ? 4 : 0};
        if (sj.n == 96) {
            TRY(split_f16_tiled_multi_launch(sj, (hipStream_t)stream), "split_f16_images");
            sj.n = 0;
        }
    }
    TRY(split_f16_tiled_multi_launch(sj, (hipStream_t)stream), "split_f16_images");
    return 0;
}
size_t gecco_split_bf16_image_bytes(int Nout, int K) { return split_bf16_image_bytes(Nout, K); }
int gecco_split_bf16_images_f32(const GeccoSplitJob* jobs, int n, void* stream) {
    if (n < 0 || (n > 0 && !jobs)) return fail(-1, "split_bf16_images: null argument");
    SplitJobs sj;
    sj.n = 0;
    for (int i = 0; i < n; ++i) {
        const GeccoSplitJob& j = jobs[i];
        if (!j.W || !j.img || j.Nout <= 0 || j.K <= 0 || (j.K % 16) || (!j.transposed && (j.ldw & 3)))
            return fail(-2, "split_bf16_images: job %d needs K %% 16 == 0 (and ldw %% 4 == 0 unless transposed)", i);
        sj.job[sj.n++] = SplitJob{j.W, static_cast<float*>(j.img), j.Nout, j.K, j.ldw, j.transposed ? 4 : 0};
        if (sj.n == 96) {
            TRY(split_bf16_tiled_multi_launch(sj, (hipStream_t)stream), "split_bf16_images");
            sj.n = 0;
        }
    }
    TRY(split_bf16_tiled_multi_launch(sj, (hipStream_t)stream), "split_bf16_images");
    return 0;
}

int gecco_linear_actbwd_ok(int rows, int K, int Nout, int precision) {
    GemmArgs g{};
    float dummy = 0.f;
    g.A = &dummy; g.W = &dummy; g.C = &dummy;
    g.B = 1; g.rows = rows; g.K = K; g.Nout = Nout; g.lda = K; g.ldw = K; g.ldc = Nout; g.ldr = Nout;
    if (precision == 2) return gemm_f16_dma_supported(g) ? 1 : 0;
    return (precision == 0 || precision == 1) && gemm_f32_dma_supported(g, precision) ? 1 : 0;
}
size_t gecco_linear_actbwd_tiles(int B, int rows, int Nout) { return (size_t)B * ((rows + 63) / 64) * ((Nout + 127) / 128); }   // one slot per 64 rows
int gecco_linear_actbwd_f32(const float* A, const float* W, const float* u, const float* alpha, int kind, const float* residual,
                            float* C, float* agrad, int B, int rows, int K, int Nout, int precision, void* wsplit, void* stream) {
    if (!A || !u || !C) return fail(-1, "linear_actbwd: null argument");
    if (kind < 1 || kind > 4) return fail(-2, "linear_actbwd: kind must be 1 / 2 (GaussianActivation), 3 (ReLU) or 4 (GELU)");
    if ((kind == 1 || kind == 2) && (!alpha || !agrad)) return fail(-1, "linear_actbwd: GaussianActivation needs alpha and the agrad partials");
    if (!gecco_linear_actbwd_ok(rows, K, Nout, precision)) return fail(-2, "linear_actbwd: shape / precision outside the LDS-DMA kernels' reach");
    if (precision >= 1 && !wsplit) return fail(-1, "linear_actbwd: precision 1 / 2 need wsplit");
    if (!W && precision == 0) return fail(-2, "linear_actbwd: W == NULL (image ready) needs precision 1 / 2");
    TRY(linear(A, W, nullptr, nullptr, nullptr, alpha, residual, C, nullptr, B, rows, K, Nout, 0, (hipStream_t)stream, precision,
               W ? static_cast<float*>(wsplit) : nullptr, W ? nullptr : static_cast<const float*>(wsplit), 0, 0, 0, 0, u, kind,
               (kind == 1 || kind == 2) ? agrad : nullptr), "linear_actbwd");
    return 0;
}

int gecco_linear_dotstats_f32(const float* A, const float* W, const float* dot_x, float* C, float* stats, int B, int rows, int K, int Nout,
                              int precision, void* wsplit, void* stream) {
    if (!A || !dot_x || !C || !stats) return fail(-1, "linear_dotstats: null argument");
    if (!gecco_linear_actbwd_ok(rows, K, Nout, precision)) return fail(-2, "linear_dotstats: shape / precision outside the LDS-DMA kernels' reach");
    if (precision >= 1 && !wsplit) return fail(-1, "linear_dotstats: precision 1 / 2 need wsplit");
    if (!W && precision == 0) return fail(-2, "linear_dotstats: W == NULL (image ready) needs precision 1 / 2");
    TRY(linear(A, W, nullptr, nullptr, nullptr, nullptr, nullptr, C, stats, B, rows, K, Nout, 0, (hipStream_t)stream, precision,
               W ? static_cast<float*>(wsplit) : nullptr, W ? nullptr : static_cast<const float*>(wsplit), 0, 0, 0, 0, nullptr, 0, nullptr, nullptr,
               dot_x), "linear_dotstats");
    return 0;
}

int gecco_linear_dotstats_a16_f32(const void* A16, const float* W, const float* dot_x, const float* residual, float* C, float* stats, int B, int rows,
                                  int K, int Nout, void* wsplit, void* stream) {
    if (!A16 || !dot_x || !C || !stats || !wsplit) return fail(-1, "linear_dotstats_a16: null argument");
    if (!gecco_linear_actbwd_ok(rows, K, Nout, 2) || (K & 7) || rows < 128) return fail(-2, "linear_dotstats_a16: shape outside the fp16 LDS-DMA kernel's reach");
    int rc = linear(static_cast<const float*>(A16), W, nullptr, nullptr, nullptr, nullptr, residual, C, stats, B, rows, K, Nout, 0, (hipStream_t)stream, 2,
                    W ? static_cast<float*>(wsplit) : nullptr, W ? nullptr : static_cast<const float*>(wsplit), 1, 0, 0, 0, nullptr, 0, nullptr, nullptr,
                    dot_x);
    if (rc == -9) return fail(-2, "linear_dotstats_a16: shape outside the fp16 LDS-DMA kernel's reach");
    TRY(rc, "linear_dotstats_a16");
    return 0;
}

int gecco_linear_act_keep_f32(const float* A, const float* W, const float* bias, const float* alpha, int act, float* pre_out,
                              float* C, int B, int rows, int K, int Nout, int precision, void* wsplit, void* stream) {
    return gecco_linear_act_keep_pro_f32(A, W, bias, nullptr, nullptr, alpha, act, pre_out, C, B, rows, K, Nout, precision, wsplit, stream);
}

int gecco_linear_act_keep_pro_f32(const float* A, const float* W, const float* bias, const float* pro_a, const float* pro_o,
                                  const float* alpha, int act, float* pre_out, float* C, int B, int rows, int K, int Nout,
                                  int precision, void* wsplit, void* stream) {
    if (!A || !pre_out || !C) return fail(-1, "linear_act_keep: null argument");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "linear_act_keep: pro_a / pro_o must both be set");
    if (pro_a && K > 1024) return fail(-2, "linear_act_keep: the AdaGN prologue needs K <= 1024");
    if (act < 1 || act > 4) return fail(-2, "linear_act_keep: act must be 1 / 2 (GaussianActivation), 3 (ReLU) or 4 (GELU)");
    if ((act == 1 || act == 2) && !alpha) return fail(-1, "linear_act_keep: GaussianActivation needs alpha");
    if (!gecco_linear_actbwd_ok(rows, K, Nout, precision)) return fail(-2, "linear_act_keep: shape / precision outside the LDS-DMA kernels' reach");
    if (precision >= 1 && !wsplit) return fail(-1, "linear_act_keep: precision 1 / 2 need wsplit");
    if (!W && precision == 0) return fail(-2, "linear_act_keep: W == NULL (image ready) needs precision 1 / 2");
    TRY(linear(A, W, bias, pro_a, pro_o, alpha, nullptr, C, nullptr, B, rows, K, Nout, act, (hipStream_t)stream, precision,
               W ? static_cast<float*>(wsplit) : nullptr, W ? nullptr : static_cast<const float*>(wsplit), 0, 0, 0, 0, nullptr, 0,
               nullptr, pre_out), "linear_act_keep");
    return 0;
}

/* ---- the training forward in h8 arithmetic with fp32 tensors (gemm_h8_astat.hip, OUT forms of gemm_h8_astat_kernel) ---- */
size_t gecco_h8_image_bytes(int Nout, int K) { return (Nout % 64 || K % 64) ? 0 : h8_image_bytes(Nout, K); }
int gecco_linear_h8_train_ok(int rows, int K, int Nout) {
    return rows >= 128 && rows % 128 == 0 && (K == 128 || K == 256 || K == 384) && Nout % 64 == 0 && Nout >= 128 && Nout <= 4096;
}
int gecco_h8_images_f32(const GeccoSplitJob* jobs, int n, void* stream) {
    if (n < 0 || (n > 0 && !jobs)) return fail(-1, "h8_images: null argument");
    SplitJobs sj;
    sj.n = 0;
    for (int i = 0; i < n; ++i) {
        const GeccoSplitJob& j = jobs[i];
        if (!j.W || !j.img || j.Nout <= 0 || (j.Nout % 64) || (j.K % 64) || j.K <= 0 || (j.ldw & 3) || j.transposed)
            return fail(-2, "h8_images: job %d needs Nout %% 64 == 0, K %% 64 == 0, ldw %% 4 == 0, not transposed", i);
        sj.job[sj.n++] = SplitJob{j.W, static_cast<float*>(j.img), j.Nout, j.K, j.ldw, 0};
        if (sj.n == 96) {
            TRY(h8_image_multi_launch(sj, (hipStream_t)stream), "h8_images");
            sj.n = 0;
        }
    }
    TRY(h8_image_multi_launch(sj, (hipStream_t)stream), "h8_images");
    return 0;
}

int gecco_linear_h8_train_f32(const float* x, const float* pro_a, const float* pro_o, const float* W1, const float* bias1, int Nout1, float* C1,
                              const float* W2, const float* bias2, int Nout2, float* C2, const float* alpha, int act, float* pre_out, int B,
                              int rows, int K, void* wsplit, void* stream) {
    if (!x || !C1 || !wsplit || (Nout2 > 0 && !C2)) return fail(-1, "linear_h8_train: null argument");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "linear_h8_train: pro_a / pro_o must both be set");
    if (Nout2 > 0 && ((W1 == nullptr) != (W2 == nullptr))) return fail(-1, "linear_h8_train: W1 / W2 both given or both ready");
    if ((act != 0) != (pre_out != nullptr)) return fail(-2, "linear_h8_train: an activation comes with pre_out (the keep form), and only with it");
    if ((act == 1 || act == 2) && !alpha) return fail(-1, "linear_h8_train: GaussianActivation needs alpha");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g{};
    g.A = x; g.pro_a = pro_a; g.pro_o = pro_o; g.bias = bias1; g.C = C1; g.alpha = alpha; g.act = act; g.pre_out = pre_out;
    g.B = B; g.rows = rows; g.K = K; g.Nout = Nout1 + (Nout2 > 0 ? Nout2 : 0); g.lda = K; g.ldw = K; g.ldc = Nout1; g.ldr = Nout1;
    g.precision = 1; g.w_img = wsplit;
    if (Nout2 > 0) { g.C2 = C2; g.bias2 = bias2; g.n_split = Nout1; g.ldc2 = Nout2; }
    if (!gemm_h8_train_supported(g))
        return fail(-2, "linear_h8_train: needs rows %% 128 == 0, K in {128, 256, 384}, Nout (each segment) %% 64 == 0, Nout >= 128, act 0 .. 3");
    if (W1) {
        SplitJobs jobs;
        jobs.n = 0;
        float* img = static_cast<float*>(wsplit);
        jobs.job[jobs.n++] = SplitJob{W1, img, Nout1, K, K, 0};
        if (Nout2 > 0) jobs.job[jobs.n++] = SplitJob{W2, img + h8_image_bytes(Nout1, K) / sizeof(float), Nout2, K, K, 0};
        TRY(h8_image_multi_launch(jobs, s), "linear_h8_train(image)");
    }
    TRY(gemm_h8_train_launch(g, s), "linear_h8_train");
    return 0;
}

/* ---- A-stationary fp16 linears with fp32 tensors: the training path's 384-wide products under autocast(float16) (gemm_h8_astat.hip,
 * OUT forms of gemm_kvq_astat_kernel) ---- */
size_t gecco_astat16_image_bytes(int Nout, int K) { return (Nout % 64 || K % 64) ? 0 : kvq_image_bytes(Nout, K, 0); }
int gecco_linear_astat16_ok(int rows, int K, int Nout) {
    return rows >= 128 && rows % 128 == 0 && (K == 128 || K == 256 || K == 384 || K == 512) && Nout % 64 == 0 && Nout >= 128 && Nout <= 4096;
}
int gecco_astat16_images_f32(const GeccoSplitJob* jobs, int n, void* stream) {
    if (n < 0 || (n > 0 && !jobs)) return fail(-1, "astat16_images: null argument");
    SplitJobs sj;
    sj.n = 0;
    for (int i = 0; i < n; ++i) {
        const GeccoSplitJob& j = jobs[i];
        if (!j.W || !j.img || j.Nout <= 0 || (j.Nout % 64) || (j.K % 64) || j.K <= 0 || (!j.transposed && (j.ldw & 3)))
            return fail(-2, "astat16_images: job %d needs Nout %% 64 == 0, K %% 64 == 0 (and ldw %% 4 == 0 unless transposed)", i);
        sj.job[sj.n++] = SplitJob{j.W, static_cast<float*>(j.img), j.Nout, j.K, j.ldw, j.transposed ? 5 : 1};
        if (sj.n == 96) {
            TRY(h8_image_multi_launch(sj, (hipStream_t)stream), "astat16_images");
            sj.n = 0;
        }
    }
    TRY(h8_image_multi_launch(sj, (hipStream_t)stream), "astat16_images");
    return 0;
}

int gecco_linear_astat16_f32(const float* x, const float* pro_a, const float* pro_o, const float* W1, const float* bias1, int Nout1, float* C1,
                             const float* W2, const float* bias2, int Nout2, float* C2, const float* residual, int transposed, int B, int rows,
                             int K, void* wsplit, void* stream) {
    if (!x || !C1 || !wsplit || (Nout2 > 0 && !C2)) return fail(-1, "linear_astat16: null argument");
#ifndef GECCO_EXPERIMENTAL
    // C1 = x W + another gradient: measured, no gain inside the step (19.31 vs 19.27 ms) — not part of the shipped surface
    if (residual) return fail(-2, "linear_astat16: the residual form is an experiment (build with -DGECCO_EXPERIMENTAL); pass NULL");
#endif
    if (residual && (Nout2 > 0 || pro_a || bias1)) return fail(-2, "linear_astat16: the residual form takes one weight, no prologue, no bias");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "linear_astat16: pro_a / pro_o must both be set");
    if (Nout2 > 0 && ((W1 == nullptr) != (W2 == nullptr))) return fail(-1, "linear_astat16: W1 / W2 both given or both ready");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g{};
    g.A = x; g.pro_a = pro_a; g.pro_o = pro_o; g.bias = bias1; g.C = C1;
    g.B = B; g.rows = rows; g.K = K; g.Nout = Nout1 + (Nout2 > 0 ? Nout2 : 0); g.lda = K; g.ldw = K; g.ldc = Nout1; g.ldr = Nout1;
    g.precision = 2; g.w_img = wsplit;
    if (Nout2 > 0) { g.C2 = C2; g.bias2 = bias2; g.n_split = Nout1; g.ldc2 = Nout2; }
    if (residual) { g.mul_u = residual; g.mul_kind = 0; }   // C1 = x W^T + residual (the epilogue form that reads a second tensor)
    if (!gemm_astat_train_supported(g))
        return fail(-2, "linear_astat16: needs rows %% 128 == 0, K in {128, 256, 384, 512}, Nout (each segment) %% 64 == 0, Nout >= 128");
    if (W1) {
        if (transposed && Nout2 > 0) return fail(-2, "linear_astat16: the transposed form takes one weight");
        SplitJobs jobs;
        jobs.n = 0;
        float* img = static_cast<float*>(wsplit);
        jobs.job[jobs.n++] = SplitJob{W1, img, Nout1, K, transposed ? Nout1 : K, transposed ? 5 : 1};
        if (Nout2 > 0) jobs.job[jobs.n++] = SplitJob{W2, img + kvq_image_bytes(Nout1, K, 0) / sizeof(float), Nout2, K, K, 1};
        TRY(h8_image_multi_launch(jobs, s), "linear_astat16(image)");
    }
    TRY(gemm_astat_train_launch(g, s), "linear_astat16");
    return 0;
}

int gecco_linear_astat16_keep(const float* x, const float* pro_a, const float* pro_o, const float* W, const float* bias, const float* alpha,
                              int act, float* pre_out, void* C16out, int B, int rows, int K, int Nout, void* wsplit, void* stream) {
    return gecco_linear_astat16_keep_y16(x, pro_a, pro_o, W, bias, alpha, act, pre_out, C16out, nullptr, B, rows, K, Nout, wsplit, stream);
}

int gecco_linear_astat16_keep_y16(const float* x, const float* pro_a, const float* pro_o, const float* W, const float* bias, const float* alpha,
                                  int act, float* pre_out, void* C16out, void* y16, int B, int rows, int K, int Nout, void* wsplit,
                                  void* stream) {
    if (!x || !pre_out || !C16out || !wsplit) return fail(-1, "linear_astat16_keep: null argument");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "linear_astat16_keep: pro_a / pro_o must both be set");
    if ((act == 1 || act == 2) && !alpha) return fail(-1, "linear_astat16_keep: GaussianActivation needs alpha");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g{};
    g.A = x; g.pro_a = pro_a; g.pro_o = pro_o; g.bias = bias; g.alpha = alpha; g.act = act; g.C = static_cast<float*>(C16out); g.pre_out = pre_out;
    g.B = B; g.rows = rows; g.K = K; g.Nout = Nout; g.lda = K; g.ldw = K; g.ldc = Nout; g.ldr = Nout;
    g.precision = 2; g.w_img = wsplit; g.y16_out = y16;
    if (!gemm_astat_train_supported(g))
        return fail(-2, "linear_astat16_keep: needs rows %% 128 == 0, K in {128, 256, 384, 512}, Nout %% 64 == 0, act 1 / 2 (GaussianActivation) or 3 (ReLU)");
    if (W) {
        SplitJobs jobs;
        jobs.n = 1;
        jobs.job[0] = SplitJob{W, static_cast<float*>(wsplit), Nout, K, K, 1};
        TRY(h8_image_multi_launch(jobs, s), "linear_astat16_keep(image)");
    }
    TRY(gemm_astat_train_launch(g, s), "linear_astat16_keep");
    return 0;
}

int gecco_linear_astat16_actbwd(const float* dy, const float* W, const float* u, const float* alpha, int kind, float* C, float* agrad, int B,
                                int rows, int K, int Nout, void* wsplit, void* stream) {
    if (!dy || !u || !C || !wsplit) return fail(-1, "linear_astat16_actbwd: null argument");
    if ((kind == 1 || kind == 2) && (!alpha || !agrad)) return fail(-1, "linear_astat16_actbwd: GaussianActivation needs alpha and the agrad partials");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g{};
    g.A = dy; g.alpha = alpha; g.C = C; g.mul_u = u; g.mul_kind = kind; g.agrad = (kind == 1 || kind == 2) ? agrad : nullptr;
    g.B = B; g.rows = rows; g.K = K; g.Nout = Nout; g.lda = K; g.ldw = K; g.ldc = Nout; g.ldr = Nout;
    g.precision = 2; g.w_img = wsplit;
    if (!gemm_astat_train_supported(g))
        return fail(-2, "linear_astat16_actbwd: needs rows %% 128 == 0, K in {128, 256, 384, 512}, Nout %% 64 == 0, kind 1 / 2 (GaussianActivation) or 3 (ReLU)");
    if (W) {   // the linear's own weight (K, Nout): the stream of its transpose, straight from it
        SplitJobs jobs;
        jobs.n = 1;
        jobs.job[0] = SplitJob{W, static_cast<float*>(wsplit), Nout, K, Nout, 5};
        TRY(h8_image_multi_launch(jobs, s), "linear_astat16_actbwd(image)");
    }
    TRY(gemm_astat_train_launch(g, s), "linear_astat16_actbwd");
    return 0;
}

int gecco_linear_astat16_actbwd_h16(const float* dy, const float* W, const float* u, const float* alpha, int kind, void* C16out, float* agrad, int B,
                                    int rows, int K, int Nout, void* wsplit, void* stream) {
    if (!dy || !u || !C16out || !wsplit) return fail(-1, "linear_astat16_actbwd_h16: null argument");
    if (kind < 1 || kind > 3) return fail(-2, "linear_astat16_actbwd_h16: kind 1 / 2 (GaussianActivation) or 3 (ReLU)");
    if ((kind == 1 || kind == 2) && (!alpha || !agrad)) return fail(-1, "linear_astat16_actbwd_h16: GaussianActivation needs alpha and the agrad partials");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g{};
    g.A = dy; g.alpha = alpha; g.C = static_cast<float*>(C16out); g.mul_u = u; g.mul_kind = kind; g.agrad = (kind == 1 || kind == 2) ? agrad : nullptr;
    g.B = B; g.rows = rows; g.K = K; g.Nout = Nout; g.lda = K; g.ldw = K; g.ldc = Nout; g.ldr = Nout;
    g.precision = 2; g.w_img = wsplit; g.c_f16 = 1;
    if (!gemm_astat_train_supported(g))
        return fail(-2, "linear_astat16_actbwd_h16: needs rows %% 128 == 0, K in {128, 256, 384, 512}, Nout %% 64 == 0");
    if (W) {
        SplitJobs jobs;
        jobs.n = 1;
        jobs.job[0] = SplitJob{W, static_cast<float*>(wsplit), Nout, K, Nout, 5};
        TRY(h8_image_multi_launch(jobs, s), "linear_astat16_actbwd_h16(image)");
    }
    TRY(gemm_astat_train_launch(g, s), "linear_astat16_actbwd_h16");
    return 0;
}

int gecco_linear_act_keep_h16(const float* A, const float* W, const float* bias, const float* pro_a, const float* pro_o,
                              const float* alpha, int act, float* pre_out, void* C16out, int B, int rows, int K, int Nout, void* wsplit,
                              void* stream) {
    if (!A || !pre_out || !C16out || !wsplit) return fail(-1, "linear_act_keep_h16: null argument");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "linear_act_keep_h16: pro_a / pro_o must both be set");
    if (act < 1 || act > 4) return fail(-2, "linear_act_keep_h16: act must be 1 / 2 (GaussianActivation), 3 (ReLU) or 4 (GELU)");
    if ((act == 1 || act == 2) && !alpha) return fail(-1, "linear_act_keep_h16: GaussianActivation needs alpha");
    int rc = linear(A, W, bias, pro_a, pro_o, alpha, nullptr, static_cast<float*>(C16out), nullptr, B, rows, K, Nout, act, (hipStream_t)stream, 2,
                    W ? static_cast<float*>(wsplit) : nullptr, W ? nullptr : static_cast<const float*>(wsplit), 0, 1, 0, 0, nullptr, 0,
                    nullptr, pre_out);
    if (rc == -9) return fail(-2, "linear_act_keep_h16: needs rows >= 128, K %% 32 == 0, K <= 1024 with a prologue, Nout %% 4 == 0");
    TRY(rc, "linear_act_keep_h16");
    return 0;
}

int gecco_linear_f16io(const void* A, const float* W, const float* bias, const float* alpha, const float* residual,
                       void* C, float* stats, int B, int rows, int K, int Nout, int act, int a_f16, int c_f16,
                       void* wsplit, void* stream) {
    if (!A || !C || !wsplit) return fail(-1, "linear_f16io: null argument");
    if (!a_f16 && !c_f16) return fail(-2, "linear_f16io: at least one of A / C must be an fp16 tensor (else gecco_linear_ex_f32)");
    // W == NULL: wsplit already holds the fp16 image of W (gecco_split_f16_images_f32): kernel launch only
    int rc = linear(static_cast<const float*>(A), W, bias, nullptr, nullptr, alpha, residual, static_cast<float*>(C),
                    stats, B, rows, K, Nout, act, (hipStream_t)stream, 2, W ? static_cast<float*>(wsplit) : nullptr,
                    W ? nullptr : static_cast<const float*>(wsplit), a_f16, c_f16);
    if (rc == -9) return fail(-2, "linear_f16io: needs rows >= 128, K %% 32 == 0, lda %% 8 == 0; fp16 C excludes residual / stats");
    TRY(rc, "linear_f16io");
    return 0;
}

int gecco_linear_pair_f16io(const void* A, const float* W1, const float* bias1, int Nout1, void* C1, const float* W2,
                            const float* bias2, int Nout2, void* C2, int B, int rows, int K, void* wsplit,
                            void* stream) {
    if (!A || !W1 || !W2 || !C1 || !C2 || !wsplit) return fail(-1, "linear_pair_f16io: null argument");
    int rc = linear_pair(static_cast<const float*>(A), W1, bias1, Nout1, static_cast<float*>(C1), W2, bias2, Nout2,
                         static_cast<float*>(C2), nullptr, nullptr, B, rows, K, (hipStream_t)stream, 2,
                         static_cast<float*>(wsplit), nullptr, 1, 1);
    if (rc == 1 || rc == -9) return fail(-2, "linear_pair_f16io: needs rows >= 128, K %% 32 == 0, Nout1 %% 128 == 0");
    TRY(rc, "linear_pair_f16io");
    return 0;
}

int gecco_linear_astat_f16(const float* x, const float* pro_a, const float* pro_o, const float* W1, const float* bias1,
                           int Nout1, void* C1, const float* W2, const float* bias2, int Nout2, void* C2,
                           const float* alpha, int act, int B, int rows, int K, int head_dim, void* wsplit, void* stream) {
    if (!x || !C1 || !wsplit) return fail(-1, "linear_astat: null argument");
    const bool image_ready = W1 == nullptr;   // wsplit holds the images a previous call made from the same weights
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "linear_astat: pro_a/pro_o must both be set");
    if (!image_ready && (W2 == nullptr) != (C2 == nullptr)) return fail(-1, "linear_astat: W2 and C2 go together");
    hipStream_t s = (hipStream_t)stream;
    float* img = static_cast<float*>(wsplit);
    if (!image_ready) {
        TRY(split_f16_tiled_launch(W1, img, Nout1, K, K, s), "linear_astat(split)");
        if (W2) TRY(split_f16_tiled_launch(W2, img + split_f16_image_bytes(Nout1, K) / sizeof(float), Nout2, K, K, s), "linear_astat(split)");
    }
    GemmArgs g{};
    g.A = x; g.pro_a = pro_a; g.pro_o = pro_o; g.bias = bias1; g.alpha = alpha; g.act = act; g.C = static_cast<float*>(C1);
    g.B = B; g.rows = rows; g.K = K; g.Nout = Nout1 + (C2 ? Nout2 : 0); g.lda = K; g.ldw = K; g.ldc = Nout1; g.ldr = Nout1;
    g.precision = 2; g.w_img = img; g.c_f16 = 1; g.hm_hd = head_dim;
    if (C2) { g.C2 = static_cast<float*>(C2); g.bias2 = bias2; g.n_split = Nout1; g.ldc2 = Nout2; }
    if ((act == 1 || act == 2) && !alpha) return fail(-6, "linear_astat: GaussianActivation needs alpha");
    if (head_dim < 0) return fail(-2, "linear_astat: head_dim < 0");
    if (!gemm_f16_astat_supported(g))
        return fail(-2, "linear_astat: needs rows %% 128 == 0, Nout %% 128 == 0, K in {128, 256, 384, 512}; head-major: "
                        "even head_dim >= 8 dividing both segment widths");
    TRY(gemm_f16_astat_launch(g, s), "linear_astat");
    return 0;
}

int gecco_linear_kvq_f16(const float* x, const float* pro_a, const float* pro_o, const float* W1, const float* bias1, int Nout1,
                         void* C1, const float* W2, const float* bias2, int Nout2, void* C2, int B, int rows, int K, int head_dim,
                         int lo_begin, int lo_end, void* wsplit, void* stream) {
    return gecco_linear_kvq_y16_f16(x, pro_a, pro_o, W1, bias1, Nout1, C1, W2, bias2, Nout2, C2, nullptr, B, rows, K, head_dim, lo_begin, lo_end,
                                    wsplit, stream);
}

int gecco_linear_kvq_y16_f16(const float* x, const float* pro_a, const float* pro_o, const float* W1, const float* bias1, int Nout1,
                             void* C1, const float* W2, const float* bias2, int Nout2, void* C2, void* y16, int B, int rows, int K,
                             int head_dim, int lo_begin, int lo_end, void* wsplit, void* stream) {
    if (!x || !C1 || !wsplit) return fail(-1, "linear_kvq: null argument");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "linear_kvq: pro_a/pro_o must both be set");
    if ((Nout2 > 0) != (C2 != nullptr)) return fail(-1, "linear_kvq: Nout2 and C2 go together");
    if (head_dim < 0 || lo_begin < 0 || lo_end < lo_begin || lo_end > Nout1 || (lo_begin & 63) || (lo_end & 63) || (Nout1 & 63) || (Nout2 & 63) ||
        K <= 0 || (K & 127))
        return fail(-2, "linear_kvq: head_dim >= 0; lo range inside the first segment, multiples of 64; Nout %% 64 == 0; K %% 128 == 0");
    hipStream_t s = (hipStream_t)stream;
    float* img = static_cast<float*>(wsplit);
    GemmArgs g{};
    g.A = x; g.pro_a = pro_a; g.pro_o = pro_o; g.bias = bias1; g.C = static_cast<float*>(C1);
    g.B = B; g.rows = rows; g.K = K; g.Nout = Nout1 + Nout2; g.lda = K; g.ldw = K; g.ldc = Nout1; g.ldr = Nout1;
    g.precision = 2; g.w_img = img; g.c_f16 = 1; g.hm_hd = head_dim; g.lo_begin = lo_begin / 64; g.lo_tiles = lo_end / 64;
    g.y16_out = y16;
    if (C2) { g.C2 = static_cast<float*>(C2); g.bias2 = bias2; g.n_split = Nout1; g.ldc2 = Nout2; }
    // (the two-term range must be whole 384-column segments: it is a range of TILES in the stream)
    g.kvq_perm = option(OPT_KVQPERM) && kvq_perm48_ok(head_dim, K, Nout1, Nout2) && lo_begin % 384 == 0 && lo_end % 384 == 0;
    const int p48 = g.kvq_perm ? 64 : 0;
    if (!gemm_kvq_astat_supported(g))
        return fail(-2, "linear_kvq: needs rows %% 128 == 0, Nout1 + Nout2 >= 128, K in {128, 256, 384, 512}; head-major: head_dim %% 8 == 0 "
                        "dividing both segment widths");
    if (W1) {   // NULL: wsplit still holds the stream a previous call made from the same weights
        if (Nout2 > 0 && !W2) return fail(-1, "linear_kvq: W2 missing");
        SplitJobs jobs;
        jobs.n = 0;
        jobs.job[jobs.n++] = SplitJob{W1, img, Nout1, K, K, 1 | p48 | ((lo_begin / 64) << 8) | ((lo_end / 64) << 20)};
        if (Nout2 > 0) jobs.job[jobs.n++] = SplitJob{W2, img + kvq_image_bytes(Nout1, K, lo_end - lo_begin) / sizeof(float), Nout2, K, K, 1 | p48};
        TRY(h8_image_multi_launch(jobs, s), "linear_kvq(image)");
    }
    TRY(gemm_kvq_astat_launch(g, s), "linear_kvq");
    return 0;
}

int gecco_linear_h8_img_f32(const float* x, const float* pro_a, const float* pro_o, const float* W, const float* bias,
                            const float* alpha, int act, void* c_img, int image_kind, int B, int rows, int K, int Nout, void* wsplit,
                            void* stream) {
    if (!x || !c_img || !wsplit) return fail(-1, "linear_h8_img: null argument");
    if (image_kind != 1 && image_kind != 2) return fail(-2, "linear_h8_img: image_kind must be 1 (tiled split image) or 2 (h8 activation image)");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "linear_h8_img: pro_a/pro_o must both be set");
    if ((act == 1 || act == 2) && !alpha) return fail(-6, "linear_h8_img: GaussianActivation needs alpha");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g{};
    g.A = x; g.pro_a = pro_a; g.pro_o = pro_o; g.bias = bias; g.alpha = alpha; g.act = act; g.C = static_cast<float*>(c_img);
    g.B = B; g.rows = rows; g.K = K; g.Nout = Nout; g.lda = K; g.ldw = K; g.ldc = Nout; g.c_img = image_kind; g.w_img = wsplit;
    g.h6 = image_kind == 2 && option(OPT_H6) ? 1 : 0;   // option "h6": the cross terms in fp6 with block scales (the network's mlp.0)
    if (!gemm_h8_astat_supported(g))
        return fail(-2, "linear_h8_img: needs rows %% 128 == 0, Nout %% 64 == 0, Nout >= 128, K in {128, 256, 384}, act in 0 .. 3");
    if (W) {   // NULL: wsplit still holds the image a previous call made from the same weights
        SplitJobs jobs;
        jobs.n = 1;
        jobs.job[0] = SplitJob{W, static_cast<float*>(wsplit), Nout, K, K, g.h6 ? 32 : 0};
        TRY(h8_image_multi_launch(jobs, s), "linear_h8_img(image)");
    }
    TRY(gemm_h8_astat_launch(g, s), "linear_h8_img");
    return 0;
}

int gecco_linear_h8_areg_f32(const void* a_img, const float* W, const float* bias, const float* residual, float* C, float* stats,
                             int B, int rows, int K, int Nout, void* wsplit, void* stream) {
    if (!a_img || !C || !wsplit) return fail(-1, "linear_h8_areg: null argument");
    hipStream_t s = (hipStream_t)stream;
    GemmArgs g{};
    g.A = static_cast<const float*>(a_img); g.bias = bias; g.residual = residual; g.C = C; g.stats = stats;
    g.B = B; g.rows = rows; g.K = K; g.Nout = Nout; g.lda = K; g.ldw = K; g.ldc = Nout; g.ldr = Nout; g.a_img = 2; g.w_img = wsplit; g.precision = 1;
    if (!gemm_h8_areg_supported(g))
        return fail(-2, "linear_h8_areg: needs rows %% 128 == 0, K in {128, 256, 384, 512, 768, 1024}, Nout %% 4 == 0");
    if (W) {   // NULL: wsplit still holds the image a previous call made from the same weights
        SplitJobs jobs;
        jobs.n = 1;
        jobs.job[0] = SplitJob{W, static_cast<float*>(wsplit), Nout, K, K, 2};
        TRY(h8_image_multi_launch(jobs, s), "linear_h8_areg(image)");
    }
    TRY(gemm_h8_areg_launch(g, s), "linear_h8_areg");
    return 0;
}

int gecco_mlp_fused_f16(float* x, const float* pro_a, const float* pro_o, const float* W0, const float* b0, const float* W2,
                        const float* b2, const float* alpha, int act, float* stats, int B, int rows, int C, int width,
                        void* wsplit, void* stream) {
    if (!x || !pro_a || !pro_o || !wsplit || ((W0 == nullptr) != (W2 == nullptr))) return fail(-1, "mlp_fused: null argument");
    if ((act == 1 || act == 2) && !alpha) return fail(-6, "mlp_fused: GaussianActivation needs alpha");
    if (!mlp_fused_f16_supported(C, width, rows))
        return fail(-2, "mlp_fused: needs C in {128, 256, 384}, width == 2 C, rows %% 128 == 0");
    hipStream_t s = (hipStream_t)stream;
    float* img = static_cast<float*>(wsplit);
    SplitJobs jobs;
    jobs.n = 0;
    const int nkb = C / 32;
    for (int jc = 0; jc < width / 128; ++jc) {   // the stream order of mlp_fused_f16.hip
        float* cb = img + (size_t)jc * 2 * nkb * 2048;
        jobs.job[jobs.n++] = SplitJob{W0 + (size_t)jc * 128 * C, cb, 128, C, C, 0};
        for (int hf = 0; hf < 2; ++hf)
            jobs.job[jobs.n++] = SplitJob{W2 + (size_t)jc * 128 + hf * 64, cb + (size_t)(nkb + hf * (nkb / 2)) * 2048, C, 64, width, 0};
    }
    if (W0) TRY(split_f16_tiled_multi_launch(jobs, s), "mlp_fused(split)");   // W0 == W2 == NULL: wsplit holds the image already
    MlpArgs ma{};
    ma.x = x; ma.pro_a = pro_a; ma.pro_o = pro_o; ma.w_stream = img; ma.b0 = b0; ma.b2 = b2; ma.alpha = alpha; ma.act = act;
    ma.stats = stats; ma.B = B; ma.rows = rows;
    TRY(mlp_fused_f16_launch(ma, C, width, s), "mlp_fused");
    return 0;
}

int gecco_unpool_outproj_f16(float* x, const void* q16, const float* kvh, const float* W, const float* bias, float* stats,
                             int B, int rows, int C, int H, void* wsplit, void* stream) {
    if (!x || !q16 || !kvh || !wsplit) return fail(-1, "unpool_outproj: null argument");
    if (!unpool_outproj_f16_supported(C, H, rows))
        return fail(-2, "unpool_outproj: needs (C, head dim) in {(128, 16), (256, 32), (384, 48)}, rows %% 128 == 0");
    hipStream_t s = (hipStream_t)stream;
    if (W) TRY(split_f16_tiled_launch(W, wsplit, C, C, C, s), "unpool_outproj(split)");   // W == NULL: image ready
    UnpoolProjArgs ua{};
    ua.x = x; ua.q16 = q16; ua.kvh = kvh; ua.w_stream = static_cast<const float*>(wsplit); ua.bias = bias; ua.stats = stats;
    ua.B = B; ua.rows = rows; ua.H = H;
    TRY(unpool_outproj_f16_launch(ua, C, s), "unpool_outproj");
    return 0;
}

int gecco_unpool_outproj_h8(float* x, const void* q16, const float* kvh, const float* W, const float* bias, float* stats,
                            int B, int rows, int C, int H, void* wsplit, void* stream) {
    if (!x || !q16 || !kvh || !wsplit) return fail(-1, "unpool_outproj_h8: null argument");
    if (!unpool_outproj_h8_supported(C, H, rows))
        return fail(-2, "unpool_outproj_h8: needs (C, head dim) in {(128, 16), (256, 32), (384, 48)}, rows %% 128 == 0");
    hipStream_t s = (hipStream_t)stream;
    if (W) {   // W == NULL: image ready
        SplitJobs jobs;
        jobs.n = 1;
        jobs.job[0] = SplitJob{W, static_cast<float*>(wsplit), C, C, C, 16};
        TRY(h8_image_multi_launch(jobs, s), "unpool_outproj_h8(split)");
    }
    void* kvimg = static_cast<char*>(wsplit) + h8_image_bytes(C, C);
    TRY(kvh_image_launch(kvh, kvimg, B, C, H, s), "unpool_outproj_h8(k | v image)");
    UnpoolH8Args ua{};
    ua.x = x; ua.q16 = q16; ua.kv_img = kvimg; ua.w_img = wsplit; ua.bias = bias; ua.stats = stats; ua.B = B; ua.rows = rows; ua.H = H;
    TRY(unpool_outproj_h8_launch(ua, C, s), "unpool_outproj_h8");
    return 0;
}

int gecco_mlp_fused_w(const float* x, float* out, const float* pro_a, const float* pro_o, const float* W0, const float* b0, const float* W2,
                      const float* b2, const float* alpha, int act, float* stats, int B, int rows, int C, int width, void* wsplit, float* dbg_u,
                      void* stream) {
    if (!x || !out || !pro_a || !pro_o || !wsplit) return fail(-1, "mlp_fused_w: null argument");
    if (!mlp_fused_w_supported(C, width, rows)) return fail(-2, "mlp_fused_w: needs C in {128, 256, 384, 512}, width == 2 C, rows %% 128 == 0");
    if (act < 0 || act > 3) return fail(-6, "mlp_fused_w: act must be 0 .. 3");
    if ((act == 1 || act == 2) && !alpha) return fail(-6, "mlp_fused_w: GaussianActivation needs alpha");
    hipStream_t s = (hipStream_t)stream;
    if (W0 && W2) TRY(mlp_fused_w_image_launch(W0, b0, W2, b2, wsplit, C, width, alpha, act, s), "mlp_fused_w(image)");   // W0 == NULL: image ready (biases included)
    MlpWArgs ma{};
    ma.x = x; ma.out = out; ma.pro_a = pro_a; ma.pro_o = pro_o; ma.w_img = wsplit; ma.alpha = alpha; ma.act = act; ma.stats = stats;
    ma.B = B; ma.rows = rows; ma.dbg_u = dbg_u; ma.share = option(OPT_MLPWSHARE);
    TRY(mlp_fused_w_launch(ma, C, width, s), "mlp_fused_w");
    return 0;
}

size_t gecco_mlp_fused_w_wsplit_bytes(int C, int width) { return mlp_fused_w_image_bytes(C, width); }

size_t gecco_unpool_outproj_h8_wsplit_bytes(int B, int C, int H) {
    if (H <= 0 || C % H) return 0;
    return h8_image_bytes(C, C) + unpool_outproj_h8_kv_bytes(B, C, H);
}

int gecco_unpool_attn_h8img(const void* q16, const float* kvh, void* out_img, int B, int N, int C, int H, void* stream) {
    if (!q16 || !kvh || !out_img) return fail(-1, "unpool_attn_h8img: null argument");
    if (N % 128 || C % 64 || H <= 0 || C % H || !attn_x3_supported(C / H)) return fail(-2, "unpool_attn_h8img: needs N %% 128 == 0, C %% 64 == 0, head dim 16 / 32 / 48 / 64");
    TRY(unpool_attn_launch(static_cast<const float*>(q16), kvh, static_cast<float*>(out_img), B, N, C, H, 64, (hipStream_t)stream, 2, 2, 1, 2),
        "unpool_attn_h8img");
    return 0;
}

int gecco_gemm_tn_x3_f32(const float* A, const float* Bm, float* parts, int Z, int R, int N, int K, int group, void* stream) {
    return gecco_gemm_tn_x3_bias_f32(A, Bm, parts, nullptr, Z, R, N, K, group, stream);
}

int gecco_gemm_tn_x3_bias_f32(const float* A, const float* Bm, float* parts, float* colsum_parts, int Z, int R, int N, int K,
                              int group, void* stream) {
    return gecco_gemm_tn_x3_pro_f32(A, Bm, nullptr, nullptr, parts, colsum_parts, Z, R, N, K, group, stream);
}

int gecco_gemm_tn_x3_pro_f32(const float* A, const float* Bm, const float* pro_a, const float* pro_o, float* parts,
                             float* colsum_parts, int Z, int R, int N, int K, int group, void* stream) {
    if (!A || !Bm || !parts) return fail(-1, "gemm_tn_x3: null argument");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "gemm_tn_x3: pro_a / pro_o must both be set");
    TnArgs g{};
    g.pro_a = pro_a; g.pro_o = pro_o;
    g.A = A; g.Bm = Bm; g.C = parts; g.Z = Z; g.R = R; g.N = N; g.K = K; g.lda = N; g.ldb = K;
    g.sA = (size_t)R * N; g.sB = (size_t)R * K; g.group = group; g.colsum = colsum_parts;
    if (!gemm_tn_x3_supported(g)) return fail(-2, "gemm_tn_x3: needs R %% 32 == 0, N %% 4 == 0, K %% 4 == 0, group > 0");
    TRY(gemm_tn_x3_launch(g, (hipStream_t)stream), "gemm_tn_x3");
    return 0;
}

int gecco_gemm_tn_f16_b16_f32(const float* A, const void* B16, float* parts, float* colsum_parts, int Z, int R, int N, int K, int group,
                              void* stream) {
    if (!A || !B16 || !parts) return fail(-1, "gemm_tn_f16_b16: null argument");
    TnArgs g{};
    g.f16 = 1; g.b_f16 = 1;
    g.A = A; g.Bm = static_cast<const float*>(B16); g.C = parts; g.Z = Z; g.R = R; g.N = N; g.K = K; g.lda = N; g.ldb = K;
    g.sA = (size_t)R * N; g.sB = (size_t)R * K; g.group = group; g.colsum = colsum_parts;
    if (!gemm_tn_f16_supported(g)) return fail(-2, "gemm_tn_f16_b16: needs R %% 32 == 0, N %% 4 == 0, K %% 8 == 0, group > 0");
    TRY(gemm_tn_f16_launch(g, (hipStream_t)stream), "gemm_tn_f16_b16");
    return 0;
}

int gecco_gemm_tn_f16_ex_f32(const void* A, int a_f16, const void* Bm, int b_f16, const float* pro_a, const float* pro_o, float* parts,
                             float* colsum_parts, float* out, float* colsum_out, unsigned* counters, int Z, int R, int N, int K, int group,
                             void* stream) {
    if (!A || !Bm || !parts) return fail(-1, "gemm_tn_f16_ex: null argument");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "gemm_tn_f16_ex: pro_a / pro_o must both be set");
    if ((counters != nullptr) != (out != nullptr) || (colsum_out && !(colsum_parts && counters)))
        return fail(-1, "gemm_tn_f16_ex: counters and out go together; colsum_out needs colsum_parts and counters");
    if (a_f16 && b_f16 && (pro_a || counters || N % 128 || K % 128))
        return fail(-2, "gemm_tn_f16_ex: both operands fp16: whole 128 x 128 tiles, no AdaGN apply, the separate reduction");
    TnArgs g{};
    g.pro_a = pro_a; g.pro_o = pro_o; g.f16 = 1; g.a_f16 = a_f16 != 0; g.b_f16 = b_f16 != 0;
    g.A = static_cast<const float*>(A); g.Bm = static_cast<const float*>(Bm); g.C = parts; g.Z = Z; g.R = R; g.N = N; g.K = K; g.lda = N; g.ldb = K;
    g.sA = (size_t)R * N; g.sB = (size_t)R * K; g.group = group; g.colsum = colsum_parts;
    g.counters = counters; g.out = out; g.colsum_out = colsum_out;
    if (!gemm_tn_f16_supported(g)) return fail(-2, "gemm_tn_f16_ex: needs R %% 32 == 0, N %% 4 == 0, K %% 4 == 0 (8 for an fp16 operand's width), group > 0");
    TRY(gemm_tn_f16_launch(g, (hipStream_t)stream), "gemm_tn_f16_ex");
    return 0;
}

int gecco_gemm_tn_f16_a16_f32(const void* A16, const float* Bm, const float* pro_a, const float* pro_o, float* parts, float* colsum_parts,
                              int Z, int R, int N, int K, int group, void* stream) {
    if (!A16 || !Bm || !parts) return fail(-1, "gemm_tn_f16_a16: null argument");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "gemm_tn_f16_a16: pro_a / pro_o must both be set");
    TnArgs g{};
    g.pro_a = pro_a; g.pro_o = pro_o; g.f16 = 1; g.a_f16 = 1;
    g.A = static_cast<const float*>(A16); g.Bm = Bm; g.C = parts; g.Z = Z; g.R = R; g.N = N; g.K = K; g.lda = N; g.ldb = K;
    g.sA = (size_t)R * N; g.sB = (size_t)R * K; g.group = group; g.colsum = colsum_parts;
    if (!gemm_tn_f16_supported(g)) return fail(-2, "gemm_tn_f16_a16: needs R %% 32 == 0, N %% 8 == 0, K %% 4 == 0, group > 0");
    TRY(gemm_tn_f16_launch(g, (hipStream_t)stream), "gemm_tn_f16_a16");
    return 0;
}

int gecco_gemm_tn_f16_f32(const float* A, const float* Bm, const float* pro_a, const float* pro_o, float* parts,
                          float* colsum_parts, int Z, int R, int N, int K, int group, void* stream) {
    if (!A || !Bm || !parts) return fail(-1, "gemm_tn_f16: null argument");
    if ((pro_a == nullptr) != (pro_o == nullptr)) return fail(-1, "gemm_tn_f16: pro_a / pro_o must both be set");
    TnArgs g{};
    g.pro_a = pro_a; g.pro_o = pro_o; g.f16 = 1;
    g.A = A; g.Bm = Bm; g.C = parts; g.Z = Z; g.R = R; g.N = N; g.K = K; g.lda = N; g.ldb = K;
    g.sA = (size_t)R * N; g.sB = (size_t)R * K; g.group = group; g.colsum = colsum_parts;
    if (!gemm_tn_f16_supported(g)) return fail(-2, "gemm_tn_f16: needs R %% 32 == 0, N %% 4 == 0, K %% 4 == 0, group > 0");
    TRY(gemm_tn_f16_launch(g, (hipStream_t)stream), "gemm_tn_f16");
    return 0;
}
int gecco_gemm_tn_f16_tiles(int N, int K) { return gemm_tn_f16_tiles(N, K); }

int gecco_affine_cast_f16(const float* x, const float* a, const float* o, void* y16, int B, int rows, int C,
                          void* stream) {
    if (C % 8) return fail(-2, "affine_cast_f16: C must be a multiple of 8");
    TRY(affine_cast_f16_launch(x, a, o, y16, B, rows, C, (hipStream_t)stream), "affine_cast_f16");
    return 0;
}

int gecco_pool_attn_f16in(const void* KV16, const float* inducers, float* merged, int B, int N, int C, int H, int I,
                          int head_major, void* ws, size_t ws_bytes, void* stream) {
    if (ws_bytes < gecco_pool_attn_workspace_bytes(B, N, C, H, I)) return fail(-7, "pool_attn: workspace too small");
    Carver c(ws);
    const int ns = pool_attn_nsplit(B, N, H);
    float* po = c.f32((size_t)B * H * ns * 64 * (C / H));
    float* pml = c.f32((size_t)B * H * ns * 64 * 2);
    int rc = pool_attn_launch(static_cast<const float*>(KV16), inducers, po, pml, merged, B, N, C, H, I, ns,
                              (hipStream_t)stream, 2, 1, head_major != 0);
    if (rc == -9) return fail(-2, "pool_attn_f16in: head dim must be 16, 32, 48 or 64");
    TRY(rc, "pool_attn_f16in");
    return 0;
}

int gecco_unpool_attn_f16io(const void* q16, const float* kvh, void* out16, int B, int N, int C, int H, int I,
                            int head_major, void* stream) {
    int rc = unpool_attn_launch(static_cast<const float*>(q16), kvh, static_cast<float*>(out16), B, N, C, H, I,
                                (hipStream_t)stream, 2, 1, head_major != 0);
    if (rc == -9) return fail(-2, "unpool_attn_f16io: head dim must be 16, 32, 48 or 64");
    TRY(rc, "unpool_attn_f16io");
    return 0;
}

int gecco_col_stats_f32(const float* x, float* stats, int B, int rows, int C, void* stream) {
    TRY(col_stats_launch(x, stats, B, rows, C, (hipStream_t)stream), "col_stats");
    return 0;
}

int gecco_adagn_coeffs_f32(const float* stats, int T, int rows, const float* t, int ctx_dim, const GeccoAdaGN* p,
                           float* a, float* o, int B, int C, int G, float eps, void* stream) {
    TRY(adagn_coeffs_launch(stats, T, rows, t, ctx_dim, p ? p->scale_w : nullptr, p ? p->scale_b : nullptr,
                            p ? p->bias_w : nullptr, p ? p->bias_b : nullptr, a, o, B, C, G, eps,
                            (hipStream_t)stream), "adagn_coeffs");
    return 0;
}

int gecco_affine_apply_f32(const float* x, const float* a, const float* o, float* y, int B, int rows, int C,
                           void* stream) {
    TRY(affine_apply_launch(x, a, o, y, B, rows, C, (hipStream_t)stream), "affine_apply");
    return 0;
}

size_t gecco_adagn_workspace_bytes(int B, int rows, int C) {
    Carver c(nullptr);
    c.f32((size_t)B * row_tiles_stats(rows) * 2 * C);
    c.f32((size_t)B * C);
    c.f32((size_t)B * C);
    return (c.off + 255) & ~size_t(255);
}

int gecco_adagn_f32(const float* x, const float* t, int ctx_dim, const GeccoAdaGN* p, float* y, int B, int rows,
                    int C, int G, float eps, void* ws, size_t ws_bytes, void* stream) {
    if (ws_bytes < gecco_adagn_workspace_bytes(B, rows, C)) return fail(-7, "adagn: workspace too small");
    Carver c(ws);
    float* stats = c.f32((size_t)B * row_tiles_stats(rows) * 2 * C);
    float* a = c.f32((size_t)B * C);
    float* o = c.f32((size_t)B * C);
    hipStream_t s = (hipStream_t)stream;
    TRY(col_stats_launch(x, stats, B, rows, C, s), "col_stats");
    TRY(adagn_coeffs_launch(stats, row_tiles_stats(rows), rows, t, ctx_dim, p ? p->scale_w : nullptr,
                            p ? p->scale_b : nullptr, p ? p->bias_w : nullptr, p ? p->bias_b : nullptr, a, o, B, C, G,
                            eps, s), "adagn_coeffs");
    TRY(affine_apply_launch(x, a, o, y, B, rows, C, s), "affine_apply");
    return 0;
}

size_t gecco_pool_attn_workspace_bytes(int B, int N, int C, int H, int I) {
    (void)I;
    Carver c(nullptr);
    const int ns = pool_attn_nsplit(B, N, H);
    c.f32((size_t)B * H * ns * 64 * (C / H));
    c.f32((size_t)B * H * ns * 64 * 2);
    return (c.off + 255) & ~size_t(255);
}

int gecco_pool_attn_ex_f32(const float* KV, const float* inducers, float* merged, int B, int N, int C, int H, int I,
                           int precision, void* ws, size_t ws_bytes, void* stream) {
    if (precision < 0 || precision > 2) return fail(-2, "pool_attn: precision must be 0 (fp32), 1 (split-bf16) or 2 (fp16)");
    if (ws_bytes < gecco_pool_attn_workspace_bytes(B, N, C, H, I)) return fail(-7, "pool_attn: workspace too small");
    Carver c(ws);
    const int ns = pool_attn_nsplit(B, N, H);
    float* po = c.f32((size_t)B * H * ns * 64 * (C / H));
    float* pml = c.f32((size_t)B * H * ns * 64 * 2);
    TRY(pool_attn_launch(KV, inducers, po, pml, merged, B, N, C, H, I, ns, (hipStream_t)stream, precision), "pool_attn");
    return 0;
}

int gecco_pool_attn_lse_f32(const void* ws, size_t ws_bytes, float* lse, int B, int N, int C, int H, int I, void* stream) {
    if (ws_bytes < gecco_pool_attn_workspace_bytes(B, N, C, H, I)) return fail(-7, "pool_attn_lse: workspace too small");
    Carver c(const_cast<void*>(ws));
    const int ns = pool_attn_nsplit(B, N, H);
    c.f32((size_t)B * H * ns * 64 * (C / H));
    const float* pml = c.f32((size_t)B * H * ns * 64 * 2);
    TRY(pool_attn_lse_launch(pml, lse, B, H, ns, (hipStream_t)stream), "pool_attn_lse");
    return 0;
}

int gecco_pool_attn_bwd_partials(int B, int N, int H) { return pool_attn_bwd_nsplit(B, N, H); }
int gecco_unpool_attn_bwd_partials(int B, int N, int H) { return unpool_attn_bwd_chunks(B, N, H, nullptr); }

int gecco_pool_attn_bwd_f32(const float* KV, const float* inducers, const float* merged, const float* lse, const float* dO,
                            float* dKV, float* dQ_partials, int B, int N, int C, int H, int I, void* stream) {
    return gecco_pool_attn_bwd_ex_f32(KV, inducers, merged, lse, dO, dKV, dQ_partials, B, N, C, H, I, 0, stream);
}
int gecco_unpool_attn_bwd_f32(const float* q, const float* kvh, const float* dO, float* dq, float* dkv_partials, int B, int N,
                              int C, int H, int I, void* stream) {
    return gecco_unpool_attn_bwd_ex_f32(q, kvh, dO, dq, dkv_partials, B, N, C, H, I, 0, stream);
}

int gecco_pool_attn_bwd_ex_f32(const float* KV, const float* inducers, const float* merged, const float* lse, const float* dO,
                               float* dKV, float* dQ_partials, int B, int N, int C, int H, int I, int precision, void* stream) {
    if (B <= 0 || N <= 0) return fail(-2, "pool_attn_bwd: empty batch");
    if (precision < 0 || precision > 3) return fail(-2, "pool_attn_bwd: precision must be 0 (fp32), 1 (split-bf16), 2 (fp16 operands) or 3 (fp16 operands, fp16 KV / dKV tensors)");
    const int rc = pool_attn_bwd_launch(KV, inducers, merged, lse, dO, dKV, dQ_partials, B, N, C, H, I, pool_attn_bwd_nsplit(B, N, H),
                                        (hipStream_t)stream, precision);
    if (rc == -3 || rc == -4) return fail(-2, "pool_attn_bwd: needs I == 64 and a head dim that is a multiple of 8 up to 64");
    TRY(rc, "pool_attn_bwd");
    return 0;
}

int gecco_unpool_attn_bwd_ex_f32(const float* q, const float* kvh, const float* dO, float* dq, float* dkv_partials, int B, int N,
                                 int C, int H, int I, int precision, void* stream) {
    if (B <= 0 || N <= 0) return fail(-2, "unpool_attn_bwd: empty batch");
    if (precision < 0 || precision > 3) return fail(-2, "unpool_attn_bwd: precision must be 0 (fp32), 1 (split-bf16), 2 (fp16 operands) or 3 (fp16 operands, fp16 q / dO / dq tensors)");
    const int rc = unpool_attn_bwd_launch(q, kvh, dO, dq, dkv_partials, B, N, C, H, I, (hipStream_t)stream, precision);
    if (rc == -3 || rc == -4) return fail(-2, "unpool_attn_bwd: needs I == 64 and a head dim that is a multiple of 8 up to 64");
    TRY(rc, "unpool_attn_bwd");
    return 0;
}

int gecco_pool_attn_f32(const float* KV, const float* inducers, float* merged, int B, int N, int C, int H, int I,
                        void* ws, size_t ws_bytes, void* stream) {
    return gecco_pool_attn_ex_f32(KV, inducers, merged, B, N, C, H, I, 0, ws, ws_bytes, stream);
}

int gecco_unpool_attn_ex_f32(const float* q, const float* kvh, float* out, int B, int N, int C, int H, int I,
                             int precision, void* stream) {
    if (precision < 0 || precision > 2) return fail(-2, "unpool_attn: precision must be 0 (fp32), 1 (split-bf16) or 2 (fp16)");
    TRY(unpool_attn_launch(q, kvh, out, B, N, C, H, I, (hipStream_t)stream, precision), "unpool_attn");
    return 0;
}

int gecco_unpool_attn_f32(const float* q, const float* kvh, float* out, int B, int N, int C, int H, int I,
                          void* stream) {
    return gecco_unpool_attn_ex_f32(q, kvh, out, B, N, C, H, I, 0, stream);
}

int gecco_edm_coeffs_f32(const float* sigma, float sigma_data, float* coef, int B, void* stream) {
    TRY(edm_coeffs_launch(sigma, sigma_data, coef, B, (hipStream_t)stream), "edm_coeffs");
    return 0;
}

int gecco_lift_f32(const float* x, const float* coef, const float* W, const float* bias, float* out, float* stats,
                   int B, int N, int C, void* stream) {
    TRY(lift_launch(x, coef, W, bias, out, stats, B, N, C, (hipStream_t)stream), "lift");
    return 0;
}

int gecco_lower_edm_f32(const float* feat, const float* x, const float* coef, const float* W, const float* bias,
                        const float* gn_a, const float* gn_o, float* out, float* raw, int B, int N, int C, float eps,
                        void* stream) {
    if (coef && !x) return fail(-1, "lower_edm: x required with coef");
    TRY(lower_edm_launch(feat, x, coef, W, bias, gn_a, gn_o, out, raw, B, N, C, eps, (hipStream_t)stream), "lower_edm");
    return 0;
}

size_t gecco_set_transformer_workspace_bytes(const GeccoSetTransformer* st, int B, int N) {
    return carve_st(st, B, N, nullptr).bytes;
}

int gecco_set_transformer_fwd_f32(const GeccoSetTransformer* st, float* x, const float* t, const float* stats_x,
                                  int stats_T, const float* const* h_in, float* const* h_out, float* stats_out,
                                  int B, int N, void* ws, size_t ws_bytes, void* stream) {
    return st_forward(st, x, t, stats_x, stats_T, h_in, h_out, stats_out, B, N, ws, ws_bytes, (hipStream_t)stream);
}

size_t gecco_linear_lift_workspace_bytes(const GeccoLinearLift* m, int B, int N) {
    return carve_ll(m, B, N, nullptr).bytes;
}

int gecco_linear_lift_fwd_f32(const GeccoLinearLift* m, const float* x, const float* sigma, float* denoised,
                              float* raw, const float* const* h_in, float* const* h_out, int B, int N, void* ws,
                              size_t ws_bytes, void* stream) {
    if (!m || !x || !sigma || !(denoised || raw)) return fail(-1, "linear_lift: null argument");
    LLWorkspace w = carve_ll(m, B, N, ws);
    if (ws_bytes < w.bytes) return fail(-7, "linear_lift: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
    hipStream_t s = (hipStream_t)stream;
    const int C = m->inner.C;
    TRY(edm_coeffs_launch(sigma, m->sigma_data, w.coef, B, s), "edm_coeffs");
    TRY(lift_launch(x, w.coef, m->lift_w, m->lift_b, w.feat, w.stats, B, N, C, s), "lift");
    // AdaGN reads t as a packed (B, ctx_dim) array; under EDMPrecond ctx_dim == 1 and t = c_noise,
    // which edm_coeffs also writes packed at coef[4B .. 5B).
    if (m->inner.ctx_dim != 1) return fail(-3, "linear_lift: t_embed_dim must be 1 under EDMPrecond");
    int rc = st_forward(&m->inner, w.feat, w.coef + 4 * (size_t)B, w.stats, row_tiles_stats(N), h_in, h_out, nullptr, B,
                        N, w.st_ws, w.st_bytes, s);
    if (rc) return rc;
    TRY(lower_edm_launch(w.feat, x, w.coef, m->lower_w, m->lower_b, nullptr, nullptr, denoised, raw, B, N, C, 1e-5f, s),
        "lower_edm");
    return 0;
}

// ---------------------------------------------------------------------------- conditional path
int gecco_nchw_to_nhwc_f32(const float* src, float* dst, int B, int C, int H, int W, void* stream) {
    TRY(nchw_to_nhwc_launch(src, dst, B, C, H, W, (hipStream_t)stream), "nchw_to_nhwc");
    return 0;
}

int gecco_bilinear_taps_f32(const float* uv, int H, int W, int* x0, int* y0, float* wx1, float* wy1, size_t n,
                            void* stream) {
    TRY(bilinear_taps_launch(uv, H, W, x0, y0, wx1, wy1, n, (hipStream_t)stream), "bilinear_taps");
    return 0;
}

int gecco_lookup_row_tiles(int N) { return (N + lookup_row_tile() - 1) / lookup_row_tile(); }

}  // extern "C"

namespace {

int make_lookup_args(const GeccoReparam* rp, const GeccoPyramid* pyr, LookupArgs* a) {
    if (!pyr || pyr->n_levels < 1 || pyr->n_levels > 4) return fail(-8, "lookup: 1..4 pyramid levels required");
    a->n_levels = pyr->n_levels;
    a->c_total = 0;
    for (int l = 0; l < 4; ++l) {
        const bool on = l < pyr->n_levels;
        a->C[l] = on ? pyr->C[l] : 0;
        a->H[l] = on ? pyr->H[l] : 1;
        a->W[l] = on ? pyr->W[l] : 1;
        a->feat[l] = on ? pyr->feat[l] : nullptr;
        if (on && !pyr->feat[l]) return fail(-1, "lookup: null pyramid level %d", l);
        a->c_total += a->C[l];
    }
    a->texel_f16 = pyr->texel_f16 ? 1 : 0;
    a->out_f16 = 0;
    a->reparam_kind = rp ? rp->kind : 0;
    a->rp_mean = rp ? rp->mean : nullptr;
    a->rp_std = rp ? rp->std : nullptr;
    a->logit_scale = rp ? rp->logit_scale : 1.1f;
    if (a->reparam_kind && (!a->rp_mean || !a->rp_std)) return fail(-1, "lookup: reparam buffers missing");
    return 0;
}

struct RNWorkspace {
    float *feat, *raw, *coef, *stats_raw, *stats_x, *stats_out, *a_raw, *o_raw, *a_out, *o_out, *wsplit, *wfold, *bfold;
    void* st_ws;
    size_t st_bytes, bytes;
};

RNWorkspace carve_rn(const GeccoRayNetwork* m, int c_total, int B, int N, void* base) {
    Carver c(base);
    RNWorkspace w;
    const size_t C = m->backbone.C;
    w.feat = c.f32((size_t)B * N * C);
    w.raw = c.f32((size_t)B * N * c_total);
    w.coef = c.f32((size_t)B * 5);
    w.stats_raw = c.f32((size_t)B * gecco_lookup_row_tiles(N) * 2 * c_total);
    w.stats_x = c.f32((size_t)B * row_tiles_gemm(N) * 2 * C);
    w.stats_out = c.f32((size_t)B * row_tiles_gemm(N) * 2 * C);
    w.a_raw = c.f32((size_t)B * c_total);
    w.o_raw = c.f32((size_t)B * c_total);
    w.a_out = c.f32((size_t)B * C);
    w.o_out = c.f32((size_t)B * C);
    w.wsplit = c.f32(((C + 127) / 128 * 128) * (size_t)c_total);   // tiled image of img_feature_proj (precision 1 / 2)
    // "w2" mode ("imgproj16"): per-sample fp16 images of img_feature_proj with GN16's scale folded in, and the biases with its offsets
    const bool fold = m->backbone.precision == 4;
    w.wfold = fold ? c.f32((size_t)B * ((C + 127) / 128 * 128) * c_total / 2) : nullptr;
    w.bfold = fold ? c.f32((size_t)B * C) : nullptr;
    c.off = (c.off + 255) & ~size_t(255);
    w.st_bytes = carve_st(&m->backbone, B, N, nullptr).bytes;
    w.st_ws = base ? static_cast<char*>(base) + c.off : nullptr;
    w.bytes = c.off + w.st_bytes;
    return w;
}

}  // namespace

extern "C" {

int gecco_cast_f16(const float* src, void* dst, size_t n, void* stream) {
    if (n && (!src || !dst)) return fail(-1, "cast_f16: null argument");
    TRY(cast_f16_launch(src, dst, n, (hipStream_t)stream), "cast_f16");
    return 0;
}

int gecco_ray_lookup_f32(const float* geom, const float* coef, const float* K, const GeccoReparam* rp,
                         const GeccoPyramid* pyr, float* out, float* stats, int B, int N, void* stream) {
    if (!geom || !K || !out) return fail(-1, "ray_lookup: null argument");
    LookupArgs a;
    int rc = make_lookup_args(rp, pyr, &a);
    if (rc) return rc;
    TRY(ray_lookup_launch(geom, coef, K, a, out, stats, B, N, (hipStream_t)stream), "ray_lookup");
    return 0;
}

int gecco_ray_lookup_taps_f32(const float* geom, const float* coef, const float* K, const GeccoReparam* rp, const GeccoPyramid* pyr, float* uv,
                              int* x0, int* y0, float* wx1, float* wy1, int B, int N, void* stream) {
    if (!geom || !K || !pyr || !uv || !x0 || !y0 || !wx1 || !wy1) return fail(-1, "ray_lookup_taps: null argument");
    LookupArgs a;
    int rc = make_lookup_args(rp, pyr, &a);
    if (rc) return rc;
    TRY(ray_lookup_taps_launch(geom, coef, K, a, uv, x0, y0, wx1, wy1, B, N, (hipStream_t)stream), "ray_lookup_taps");
    return 0;
}

int gecco_ray_lookup_bwd_f32(const float* geom, const float* coef, const float* K, const GeccoReparam* rp,
                             const GeccoPyramid* pyr, const float* dout, float* const* dfeat, int B, int N,
                             void* stream) {
    if (!geom || !K || !dout || !dfeat || !pyr) return fail(-1, "ray_lookup_bwd: null argument");
    LookupArgs a;
    int rc = make_lookup_args(rp, pyr, &a);
    if (rc) return rc;
    if (a.texel_f16) return fail(-2, "ray_lookup_bwd: fp16 texel pyramids serve the forward lookup only (pass the fp32 levels)");
    for (int l = 0; l < a.n_levels; ++l)
        if (!dfeat[l]) return fail(-1, "ray_lookup_bwd: null gradient level");
    TRY(ray_lookup_bwd_launch(geom, coef, K, a, dfeat, dout, B, N, (hipStream_t)stream), "ray_lookup_bwd");
    return 0;
}

int gecco_ray_lookup_dgeom_f32(const float* geom, const float* K, const GeccoReparam* rp, const GeccoPyramid* pyr, const float* dout,
                               float* dgeom, float* dK_partials, int B, int N, void* stream) {
    if (!geom || !K || !dout || !pyr || (!dgeom && !dK_partials)) return fail(-1, "ray_lookup_dgeom: null argument");
    LookupArgs a;
    int rc = make_lookup_args(rp, pyr, &a);
    if (rc) return rc;
    if (a.texel_f16) return fail(-2, "ray_lookup_dgeom: fp16 texel pyramids serve the forward lookup only (pass the fp32 levels)");
    TRY(ray_lookup_dgeom_launch(geom, K, a, dout, dgeom, dK_partials, B, N, (hipStream_t)stream), "ray_lookup_dgeom");
    return 0;
}

size_t gecco_ray_lookup_bwd_sorted_workspace_bytes(const GeccoPyramid* pyr, int B, int N) {
    LookupArgs a;
    if (!pyr || make_lookup_args(nullptr, pyr, &a) || !ray_lookup_bwd_sorted_supported(a, N)) return 0;
    return ray_lookup_bwd_sorted_ws_bytes(a, B, N);
}
int gecco_ray_lookup_bwd_sorted_f32(const float* geom, const float* coef, const float* K, const GeccoReparam* rp,
                                    const GeccoPyramid* pyr, const float* dout, float* const* dfeat, int B, int N, void* ws,
                                    size_t ws_bytes, void* stream) {
    if (!geom || !K || !dout || !dfeat || !pyr || !ws) return fail(-1, "ray_lookup_bwd_sorted: null argument");
    LookupArgs a;
    int rc = make_lookup_args(rp, pyr, &a);
    if (rc) return rc;
    if (a.texel_f16) return fail(-2, "ray_lookup_bwd_sorted: fp16 texel pyramids serve the forward lookup only (pass the fp32 levels)");
    for (int l = 0; l < a.n_levels; ++l)
        if (!dfeat[l]) return fail(-1, "ray_lookup_bwd_sorted: null gradient level");
    if (!ray_lookup_bwd_sorted_supported(a, N)) return fail(-2, "ray_lookup_bwd_sorted: needs N <= 4096 and H W <= 2^17 per level");
    if (ws_bytes < ray_lookup_bwd_sorted_ws_bytes(a, B, N)) return fail(-3, "ray_lookup_bwd_sorted: workspace too small");
    TRY(ray_lookup_bwd_sorted_launch(geom, coef, K, a, dfeat, dout, B, N, ws, (hipStream_t)stream), "ray_lookup_bwd_sorted");
    return 0;
}

size_t gecco_ray_network_workspace_bytes(const GeccoRayNetwork* m, const GeccoPyramid* pyr, int B, int N) {
    int ct = 0;
    for (int l = 0; l < pyr->n_levels && l < 4; ++l) ct += pyr->C[l];
    return carve_rn(m, ct, B, N, nullptr).bytes;
}

int gecco_ray_network_fwd_f32(const GeccoRayNetwork* m, const float* x, const float* sigma, const float* K,
                              const GeccoPyramid* pyr, float* denoised, float* raw, const float* const* h_in,
                              float* const* h_out, int B, int N, void* ws, size_t ws_bytes, void* stream) {
    if (!m || !x || !sigma || !K || !(denoised || raw)) return fail(-1, "ray_network: null argument");
    if (m->backbone.ctx_dim != 1) return fail(-3, "ray_network: t_embed_dim must be 1 under EDMPrecond");
    LookupArgs a;
    int rc = make_lookup_args(&m->reparam, pyr, &a);
    if (rc) return rc;
    RNWorkspace w = carve_rn(m, a.c_total, B, N, ws);
    if (ws_bytes < w.bytes) return fail(-7, "ray_network: workspace too small (%zu < %zu)", ws_bytes, w.bytes);
    hipStream_t s = (hipStream_t)stream;
    const int C = m->backbone.C;
    TRY(edm_coeffs_launch(sigma, m->sigma_data, w.coef, B, s), "edm_coeffs");
    // xyz_embed(c_in * x)  (models/ray.py:99)
    TRY(lift_launch(x, w.coef, m->xyz_w, m->xyz_b, w.feat, nullptr, B, N, C, s), "xyz_embed");
    // projective lookup on c_in * x, fp32 always (models/ray.py:103-109) + GN(16) partials
    // "w2" with option "imgproj16" (opt-in: it moves C3's F_x from 2.3e-4 to 2.7e-4 of the mode's 5e-4 for 2 % of the evaluation): the lookup leaves halves, GN16's apply goes INTO the weights (per-sample images of W * a, biases + W o) and
    // img_feature_proj multiplies fp16(lookup) by them, one term each, on the fp16-operand streaming kernel: a third of split-bf16's matrix
    // work on half its operand bytes (the mode's one-term operands are the hidden layer, K and q already)
    PlanScope plan_scope(&m->backbone);
    const bool img16 = m->backbone.precision == 4 && option(OPT_IMGPROJ16) && a.c_total % 32 == 0 && N >= 128 && C % 4 == 0;
    a.out_f16 = img16;
    TRY(ray_lookup_launch(x, w.coef, K, a, w.raw, w.stats_raw, B, N, s), "ray_lookup");
    TRY(adagn_coeffs_launch(w.stats_raw, gecco_lookup_row_tiles(N), N, nullptr, 0, nullptr, nullptr, nullptr, nullptr,
                            w.a_raw, w.o_raw, B, a.c_total, 16, 1e-5f, s), "gn16(img)");
    // point_features = xyz_features + Linear(GN16(lookup))  (models/ray.py:112-113): GN apply in the GEMM
    // prologue, the add as its residual, the first AdaGN's statistics in its epilogue
    if (img16) {
        TRY(fold_f16_image_launch(m->img_w, m->img_b, w.a_raw, w.o_raw, w.wfold, w.bfold, B, C, a.c_total, a.c_total, s), "img_feature_proj fold");
        GemmArgs g{};
        g.A = w.raw; g.a_f16 = 1; g.bias = w.bfold; g.bias_bstride = C; g.residual = w.feat; g.C = w.feat; g.stats = w.stats_x;
        g.B = B; g.rows = N; g.K = a.c_total; g.Nout = C; g.lda = a.c_total; g.ldw = a.c_total; g.ldc = C; g.ldr = C;
        g.precision = 2; g.w_img = w.wfold; g.w_img_bstride = (size_t)((C + 127) / 128 * 128) * a.c_total / 2;
        if (!gemm_f16_dma_supported(g)) return fail(-9, "img_feature_proj: shape outside the fp16 streaming kernel");
        TRY(gemm_f32_launch(g, s), "img_feature_proj");
    } else {
        TRY(linear(w.raw, m->img_w, m->img_b, w.a_raw, w.o_raw, nullptr, w.feat, w.feat, w.stats_x, B, N, a.c_total, C, 0,
                   s, m->backbone.precision >= 3 ? 1 : m->backbone.precision, w.wsplit), "img_feature_proj");   // mixed mode: split-bf16
    }
    rc = st_forward(&m->backbone, w.feat, w.coef + 4 * (size_t)B, w.stats_x, row_tiles_gemm(N), h_in, h_out,
                    w.stats_out, B, N, w.st_ws, w.st_bytes, s);
    if (rc) return rc;
    // output_proj = Linear(GN16(.)) (models/ray.py:56-59,120) + EDM combine (diffusion.py:57)
    TRY(adagn_coeffs_launch(w.stats_out, row_tiles_gemm(N), N, nullptr, 0, nullptr, nullptr, nullptr, nullptr, w.a_out,
                            w.o_out, B, C, 16, 1e-5f, s), "gn16(out)");
    TRY(lower_edm_launch(w.feat, x, w.coef, m->out_w, m->out_b, w.a_out, w.o_out, denoised, raw, B, N, C, 1e-5f, s),
        "output_proj+edm");
    return 0;
}

// ---------------------------------------------------------------------------- reparam / activation / sampler
int gecco_gaussian_reparam(const void* x, const float* mean, const float* sigma, void* y, size_t n_elems, int dim,
                           int inverse, int is_f64, void* stream) {
    TRY(gaussian_reparam_launch(x, mean, sigma, y, n_elems, dim, inverse, is_f64, (hipStream_t)stream),
        "gaussian_reparam");
    return 0;
}
int gecco_uvl_reparam(const void* x, const float* K, const float* uvl_mean, const float* uvl_std, double logit_scale,
                      void* y, int B, int N, int inverse, int is_f64, void* stream) {
    TRY(uvl_reparam_launch(x, K, uvl_mean, uvl_std, logit_scale, y, B, N, inverse, is_f64, (hipStream_t)stream),
        "uvl_reparam");
    return 0;
}
int gecco_relu_f32(const float* x, float* y, size_t n, void* stream) {
    TRY(relu_launch(x, y, n, (hipStream_t)stream), "relu");
    return 0;
}
int gecco_relu_bwd_f32(const float* y, const float* dy, float* du, size_t n, void* stream) {
    TRY(relu_bwd_launch(y, dy, du, n, (hipStream_t)stream), "relu_bwd");
    return 0;
}
int gecco_gaussian_act_f32(const float* x, const float* alpha, float* y, size_t n, int normalized, void* stream) {
    TRY(gaussian_act_launch(x, alpha, y, n, normalized, (hipStream_t)stream), "gaussian_act");
    return 0;
}
int gecco_sampler_add_noise_f64(const double* x_cur, const float* noise, size_t noise_step_stride, const double* sched,
                                const int* step, int col, int sigma_col, double* x_out, float* x_in, float* sigma,
                                size_t n, int B, void* stream) {
    TRY(sampler_add_noise_f64_launch(x_cur, noise, noise_step_stride, sched, step, col, sigma_col, x_out, x_in, sigma,
                                     n, B, (hipStream_t)stream), "sampler_add_noise_f64");
    return 0;
}
int gecco_sampler_add_noise_f32(const float* x, const float* noise, size_t noise_step_stride, const double* sched,
                                const int* step, int col, float* out, float* sigma, size_t n, int B, void* stream) {
    TRY(sampler_add_noise_f32_launch(x, noise, noise_step_stride, sched, step, col, out, sigma, n, B,
                                     (hipStream_t)stream), "sampler_add_noise_f32");
    return 0;
}
int gecco_sampler_euler_f64(const double* x_hat, const float* den, const double* sched, const int* step, double* d_cur,
                            double* x_next, float* x_in, float* sigma, size_t n, int B, void* stream) {
    TRY(sampler_euler_launch(x_hat, den, sched, step, d_cur, x_next, x_in, sigma, n, B, (hipStream_t)stream),
        "sampler_euler");
    return 0;
}
int gecco_sampler_heun_f64(const double* x_hat, const double* x_next, const float* den, const double* d_cur,
                           const double* sched, const int* step, double* x_out, size_t n, void* stream) {
    TRY(sampler_heun_launch(x_hat, x_next, den, d_cur, sched, step, x_out, n, (hipStream_t)stream), "sampler_heun");
    return 0;
}
int gecco_sampler_advance(int* step, int delta, void* stream) {
    TRY(sampler_advance_launch(step, delta, (hipStream_t)stream), "sampler_advance");
    return 0;
}
int gecco_sampler_scale_f64(const float* latents, double t0, double* x, size_t n, void* stream) {
    TRY(sampler_scale_launch(latents, t0, x, n, (hipStream_t)stream), "sampler_scale");
    return 0;
}

// ---------------------------------------------------------------------------- training path
int gecco_gemm_f32(const GeccoGemm* g, void* stream) {
    if (!g || !g->A || !g->B || !g->C) return fail(-1, "gemm: null argument");
    GemmGeneralArgs a;
    a.A = g->A; a.B = g->B; a.bias = g->bias; a.C = g->C; a.Z = g->Z; a.zdiv = g->zdiv > 0 ? g->zdiv : 1;
    a.M = g->M; a.N = g->N; a.K = g->K; a.lda = g->lda; a.ldb = g->ldb; a.ldc = g->ldc;
    a.sA1 = g->sA1; a.sA2 = g->sA2; a.sB1 = g->sB1; a.sB2 = g->sB2; a.sC1 = g->sC1; a.sC2 = g->sC2;
    a.a_kmajor = g->a_kmajor; a.b_kmajor = g->b_kmajor; a.scale = g->scale;
    TRY(gemm_general_launch(a, (hipStream_t)stream), "gemm");
    return 0;
}
int gecco_reduce_batch_f32(const float* parts, float* out, size_t n, int Z, size_t stride, int accumulate, void* stream) {
    TRY(reduce_batch_launch(parts, out, n, Z, stride, accumulate, (hipStream_t)stream), "reduce_batch");
    return 0;
}
int gecco_softmax_fwd_f32(const float* S, float* P, size_t rows, int n, float scale, void* stream) {
    TRY(softmax_fwd_launch(S, P, rows, n, scale, (hipStream_t)stream), "softmax_fwd");
    return 0;
}
int gecco_softmax_bwd_f32(const float* P, const float* dP, float* dS, size_t rows, int n, float scale, void* stream) {
    TRY(softmax_bwd_launch(P, dP, dS, rows, n, scale, (hipStream_t)stream), "softmax_bwd");
    return 0;
}
int gecco_gauss_act_bwd_blocks(size_t n) { return gauss_act_bwd_blocks(n); }
int gecco_gauss_act_bwd_f32(const float* u, const float* dy, const float* alpha, float* du, float* partial, size_t n,
                            int normalized, void* stream) {
    TRY(gauss_act_bwd_launch(u, dy, alpha, du, partial, n, normalized, (hipStream_t)stream), "gauss_act_bwd");
    return 0;
}
int gecco_col_dot_stats_f32(const float* dy, const float* x, float* gstats, int B, int rows, int C, void* stream) {
    TRY(col_dot_stats_launch(dy, x, gstats, B, rows, C, (hipStream_t)stream), "col_dot_stats");
    return 0;
}
int gecco_adagn_bwd_coeffs_f32(const float* xstats, int Tx, const float* gstats, int Tg, int rows, const float* t,
                               int ctx_dim, const GeccoAdaGN* p, float* cA, float* cB, float* cC, float* ds, float* dz,
                               int B, int C, int G, float eps, void* stream) {
    TRY(adagn_bwd_coeffs_launch(xstats, Tx, gstats, Tg, rows, t, ctx_dim, p ? p->scale_w : nullptr,
                                p ? p->scale_b : nullptr, cA, cB, cC, ds, dz, B, C, G, eps, (hipStream_t)stream),
        "adagn_bwd_coeffs");
    return 0;
}
int gecco_affine2_apply_f32(const float* dy, const float* x, const float* cA, const float* cB, const float* cC,
                            float* dx, int B, int rows, int C, void* stream) {
    TRY(affine2_apply_launch(dy, x, cA, cB, cC, dx, B, rows, C, (hipStream_t)stream), "affine2_apply");
    return 0;
}
int gecco_affine2_apply_add_f32(const float* dy, const float* x, const float* cA, const float* cB, const float* cC,
                                const float* add, float* dx, int B, int rows, int C, void* stream) {
    TRY(affine2_apply_launch(dy, x, cA, cB, cC, dx, B, rows, C, (hipStream_t)stream, add), "affine2_apply_add");
    return 0;
}
int gecco_adagn_param_grads_f32(const float* ds, const float* dz, const float* t, int B, int C, int ctx_dim,
                                float* d_scale_w, float* d_scale_b, float* d_bias_w, float* d_bias_b, void* stream) {
    TRY(adagn_param_grads_launch(ds, dz, t, B, C, ctx_dim, d_scale_w, d_scale_b, d_bias_w, d_bias_b,
                                 (hipStream_t)stream), "adagn_param_grads");
    return 0;
}
int gecco_lift_bwd_f32(const float* dY, const float* xin, float* partial, int B, int N, int C, void* stream) {
    TRY(lift_bwd_launch(dY, xin, partial, B, N, C, (hipStream_t)stream), "lift_bwd");
    return 0;
}
int gecco_lower_bwd_blocks(size_t rows) { return lower_bwd_blocks(rows); }
int gecco_lower_bwd_f32(const float* feat, const float* dF, const float* W, float* dfeat, float* partial, size_t rows,
                        int C, float eps, void* stream) {
    TRY(lower_bwd_launch(feat, dF, W, dfeat, partial, rows, C, eps, (hipStream_t)stream), "lower_bwd");
    return 0;
}

// ---------------------------------------------------------------------------- optimizer
int gecco_adam_ema_step_f32(const GeccoAdamEma* a, void* stream) {
    return gecco_adam_ema_step_amp_f32(a, nullptr, nullptr, nullptr, stream);
}

int gecco_adam_ema_step_amp_f32(const GeccoAdamEma* a, const float* amp_scale, const float* found_inf, int* skipped, void* stream) {
    if (!a || !a->p || !a->g || !a->m || !a->v) return fail(-1, "adam_ema: null argument");
    if ((amp_scale || skipped) && !found_inf) return fail(-1, "adam_ema: amp_scale / skipped come with found_inf (the GradScaler protocol)");
    if (a->do_ema && !a->ema) return fail(-1, "adam_ema: do_ema needs the ema buffer");
    if (a->n % 4) return fail(-2, "adam_ema: n must be a multiple of 4 (pad the flat buffers)");
    if (a->step < 1) return fail(-2, "adam_ema: step is 1-based");
    const uintptr_t al = (uintptr_t)a->p | (uintptr_t)a->g | (uintptr_t)a->m | (uintptr_t)a->v | (uintptr_t)a->ema;
    if (al & 15) return fail(-2, "adam_ema: buffers must be 16-byte aligned");
    AdamEmaArgs k{};
    k.p = a->p; k.g = a->g; k.m = a->m; k.v = a->v; k.ema = a->ema; k.n = a->n;
    // every derived scalar in double first, like torch's Python scalars (torch/optim/adam.py _single_tensor_adam:
    // 1 - beta1, 1 - beta2, bias corrections, lr / bc1; ema.py:187-194: 1 - decay), then one rounding to fp32
    k.beta2 = (float)a->beta2; k.eps = (float)a->eps; k.weight_decay = (float)a->weight_decay;
    k.w1 = (float)(1.0 - a->beta1); k.w2 = (float)(1.0 - a->beta2);
    const double bc1 = 1.0 - pow(a->beta1, (double)a->step), bc2 = 1.0 - pow(a->beta2, (double)a->step);
    k.step_size = (float)(a->lr / bc1);
    k.bc2_sqrt = (float)sqrt(bc2);
    k.grad_scale = a->grad_scale; k.ema_decay = (float)a->ema_decay; k.ema_w = (float)(1.0 - a->ema_decay); k.do_ema = a->do_ema;
    k.amp_scale = amp_scale; k.found_inf = found_inf; k.skipped = skipped;
    k.lr = a->lr; k.beta1 = a->beta1; k.beta2d = a->beta2; k.step = a->step;
    TRY(adam_ema_launch(k, (hipStream_t)stream), "adam_ema");
    return 0;
}
int gecco_ema_update_f32(const float* p, float* ema, size_t n, double decay, void* stream) {
    if (!p || !ema) return fail(-1, "ema_update: null argument");
    if ((n % 4) || (((uintptr_t)p | (uintptr_t)ema) & 15)) return fail(-2, "ema_update: n %% 4 == 0 and 16-byte aligned buffers");
    TRY(ema_update_launch(p, ema, n, decay, (hipStream_t)stream), "ema_update");
    return 0;
}

// ---------------------------------------------------------------------------- samplers / metrics of the "next" rows
int gecco_sampler_refresh_known_f64(double* x, const float* known, const float* noise, const double* sched, const int* step,
                                    int col, int m, int n_known, int B, void* stream) {
    if (!x || !known || !noise || !sched || !step) return fail(-1, "sampler_refresh_known: null argument");
    TRY(sampler_refresh_known_launch(x, known, noise, sched, step, col, m, n_known, B, (hipStream_t)stream), "sampler_refresh_known");
    return 0;
}
int gecco_distance_matrix_f32(const float* a, const float* b, float* D, int B, int N, int M, int squared, void* stream) {
    if (!a || !b || !D) return fail(-1, "distance_matrix: null argument");
    TRY(dist_matrix_launch(a, b, D, B, N, M, squared, (hipStream_t)stream), "distance_matrix");
    return 0;
}
int gecco_set_chamfer_f32(const float* a, const float* b, float* out, int S, int T, int N, int M, int squared, void* stream) {
    if (!a || !b || !out) return fail(-1, "set_chamfer: null argument");
    if (S <= 0 || T <= 0 || N <= 0 || M <= 0) return fail(-2, "set_chamfer: empty set or cloud");
    hipStream_t s = (hipStream_t)stream;
    TRY(set_nearest_mean_launch(a, b, out, S, T, N, M, squared, T, 1, 0.5f, 0, s), "set_chamfer(a -> b)");
    TRY(set_nearest_mean_launch(b, a, out, T, S, M, N, squared, 1, T, 0.5f, 1, s), "set_chamfer(b -> a)");
    return 0;
}

int gecco_set_metrics_f32(const float* ss, const float* sd, const float* dd, int n, float* out3, int* flags, void* stream) {
    if (!ss || !sd || !dd || !out3 || !flags) return fail(-1, "set_metrics: null argument");
    if (n <= 0) return fail(-2, "set_metrics: empty set");
    TRY(set_metrics_launch(ss, sd, dd, n, out3, flags, (hipStream_t)stream), "set_metrics");
    return 0;
}

int gecco_chamfer_f32(const float* a, const float* b, float* out, float* ws, int B, int N, int M, int squared, void* stream) {
    if (!a || !b || !out || !ws) return fail(-1, "chamfer: null argument");
    hipStream_t s = (hipStream_t)stream;
    float* min_ab = ws;                       // (B, N): nearest b of every a
    float* min_ba = ws + (size_t)B * N;       // (B, M): nearest a of every b
    TRY(nearest_dist_launch(a, b, min_ab, B, N, M, squared, s), "chamfer(a -> b)");
    TRY(nearest_dist_launch(b, a, min_ba, B, M, N, squared, s), "chamfer(b -> a)");
    TRY(row_mean_launch(min_ab, out, B, N, 0.5f, 0, s), "chamfer(mean a)");
    TRY(row_mean_launch(min_ba, out, B, M, 0.5f, 1, s), "chamfer(mean b)");
    return 0;
}
int gecco_sinkhorn_f32(const float* C, float* f, float* g, float* rowcost, float* out, int B, int N, int M, float epsilon,
                       int iterations, void* stream) {
    if (!C || !f || !g || !rowcost || !out) return fail(-1, "sinkhorn: null argument");
    if (epsilon <= 0.f || iterations < 1) return fail(-2, "sinkhorn: epsilon > 0 and iterations >= 1");
    hipStream_t s = (hipStream_t)stream;
    TRY((int)hipMemsetAsync(g, 0, (size_t)B * M * sizeof(float), s), "sinkhorn(g = 0)");
    for (int it = 0; it < iterations; ++it) TRY(sinkhorn_step_launch(C, f, g, B, N, M, epsilon, s), "sinkhorn(step)");
    TRY(sinkhorn_cost_launch(C, f, g, rowcost, out, B, N, M, epsilon, s), "sinkhorn(cost)");
    return 0;
}

// ---------------------------------------------------------------------------- ConvNeXt conditioner (channels-last)
int gecco_convnext_stem_f32(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b, float* out,
                            int B, int H, int W, int C, float eps, void* stream) {
    if (!x || !w || !bias || !ln_w || !ln_b || !out) return fail(-1, "convnext_stem: null argument");
    int rc = cnx_stem_launch(x, w, bias, ln_w, ln_b, out, nullptr, B, H, W, C, eps, (hipStream_t)stream);
    if (rc == -9) return fail(-2, "convnext_stem: needs C == 96 and H, W multiples of 4");
    TRY(rc, "convnext_stem");
    return 0;
}
int gecco_convnext_dwconv_ln_f32(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b, float* out,
                                 int B, int H, int W, int C, float eps, void* stream) {
    if (!x || !w || !bias || !ln_w || !ln_b || !out) return fail(-1, "convnext_dwconv_ln: null argument");
    int rc = cnx_dwconv_ln_launch(x, w, bias, ln_w, ln_b, out, nullptr, B, H, W, C, eps, (hipStream_t)stream);
    if (rc == -9) return fail(-2, "convnext_dwconv_ln: C must be 96, 192 or 384");
    TRY(rc, "convnext_dwconv_ln");
    return 0;
}
int gecco_convnext_ln_patch2_f32(const float* x, const float* ln_w, const float* ln_b, float* out, int B, int H, int W, int C,
                                 float eps, void* stream) {
    if (!x || !ln_w || !ln_b || !out) return fail(-1, "convnext_ln_patch2: null argument");
    int rc = cnx_ln_patch2_launch(x, ln_w, ln_b, out, B, H, W, C, eps, (hipStream_t)stream);
    if (rc == -9) return fail(-2, "convnext_ln_patch2: C must be 96, 192 or 384 and H, W even");
    TRY(rc, "convnext_ln_patch2");
    return 0;
}
int gecco_convnext_fold_scale_f32(const float* W, const float* b, const float* s, float* Wo, float* bo, int N, int K, void* stream) {
    if (!W || !b || !s || !Wo || !bo) return fail(-1, "convnext_fold_scale: null argument");
    TRY(cnx_fold_scale_launch(W, b, s, Wo, bo, N, K, (hipStream_t)stream), "convnext_fold_scale");
    return 0;
}

// ---- the conditioner's training path
int gecco_convnext_stem_train_f32(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b, float* out,
                                  float* z, int B, int H, int W, int C, float eps, void* stream) {
    if (!x || !w || !bias || !ln_w || !ln_b || !out || !z) return fail(-1, "convnext_stem_train: null argument");
    int rc = cnx_stem_launch(x, w, bias, ln_w, ln_b, out, z, B, H, W, C, eps, (hipStream_t)stream);
    if (rc == -9) return fail(-2, "convnext_stem_train: needs C == 96 and H, W multiples of 4");
    TRY(rc, "convnext_stem_train");
    return 0;
}
int gecco_convnext_dwconv_ln_train_f32(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b,
                                       float* out, float* z, int B, int H, int W, int C, float eps, void* stream) {
    if (!x || !w || !bias || !ln_w || !ln_b || !out || !z) return fail(-1, "convnext_dwconv_ln_train: null argument");
    int rc = cnx_dwconv_ln_launch(x, w, bias, ln_w, ln_b, out, z, B, H, W, C, eps, (hipStream_t)stream);
    if (rc == -9) return fail(-2, "convnext_dwconv_ln_train: C must be 96, 192 or 384");
    TRY(rc, "convnext_dwconv_ln_train");
    return 0;
}
int gecco_convnext_dwconv_f32(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int C, void* stream) {
    if (!x || !w || !out) return fail(-1, "convnext_dwconv: null argument");
    int rc = cnx_dwconv_ln_launch(x, w, bias, nullptr, nullptr, out, nullptr, B, H, W, C, 0.f, (hipStream_t)stream);
    if (rc == -9) return fail(-2, "convnext_dwconv: C must be 96, 192 or 384");
    TRY(rc, "convnext_dwconv");
    return 0;
}
int gecco_convnext_dwconv_bwd_f32(const float* dz, const float* w, const float* add, float* dx, int B, int H, int W, int C, void* stream) {
    if (!dz || !w || !dx) return fail(-1, "convnext_dwconv_bwd: null argument");
    int rc = cnx_dwconv_ln_launch(dz, w, nullptr, nullptr, nullptr, dx, nullptr, B, H, W, C, 0.f, (hipStream_t)stream, add, 1);
    if (rc == -9) return fail(-2, "convnext_dwconv_bwd: C must be 96, 192 or 384");
    TRY(rc, "convnext_dwconv_bwd");
    return 0;
}
int gecco_convnext_fold_scale_bwd_f32(const float* dWp, const float* dbp, const float* W, const float* b, const float* s, float* dW,
                                      float* db, float* ds, int N, int K, void* stream) {
    if (!dWp || !dbp || !W || !b || !s || !dW || !db || !ds) return fail(-1, "convnext_fold_scale_bwd: null argument");
    TRY(cnx_fold_scale_bwd_launch(dWp, dbp, W, b, s, dW, db, ds, N, K, (hipStream_t)stream), "convnext_fold_scale_bwd");
    return 0;
}
int gecco_convnext_ln_bwd_blocks(int B, int H, int W, int C) { return cnx_ln_bwd_blocks(B, H, W, C); }
int gecco_convnext_ln_bwd_f32(const float* z, const float* dy, const float* ln_w, float* dz, float* parts, int B, int H, int W, int C,
                              float eps, int patch2, void* stream) {
    if (!z || !dy || !ln_w || !dz || !parts) return fail(-1, "convnext_ln_bwd: null argument");
    int rc = cnx_ln_bwd_launch(z, dy, ln_w, dz, parts, B, H, W, C, eps, patch2, (hipStream_t)stream);
    if (rc == -9) return fail(-2, "convnext_ln_bwd: C must be 96, 192 or 384 (and H, W even for the patch layout)");
    TRY(rc, "convnext_ln_bwd");
    return 0;
}
int gecco_convnext_dwconv_dw_blocks(int B, int H, int W, int C) { return cnx_dwconv_dw_blocks(B, H, W, C); }
int gecco_convnext_dwconv_dw_f32(const float* x, const float* dz, float* parts, int B, int H, int W, int C, void* stream) {
    if (!x || !dz || !parts) return fail(-1, "convnext_dwconv_dw: null argument");
    int rc = cnx_dwconv_dw_launch(x, dz, parts, B, H, W, C, (hipStream_t)stream);
    if (rc == -9) return fail(-2, "convnext_dwconv_dw: C must be 96, 192 or 384");
    TRY(rc, "convnext_dwconv_dw");
    return 0;
}
int gecco_gelu_f32(const float* u, float* y, size_t n, void* stream) {
    if (!u || !y) return fail(-1, "gelu: null argument");
    int rc = gelu_launch(u, y, n, (hipStream_t)stream);
    if (rc == -9) return fail(-2, "gelu: n must be a multiple of 4");
    TRY(rc, "gelu");
    return 0;
}
int gecco_gelu_bwd_f32(const float* u, const float* dy, float* du, size_t n, void* stream) {
    if (!u || !dy || !du) return fail(-1, "gelu_bwd: null argument");
    int rc = gelu_bwd_launch(u, dy, du, n, (hipStream_t)stream);
    if (rc == -9) return fail(-2, "gelu_bwd: n must be a multiple of 4");
    TRY(rc, "gelu_bwd");
    return 0;
}
int gecco_convnext_im2col4_f32(const float* x, float* out, int B, int H, int W, void* stream) {
    if (!x || !out) return fail(-1, "convnext_im2col4: null argument");
    int rc = cnx_im2col4_launch(x, out, B, H, W, (hipStream_t)stream);
    if (rc == -9) return fail(-2, "convnext_im2col4: H, W must be multiples of 4");
    TRY(rc, "convnext_im2col4");
    return 0;
}

}  // extern "C"
