/* gecco_hip.h — C ABI of libgecco_hip.so: the MI355X (gfx950) implementation of the GECCO
 * denoiser hot path.
 *
 * The reference (cvlab-epfl/gecco, gecco-torch) has no FFI/plugin layer: its extension surface
 * is "Python that constructs nn.Modules".  This header is therefore the boundary a maintainer
 * binds (ctypes stub in INTEGRATION.md) underneath the same-named Python modules; every entry
 * point cites the reference code it replaces (paths relative to
 * gecco-torch/src/gecco_torch/).
 *
 * Conventions
 *  - all pointers are raw DEVICE pointers to contiguous fp32 (fp64 where the name says so);
 *    activations are channels-last (B, n, C) row-major, weights are nn.Linear layout (out, in);
 *  - `stream` is a hipStream_t (0 = default stream); calls only enqueue work: no allocation, no
 *    synchronisation, no host reads — they are hipGraph-capture safe;
 *  - scratch memory is caller-provided (`ws`, size from the matching *_workspace_bytes());
 *  - return value 0 = success; negative = argument error (see gecco_last_error()); positive =
 *    hipError_t of the launch.
 */
#ifndef GECCO_HIP_H
#define GECCO_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GECCO_ABI_VERSION 14

int gecco_abi_version(void);
const char* gecco_build_arch(void);   /* "gfx950" */
const char* gecco_last_error(void);   /* thread-local description of the last non-zero return */

/* ---- parameter tables (device pointers into the module's own state-dict tensors) ------------ */
typedef struct GeccoAdaGN {   /* AdaGN, models/normalization.py:14-44 */
    const float* scale_w;     /* scale.weight (C, ctx_dim) */
    const float* scale_b;     /* scale.bias   (C)          */
    const float* bias_w;      /* bias.weight  (C, ctx_dim) */
    const float* bias_b;      /* bias.bias    (C)          */
} GeccoAdaGN;

typedef struct GeccoMLP {     /* MLP(depth=1) + GaussianActivation, models/mlp.py:5-39, activation.py */
    const float* w0;          /* 0.weight (width, C) */
    const float* b0;          /* 0.bias   (width)    */
    const float* alpha;       /* 1.alpha  ()         */
    const float* w2;          /* 2.weight (C, width) */
    const float* b2;          /* 2.bias   (C)        */
} GeccoMLP;

typedef struct GeccoLayer {   /* BroadcastingLayer, models/set_transformer.py:120-168 */
    GeccoAdaGN broadcast_norm;
    const float* inducers;    /* broadcast.pool.inducers (1, H, I, hd)   */
    const float* kv_proj_w;   /* broadcast.pool.kv_proj.weight (2C, C)   */
    const float* pool_out_w;  /* broadcast.pool.out_proj.weight (C, C)   */
    GeccoAdaGN norm_1;
    GeccoMLP bmlp;            /* broadcast.mlp */
    GeccoAdaGN norm_2;
    const float* in_proj_w;   /* broadcast.unpool.in_proj_weight (3C, C) */
    const float* in_proj_b;   /* broadcast.unpool.in_proj_bias (3C)      */
    const float* unpool_out_w;/* broadcast.unpool.out_proj.weight (C, C) */
    const float* unpool_out_b;/* broadcast.unpool.out_proj.bias (C)      */
    GeccoAdaGN mlp_norm;
    GeccoMLP mlp;
} GeccoLayer;

typedef struct GeccoSetTransformer {  /* SetTransformer, models/set_transformer.py:171-216 */
    int n_layers, C, H, I, ctx_dim, G, width, act;  /* act of the MLPs: 0 identity, 1 GaussianActivation(normalized), 2 raw, 3 nn.ReLU */
    int precision;  /* arithmetic of the linears and attention products: 0 exact fp32 MFMA (~1e-6 vs the fp32
                     * reference), 1 split-bf16 on bf16 MFMA, fp32 accumulate (a = hi + lo; 3 MFMAs; ~2e-5),
                     * 2 fp16 operands (round to nearest even), fp32 accumulate, fp16-stored intermediates (D ~4e-4, F_x ~1e-3),
                     * 3 "mixed": kv_proj | q_proj with fp16 activations and two-term fp16 weights, fp16 K | V / q and attention
                     *   products; every product that feeds the residual stream or the shared inducer states (the 64-inducer
                     *   chain, unpool.out_proj, the point MLP) in split-bf16 (~6e-5 on both outputs); point counts that are
                     *   not a multiple of 128 (and widths / head dims outside the fp16 kernels' set) run as mode 1,
                     * 4 "w2": mode 3 with the point MLP of every layer as ONE launch whose hidden layer stays in registers as fp16
                     *   (gecco_mlp_fused_w: two-term weights and AdaGN(x), one-term hidden layer): F_x ~3.5e-4, inside a 5e-4 bar;
                     *   feature_dim 128, 256, 384 or 512 with point counts in multiples of 128, anything else runs as mode 3 */
    int images_ready;  /* 0: every forward rebuilds the weight images it streams (weights may change between calls; the default).
                        * 1: the caller guarantees that the workspace of this call still holds the images the previous forward with this
                        *    table, the same (B, N), the same options and the same cached / uncached form built in it, and that no weight
                        *    changed since: the forward skips the image launches (a sampler's 2nd .. 255th evaluation) */
    unsigned opt_mask, opt_vals;  /* this plan's own path switches (ABI 14): bit gecco_option_index(name) of opt_mask set = the option is
                                   * pinned to the same bit of opt_vals for every forward with this table, whatever gecco_set_option /
                                   * the environment say; clear = follow the process-wide default.  The reference's modules are
                                   * self-contained (models/set_transformer.py:176-216): so are these tables. */
    const GeccoLayer* layers;                       /* HOST array of n_layers tables */
} GeccoSetTransformer;

/* ---- unit operators --------------------------------------------------------------------- */

/* C = residual + act((A*pro_a + pro_o) @ W^T + bias); optional GroupNorm partial statistics of C.
 * Replaces nn.Linear call sites of models/set_transformer.py:49,65,112,165-166 and models/mlp.py.
 * stats: (B, T, 2, Nout) with T = gecco_linear_row_tiles(rows).  Any of bias/pro/alpha/residual/stats
 * may be NULL.  act: 0 none, 1 / 2 GaussianActivation normalized / raw (alpha required), 3 ReLU (the reference's
 * default activation, models/mlp.py:12), 4 GELU in its exact erf form (the conditioner's CNBlocks). */
int gecco_linear_f32(const float* A, const float* W, const float* bias, const float* pro_a, const float* pro_o,
                     const float* alpha, const float* residual, float* C, float* stats, int B, int rows, int K,
                     int Nout, int act, void* stream);
/* Path switches for A/B measurements and tests (the results agree to rounding):
 *   "astat" (default 1): fp16 mode runs AdaGN + kv|q and AdaGN + mlp.0 as one A-stationary pass over x;
 *   "chain" (default 1): fp16 mode runs the 64-inducer chain of a layer (pool merge .. unpool k|v) as one launch;
 *   "headmajor" (default 1): fp16 mode stores K | V and q head-major (needs "astat");
 *   "mlpfused" (default 1): fp16 mode runs AdaGN + mlp.0 + activation + mlp.2 + residual + statistics as one launch;
 *   "unpoolfused" (default 1): fp16 mode runs unpool attention + out_proj + residual + statistics as one launch
 *   (needs "headmajor").
 *   "lo8" (default 1): mixed mode streams the second term of the V projection's two-term weights as fp8 (e4m3, x 2^19) on
 *   v_mfma_scale_f32_32x32x64_f8f6f4 (d = 256, 384) instead of as fp16: same result to ~1e-6 of the output.
 *   "actimg" (default 1): split-bf16 / mixed modes hand the MLP hidden layer and the unpool attention output to the next
 *   GEMM as tiled split images (bf16 hi | lo planes in 8 KiB blocks, same bytes as the fp32 tensor) which that GEMM loads
 *   global -> registers (gemm_x3_areg.hip): same bits as the fp32 hand-over.
 *   "h8" (default 1): mixed mode runs mlp.0 as an fp16 main product plus two fp8 cross terms (v_mfma_scale_f32_32x32x64_f8f6f4)
 *   on the A-stationary kernel (gemm_h8_astat.hip: 128-row blocks; needs "actimg", rows % 128 == 0, feature_dim 128 / 256 / 384, or 512 at one block per CU) instead of as a
 *   split-bf16 product: 2 instead of 3 matrix-pipe units per product, same accuracy (~6e-5 on F_x).
 *   "h8areg" (default 1): mixed mode hands the MLP hidden layer and the unpool attention output on as h8 activation images
 *   (fp16 hi + fp8 lo, 3 bytes per element) and runs mlp.2 / out_proj as h8 products (gecco_linear_h8_areg_f32) instead of
 *   split-bf16 products on tiled split images: 2 instead of 3 matrix-pipe units per product, same accuracy (needs "h8").
 *   "unpoolh8" (default 1): mixed mode runs unpool attention + out_proj (h8) + residual + statistics as one launch
 *   (gecco_unpool_outproj_h8) instead of the attention writing an h8 activation image for gecco_linear_h8_areg_f32; needs
 *   "kvq64" (head-major fp16 q), "h8areg", feature_dim in {128, 256, 384, 512} with 8 heads.
 *   "mlpw" (default 1): the "w2" mode (precision 4) runs the point MLP of a layer (AdaGN apply, mlp.0, activation, mlp.2, residual,
 *   statistics) as one launch (gecco_mlp_fused_w); 0: as the mixed mode does (gecco_linear_h8_img_f32 + gecco_linear_h8_areg_f32).
 *   "mlpwshare" (default 0): gecco_mlp_fused_w launches (a block fills its CU) take three quarters of the CUs instead of all of them, so
 *   that another stream's kernels run beside them: set around an evaluation issued as two half batches on two streams.
 *   "h6" (default 1): mixed mode computes mlp.0's two cross terms as fp6 (e2m3) x fp6 with one E8M0 scale per lane and 64-k group (the
 *   scale blocks of v_mfma_scale_f32_32x32x64_f8f6f4) instead of fp8 with fixed scales: half their matrix cycles, the same accuracy;
 *   gecco_linear_h8_img_f32 with image_kind 2 follows it too.
 *   "chaincl" (default 1): the one-launch inducer chain ("chain" / "chain2") runs as a cluster of feature_dim / 128 blocks per sample that
 *   hand each other their column tiles through L2 (bit-identical to one block per sample; feature_dim >= 256); with it the mixed mode
 *   runs the one-launch chain at feature_dim 512 too.
 *   "kvfold" (default 1): mixed mode with "unpoolh8": the chain's last epilogue writes the fp16 k | v image gecco_unpool_outproj_h8
 *   streams instead of fp32 k | v for a reformatting launch (same bits).
 *   "chain2" (default 1): mixed mode runs the 64-inducer chain of a layer (pool merge .. unpool k|v) as the ONE launch of the fp16
 *   mode with two-term fp16 weights (hi | lo blocks per column tile) instead of five 64-row split-bf16 GEMMs + their AdaGN
 *   coefficient launches: 16 -> 9 launches per layer; F_x 6e-5 -> ~1e-4 (its activations are rounded to fp16 once).
 *   "kvq64" (default 1): mixed mode runs kv_proj | q_proj on the 64-column-tile A-stationary kernel (gecco_linear_kvq_f16; needs
 *   "headmajor") instead of the 128-column-tile one (gecco_linear_astat_f16 + lo image): same arithmetic, also at feature_dim 512.
 *   "kvqperm" (default 1): mixed mode at head dim 48 (feature_dim 384): the kvq stream deals the columns of K, V and q to the 64-column tiles
 *   head-aligned (a tile = one head + a third of another) and gecco_linear_kvq_f16's epilogue writes a head's (32 rows, 48) slab as three
 *   contiguous 1 KiB stores through LDS instead of as 32-byte pieces: the same bits, fewer partial cache lines.
 *   "imgproj16" (default 0: opt-in): "w2" mode of gecco_ray_network_fwd_f32: the lookup leaves halves, GroupNorm(16)'s apply is folded into
 *   per-sample fp16 images of img_feature_proj's weight (and its offsets into the bias), and the product fp16(lookup) x fp16(W a) runs one
 *   term each on the fp16-operand streaming kernel instead of split-bf16: 280 -> 154 + 24 us at C3 (2 % of an evaluation) for F_x 2.3e-4 ->
 *   2.7e-4 at C3 and up to 3.2e-4 -> 4.7e-4 on small pyramids (the mode's bar is 5e-4) — off unless asked for.
 * value < 0 returns the option to its default / environment (GECCO_ASTAT, GECCO_CHAIN, GECCO_HEADMAJOR, GECCO_MLPFUSED,
 * GECCO_UNPOOLFUSED, GECCO_LO8, GECCO_ACTIMG, GECCO_H8, GECCO_KVQ64, GECCO_H8AREG, GECCO_CHAIN2).
 * Process-wide DEFAULT: a network forward consults its own table first (GeccoSetTransformer.opt_mask / opt_vals, ABI 14), so two
 * plans — or two host threads — with different settings never alias; the unit operators below follow the process-wide value. */
int gecco_set_option(const char* name, int value);
/* Bit position of an option in GeccoSetTransformer.opt_mask / opt_vals (-1: unknown name). */
int gecco_option_index(const char* name);

int gecco_linear_row_tiles(int rows);
/* The same with the arithmetic selectable: precision 0 = exact fp32 MFMA, 1 = split-bf16, 2 = fp16 (see GeccoSetTransformer);
 * wsplit: scratch of >= ceil(Nout/128)*128*K*4 bytes for the tiled image of W (bf16 hi | lo, or fp16; precision 1 / 2). */
int gecco_linear_ex_f32(const float* A, const float* W, const float* bias, const float* pro_a, const float* pro_o,
                        const float* alpha, const float* residual, float* C, float* stats, int B, int rows, int K,
                        int Nout, int act, int precision, void* wsplit, void* stream);
/* Two linears over the same (optionally AdaGN-modulated) input in one launch, A read once:
 *   C1 (B, rows, Nout1) = A' W1^T + bias1,  C2 (B, rows, Nout2) = A' W2^T + bias2,  A' = A*pro_a + pro_o.
 * The layer uses it for AttentionPool.kv_proj and the q rows of nn.MultiheadAttention.in_proj, which both read
 * broadcast_norm(x) (models/set_transformer.py:49 and :112 under :150-155).  Falls back to two launches when
 * Nout1 % 128 != 0 or rows < 128.  wsplit: >= (ceil(Nout1/128) + ceil(Nout2/128))*128*K*4 bytes (precision 1). */
int gecco_linear_pair_f32(const float* A, const float* W1, const float* bias1, int Nout1, float* C1, const float* W2,
                          const float* bias2, int Nout2, float* C2, const float* pro_a, const float* pro_o, int B,
                          int rows, int K, int precision, void* wsplit, void* stream);
/* Weight images prepared AHEAD of the linears that use them (the training step: one launch per <= 96 weights at its start
 * instead of one per linear call, and the dX products' W^T images straight from W — no transposed copy).  A job writes the
 * tiled split-bf16 image (gecco_split_bf16_image_bytes(Nout, K) bytes) of the (Nout, K) matrix W (row stride ldw), or, with
 * transposed != 0, of W^T where W is (K, ldw >= Nout).  gecco_linear_ex_f32 / gecco_linear_pair_f32 with W == NULL (W1 ==
 * W2 == NULL) and precision 1 take `wsplit` as that READY image (pair: the image of W1, then from the next 128-row tile
 * boundary that of W2) and only launch the GEMM; the shape must satisfy gecco_linear_image_ok(rows, K, Nout, prologue?). */
typedef struct GeccoSplitJob { const float* W; void* img; int Nout, K, ldw, transposed; } GeccoSplitJob;
int gecco_split_bf16_images_f32(const GeccoSplitJob* jobs, int n, void* stream);
size_t gecco_split_bf16_image_bytes(int Nout, int K);
int gecco_linear_image_ok(int rows, int K, int Nout, int with_prologue);
/* The same for precision 2 — fp16 operands, fp32 accumulation: the arithmetic of the reference's own trainer setting
 * (example_configs: precision="16-mixed" = torch.autocast(float16) around training_step, diffusion.py:213-222), which the
 * training path selects when it runs under that autocast.  Images are 2 bytes per weight element; K % 32 == 0. */
int gecco_split_f16_images_f32(const GeccoSplitJob* jobs, int n, void* stream);
size_t gecco_split_f16_image_bytes(int Nout, int K);
int gecco_linear_image_ok_f16(int rows, int K, int Nout, int with_prologue);
/* The dX product of the linear that FOLLOWS an activation, with the activation's backward as its epilogue (training:
 * autograd of models/mlp.py's Linear -> act -> Linear, and of a CNBlock's Linear -> GELU -> Linear):
 *   C = residual + (A W^T) * act'(u),   A = dY (B, rows, K), W (Nout, K) = W2^T (or NULL: its ready image in wsplit),
 *   u (B, rows, Nout) the pre-activation the forward kept; dh = dY W2 is never written.
 * kind: 1 / 2 GaussianActivation normalized / raw, 3 ReLU, 4 GELU.  GaussianActivation: alpha (device scalar), and agrad —
 * gecco_linear_actbwd_tiles(B, rows, Nout) floats, ZEROED by the caller — receives per output tile
 * sum (A W^T) * d act / d alpha (u): d alpha = their sum (gecco_reduce_batch_f32).  Shapes: gecco_linear_actbwd_ok. */
int gecco_linear_actbwd_ok(int rows, int K, int Nout, int precision);
size_t gecco_linear_actbwd_tiles(int B, int rows, int Nout);
int gecco_linear_actbwd_f32(const float* A, const float* W, const float* u, const float* alpha, int kind, const float* residual,
                            float* C, float* agrad, int B, int rows, int K, int Nout, int precision, void* wsplit, void* stream);
/* A dX product that also leaves the column statistics the AdaGN backward of its consumer needs: C = A W^T and
 * stats (B, gecco_linear_row_tiles(rows), 2, Nout) = per row tile {sum_rows C, sum_rows C * dot_x}, dot_x (B, rows, Nout) the tensor
 * that AdaGN normalised — what gecco_col_dot_stats_f32 computes from C and dot_x in a pass of its own (autograd of
 * models/normalization.py:36-44 behind a linear: set_transformer.py:165-166).  Shapes: gecco_linear_actbwd_ok; W == NULL: image ready. */
int gecco_linear_dotstats_f32(const float* A, const float* W, const float* dot_x, float* C, float* stats, int B, int rows, int K, int Nout,
                              int precision, void* wsplit, void* stream);
/* The same from an fp16 A tensor (the du of gecco_linear_astat16_actbwd_h16; the dq of an fp16-tensor layer), fp16 arithmetic:
 * C = A16 W^T + residual (fp32; residual may be NULL: another gradient contribution to the same tensor) + the {sum C, sum C x} partials of
 * the SUM.  rows >= 128, K % 32 == 0.  W == NULL: wsplit holds the ready fp16 image. */
int gecco_linear_dotstats_a16_f32(const void* A16, const float* W, const float* dot_x, const float* residual, float* C, float* stats, int B,
                                  int rows, int K, int Nout, void* wsplit, void* stream);
/* Its forward companion: C = act(A W^T + bias) AND pre_out = A W^T + bias (the u the backward needs) from one epilogue — the
 * training forward of Linear -> act without a separate activation pass.  act 1 / 2 / 3 / 4 as above; W == NULL: image ready. */
int gecco_linear_act_keep_f32(const float* A, const float* W, const float* bias, const float* alpha, int act, float* pre_out,
                              float* C, int B, int rows, int K, int Nout, int precision, void* wsplit, void* stream);
/* the same with the AdaGN apply as the prologue of the product: A' = A * pro_a[b, k] + pro_o[b, k] (K <= 1024) */
int gecco_linear_act_keep_pro_f32(const float* A, const float* W, const float* bias, const float* pro_a, const float* pro_o,
                                  const float* alpha, int act, float* pre_out, float* C, int B, int rows, int K, int Nout,
                                  int precision, void* wsplit, void* stream);
/* the same in fp16 arithmetic (precision 2) with act(u) STORED as fp16 (C16out: (B, rows, Nout) halves) beside the fp32
 * pre-activation: under the reference's autocast(float16) trainer setting the hidden layer of an MLP is read again only by the
 * matrix pipe (mlp.2's forward through gecco_linear_f16io with an fp16 A, its weight gradient through
 * gecco_gemm_tn_f16_b16_f32) — as fp16 either way; storing it so halves its bytes.  pro_a / pro_o may be NULL; W == NULL: the fp16
 * image of W is ready in wsplit.  rows >= 128. */
int gecco_linear_act_keep_h16(const float* A, const float* W, const float* bias, const float* pro_a, const float* pro_o,
                              const float* alpha, int act, float* pre_out, void* C16out, int B, int rows, int K, int Nout, void* wsplit,
                              void* stream);

/* ---- The training FORWARD's AdaGN-prologue products in h8 arithmetic (fp16 main product + two fp8 cross terms: as accurate as split-bf16,
 * two matrix units instead of three) on the A-stationary kernel, fp32 tensors: C1 (| C2) = x' W1^T + bias1 (| x' W2^T + bias2),
 * x' = x * pro_a[b] + pro_o[b] or x — broadcast_norm -> kv_proj | q (models/set_transformer.py:161-162 -> :49, :112); with act != 0
 * (1 / 2 GaussianActivation normalized / raw, 3 ReLU; one weight) pre_out = the pre-activation and C1 = act(pre_out): the first linear of
 * an MLP (models/mlp.py:5-39) with what its backward needs.  rows % 128 == 0, K in {128, 256, 384}, Nout % 64 == 0, Nout >= 128
 * (gecco_linear_h8_train_ok).  wsplit: gecco_h8_image_bytes per weight (4 bytes per element); NULL weights: the streams are ready
 * (gecco_h8_images_f32).  The backward products keep split-bf16: unscaled gradients do not fit fp16 / fp8 operands. */
size_t gecco_h8_image_bytes(int Nout, int K);
int gecco_linear_h8_train_ok(int rows, int K, int Nout);
int gecco_h8_images_f32(const GeccoSplitJob* jobs, int n, void* stream);
int gecco_linear_h8_train_f32(const float* x, const float* pro_a, const float* pro_o, const float* W1, const float* bias1, int Nout1, float* C1,
                              const float* W2, const float* bias2, int Nout2, float* C2, const float* alpha, int act, float* pre_out, int B,
                              int rows, int K, void* wsplit, void* stream);

/* ---- A-stationary forms of the training path's products out of a <= 512-wide operand, in the autocast(float16) arithmetic (fp16
 * operands, fp32 accumulation, fp32 tensors): a 128-row block keeps its rows of x (after the optional AdaGN apply) in registers for the
 * whole launch and streams W once — the LDS-DMA GEMM re-fetches the rows for every 128-column tile and is bound by that fill, not by
 * the matrix pipe, once a product is one MFMA.  rows % 128 == 0, K in {128, 256, 384, 512}, Nout % 64 == 0,
 * Nout >= 128 (gecco_linear_astat16_ok).  wsplit: gecco_astat16_image_bytes(Nout, K) bytes per weight (2 per element); a NULL weight
 * means wsplit holds the stream already (gecco_astat16_images_f32: batched, transposed != 0 = the stream of W^T from W (K, ldw)).
 *   _f32:    C1 (| C2) = x' W1^T + bias1 (| x' W2^T + bias2), x' = x * pro_a[b] + pro_o[b] or x — broadcast_norm -> kv_proj | q
 *            (models/set_transformer.py:161-162 -> :49, :112); transposed != 0 (one weight): W1 is (K, Nout1) and C1 = x W1 — a dX product;
 *            residual: pass NULL (a form that adds another gradient contribution to C1 exists only in -DGECCO_EXPERIMENTAL builds: it
 *            bought nothing inside the training step and is not part of the shipped surface);
 *   _keep:   pre_out = u = x' W^T + bias (fp32) and C16out = fp16(act(u)) — the first linear of an MLP (models/mlp.py:5-39), act 1 / 2
 *            GaussianActivation (normalized / raw), 3 ReLU;
 *   _actbwd: C = (dy W) * act'(u), W the linear's own (K, Nout) weight, + for GaussianActivation agrad[B * rows / 128] = per-block
 *            partials of d alpha (their sum is the gradient) — the dX product through the activation. */
size_t gecco_astat16_image_bytes(int Nout, int K);
int gecco_linear_astat16_ok(int rows, int K, int Nout);
int gecco_astat16_images_f32(const GeccoSplitJob* jobs, int n, void* stream);
int gecco_linear_astat16_f32(const float* x, const float* pro_a, const float* pro_o, const float* W1, const float* bias1, int Nout1, float* C1,
                             const float* W2, const float* bias2, int Nout2, float* C2, const float* residual, int transposed, int B, int rows,
                             int K, void* wsplit, void* stream);
int gecco_linear_astat16_keep(const float* x, const float* pro_a, const float* pro_o, const float* W, const float* bias, const float* alpha,
                              int act, float* pre_out, void* C16out, int B, int rows, int K, int Nout, void* wsplit, void* stream);
/* ... and with y16 = fp16(x * pro_a + pro_o) stored as well (see gecco_linear_kvq_y16_f16). */
int gecco_linear_astat16_keep_y16(const float* x, const float* pro_a, const float* pro_o, const float* W, const float* bias, const float* alpha,
                                  int act, float* pre_out, void* C16out, void* y16, int B, int rows, int K, int Nout, void* wsplit,
                                  void* stream);
int gecco_linear_astat16_actbwd(const float* dy, const float* W, const float* u, const float* alpha, int kind, float* C, float* agrad, int B,
                                int rows, int K, int Nout, void* wsplit, void* stream);
/* The same with the result stored as HALVES (round 6): du = (dy W) act'(u) of an MLP's backward is read again only by the matrix pipe — the
 * first linear's weight gradient (gecco_gemm_tn_f16_a16_f32) and its dX product (gecco_linear_dotstats_a16_f32) — as an fp16 operand either
 * way; storing it so halves its three crossings of HBM.  The reference's autocast(float16) backward holds this gradient as an fp16
 * tensor too (autograd of models/mlp.py:5-39 under Lightning's precision="16-mixed", diffusion.py:213-222).  kind 1 / 2 / 3. */
int gecco_linear_astat16_actbwd_h16(const float* dy, const float* W, const float* u, const float* alpha, int kind, void* C16, float* agrad,
                                    int B, int rows, int K, int Nout, void* wsplit, void* stream);

/* GroupNorm partial statistics of x (B, rows, C): stats (B, T, 2, C), T = gecco_stats_row_tiles(rows). */
int gecco_col_stats_f32(const float* x, float* stats, int B, int rows, int C, void* stream);
int gecco_stats_row_tiles(int rows);

/* AdaGN folded to y = a*x + o (models/normalization.py:36-44); p == NULL -> plain GroupNorm (models/ray.py:20-30).
 * t: (B, ctx_dim).  a, o: (B, C). */
int gecco_adagn_coeffs_f32(const float* stats, int T, int rows, const float* t, int ctx_dim, const GeccoAdaGN* p,
                           float* a, float* o, int B, int C, int G, float eps, void* stream);
int gecco_affine_apply_f32(const float* x, const float* a, const float* o, float* y, int B, int rows, int C,
                           void* stream);
/* Full AdaGN.forward: y = AdaGN(x, t).  ws >= gecco_adagn_workspace_bytes(B, rows, C). */
int gecco_adagn_f32(const float* x, const float* t, int ctx_dim, const GeccoAdaGN* p, float* y, int B, int rows,
                    int C, int G, float eps, void* ws, size_t ws_bytes, void* stream);
size_t gecco_adagn_workspace_bytes(int B, int rows, int C);

/* ---- fp16-stored intermediates of the fp16 mode (precision 2) ------------------------------------------------
 * In that mode every point-stream tensor whose only consumer rounds it to fp16 anyway (AdaGN(x) as a GEMM operand,
 * K|V, q, the attention output, the MLP hidden layer) is STORED as fp16: the matrix pipe sees the same bits, a
 * third of the layer's HBM bytes never move.  The residual stream and all statistics stay fp32.  These are the unit
 * forms of the launches gecco_set_transformer_fwd_f32 makes; leading dimensions count elements of the tensor's type. */
/* C = residual + act(A @ W^T + bias) with A and / or C an fp16 tensor (a_f16 / c_f16).  fp16 A: no AdaGN prologue
 * (use gecco_affine_cast_f16 first); fp16 C: no residual, no stats.  rows >= 128, K % 32 == 0.  wsplit as above. */
int gecco_linear_f16io(const void* A, const float* W, const float* bias, const float* alpha, const float* residual,
                       void* C, float* stats, int B, int rows, int K, int Nout, int act, int a_f16, int c_f16,
                       void* wsplit, void* stream);
/* gecco_linear_pair_f32 with fp16 A and fp16 outputs (kv_proj | q_proj of the fp16 mode). */
int gecco_linear_pair_f16io(const void* A, const float* W1, const float* bias1, int Nout1, void* C1, const float* W2,
                            const float* bias2, int Nout2, void* C2, int B, int rows, int K, void* wsplit,
                            void* stream);
/* Image-ready calls (gecco_linear_astat_f16, gecco_mlp_fused_f16, gecco_unpool_outproj_f16): a NULL weight pointer
 * (W1 / W0 and W2 / W) means "wsplit still holds the fp16 weight image a previous call of the same function made from
 * the same weights" — the call then launches the kernel alone (what the network entry points do once per forward, and
 * what bench.py times).
 *
The one-pass form the network uses for kv_proj | q_proj and mlp.0 in the fp16 mode: AdaGN apply + fp16 rounding +
 * all output columns in one pass over x (the block keeps fp16(x*pro_a + pro_o) of its 128 rows in registers and walks
 * every 128-column tile of W1 | W2).  C1 (B, rows, Nout1) and C2 (B, rows, Nout2; W2/bias2/C2 may be NULL) are fp16;
 * act as in gecco_linear_f32.  Bit-identical to gecco_affine_cast_f16 + gecco_linear(_pair)_f16io.
 * rows % 128 == 0, Nout1 % 128 == 0, Nout2 % 128 == 0, K in {128, 256, 384, 512}; wsplit as for the pair.
 * head_dim > 0: head-major outputs — C1 is (B, Nout1 / head_dim, rows, head_dim) and C2 (B, Nout2 / head_dim, rows,
 * head_dim), i.e. "b n (g d) -> b g n d" applied to the row-major result: for kv_proj | q_proj that is one contiguous
 * (rows, head_dim) slab per (sample, K or V, head), the unit a pool / unpool attention block streams (the einops
 * rearranges of models/set_transformer.py:51-52,70 done by the store addressing).  head_dim even, >= 8, dividing
 * Nout1 and Nout2.  0: row-major. */
int gecco_linear_astat_f16(const float* x, const float* pro_a, const float* pro_o, const float* W1, const float* bias1,
                           int Nout1, void* C1, const float* W2, const float* bias2, int Nout2, void* C2,
                           const float* alpha, int act, int B, int rows, int K, int head_dim, void* wsplit,
                           void* stream);
/* kv_proj | q_proj of the mixed mode (models/set_transformer.py:49-52 `kv_proj`, :65-70 the q rows of nn.MultiheadAttention's
 * in_proj, both over AdaGN(x), models/normalization.py:36-44) on the 64-column-tile A-stationary kernel:
 *   C1 (B, rows, Nout1) | C2 (B, rows, Nout2) = fp16( fp16(x*pro_a + pro_o) @ fp16(W)^T + bias ),  fp32 accumulate,
 * where the columns [lo_begin, lo_end) of the first segment (the V projection) add the second weight term
 * fp8(y) @ fp8(2^19 (W - fp16(W)))^T on v_mfma_scale_f32_32x32x64_f8f6f4 (two-term weights: their rounding is the part of this
 * product's error that reaches the output).  head_dim > 0: head-major outputs as gecco_linear_astat_f16.
 * rows % 128 == 0, Nout1, Nout2, lo_begin, lo_end % 64 == 0, K in {128, 256, 384, 512}, head_dim % 8 == 0.
 * wsplit: (Nout1 + Nout2) * K * 2 + (lo_end - lo_begin) * K bytes; W1 == NULL: image-ready call.  Option "kvq64". */
int gecco_linear_kvq_f16(const float* x, const float* pro_a, const float* pro_o, const float* W1, const float* bias1, int Nout1,
                         void* C1, const float* W2, const float* bias2, int Nout2, void* C2, int B, int rows, int K, int head_dim,
                         int lo_begin, int lo_end, void* wsplit, void* stream);
/* The same with the operand the kernel forms, y16 = fp16(x * pro_a + pro_o) (B, rows, K), stored beside the products (round 6): the weight
 * gradients of these very linears read it back as their fp16 X operand (gecco_gemm_tn_f16_ex_f32 with both operands fp16).  y16 may be NULL. */
int gecco_linear_kvq_y16_f16(const float* x, const float* pro_a, const float* pro_o, const float* W1, const float* bias1, int Nout1,
                             void* C1, const float* W2, const float* bias2, int Nout2, void* C2, void* y16, int B, int rows, int K,
                             int head_dim, int lo_begin, int lo_end, void* wsplit, void* stream);
/* mlp.0 of a BroadcastingLayer's point MLP in the mixed mode (models/set_transformer.py:164-166: the first linear of
 * `x + mlp(mlp_norm(x))` with the AdaGN apply of models/normalization.py:44 folded in; models/mlp.py:5-39; activation.py:17-24):
 *   u = act((x*pro_a + pro_o) @ W^T + bias),   product = fp16(y) fp16(W) + fp8(y) fp8(W - fp16(W)) + fp8(y - fp16(y)) fp8(W)
 * (fp32 accumulate; the cross terms on v_mfma_scale_f32_32x32x64_f8f6f4), written as the TILED SPLIT IMAGE the next linear
 * loads into registers.  image_kind 1 — the tiled split image of the split-bf16 consumer: per (sample, 128-row tile, 16-column
 * step) one 8 KiB block, bf16 hi plane [128][16] (the 16-byte half of a row swapped when (row >> 3) & 1), then the lo plane;
 * B * rows * Nout * 4 bytes.  image_kind 2 — the h8 activation image of gecco_linear_h8_areg_f32: per (sample, 128-row tile,
 * 64-column group) one 24 KiB block: 16 KiB of fp16 MFMA fragments [32-row tile][sub][c][lane] (row 32 rt + (lane & 31), columns
 * 32 sub + 16 (lane >> 5) + 8 c + 0 .. 7), then 8 KiB of fp8(2^14 (u - fp16(u))) halves [32-row tile][t][lane] (columns 32 t +
 * 16 (lane >> 5) + 0 .. 15); B * rows * Nout * 3 bytes.
 * rows % 128 == 0, Nout % 64 == 0, K in {128, 256, 384}, act 0 .. 3.  wsplit: Nout * K * 4 bytes; W == NULL: image-ready call. */
int gecco_linear_h8_img_f32(const float* x, const float* pro_a, const float* pro_o, const float* W, const float* bias,
                            const float* alpha, int act, void* c_img, int image_kind, int B, int rows, int K, int Nout,
                            void* wsplit, void* stream);
/* The linear that consumes an h8 activation image (mixed mode: mlp.2 of the point MLP and the unpool attention's out_proj;
 * models/set_transformer.py:112,164-166, models/mlp.py:5-39):  C = residual + A @ W^T + bias  (+ GroupNorm partials `stats`
 * (B, rows / 128, 2, Nout) of C), A = hi + 2^-14 lo read from the image, product = fp16 main term + two fp8 cross terms as in
 * gecco_linear_h8_img_f32, fp32 accumulate.  C may alias residual.  rows % 128 == 0, K in {128, 256, 384, 512, 768, 1024},
 * Nout % 4 == 0.  wsplit: ceil(Nout / 128) * 128 * K * 4 bytes; W == NULL: image-ready call.  Option "h8areg". */
int gecco_linear_h8_areg_f32(const void* a_img, const float* W, const float* bias, const float* residual, float* C, float* stats,
                             int B, int rows, int K, int Nout, void* wsplit, void* stream);
/* The point-stream MLP of a BroadcastingLayer in one launch (fp16 mode; models/set_transformer.py:165-166 with the
 * AdaGN apply of :164 folded in): x += (GaussianActivation(fp16(x*pro_a + pro_o) @ W0^T + b0)) @ W2^T + b2, in place on
 * the fp32 x (B, rows, C); the 2C-wide hidden layer never leaves the CU.  stats (B, rows / 128, 2, C) or NULL: GroupNorm
 * partials of the updated x.  Bit-identical x and stats to gecco_linear_astat_f16 (act) followed by gecco_linear_f16io
 * (fp16 A, residual, stats).  C in {128, 256, 384}, width == 2 C, rows % 128 == 0; wsplit: 4 * C * width bytes. */
int gecco_mlp_fused_f16(float* x, const float* pro_a, const float* pro_o, const float* W0, const float* b0, const float* W2,
                        const float* b2, const float* alpha, int act, float* stats, int B, int rows, int C, int width,
                        void* wsplit, void* stream);
/* The unpool half of a BroadcastingLayer in one launch (fp16 mode; models/set_transformer.py:112 = nn.MultiheadAttention
 * with the 64 inducer states as keys / values, its out_proj, and the residual of :164): x += softmax(q k^T / sqrt(hd)) v @
 * W^T + bias, in place on the fp32 x (B, rows, C).  q16: head-major fp16 (B, H, rows, hd) as gecco_linear_astat_f16
 * (head_dim = hd) writes it; kvh (B, 64, 2C) fp32; stats (B, rows / 128, 2, C) or NULL.  Bit-identical x and stats to
 * gecco_unpool_attn_f16io (head_major) followed by gecco_linear_f16io (fp16 A, residual, stats).
 * (C, hd) in {(128, 16), (256, 32), (384, 48)}, rows % 128 == 0; wsplit: 2 * C * C bytes. */
int gecco_unpool_outproj_f16(float* x, const void* q16, const float* kvh, const float* W, const float* bias, float* stats,
                             int B, int rows, int C, int H, void* wsplit, void* stream);
/* The point MLP of a layer in the "w2" mode, one launch (mlp_fused_w.hip): out = x + mlp.2(act(mlp.0(x * pro_a + pro_o))) — out may be
 * x — with the 2 C wide hidden layer kept in registers.  Arithmetic: fp16 products with fp6 (e2m3, block-scaled) second terms; AdaGN(x)
 * and both weights carry two terms, the hidden layer one (fp16): ~2e-4 of the MLP's output scale against fp32, ~3.5e-4 on a network's
 * F_x (the mixed mode: 6e-5).  stats (B, rows / 128, 2, C) or NULL: GroupNorm partials of out.  Replaces
 * models/set_transformer.py:164-166, models/mlp.py:5-39, models/activation.py:17-24, models/normalization.py:36-44.
 * C in {128, 256, 384, 512} (512: two passes over the hidden width, mlp_fused_w.hip), width == 2 C, rows % 128 == 0; act 0 .. 3; wsplit: gecco_mlp_fused_w_wsplit_bytes(C, width) bytes; W0 == NULL: the weight
 * stream of an earlier call (same weights, biases, activation and alpha: the stream holds them all) is in wsplit.  dbg_u: NULL, or (B, rows, width) receiving mlp.0's
 * pre-activations times sqrt(log2(e) / 2) / |alpha| for the Gaussian activations (the form the kernel computes them in; diagnostics). */
int gecco_mlp_fused_w(const float* x, float* out, const float* pro_a, const float* pro_o, const float* W0, const float* b0, const float* W2,
                      const float* b2, const float* alpha, int act, float* stats, int B, int rows, int C, int width, void* wsplit, float* dbg_u,
                      void* stream);
size_t gecco_mlp_fused_w_wsplit_bytes(int C, int width);
/* The same half of the layer in the MIXED mode, one launch (unpool_outproj_h8.hip; option "unpoolh8", default 1): attention in
 * fp16 (the bits of gecco_unpool_attn_h8img), out_proj as the h8 product (fp16 main product + two fp8 cross terms, as
 * gecco_linear_h8_areg_f32), + bias + residual, in place on x, + GroupNorm partials; the attention output stays in registers as
 * the stationary operand.  Replaces models/set_transformer.py:70-75, 112 and the residual of :164.  Equal to
 * gecco_unpool_attn_h8img followed by gecco_linear_h8_areg_f32 up to fp32 summation order (~1e-6).  (C, hd) in {(128, 16),
 * (256, 32), (384, 48)}, rows % 128 == 0; wsplit: gecco_unpool_outproj_h8_wsplit_bytes(B, C, H) (weight image, then the fp16
 * k | v image of the inducers); W == NULL: the weight image is ready. */
int gecco_unpool_outproj_h8(float* x, const void* q16, const float* kvh, const float* W, const float* bias, float* stats,
                            int B, int rows, int C, int H, void* wsplit, void* stream);
size_t gecco_unpool_outproj_h8_wsplit_bytes(int B, int C, int H);
/* The mixed mode's unpool attention alone (nn.MultiheadAttention core, models/set_transformer.py:112): head-major fp16 q16
 * (B, H, N, hd), kvh (B, 64, 2C) fp32 -> the h8 activation image (image_kind 2 of gecco_linear_h8_img_f32: B * N * C * 3 bytes)
 * that gecco_linear_h8_areg_f32 consumes.  N % 128 == 0, C % 64 == 0, hd in {16, 32, 48, 64}. */
int gecco_unpool_attn_h8img(const void* q16, const float* kvh, void* out_img, int B, int N, int C, int H, void* stream);
/* y16[b, m, c] = fp16(a[b, c] * x[b, m, c] + o[b, c]) — the AdaGN apply (models/normalization.py:44) rounded once,
 * exactly the operand the fp16 GEMM's prologue would form.  C % 8 == 0. */
int gecco_affine_cast_f16(const float* x, const float* a, const float* o, void* y16, int B, int rows, int C,
                          void* stream);
/* gecco_pool_attn_ex_f32 (precision 2) reading an fp16 KV; gecco_unpool_attn_ex_f32 (precision 2) with fp16 q / out.
 * head_major != 0: KV16 is (B, 2H, N, hd) = K heads then V heads, q16 is (B, H, N, hd) (gecco_linear_astat_f16 with
 * head_dim = hd); out16 stays (B, N, C).  Same bits either way. */
int gecco_pool_attn_f16in(const void* KV16, const float* inducers, float* merged, int B, int N, int C, int H, int I,
                          int head_major, void* ws, size_t ws_bytes, void* stream);
int gecco_unpool_attn_f16io(const void* q16, const float* kvh, void* out16, int B, int N, int C, int H, int I,
                            int head_major, void* stream);

/* AttentionPool core (models/set_transformer.py:47-63, without out_proj): KV (B, N, 2C) -> merged (B, I, C). */
int gecco_pool_attn_f32(const float* KV, const float* inducers, float* merged, int B, int N, int C, int H, int I,
                        void* ws, size_t ws_bytes, void* stream);
/* Same with the arithmetic selected: precision 0 = exact fp32 MFMA, 1 = split-bf16 (head dims 16/32/48/64;
 * other head dims run the fp32 kernel).  Same workspace. */
int gecco_pool_attn_ex_f32(const float* KV, const float* inducers, float* merged, int B, int N, int C, int H, int I,
                           int precision, void* ws, size_t ws_bytes, void* stream);
size_t gecco_pool_attn_workspace_bytes(int B, int N, int C, int H, int I);

/* Training path: backward of the two attention cores (autograd through F.scaled_dot_product_attention,
 * models/set_transformer.py:55-63, and through nn.MultiheadAttention, :112, under loss.backward(), diffusion.py:213-222),
 * fused: probabilities are recomputed per tile, nothing of shape (B, H, N, I) touches HBM.
 *   gecco_pool_attn_lse_f32: after gecco_pool_attn_ex_f32 on the same workspace, lse (B, H, I) = log2-domain
 *     log-sum-exp of the scaled scores (what the backward needs to recompute P).
 *   gecco_pool_attn_bwd_f32: dO (B, I, C) -> dKV (B, N, 2C) and dQ_partials (P, H, I, hd) with
 *     P = B * gecco_pool_attn_bwd_partials(B, N, H); the caller sums the P partials in order (gecco_reduce_batch_f32).
 *   gecco_unpool_attn_bwd_f32: dO (B, N, C) -> dq (B, N, C) and dkv_partials (P, B, I, 2C) with
 *     P = gecco_unpool_attn_bwd_partials(B, N, H); summed over P by the caller. */
int gecco_pool_attn_lse_f32(const void* ws, size_t ws_bytes, float* lse, int B, int N, int C, int H, int I, void* stream);
int gecco_pool_attn_bwd_partials(int B, int N, int H);
int gecco_pool_attn_bwd_f32(const float* KV, const float* inducers, const float* merged, const float* lse, const float* dO,
                            float* dKV, float* dQ_partials, int B, int N, int C, int H, int I, void* stream);
int gecco_unpool_attn_bwd_partials(int B, int N, int H);
int gecco_unpool_attn_bwd_f32(const float* q, const float* kvh, const float* dO, float* dq, float* dkv_partials, int B, int N,
                              int C, int H, int I, void* stream);
/* the same with the arithmetic selected: precision 0 = exact fp32 MFMA, 1 = split-bf16 (head dims 16 / 32 / 48 / 64; other head
 * dims run the fp32 kernels), 2 = fp16 operands (the reference's autocast(float16) backward: one MFMA per product),
 * 3 (round 6) = 2 with the point-stream tensors as fp16 TENSORS, row-major: pool — KV and dKV (B, N, 2C) halves; unpool — q, dO and dq
 * (B, N, C) halves (pass the pointers through the float* parameters); the inducer-side tensors (inducers, merged, lse, dO of the pool;
 * kvh, the dkv partials) stay fp32.  Head dims 16 / 32 / 48 / 64 only. */
int gecco_pool_attn_bwd_ex_f32(const float* KV, const float* inducers, const float* merged, const float* lse, const float* dO,
                               float* dKV, float* dQ_partials, int B, int N, int C, int H, int I, int precision, void* stream);
int gecco_unpool_attn_bwd_ex_f32(const float* q, const float* kvh, const float* dO, float* dq, float* dkv_partials, int B, int N,
                                 int C, int H, int I, int precision, void* stream);

/* nn.MultiheadAttention core (models/set_transformer.py:112, between in_proj and out_proj):
 * q (B, N, C) projected queries, kvh (B, I, 2C) projected inducer keys|values -> out (B, N, C). */
int gecco_unpool_attn_f32(const float* q, const float* kvh, float* out, int B, int N, int C, int H, int I,
                          void* stream);
int gecco_unpool_attn_ex_f32(const float* q, const float* kvh, float* out, int B, int N, int C, int H, int I,
                             int precision, void* stream);

/* coef: 5*B floats: coef[4b..4b+3] = {c_skip, c_out, c_in, c_noise} (diffusion.py:46-51) and
 * coef[4B + b] = c_noise packed (the AdaGN `t` input for t_embed_dim == 1). */
int gecco_edm_coeffs_f32(const float* sigma, float sigma_data, float* coef, int B, void* stream);
/* out = (c_in * x) @ W^T + b  (linear_lift.py:44 / ray.py:99); coef may be NULL (c_in = 1);
 * stats (B, gecco_stats_row_tiles(N), 2, C) may be NULL. */
int gecco_lift_f32(const float* x, const float* coef, const float* W, const float* bias, float* out, float* stats,
                   int B, int N, int C, void* stream);
/* F = Linear(C->3)(norm(feat)); D = c_skip*x + c_out*F (linear_lift.py:25-29,46; ray.py:56-59,120;
 * diffusion.py:57).  gn_a/gn_o NULL -> per-point LayerNorm; else y = gn_a*feat + gn_o.  out and/or raw. */
int gecco_lower_edm_f32(const float* feat, const float* x, const float* coef, const float* W, const float* bias,
                        const float* gn_a, const float* gn_o, float* out, float* raw, int B, int N, int C, float eps,
                        void* stream);

/* ---- network-level entry points ------------------------------------------------------------ */

/* SetTransformer.forward (models/set_transformer.py:198-216), in place on x (B, N, C).
 * t (B, ctx_dim).  stats_x/stats_T: GroupNorm partials of x from its producer (NULL -> computed here).
 * h_in: NULL or host array of n_layers device pointers (B, I, C) = cached inducer states (`hs`).
 * h_out: NULL or host array of n_layers device pointers to receive them (`return_h`).
 * stats_out: NULL or (B, gecco_linear_row_tiles(N), 2, C) partials of the output (for a GN head). */
int gecco_set_transformer_fwd_f32(const GeccoSetTransformer* st, float* x, const float* t, const float* stats_x,
                                  int stats_T, const float* const* h_in, float* const* h_out, float* stats_out,
                                  int B, int N, void* ws, size_t ws_bytes, void* stream);
size_t gecco_set_transformer_workspace_bytes(const GeccoSetTransformer* st, int B, int N);

typedef struct GeccoLinearLift {   /* EDMPrecond(LinearLift(SetTransformer)), diffusion.py:22-62, linear_lift.py */
    GeccoSetTransformer inner;
    const float* lift_w;    /* lift.weight (C, 3)    */
    const float* lift_b;    /* lift.bias (C)         */
    const float* lower_w;   /* lower.1.weight (3, C) */
    const float* lower_b;   /* lower.1.bias (3)      */
    float sigma_data;
} GeccoLinearLift;

/* Diffusion.forward for the unconditional model (diffusion.py:233-247): x (B, N, 3), sigma (B) ->
 * denoised (B, N, 3); raw (optional) receives F_x. */
int gecco_linear_lift_fwd_f32(const GeccoLinearLift* m, const float* x, const float* sigma, float* denoised,
                              float* raw, const float* const* h_in, float* const* h_out, int B, int N, void* ws,
                              size_t ws_bytes, void* stream);
size_t gecco_linear_lift_workspace_bytes(const GeccoLinearLift* m, int B, int N);

/* ---- image-conditional path ------------------------------------------------------------------- */

/* dst[i] = fp16(src[i]) (round to nearest even), n elements: the fp16 texel image of a channels-last pyramid level. */
int gecco_cast_f16(const float* src, void* dst, size_t n, void* stream);

typedef struct GeccoPyramid {     /* FeaturePyramidContext.features, models/feature_pyramid.py:17-20 */
    int n_levels;                 /* <= 4 */
    int C[4], H[4], W[4];
    const float* feat[4];         /* CHANNELS-LAST (B, H, W, C) per level (see gecco_nchw_to_nhwc_f32): fp32, or — texel_f16 — fp16 */
    int texel_f16;                /* 1: the levels hold fp16 texels (gecco_cast_f16 of the fp32 image, made once per conditioner call): the
                                   * forward lookups (gecco_ray_lookup_f32, gecco_ray_network_fwd_f32) gather half the bytes; coordinates, taps
                                   * and weights stay fp32 and bit-exact, the interpolation runs in fp32 on the converted texels.  The gradient
                                   * entry points take fp32 levels only.  The "w2" plans use it; fp32 / bf16x3 / mixed keep fp32 texels. */
} GeccoPyramid;

typedef struct GeccoReparam {     /* reparam.py: 0 NoReparam, 1 GaussianReparam(mean, sigma), 2 UVLReparam */
    int kind;
    const float* mean;            /* (3) mean | uvl_mean */
    const float* std;             /* (3) sigma | uvl_std */
    float logit_scale;            /* UVLReparam.logit_scale (1.1) */
} GeccoReparam;

/* (B, C, H, W) -> (B, H, W, C).  Once per conditioner call; the lookup then runs 2*steps-1 times. */
int gecco_nchw_to_nhwc_f32(const float* src, float* dst, int B, int C, int H, int W, void* stream);

/* F.grid_sample(bilinear, zeros, align_corners=False) tap indices for uv in [0,1]^2 (models/ray.py:79-82):
 * x0,y0 int32 (n), wx1 = ix - x0, wy1 = iy - y0.  Bit-exact against the oracle's op order. */
int gecco_bilinear_taps_f32(const float* uv, int H, int W, int* x0, int* y0, float* wx1, float* wy1, size_t n,
                            void* stream);
/* The fused lookup's whole coordinate chain for every point — geometry (x c_in[b] when coef is given) -> reparametrisation ->
 * project_points -> the bilinear taps of every pyramid level — computed by the SAME device functions gecco_ray_lookup_f32's kernel
 * calls (models/ray.py:64-87, reparam.py:102-110, 159-177, 131-137).  Diagnostics and the bit-exactness tests of the index math:
 * uv (B, N, 2); x0 / y0 int32 and wx1 = ix - x0 / wy1 = iy - y0, each (n_levels, B, N).  Pyramid pointers are not read. */
int gecco_ray_lookup_taps_f32(const float* geom, const float* coef, const float* K, const GeccoReparam* rp, const GeccoPyramid* pyr, float* uv,
                              int* x0, int* y0, float* wx1, float* wy1, int B, int N, void* stream);

/* RayNetwork.extract_image_features (models/ray.py:64-87): geom (B, N, 3) diffusion-space geometry
 * (multiplied by coef's c_in when coef != NULL), K (B, 3, 3) -> out (B, N, sum C).  stats: NULL or
 * (B, gecco_lookup_row_tiles(N), 2, sum C) GroupNorm partials of out. */
int gecco_ray_lookup_f32(const float* geom, const float* coef, const float* K, const GeccoReparam* rp,
                         const GeccoPyramid* pyr, float* out, float* stats, int B, int N, void* stream);
int gecco_lookup_row_tiles(int N);
/* Its backward w.r.t. the pyramids (autograd of F.grid_sample's input under loss.backward(), models/ray.py:82-85):
 * dfeat[l] (B, H_l, W_l, C_l) channels-last, ZEROED by the caller, += tap weight * dout (B, N, sum C).  pyr gives the
 * level shapes (its feat pointers are not read).  Float atomics, like torch's grid_sampler backward: any N; the sorted form
 * below is the one the training path uses for N <= 4096. */
int gecco_ray_lookup_bwd_f32(const float* geom, const float* coef, const float* K, const GeccoReparam* rp,
                             const GeccoPyramid* pyr, const float* dout, float* const* dfeat, int B, int N,
                             void* stream);
/* The lookup's gradient with respect to the GEOMETRY: dgeom (B, N, 3) = autograd of F.grid_sample w.r.t. its grid through the
 * projection and the reparametrisation (models/ray.py:64-87), for a caller that differentiates the conditional denoiser with respect
 * to its input cloud; geom is the diffusion-space geometry the forward saw (c_in already applied), pyr the pyramid it read.
 * dK_partials (optional; dgeom may then be NULL): (B, gecco_lookup_row_tiles(N), 4) partial sums of the gradient with respect to
 * (fx, cx, fy, cy) of each sample's camera matrix — the projection and, for the UVL reparametrisation, the unprojection in front. */
int gecco_ray_lookup_dgeom_f32(const float* geom, const float* K, const GeccoReparam* rp, const GeccoPyramid* pyr, const float* dout,
                               float* dgeom, float* dK_partials, int B, int N, void* stream);
/* The same gradient by SORT + GATHER: per (image, level) the 4 N (texel, point, tap) entries are sorted by texel in LDS, then
 * every texel's threads walk its list — no atomics, a fixed summation order (bit-reproducible), and every texel of dfeat is
 * WRITTEN (no zero fill).  Needs N <= 4096 and H_l W_l <= 2^17 (gecco_ray_lookup_bwd_sorted_workspace_bytes returns 0
 * otherwise: use the atomic form); ws: that many bytes of scratch. */
size_t gecco_ray_lookup_bwd_sorted_workspace_bytes(const GeccoPyramid* pyr, int B, int N);
int gecco_ray_lookup_bwd_sorted_f32(const float* geom, const float* coef, const float* K, const GeccoReparam* rp,
                                    const GeccoPyramid* pyr, const float* dout, float* const* dfeat, int B, int N, void* ws,
                                    size_t ws_bytes, void* stream);

typedef struct GeccoRayNetwork {  /* EDMPrecond(RayNetwork(SetTransformer, reparam)), models/ray.py:33-120 */
    GeccoSetTransformer backbone;
    const float* xyz_w;  const float* xyz_b;   /* xyz_embed (C, 3), (C)                  */
    const float* img_w;  const float* img_b;   /* img_feature_proj.1 (C, sum C_l), (C)   */
    const float* out_w;  const float* out_b;   /* output_proj.1 (3, C), (3)              */
    GeccoReparam reparam;
    float sigma_data;
} GeccoRayNetwork;

/* Diffusion.forward for the image-conditional model with a precomputed feature pyramid
 * (diffusion.py:233-247, post_context given). */
int gecco_ray_network_fwd_f32(const GeccoRayNetwork* m, const float* x, const float* sigma, const float* K,
                              const GeccoPyramid* pyr, float* denoised, float* raw, const float* const* h_in,
                              float* const* h_out, int B, int N, void* ws, size_t ws_bytes, void* stream);
size_t gecco_ray_network_workspace_bytes(const GeccoRayNetwork* m, const GeccoPyramid* pyr, int B, int N);

/* ---- reparameterisations and activation (reparam.py, models/activation.py) -------------------- */
/* inverse = 0: data_to_diffusion, 1: diffusion_to_data.  is_f64 selects float/double x, y (the sampler state is fp64). */
int gecco_gaussian_reparam(const void* x, const float* mean, const float* sigma, void* y, size_t n_elems, int dim,
                           int inverse, int is_f64, void* stream);
int gecco_uvl_reparam(const void* x, const float* K, const float* uvl_mean, const float* uvl_std, double logit_scale,
                      void* y, int B, int N, int inverse, int is_f64, void* stream);
/* nn.ReLU, the reference's default `activation` (models/mlp.py:12, set_transformer.py:81,133), stand-alone and its
 * backward du = dy * (y > 0) (training path).  In the fused kernels ReLU is epilogue code act = 3. */
int gecco_relu_f32(const float* x, float* y, size_t n, void* stream);
int gecco_relu_bwd_f32(const float* y, const float* dy, float* du, size_t n, void* stream);
int gecco_gaussian_act_f32(const float* x, const float* alpha, float* y, size_t n, int normalized, void* stream);

/* ---- sampler state kernels (diffusion.py:271-352, 354-470) ------------------------------------- */
/* sched: DEVICE table of doubles, GECCO_SCHED_COLS per step: {t_cur, t_hat, t_next, churn, redo, 0, 0, 0} with
 * churn = sqrt(t_hat^2 - t_cur^2) * S_noise, redo = sqrt(t_cur^2 - t_next^2); step: DEVICE int (current row).
 * noise pointers are advanced by step * noise_step_stride elements inside the kernel. */
#define GECCO_SCHED_COLS 8
/* x_out = x_cur + (double)((float)sched[step][col] * noise); x_in = (float)x_out; sigma[b] = (float)sched[step][sigma_col] */
int gecco_sampler_add_noise_f64(const double* x_cur, const float* noise, size_t noise_step_stride, const double* sched,
                                const int* step, int col, int sigma_col, double* x_out, float* x_in, float* sigma,
                                size_t n, int B, void* stream);
/* out = x + noise * (float)sched[step][col]; sigma[b] = that coefficient (upsampler's data_ctx, diffusion.py:430) */
int gecco_sampler_add_noise_f32(const float* x, const float* noise, size_t noise_step_stride, const double* sched,
                                const int* step, int col, float* out, float* sigma, size_t n, int B, void* stream);
/* d_cur = (x_hat - den)/t_hat; x_next = x_hat + (t_next - t_hat) d_cur; x_in = (float)x_next; sigma[b] = t_next */
int gecco_sampler_euler_f64(const double* x_hat, const float* den, const double* sched, const int* step, double* d_cur,
                            double* x_next, float* x_in, float* sigma, size_t n, int B, void* stream);
/* x_out = x_hat + (t_next - t_hat) (0.5 d_cur + 0.5 (x_next - den)/t_next) */
int gecco_sampler_heun_f64(const double* x_hat, const double* x_next, const float* den, const double* d_cur,
                           const double* sched, const int* step, double* x_out, size_t n, void* stream);
int gecco_sampler_advance(int* step, int delta, void* stream);
/* x = (double)latents * t0 */
int gecco_sampler_scale_f64(const float* latents, double t0, double* x, size_t n, void* stream);

/* ---- training path (EDMLoss.backward: diffusion.py:136-143 through autograd) ------------------------ */

/* C[z](M x N) = scale * op(A[z]) op(B[z]) (+ bias[n]);  X row-major in its own index (k contiguous) or "k-major"
 * (X[k][row]).  z = z1*zdiv + z2 with element strides (s?1, s?2) per operand.  Backward products use the SAME
 * tensors as the forward ones, only with other layout flags (dX = dY W: B k-major; dW = dY^T X: both k-major). */
typedef struct GeccoGemm {
    const float* A; const float* B; const float* bias; float* C;
    int Z, zdiv, M, N, K, lda, ldb, ldc;
    long long sA1, sA2, sB1, sB2, sC1, sC2;
    int a_kmajor, b_kmajor;
    float scale;
} GeccoGemm;
int gecco_gemm_f32(const GeccoGemm* g, void* stream);
/* out[i] (+)= sum_z parts[z*stride + i], fixed order (deterministic parameter gradients). */
/* Weight gradient of a linear in split-bf16 arithmetic (training path; autograd of every nn.Linear on the point stream):
 * parts[g] = sum over the samples z of group g of A[z]^T @ B[z], A (Z, R, N) = dY, B (Z, R, K) = X, parts
 * (ceil(Z / group), N, K); dW = gecco_reduce_batch_f32 over the groups (fixed order: bit-reproducible).
 * R % 32 == 0, N % 4 == 0, K % 4 == 0 (128 x 128 output tiles; the last tile of a dimension may be partial). */
int gecco_gemm_tn_x3_f32(const float* A, const float* Bm, float* parts, int Z, int R, int N, int K, int group, void* stream);
/* the same, also leaving colsum_parts[g] (ceil(Z / group), N) = column sums of A over the group's rows: the bias gradient
 * db = sum_rows dY comes out of the pass that reads dY for dW (colsum_parts may be NULL) */
int gecco_gemm_tn_x3_bias_f32(const float* A, const float* Bm, float* parts, float* colsum_parts, int Z, int R, int N, int K,
                              int group, void* stream);
/* the same with B read through an AdaGN apply, B'[z, m, k] = B[z, m, k] * pro_a[z, k] + pro_o[z, k] (pro_a / pro_o (Z, K) or
 * both NULL): the weight gradient of a linear whose input was AdaGN(x), formed from x — AdaGN(x) is never materialised */
int gecco_gemm_tn_x3_pro_f32(const float* A, const float* Bm, const float* pro_a, const float* pro_o, float* parts,
                             float* colsum_parts, int Z, int R, int N, int K, int group, void* stream);
/* the same product with both operands rounded to fp16 and ONE MFMA per product (fp32 accumulation, fp32 partials): the weight
 * gradient under the reference's autocast(float16) trainer setting (torch computes it as an fp16 matmul there); pro_a / pro_o
 * and colsum_parts may be NULL.  The caller's GradScaler keeps dY inside fp16's range, as it does for the reference. */
int gecco_gemm_tn_f16_f32(const float* A, const float* Bm, const float* pro_a, const float* pro_o, float* parts,
                          float* colsum_parts, int Z, int R, int N, int K, int group, void* stream);
/* output tiles per sample group of that kernel for an (N, K) gradient (its block tile is 128 x 128, 256 x 128 or 128 x 256 by shape):
 * the caller sizes `group` so that groups x tiles fill the chip */
int gecco_gemm_tn_f16_tiles(int N, int K);
/* the same with B already an fp16 tensor (Z, R, K) — the hidden layer of an MLP that gecco_linear_act_keep_h16 stored that way:
 * its tiles go to LDS as they are (no AdaGN apply) */
int gecco_gemm_tn_f16_b16_f32(const float* A, const void* B16, float* parts, float* colsum_parts, int Z, int R, int N, int K, int group,
                              void* stream);
/* ... and with A (dY) an fp16 tensor (N % 8 == 0), B fp32 with the optional AdaGN apply: the weight gradient of an MLP's first linear from du
 * stored as halves; the bias gradient's column sums are formed from those halves. */
int gecco_gemm_tn_f16_a16_f32(const void* A16, const float* Bm, const float* pro_a, const float* pro_o, float* parts, float* colsum_parts,
                              int Z, int R, int N, int K, int group, void* stream);
/* The general form (round 6): either operand may be the fp16 tensor (a_f16 / b_f16, not both), and — with `counters`, `out` — the
 * fixed-order sum of the group partials happens INSIDE the launch: counters = one zeroed unsigned per output tile
 * (gecco_gemm_tn_f16_tiles(N, K) of them; the kernel leaves them zero), out (N, K) = sum over groups of parts in group order,
 * colsum_out (N) likewise of colsum_parts — the bits gecco_reduce_batch_f32 gives, without its launches (autograd of every nn.Linear
 * under the reference's precision="16-mixed": diffusion.py:213-222). */
int gecco_gemm_tn_f16_ex_f32(const void* A, int a_f16, const void* Bm, int b_f16, const float* pro_a, const float* pro_o, float* parts,
                             float* colsum_parts, float* out, float* colsum_out, unsigned* counters, int Z, int R, int N, int K, int group,
                             void* stream);
int gecco_reduce_batch_f32(const float* parts, float* out, size_t n, int Z, size_t stride, int accumulate, void* stream);

/* Row softmax of the materialised attention scores: P = softmax(scale*S) over the last dim n; and its backward
 * dS = scale * P * (dP - sum(P*dP)). */
int gecco_softmax_fwd_f32(const float* S, float* P, size_t rows, int n, float scale, void* stream);
int gecco_softmax_bwd_f32(const float* P, const float* dP, float* dS, size_t rows, int n, float scale, void* stream);

/* GaussianActivation backward: du = dy*g'(u); partial (gecco_gauss_act_bwd_blocks(n)) per-block sums of dy*dg/dalpha. */
int gecco_gauss_act_bwd_f32(const float* u, const float* dy, const float* alpha, float* du, float* partial, size_t n,
                            int normalized, void* stream);
int gecco_gauss_act_bwd_blocks(size_t n);

/* AdaGN / GroupNorm backward.  gstats (B, gecco_stats_row_tiles(rows), 2, C) = {sum dy, sum dy*x};
 * coefficients give dx = dy*cA + x*cB + cC; ds, dz (B, C) are the grads of the per-(b,c) scale and shift. */
int gecco_col_dot_stats_f32(const float* dy, const float* x, float* gstats, int B, int rows, int C, void* stream);
int gecco_adagn_bwd_coeffs_f32(const float* xstats, int Tx, const float* gstats, int Tg, int rows, const float* t,
                               int ctx_dim, const GeccoAdaGN* p, float* cA, float* cB, float* cC, float* ds, float* dz,
                               int B, int C, int G, float eps, void* stream);
int gecco_affine2_apply_f32(const float* dy, const float* x, const float* cA, const float* cB, const float* cC,
                            float* dx, int B, int rows, int C, void* stream);
/* the same with the gradient that reaches x through the residual connection added in the same pass
 * (x = x + f(norm(x)), models/set_transformer.py:164-166): dx = dy*cA + x*cB + cC + add; add may be NULL */
int gecco_affine2_apply_add_f32(const float* dy, const float* x, const float* cA, const float* cB, const float* cC,
                                const float* add, float* dx, int B, int rows, int C, void* stream);
int gecco_adagn_param_grads_f32(const float* ds, const float* dz, const float* t, int B, int C, int ctx_dim,
                                float* d_scale_w, float* d_scale_b, float* d_bias_w, float* d_bias_b, void* stream);

/* lift backward: partial (B, gecco_stats_row_tiles(N), 4, C) = {dW[:,0], dW[:,1], dW[:,2], db} per row tile. */
int gecco_lift_bwd_f32(const float* dY, const float* xin, float* partial, int B, int N, int C, void* stream);
/* lower backward (LayerNorm + Linear(C->3)): dfeat (rows, C); partial (gecco_lower_bwd_blocks(rows), 3*C + 4). */
int gecco_lower_bwd_f32(const float* feat, const float* dF, const float* W, float* dfeat, float* partial, size_t rows,
                        int C, float eps, void* stream);
int gecco_lower_bwd_blocks(size_t rows);

/* ---- optimizer step -------------------------------------------------------------------
 * torch.optim.Adam(lr=1e-4) of Diffusion.configure_optimizers (diffusion.py:210-211) fused with the EMA shadow-weight
 * update the reference runs after every step (EMAOptimizer.step / update, ema.py:273-325; ema_update, ema.py:187-194:
 * ema = ema * decay + (1 - decay) * param): ONE pass over flat, 16-byte aligned fp32 buffers of n elements (n % 4 == 0),
 * 36 bytes per parameter.  Same operation order as torch's single-tensor Adam, in fp32:
 *   g' = g * grad_scale (+ weight_decay * p);  m += (1 - beta1) (g' - m);  v = beta2 v + (1 - beta2) g'^2;
 *   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps);  ema = decay * ema + (1 - decay) * p.
 * `step` is the 1-based count of this update (Adam's state["step"] after it).  `grad_scale` folds the 1 / world_size of
 * a summing gradient all-reduce into the read of g.  ema may be NULL when do_ema == 0. */
typedef struct GeccoAdamEma {
    float* p; const float* g; float* m; float* v; float* ema;
    size_t n;
    double lr, beta1, beta2, eps, weight_decay;   /* doubles: torch derives 1 - beta, lr / bc1, 1 - decay in double, then rounds */
    double ema_decay;
    float grad_scale;
    int step;
    int do_ema;
} GeccoAdamEma;
int gecco_adam_ema_step_f32(const GeccoAdamEma* a, void* stream);
/* The same step inside torch.amp.GradScaler's protocol for optimizers that handle the scale themselves
 * (`_step_supports_amp_scaling`, torch/amp/grad_scaler.py: the scaler hands over its scale and found_inf tensors instead of
 * unscaling and reading found_inf back on the host — the reference's `precision="16-mixed"` trainer, example_configs): every
 * gradient is divided by *amp_scale while it is read (NULL: already unscaled), and when *found_inf != 0 NOTHING is written
 * (torch skips optimizer.step(), and the EMA update inside it, then) and *skipped is incremented; the bias corrections use
 * step - *skipped, Adam's own step count.  All three are device scalars; the host never waits for the gradients. */
int gecco_adam_ema_step_amp_f32(const GeccoAdamEma* a, const float* amp_scale, const float* found_inf, int* skipped, void* stream);
/* ema = ema * decay + (1 - decay) * p alone (ema_update, ema.py:187-194), for optimizers other than the fused Adam. */
int gecco_ema_update_f32(const float* p, float* ema, size_t n, double decay, void* stream);

/* ---- samplers and metrics behind the hot path (gecco-torch/README.md:49-52 lists them as absent
 * from the torch package, the JAX package has them) ------------------------------------------------------------------ */

/* Inpainting sampler (gecco-jax models/stochastic.py:101-187): every sub-step re-draws the KNOWN points of the state at
 * the current noise level: x[b, m + j, :] = known[b, j, :] + noise[b, j, :] * sched[step][col]; x (B, m + n_known, 3)
 * fp64, known / noise (B, n_known, 3) fp32 (diffusion space). */
int gecco_sampler_refresh_known_f64(double* x, const float* known, const float* noise, const double* sched, const int* step,
                                    int col, int m, int n_known, int B, void* stream);
/* D[b, i, j] = |a[b, i] - b[b, j]| formed as sqrt(max(|a|^2 + |b|^2 - 2 a.b, 0)) (gecco-jax geometry.py:8-24); squared != 0:
 * no square root.  a (B, N, 3), b (B, M, 3), D (B, N, M). */
int gecco_distance_matrix_f32(const float* a, const float* b, float* D, int B, int N, int M, int squared, void* stream);
/* Chamfer distance per sample (gecco-jax metrics.py:92-103): out[b] = (mean_i min_j d(a_i, b_j) + mean_j min_i d) / 2.
 * ws: B * (N + M) floats. */
int gecco_chamfer_f32(const float* a, const float* b, float* out, float* ws, int B, int N, int M, int squared, void* stream);
/* Set-vs-set Chamfer distances (gecco-jax benchmark.py:21-39 `batched_pairwise_distance` with `chamfer_distance` / `_squared`):
 * out (S, T) row-major, out[s, t] = Chamfer(a[s], b[t]) for a (S, N, 3) and b (T, M, 3) — two launches (one per direction), no
 * N x M matrix anywhere.  The minimum is taken over |b|^2 - 2 a.b with |a|^2 added after it (the reference adds it before: equal up
 * to the rounding of that addition). */
int gecco_set_chamfer_f32(const float* a, const float* b, float* out, int S, int T, int N, int M, int squared, void* stream);
/* 1-NN accuracy, minimum matching distance, coverage of a generated set against a reference set of n clouds each (gecco-jax
 * benchmark.py:128-156: `_assemble_dist_m`, `_one_nn_acc`, `_mmd`, `_cov`, reference semantics to the letter incl. the `<= n` of
 * `_one_nn_acc` and numpy's first-of-equals argmin): ss (n, n) sample-sample, sd (n, n) sample-data, dd (n, n) data-data distances;
 * out3 = {1-NNA, MMD, COV}; flags: n ints of scratch. */
int gecco_set_metrics_f32(const float* ss, const float* sd, const float* dd, int n, float* out3, int* flags, void* stream);
/* Entropic optimal transport between uniform marginals on a cost matrix C (B, N, M) (gecco-jax metrics.py:141-156:
 * `sinkhorn_emd` through ott): `iterations` log-domain Sinkhorn sweeps, then out[b] = <P, C> with
 * P_ij = exp((f_i + g_j - C_ij) / epsilon) / (N M).  f (B, N), g (B, M), rowcost (B, N) are caller scratch / outputs. */
int gecco_sinkhorn_f32(const float* C, float* f, float* g, float* rowcost, float* out, int B, int N, int M, float epsilon,
                       int iterations, void* stream);

/* ---- ConvNeXt conditioner, channels-last on the device (ConvNeXtExtractor, models/feature_pyramid.py:28-73,
 * = torchvision's ConvNeXt stages).  Activations are (B, H, W, C) fp32.  The pointwise linears of a CNBlock run through
 * gecco_linear_ex_f32 on rows = B H W (act = 4: exact-erf GELU; the second one with the block input as residual and
 * layer_scale folded into its weights by gecco_convnext_fold_scale_f32); these are the rest of a block. */
/* stem: out = LayerNorm_C(Conv2d(3 -> C, k4, s4)(x) + bias); x NCHW (B, 3, H, W), w (C, 3, 4, 4), out (B, H/4, W/4, C); C == 96 */
int gecco_convnext_stem_f32(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b, float* out,
                            int B, int H, int W, int C, float eps, void* stream);
/* CNBlock front half: out = LayerNorm_C(dwconv7x7(x, padding 3) + bias); w TAP-MAJOR (49, C) = the module's weight
 * (C, 1, 7, 7) reshaped to (C, 49) and transposed (a re-layout of 19 .. 75 KB of parameters, like the 2 x 2 downsample
 * weight's); C in {96, 192, 384} */
int gecco_convnext_dwconv_ln_f32(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b, float* out,
                                 int B, int H, int W, int C, float eps, void* stream);
/* downsample front half: LayerNorm_C per texel, written as the 2 x 2 stride-2 conv's GEMM operand (B, H/2, W/2, (dy, dx, c)) */
int gecco_convnext_ln_patch2_f32(const float* x, const float* ln_w, const float* ln_b, float* out, int B, int H, int W, int C,
                                 float eps, void* stream);
/* Wo[n, k] = s[n] W[n, k], bo[n] = s[n] b[n] (CNBlock.layer_scale folded into its second linear) */
int gecco_convnext_fold_scale_f32(const float* W, const float* b, const float* s, float* Wo, float* bo, int N, int K, void* stream);

/* ---- the conditioner's TRAINING path.  The reference trains ConvNeXtExtractor with the denoiser (it is a sub-module of
 * Diffusion: diffusion.py:203-222 optimises self.parameters(); feature_pyramid.py:55-59 only removes stochastic depth), so
 * the pyramid gradient of gecco_ray_lookup_bwd_f32 continues through these.  The pointwise linears' dX / dW / db are the
 * GEMM entries of the denoiser's training path (gecco_linear_ex_f32 on W^T, gecco_gemm_tn_x3_bias_f32, gecco_gemm_f32). */
/* the two forward entries above that also keep z, the LayerNorm's input (same shape as out) */
int gecco_convnext_stem_train_f32(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b, float* out,
                                  float* z, int B, int H, int W, int C, float eps, void* stream);
int gecco_convnext_dwconv_ln_train_f32(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b,
                                       float* out, float* z, int B, int H, int W, int C, float eps, void* stream);
/* out = dwconv7x7(x, padding 3) (+ bias when non-null), w tap-major (49, C).  With the taps reversed (w[48 - tap]) and dz as
 * x this is the depthwise convolution's input gradient. */
int gecco_convnext_dwconv_f32(const float* x, const float* w, const float* bias, float* out, int B, int H, int W, int C, void* stream);
/* the depthwise convolution's input gradient in one launch: dx = dwconv7x7(dz, taps reversed) (+ add: the gradient that reached
 * the block input through its skip connection); w is the FORWARD tap-major weight (the kernel stages it reversed) */
int gecco_convnext_dwconv_bwd_f32(const float* dz, const float* w, const float* add, float* dx, int B, int H, int W, int C, void* stream);
/* backward of gecco_convnext_fold_scale_f32: dW = s dW', db = s db', ds[n] = sum_k dW'[n, k] W[n, k] + db'[n] b[n] */
int gecco_convnext_fold_scale_bwd_f32(const float* dWp, const float* dbp, const float* W, const float* b, const float* s, float* dW,
                                      float* db, float* ds, int N, int K, void* stream);
/* LayerNorm_C backward per texel from its input z (B, H, W, C) (statistics recomputed): dz, and per-block column partials
 * parts (gecco_convnext_ln_bwd_blocks(B, H, W, C), 3, C) = [d ln_w | d ln_b | column sums of dz] (the last is the bias
 * gradient of the convolution that produced z); reduce with gecco_reduce_batch_f32.  patch2 != 0: dy is laid out as the
 * downsample GEMM's operand (B, H/2, W/2, (dy, dx, c)) — the backward of gecco_convnext_ln_patch2_f32. */
int gecco_convnext_ln_bwd_blocks(int B, int H, int W, int C);
int gecco_convnext_ln_bwd_f32(const float* z, const float* dy, const float* ln_w, float* dz, float* parts, int B, int H, int W, int C,
                              float eps, int patch2, void* stream);
/* depthwise 7 x 7 weight gradient, tap-major: parts (gecco_convnext_dwconv_dw_blocks(B, H, W, C), 49, C) per-block partials of
 * dW[tap][c] = sum_texels dz[b, y, x, c] x[b, y + dy - 3, x + dx - 3, c] */
int gecco_convnext_dwconv_dw_blocks(int B, int H, int W, int C);
int gecco_convnext_dwconv_dw_f32(const float* x, const float* dz, float* parts, int B, int H, int W, int C, void* stream);
/* y = GELU(u) (exact erf form, nn.GELU()) and du = dy GELU'(u) on n values (n % 4 == 0) */
int gecco_gelu_f32(const float* u, float* y, size_t n, void* stream);
int gecco_gelu_bwd_f32(const float* u, const float* dy, float* du, size_t n, void* stream);
/* the stem's patch matrix (B H/4 W/4, 48), k = (ci, dy, dx) as in conv.weight.reshape(C, 48): its weight gradient is dz^T @ patches */
int gecco_convnext_im2col4_f32(const float* x, float* out, int B, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif
