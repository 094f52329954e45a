// Internal launcher interface between the kernel translation units and the C ABI (api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct GemmArgs {
    const float* A;         // (B, rows, lda)
    const float* W;         // (Nout, ldw)
    const float* bias;      // (Nout) or null
    const float* pro_a;     // (B, K) or null: A' = A * pro_a + pro_o (AdaGN apply)
    const float* pro_o;     // (B, K)
    const float* alpha;     // device scalar (GaussianActivation.alpha) when act != 0
    const float* residual;  // (B, rows, ldr) or null
    float* C;               // (B, rows, ldc)
    float* stats;           // (B, tilesM, 2, Nout) or null
    int B, rows, K, Nout;
    int lda, ldw, ldc, ldr;
    int act;                // 0 none, 1 gaussian normalized, 2 gaussian raw, 3 ReLU, 4 GELU (erf)
    int precision;          // 0 exact fp32 MFMA, 1 split-bf16, 2 fp16 (1 and 2 need w_img)
    const void* w_img;      // tiled image of W: bf16 hi | lo (split_bf16_tiled_launch) or fp16 (split_f16_tiled_launch)
    // optional second output segment, LDS-DMA kernel only: columns [n_split, Nout) are a second linear over the same
    // A (weights W2 (Nout - n_split, ldw), bias2) written to C2 (B, rows, ldc2); n_split % 128 == 0, no stats/residual
    float* C2;
    const float* W2;
    const float* bias2;
    int n_split, ldc2;
    // fp16 kernel only: A is an fp16 tensor (lda in fp16 elements; no prologue) / C (and C2) are stored as fp16
    // (ldc, ldc2 in fp16 elements; no residual, no stats).  Pointers are passed through the float* fields.
    int a_f16, c_f16;
    // A-stationary fp16 kernel only: head-major output.  hm_hd = head dim (0: row-major).  Column n of a segment with
    // nseg columns goes to element ((b * nseg / hd + n / hd) * rows + row) * hd + n % hd: one contiguous (rows, hd)
    // slab per (sample, head) — what the attention kernels stream — instead of hd-wide pieces of (rows, nseg) rows.
    int hm_hd;
    // A-stationary fp16 kernel only: image of the weights' LOW part fp16(W - fp16(W)) (same layout as w_img; split jobs with
    // pad_ = 1): two-term fp16 weights, 2 MFMAs per product ("mixed" mode).  Null: one-term weights.
    const void* w_img2;
    int lo_tiles;           // with w_img2: only the 128-column tiles [lo_begin, lo_tiles) have a lo part (lo_tiles 0: all)
    int lo_begin;
    int lo_fp8;             // with w_img2: the lo image holds fp8 (e4m3) values scaled by 2^19 in 64-k stages (gemm_f16_astat.hip WS = 3)
    // split-bf16 LDS-DMA kernel only: an activation handed from one GEMM to the next as a TILED SPLIT IMAGE instead of an
    // fp32 tensor — per (sample, 128-row tile, 16-k step) one 8 KiB block that IS the consumer's LDS A tile: bf16 hi plane
    // [128][16] then lo plane, the 16-byte chunk of a row swapped when (row >> 3) & 1 (the W image's layout, gemm_f32_dma.hip).
    // c_img: C is written that way (no residual, no statistics, rows % 128 == 0, Nout % 16 == 0);  a_img: A is read that way
    // (no prologue; K % 16 == 0): contiguous 1 KiB DMA pieces instead of 64-byte row pieces, and no hi / lo split in the K loop.
    int a_img, c_img;       // 1: tiled split image (bf16 hi | lo planes); 2: h8 activation image (gemm_h8_areg.hip: fp16 hi + fp8 lo)
    // Activation BACKWARD as an epilogue (training: the dX product of the linear that FOLLOWS an activation).  C = (A W^T) *
    // act'(u) with u (B, rows, ldc) the pre-activation the forward kept: dh = dy W2 never exists, du leaves directly.
    // mul_kind: 1 / 2 GaussianActivation normalized / raw (alpha in `alpha`), 3 ReLU, 4 GELU.  agrad (Gaussian only):
    // one float per 128-row x 128-column output tile, (b * ceil(rows / 128) + m0 / 128) * ceil(Nout / 128) + n0 / 128,
    // = sum over the tile of (A W^T) * d act / d alpha (u) — the alpha gradient's partials (zero-initialised by the caller:
    // one slot per 64 rows: 128- / 256-row tiles write every second / fourth).  LDS-DMA kernels only; no bias, no forward activation, no statistics.
    const float* mul_u;
    int mul_kind;
    float* agrad;
    // Activation FORWARD that keeps its input (training: the backward needs u): C = act(A W^T + bias) as with `act`, and
    // pre_out (B, rows, ldc) = A W^T + bias.  Same kernel instantiation as mul_u (the training path's); not with c_img.
    float* pre_out;
    // Column statistics of the training path's AdaGN backward from the GEMM that produces its dy (same kernel instantiation as mul_u):
    // with `stats`, the second statistic becomes sum_rows C * dot_x instead of sum_rows C^2 — {sum dy, sum dy x} of col_dot_stats_kernel,
    // dot_x (B, rows, ldc) the tensor the AdaGN normalised.  Not with residual / mul_u / pre_out.
    const float* dot_x;
    // fp16 LDS-DMA kernel only: PER-SAMPLE weights — sample b streams the image at w_img + b * w_img_bstride floats and adds bias + b *
    // bias_bstride (0: one image / bias for all samples).  img_feature_proj with the GroupNorm apply folded into the weights (fold_f16_image_launch)
    size_t w_img_bstride;
    int bias_bstride;
    int h8_rev;                // gemm_h8_astat.hip: blocks walk the row panels last to first (the producer wrote them first to last)
    // gemm_kvq_astat_kernel, head-major fp16 outputs at head dim 48 (feature_dim 384, 8 heads): the columns of every 384-column segment
    // (K, V, q) are dealt to its six 64-column tiles HEAD-ALIGNED — tile t = head t's 48 columns + the (t % 3)-th 16-column third of head
    // 6 + t / 3 — by the stream builder (SplitJob::pad_ | 64; kvq_perm48_col); a wave then owns a head's whole (32 rows, 48) slab = 3 KiB
    // of CONTIGUOUS head-major memory per tile, written as three 1 KiB stores through its LDS tile instead of as 32-byte pieces.  A
    // column's dot product does not depend on where in a tile it sits: the outputs are bit-identical to the plain order.
    int kvq_perm;
    // gemm_kvq_astat_kernel (training forward, round 6): also store the A operand the kernel forms, y16 = fp16(A * pro_a + pro_o), as an
    // fp16 tensor (B, rows, K) — the weight gradient of this very linear reads it back as its fp16 X operand (gemm_tn_f16.hip, DMA form)
    void* y16_out;
    int h6;                    // gemm_h8_astat.hip (c_img == 2): the cross terms in fp6 with block scales; w_img is the h6 stream (SplitJob::pad_ 32)
    int h8_stagger, h8_pair;   // gemm_h8_astat.hip: start offset of every second block of a CU (set by its launcher)
};

struct SplitJob { const float* W; float* img; int Nout, K, ldw, pad_; };   // pad_ = 1 (fp16 images): the LOW part fp16(W - fp16(W)); 2: the same as fp8 x 2^19 in 64-k blocks; 8: hi | lo blocks interleaved per column tile; 4 (bf16 images): W is (K, ldw) and the image is of W^T; h8 images (h8_image_multi_launch): see gemm_h8_astat.hip
struct SplitJobs { SplitJob job[96]; int n; };   // 3 KiB of kernel arguments
int gemm_row_tile(int rows);  // row-tile height the GEMM uses for `rows` rows per sample
int gemm_f32_launch(const GemmArgs& g, hipStream_t st);
// gemm_f32_dma.hip — LDS-DMA fast path of the same contract
bool gemm_f32_dma_supported(const GemmArgs& g, int precision = -1);   // -1: g.precision; split-bf16 also takes 64 <= rows < 128
int gemm_f32_dma_launch(const GemmArgs& g, hipStream_t st);
size_t split_bf16_image_bytes(int Nout, int K);   // ceil(Nout / 128) * 128 * K * 4
int split_bf16_tiled_launch(const float* W, void* img, int Nout, int K, int ldw, hipStream_t st);
int split_bf16_tiled_multi_launch(const SplitJobs& jobs, hipStream_t st);   // K % 16 == 0, ldw % 4 == 0 per job

// gemm_f16_dma.hip — the same contract in fp16 arithmetic (precision 2): K % 32 == 0, rows >= 64
bool gemm_f16_dma_supported(const GemmArgs& g);
int gemm_f16_dma_launch(const GemmArgs& g, hipStream_t st);
size_t split_f16_image_bytes(int Nout, int K);   // ceil(Nout / 128) * 128 * K * 2
// Per-sample folded fp16 weight images: img_b = the split_f16_tiled image of W * pa[b, :] (column scale), bias_out[b, n] = bias[n] + sum_k po[b, k]
// W[n, k] (fp32): a linear over A * pa + po as a linear over A itself.  img: B images of ceil(Nout / 128) * 128 * K halves; K % 32 == 0
int fold_f16_image_launch(const float* W, const float* bias, const float* pa, const float* po, void* img, float* bias_out, int B, int Nout, int K,
                          int ldw, hipStream_t st);
int split_f16_tiled_launch(const float* W, void* img, int Nout, int K, int ldw, hipStream_t st);
int split_f16_tiled_multi_launch(const SplitJobs& jobs, hipStream_t st);

// inducer_chain_f16.hip — fp16 mode: pool merge, pool.out_proj, norm_1, broadcast.mlp, norm_2 and the unpool k|v
// projection of the 64 inducers of every sample in one launch (one block per sample)
struct ChainArgs {
    const float* part_o;      // pool partials (B, H, nsplit, 64, hd)
    const float* part_ml;     // (B, H, nsplit, 64, 2)
    int nsplit, H;
    const float* w_stream;    // fp16 tiled images of pool.out_proj | mlp.0 | mlp.2 | unpool k|v, back to back
    int two_term;             // the images carry hi | lo blocks per column tile (SplitJob::pad_ = 8): two-term weights (mixed mode)
    const float *b0, *b2, *bkv, *alpha;
    int act;
    const float *n1_scale_w, *n1_scale_b, *n1_bias_w, *n1_bias_b;
    const float *n2_scale_w, *n2_scale_b, *n2_bias_w, *n2_bias_b;
    const float* t;
    int ctx_dim, G;
    float eps;
    float* h_out;             // (B, 64, C) fp32
    float* kvh;               // (B, 64, 2C) fp32
    int B;
    // the cluster form (C / 128 blocks per sample): exchange buffers (B, 64, C) fp32 x 2 and (B, 64, 2C) fp16, 8 counters per sample
    // (zeroed by the host before the launch; only ever counted up inside it)
    int cluster;
    float *x1, *x3;
    unsigned* xu;
    unsigned* flags;
    // mixed mode with the fused unpool + out_proj kernel: the last product's epilogue writes k | v straight as that kernel's fp16 image
    // (unpool_outproj_h8.hip: per (sample, head) K [64][hd + 8] | V^T [ceil(hd / 32) 32][72], key-permuted; `kv_img_bytes` per head, the pad
    // positions zeroed by the host) instead of fp32 kvh for kvh_image_kernel to reformat
    unsigned short* kv_img;
    int kv_img_bytes;
};
bool inducer_chain_f16_supported(int C, int Wd, int H, int G, int I);
int inducer_chain_f16_launch(const ChainArgs& g, int C, int Wd, hipStream_t st);

// mlp_fused_f16.hip — fp16 mode: x += mlp.2(act(mlp.0(AdaGN(x)))) + GroupNorm partials in one launch (128 rows per block)
struct MlpArgs {
    float* x;                 // (B, rows, C) fp32, updated in place
    const float *pro_a, *pro_o;   // (B, C) AdaGN coefficients of mlp_norm
    const float* w_stream;    // per hidden chunk j of 128: fp16 image of W0 tile j | W2[:, chunk j] K-half 0 | K-half 1
    const float *b0, *b2, *alpha;
    int act;
    float* stats;             // (B, rows / 128, 2, C) or null
    int B, rows;
    int stagger;              // cycles between the start offsets of the first blocks (see mlp_fused_f16.hip)
};
bool mlp_fused_f16_supported(int C, int Wd, int rows);
int mlp_fused_f16_launch(const MlpArgs& g, int C, int Wd, hipStream_t st);

// unpool_outproj_f16.hip — fp16 mode: unpool attention + out_proj + residual + GroupNorm partials in one launch
struct UnpoolProjArgs {
    float* x;                 // (B, rows, C) fp32 residual stream, updated in place
    const void* q16;          // head-major fp16 q (B, H, rows, hd)
    const float* kvh;         // (B, 64, 2C) fp32: k | v of the inducer states
    const float* w_stream;    // fp16 tiled image of out_proj.weight (C, C)
    const float* bias;        // (C) or null
    float* stats;             // (B, rows / 128, 2, C) or null
    int B, rows, H;
    int stagger;
};
bool unpool_outproj_f16_supported(int C, int H, int rows);
int unpool_outproj_f16_launch(const UnpoolProjArgs& g, int C, hipStream_t st);

// unpool_outproj_h8.hip — mixed mode: unpool attention + out_proj (h8 arithmetic) + residual + GroupNorm partials in one launch
struct UnpoolH8Args {
    float* x;                 // (B, rows, C) fp32 residual stream, updated in place
    const void* q16;          // head-major fp16 q (B, H, rows, hd)
    const void* kv_img;       // kvh_image_launch: per (sample, head) the fp16 LDS image of the inducers' k | v
    const void* w_img;        // h8 stream of out_proj.weight, 64-column tiles, attention k order (SplitJob::pad_ = 16)
    const float* bias;        // (C) or null
    float* stats;             // (B, rows / 128, 2, C) or null
    int B, rows, H;
    int rev;                  // set by the launcher: blocks walk the row panels last to first
    int stagger, pair;        // set by the launcher: start offset (s_memtime ticks) of every second block of a CU
};
bool unpool_outproj_h8_supported(int C, int H, int rows);
size_t unpool_outproj_h8_kv_bytes(int B, int C, int H);
int kvh_image_launch(const float* kvh, void* img, int B, int C, int H, hipStream_t st);   // kvh (B, 64, 2C) fp32
int unpool_outproj_h8_launch(const UnpoolH8Args& g, int C, hipStream_t st);

// mlp_fused_w.hip — "w2" mode: x (or out) = x + mlp.2(act(mlp.0(AdaGN(x)))) + GroupNorm partials in one launch: 4 waves of 512 registers per
// 128-row tile, the hidden layer kept as register fragments; fp16 products with fp6 block-scaled second terms (y and both weights two-term,
// the hidden layer one-term)
struct MlpWArgs {
    const float* x;           // (B, rows, C) fp32
    float* out;               // (B, rows, C) fp32; may be x (a block reads a row tile before it writes it)
    const float *pro_a, *pro_o;   // (B, C) AdaGN coefficients of mlp_norm
    const void* w_img;        // mlp_fused_w_image_launch: the layer's weight stream in the kernel's consumption order, biases included
    const float* alpha;
    int act;                  // 0 none, 1 / 2 GaussianActivation normalized / raw, 3 ReLU
    float* stats;             // (B, rows / 128, 2, C) or null
    int B, rows;
    float* dbg_u;             // diagnostics: (B, rows, width) pre-activations of mlp.0 x the activation's argument scale (1 but for act 1 / 2), or null
    int share;                // 1: another stream's kernels run beside this launch (two-stream evaluation): leave a quarter of the CUs to them
};
bool mlp_fused_w_supported(int C, int Wd, int rows);
size_t mlp_fused_w_image_bytes(int C, int Wd);
// (the stream depends on the activation: mlp.0's weights carry the Gaussian activation's argument scale)
struct MlpWImageJob { const float *W0, *b0, *W2, *b2; void* img; const float* alpha; };   // one layer's weights -> its stream
int mlp_fused_w_images_launch(const MlpWImageJob* jobs, int n, int C, int Wd, int act, hipStream_t st);   // all layers in one launch
int mlp_fused_w_image_launch(const float* W0, const float* b0, const float* W2, const float* b2, void* img, int C, int Wd, const float* alpha, int act,
                             hipStream_t st);
int mlp_fused_w_launch(const MlpWArgs& g, int C, int Wd, hipStream_t st);

// gemm_tn_x3.hip — split-bf16 weight gradients: C[g] = sum over the samples of group g of A[z]^T B[z]
struct TnArgs {
    const float* A;    // (Z, R, lda): dY, columns n
    const float* Bm;   // (Z, R, ldb): X, columns k
    float* C;          // (ceil(Z / group), N, K) partials
    int Z, R, N, K, lda, ldb;
    size_t sA, sB;     // sample strides (elements)
    int group;
    float* colsum;     // optional (ceil(Z / group), N): column sums of A per group (the bias gradient), or null
    int xcd;           // set by the launcher: XCD-aware block -> (tile, group) mapping
    // optional AdaGN apply on the B operand: B'[z, m, k] = B[z, m, k] * pro_a[z, k] + pro_o[z, k] (the weight gradient of a linear
    // whose input was AdaGN(x), from x itself: the normalised tensor is never materialised)
    const float* pro_a;
    const float* pro_o;
    int f16;           // 1: both operands rounded to fp16, one MFMA per product (the reference's autocast(float16) trainer arithmetic)
    int b_f16;         // (with f16) Bm is an fp16 tensor already: its tiles go to LDS as they are
    int a_f16;         // (with f16) A is an fp16 tensor already (the MLP backward's du, stored as halves by the dX product's epilogue)
    // (with f16; round 6) the fixed-order sum of the group partials inside the launch: `counters` (one zeroed unsigned per output tile;
    // the last group block to finish a tile — a device-wide ticket — adds the tile's partials in group order 0, 1, .. and resets the
    // counter) -> out (N, K) and colsum_out (N): the bits reduce_batch_kernel produces, without its launches
    unsigned* counters;
    float* out;
    float* colsum_out;
};
// gemm_tn_f16.hip: the same product with fp16 operands (TnArgs::f16 / b_f16), block tile chosen per shape
bool gemm_tn_f16_supported(const TnArgs& g);
int gemm_tn_f16_tiles(int N, int K);
int gemm_tn_f16_launch(const TnArgs& g, hipStream_t st);
// gemm_x3_areg.hip: split-bf16 GEMM whose A operand is a tiled split image loaded global -> registers (GemmArgs::a_img)
bool gemm_x3_areg_supported(const GemmArgs& g);
int gemm_x3_areg_launch(const GemmArgs& g, hipStream_t st);
bool gemm_tn_x3_supported(const TnArgs& g);
int gemm_tn_x3_launch(const TnArgs& g, hipStream_t st);

// gemm_f16_astat.hip — fp16 mode, A-stationary: AdaGN apply + fp16 rounding + all column tiles in one pass over x
// (fp32 A with optional prologue, fp16 outputs, one or two segments; K <= 384, full 128-tiles)
bool gemm_f16_astat_lo8_supported(int K);   // the fp8 form of the lo image exists for this K
bool gemm_f16_astat_supported(const GemmArgs& g);
int gemm_f16_astat_launch(const GemmArgs& g, hipStream_t st);

// gemm_h8_astat.hip — mixed mode's mlp.0: AdaGN apply + fp16 main product + two fp8 cross terms (2 matrix-pipe units per
// product at split-bf16 accuracy), A-stationary over 256-row blocks, output as the tiled split image (c_img)
size_t h8_image_bytes(int Nout, int K);   // Nout * K * 4 (Nout % 64 == 0)
int h8_image_multi_launch(const SplitJobs& jobs, hipStream_t st);   // K % 64 == 0, Nout % 64 == 0, ldw % 4 == 0 per job
bool gemm_h8_astat_supported(const GemmArgs& g);
// kv_proj | q_proj of the mixed mode on the same structure: fp16 outputs, w_img = the stream of kvq image jobs (SplitJob::pad_ =
// 1 | lo_begin << 8 | lo_end << 20, 64-column tiles with a second fp8 weight term); GemmArgs::lo_begin / lo_tiles in 64-column tiles
size_t kvq_image_bytes(int Nout, int K, int lo_cols);
bool gemm_kvq_astat_supported(const GemmArgs& g);
// head-aligned column order (GemmArgs::kvq_perm, SplitJob::pad_ | 64): head dim 48, both segments whole multiples of 384 columns
bool kvq_perm48_ok(int hm_hd, int K, int n_first, int n_second);
int gemm_kvq_astat_launch(const GemmArgs& g, hipStream_t st);
// gemm_h8_astat_kernel with fp32 outputs: the training forward in h8 arithmetic (w_img = the h8 stream)
bool gemm_h8_train_supported(const GemmArgs& g);
int gemm_h8_train_launch(const GemmArgs& g, hipStream_t st);
// the same structure with fp32 outputs for the training path (OUT forms of gemm_kvq_astat_kernel): w_img = the one-term kvq stream
bool gemm_astat_train_supported(const GemmArgs& g);
int gemm_astat_train_launch(const GemmArgs& g, hipStream_t st);
int gemm_h8_astat_launch(const GemmArgs& g, hipStream_t st);

// gemm_h8_areg.hip — mixed mode's mlp.2 / out_proj in h8 arithmetic: A is an h8 activation image (a_img == 2: fp16 hi + fp8 lo,
// written by gemm_h8_astat.hip with c_img == 2 or by the unpool attention with out_img == 2), W the 128-column-tile h8 stream
// (SplitJob::pad_ = 2); residual / bias / statistics epilogue of the LDS-DMA kernels
size_t h8_w128_image_bytes(int Nout, int K);
bool gemm_h8_areg_supported(const GemmArgs& g);
int gemm_h8_areg_launch(const GemmArgs& g, hipStream_t st);

// gemm_general_f32.hip — C[z] = scale * op(A[z]) op(B[z]) (+ bias), per-operand layout flag, two-level batch strides
struct GemmGeneralArgs {
    const float* A;   // a_kmajor ? (K, lda>=M) : (M, lda>=K)
    const float* B;   // b_kmajor ? (K, ldb>=N) : (N, ldb>=K)
    const float* bias;  // (N) or null
    float* C;         // (M, ldc>=N)
    int Z, zdiv;      // batch count; z = z1 * zdiv + z2
    int M, N, K;
    int lda, ldb, ldc;
    long long sA1, sA2, sB1, sB2, sC1, sC2;  // element strides of the outer / inner batch index
    int a_kmajor, b_kmajor;
    float scale;
};
int gemm_general_launch(const GemmGeneralArgs& g, hipStream_t st);
int reduce_batch_launch(const float* parts, float* out, size_t n, int Z, size_t stride, int accumulate, hipStream_t st);

// backward.hip
int softmax_fwd_launch(const float* S, float* P, size_t rows, int n, float scale, hipStream_t st);
int softmax_bwd_launch(const float* P, const float* dP, float* dS, size_t rows, int n, float scale, hipStream_t st);
int gauss_act_bwd_blocks(size_t n);
int gauss_act_bwd_launch(const float* u, const float* dy, const float* alpha, float* du, float* partial, size_t n,
                         int normalized, hipStream_t st);
int col_dot_stats_launch(const float* dy, const float* x, float* stats, int B, int rows, int C, hipStream_t st);
int adagn_bwd_coeffs_launch(const float* xstats, int Tx, const float* gstats, int Tg, int rows, const float* t,
                            int ctx_dim, const float* scale_w, const float* scale_b, float* cA, float* cB, float* cC,
                            float* ds, float* dz, int B, int C, int G, float eps, hipStream_t st);
int affine2_apply_launch(const float* dy, const float* x, const float* cA, const float* cB, const float* cC, float* dx,
                         int B, int rows, int C, hipStream_t st, const float* add = nullptr);
int adagn_param_grads_launch(const float* ds, const float* dz, const float* t, int B, int C, int ctx_dim,
                             float* d_scale_w, float* d_scale_b, float* d_bias_w, float* d_bias_b, hipStream_t st);
int lift_bwd_launch(const float* dY, const float* xin, float* partial, int B, int N, int C, hipStream_t st);
int lower_bwd_blocks(size_t rows);
int lower_bwd_launch(const float* feat, const float* dF, const float* W, float* dfeat, float* partial, size_t rows,
                     int C, float eps, hipStream_t st);

// attention_f32.hip
// precision 1 = split-bf16, 2 = fp16 arithmetic (attention_x3.hip) when the head dim allows, else the exact fp32 kernels
int pool_attn_launch(const float* KV, const float* inducers, float* part_o, float* part_ml, float* merged,
                     int B, int N, int C, int H, int I, int nsplit, hipStream_t st, int precision = 0, int io16 = 0,
                     int hm = 0);   // hm (with io16): K | V / q are head-major (GemmArgs::hm_hd)
int pool_attn_nsplit(int B, int N, int H);
int unpool_attn_launch(const float* q, const float* kvh, float* out, int B, int N, int C, int H, int I,
                       hipStream_t st, int precision = 0, int io16 = 0, int hm = 0, int out_img = 0);   // out_img: GemmArgs::a_img layout
// attention_bwd_f32.hip (training path)
int pool_attn_lse_launch(const float* part_ml, float* lse, int B, int H, int nsplit, hipStream_t st);
int pool_attn_bwd_nsplit(int B, int N, int H);
int unpool_attn_bwd_chunks(int B, int N, int H, int* tiles_per_wave);
int pool_attn_bwd_launch(const float* KV, const float* inducers, const float* merged, const float* lse, const float* dO,
                         float* dKV, float* dQpart, int B, int N, int C, int H, int I, int nsplit, hipStream_t st,
                         int precision = 0);
int unpool_attn_bwd_launch(const float* q, const float* kvh, const float* dO, float* dq, float* dkv_part, int B, int N, int C,
                           int H, int I, hipStream_t st, int precision = 0);
// attention_bwd_x3.hip: the same two kernels in split-bf16 arithmetic (head dims 16, 32, 48, 64)
bool attn_bwd_x3_supported(int HD);
int pool_attn_bwd_x3_launch(const float* KV, const float* inducers, const float* merged, const float* lse, const float* dO,
                            float* dKV, float* dQpart, int B, int N, int C, int H, int nsplit, hipStream_t st, int f16 = 0);
int unpool_attn_bwd_x3_launch(const float* q, const float* kvh, const float* dO, float* dq, float* dkv_part, int B, int N, int C,
                              int H, int tpw, int nchunk, hipStream_t st, int f16 = 0);   // f16: fp16 operands, one MFMA per product
// attention_x3.hip
bool attn_x3_supported(int HD);
int pool_attn_x3_partials_launch(const float* KV, const float* inducers, float* part_o, float* part_ml, int B, int N,
                                 int C, int H, int nsplit, hipStream_t st, int precision, int io16 = 0, int hm = 0);   // 1 split-bf16, 2 fp16
int unpool_attn_x3_launch(const float* q, const float* kvh, float* out, int B, int N, int C, int H, hipStream_t st,
                          int precision, int io16 = 0, int hm = 0, int out_img = 0);   // io16: KV / q / out are fp16 tensors (fp16 mode)

// lookup.hip
struct LookupArgs {
    int n_levels, c_total;
    int C[4], H[4], W[4];
    const float* feat[4];  // channels-last (B, H, W, C) per level
    int texel_f16;         // the levels hold fp16 texels (forward lookups only)
    int out_f16;           // ray_lookup_launch writes `out` as halves (statistics from the fp32 values); the network forward's "imgproj16"
    int reparam_kind;      // 0 none, 1 gaussian (mean, sigma), 2 UVL (uvl_mean, uvl_std, logit_scale)
    const float* rp_mean;
    const float* rp_std;
    float logit_scale;
};
int lookup_row_tile();
int cast_f16_launch(const float* src, void* dst, size_t n, hipStream_t st);
int ray_lookup_launch(const float* geom, const float* coef, const float* K, const LookupArgs& a, float* out,
                      float* stats, int B, int N, hipStream_t st);
int ray_lookup_taps_launch(const float* geom, const float* coef, const float* K, const LookupArgs& a, float* uv, int* x0, int* y0, float* wx1,
                           float* wy1, int B, int N, hipStream_t st);   // the lookup's coordinate chain alone (uv, taps per level)
int ray_lookup_dgeom_launch(const float* geom, const float* K, const LookupArgs& a, const float* dout, float* dgeom, float* dKpart, int B,
                            int N, hipStream_t st);   // gradients with respect to the geometry (B, N, 3) and (partials, (B, T, 4)) fx, cx, fy, cy
int ray_lookup_bwd_launch(const float* geom, const float* coef, const float* K, const LookupArgs& a,
                          float* const* dfeat, const float* dout, int B, int N, hipStream_t st);   // dfeat: zeroed, NHWC
// the same gradient by sort + gather (no atomics, fixed summation order, dfeat fully written: no zero fill needed)
bool ray_lookup_bwd_sorted_supported(const LookupArgs& a, int N);
size_t ray_lookup_bwd_sorted_ws_bytes(const LookupArgs& a, int B, int N);
int ray_lookup_bwd_sorted_launch(const float* geom, const float* coef, const float* K, const LookupArgs& a, float* const* dfeat,
                                 const float* dout, int B, int N, void* ws, hipStream_t st);
int bilinear_taps_launch(const float* uv, int Hh, int Ww, int* x0, int* y0, float* wx1, float* wy1, size_t n,
                         hipStream_t st);
int nchw_to_nhwc_launch(const float* src, float* dst, int B, int C, int Hh, int Ww, hipStream_t st);

// sampler.hip
int sampler_add_noise_f64_launch(const double* x_cur, const float* noise, size_t noise_step_stride, const double* sched,
                                 const int* step, int col, int sigma_col, double* x_out, float* x_in, float* sigma,
                                 size_t n, int B, hipStream_t st);
int sampler_add_noise_f32_launch(const float* x, const float* noise, size_t noise_step_stride, const double* sched,
                                 const int* step, int col, float* out, float* sigma, size_t n, int B, hipStream_t st);
int sampler_euler_launch(const double* x_hat, const float* den, const double* sched, const int* step, double* d_cur,
                         double* x_next, float* x_in, float* sigma, size_t n, int B, hipStream_t st);
int sampler_heun_launch(const double* x_hat, const double* x_next, const float* den, const double* d_cur,
                        const double* sched, const int* step, double* x_out, size_t n, hipStream_t st);
int sampler_advance_launch(int* step, int delta, hipStream_t st);
int sampler_scale_launch(const float* latents, double t, double* x, size_t n, hipStream_t st);
int gaussian_reparam_launch(const void* x, const float* mean, const float* sigma, void* y, size_t n, int dim,
                            int inverse, int is_f64, hipStream_t st);
int uvl_reparam_launch(const void* x, const float* K, const float* mean, const float* std_, double logit_scale, void* y,
                       int B, int N, int inverse, int is_f64, hipStream_t st);
int gaussian_act_launch(const float* x, const float* alpha, float* y, size_t n, int normalized, hipStream_t st);
int relu_launch(const float* x, float* y, size_t n, hipStream_t st);
int relu_bwd_launch(const float* y, const float* dy, float* du, size_t n, hipStream_t st);   // du = dy * (y > 0)

// pointwise.hip
int stats_row_tile(int rows);
int col_stats_launch(const float* x, float* stats, int B, int rows, int C, hipStream_t st);
int adagn_coeffs_launch(const float* stats, int T, int rows, const float* t, int ctx_dim, const float* scale_w,
                        const float* scale_b, const float* bias_w, const float* bias_b, float* a, float* o, int B,
                        int C, int G, float eps, hipStream_t st);
int affine_cast_f16_launch(const float* x, const float* a, const float* o, void* y16, int B, int rows, int C,
                           hipStream_t st);   // y16 = fp16(a * x + o), C % 8 == 0
int affine_apply_launch(const float* x, const float* a, const float* o, float* y, int B, int rows, int C,
                        hipStream_t st);
int edm_coeffs_launch(const float* sigma, float sigma_data, float* coef, int B, hipStream_t st);
int lift_launch(const float* x, const float* coef, const float* W, const float* bias, float* out, float* stats,
                int B, int N, int C, hipStream_t st);
int lower_edm_launch(const float* feat, const float* x, const float* coef, const float* W, const float* bias,
                     const float* gn_a, const float* gn_o, float* out, float* raw, int B, int N, int C, float eps,
                     hipStream_t st);

// optim.hip — Adam + EMA shadow weights in one pass over flat fp32 buffers (n % 4 == 0, 16-byte aligned)
struct AdamEmaArgs {
    float* p;          // parameters
    const float* g;    // gradients (summed over ranks when grad_scale = 1 / world)
    float* m;          // exp_avg
    float* v;          // exp_avg_sq
    float* ema;        // EMA shadow weights (read / written only when do_ema)
    size_t n;
    float beta2, eps, weight_decay;
    float w1, w2;      // 1 - beta1, 1 - beta2 (formed in double like torch's Python scalars, then rounded)
    float step_size;   // lr / (1 - beta1^step)
    float bc2_sqrt;    // sqrt(1 - beta2^step)
    float grad_scale, ema_decay, ema_w;   // ema_w = 1 - decay
    int do_ema;
    // torch.amp.GradScaler protocol (an optimizer with _step_supports_amp_scaling): all NULL / 0 outside of it.
    const float* amp_scale;   // device scalar: the loss scale the gradients carry (g is divided by it while it is read)
    const float* found_inf;   // device scalar: != 0 -> the step is SKIPPED (nothing is written; *skipped += 1)
    int* skipped;             // device counter of skipped steps: the bias corrections use step - *skipped
    double lr, beta1, beta2d; // for the on-device bias corrections once *skipped > 0
    int step;
};
int adam_ema_launch(const AdamEmaArgs& a, hipStream_t st);
int ema_update_launch(const float* p, float* ema, size_t n, double decay, hipStream_t st);

// metrics.hip — evaluation metrics on (B, N, 3) clouds (gecco-jax metrics.py:92-156, geometry.py:8-24)
int dist_matrix_launch(const float* A, const float* Bp, float* D, int B, int N, int M, int squared, hipStream_t st);
int nearest_dist_launch(const float* A, const float* Bp, float* mins, int B, int N, int M, int squared, hipStream_t st);
int row_mean_launch(const float* v, float* out, int B, int n, float scale, int accumulate, hipStream_t st);
int sinkhorn_step_launch(const float* C, float* f, float* g, int B, int N, int M, float eps, hipStream_t st);
// set-vs-set: out[s * ld_s + t * ld_t] (+)= scale * mean_i min_j d(a[s, i], b[t, j]) for every pair of S x T clouds; 1-NNA / MMD / COV
int set_nearest_mean_launch(const float* A, const float* Bp, float* out, int S, int T, int N, int M, int squared, int ld_s, int ld_t, float scale,
                            int accumulate, hipStream_t st);
int set_metrics_launch(const float* ss, const float* sd, const float* dd, int n, float* out, int* flags, hipStream_t st);
int sinkhorn_cost_launch(const float* C, const float* f, const float* g, float* rowcost, float* out, int B, int N, int M, float eps,
                         hipStream_t st);
// sampler.hip — inpainting: re-draw the known points of the fp64 state at the current noise level
int sampler_refresh_known_launch(double* x, const float* known, const float* noise, const double* sched, const int* step, int col,
                                 int m, int n_known, int B, hipStream_t st);

// convnext.hip — channels-last ConvNeXt conditioner pieces (the pointwise linears run on the fused GEMM)
// zout (optional): the LayerNorm's input as well (training).  dwconv: ln_w == null -> out is the plain convolution (bias optional)
int cnx_stem_launch(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b, float* out, float* zout,
                    int B, int H, int W, int C, float eps, hipStream_t st);
int cnx_dwconv_ln_launch(const float* x, const float* w, const float* bias, const float* ln_w, const float* ln_b, float* out,
                         float* zout, int B, int H, int W, int C, float eps, hipStream_t st, const float* addp = nullptr, int flip = 0);
int cnx_fold_scale_bwd_launch(const float* dWp, const float* dbp, const float* Wm, const float* b, const float* s, float* dW, float* db,
                              float* ds, int N, int K, hipStream_t st);
int cnx_ln_patch2_launch(const float* x, const float* ln_w, const float* ln_b, float* out, int B, int H, int W, int C, float eps,
                         hipStream_t st);
int cnx_fold_scale_launch(const float* Wm, const float* b, const float* s, float* Wo, float* bo, int N, int K, hipStream_t st);
// convnext_bwd.hip — the conditioner's backward (HBM-bound pieces; partial sums per block, reduced by reduce_batch)
int cnx_ln_bwd_blocks(int B, int H, int W, int C);   // rows of the (blocks, 3, C) partial buffer
int cnx_ln_bwd_launch(const float* z, const float* dy, const float* ln_w, float* dz, float* parts, int B, int H, int W, int C,
                      float eps, int patch2, hipStream_t st);
int cnx_dwconv_dw_blocks(int B, int H, int W, int C);   // rows of the (blocks, 49, C) partial buffer
int cnx_dwconv_dw_launch(const float* x, const float* dz, float* parts, int B, int H, int W, int C, hipStream_t st);
int gelu_launch(const float* u, float* y, size_t n, hipStream_t st);
int gelu_bwd_launch(const float* u, const float* dy, float* du, size_t n, hipStream_t st);
int cnx_im2col4_launch(const float* x, float* out, int B, int H, int W, hipStream_t st);
