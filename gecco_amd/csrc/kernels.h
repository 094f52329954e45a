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
    int act;                // 0 none, 1 gaussian normalized, 2 gaussian raw
};

int gemm_row_tile(int rows);  // row-tile height the GEMM uses for `rows` rows per sample
int gemm_f32_launch(const GemmArgs& g, hipStream_t st);

// attention_f32.hip
int pool_attn_launch(const float* KV, const float* inducers, float* part_o, float* part_ml, float* merged,
                     int B, int N, int C, int H, int I, int nsplit, hipStream_t st);
int pool_attn_nsplit(int B, int N, int H);
int unpool_attn_launch(const float* q, const float* kvh, float* out, int B, int N, int C, int H, int I,
                       hipStream_t st);

// pointwise.hip
int stats_row_tile(int rows);
int col_stats_launch(const float* x, float* stats, int B, int rows, int C, hipStream_t st);
int adagn_coeffs_launch(const float* stats, int T, int rows, const float* t, int ctx_dim, const float* scale_w,
                        const float* scale_b, const float* bias_w, const float* bias_b, float* a, float* o, int B,
                        int C, int G, float eps, hipStream_t st);
int affine_apply_launch(const float* x, const float* a, const float* o, float* y, int B, int rows, int C,
                        hipStream_t st);
int edm_coeffs_launch(const float* sigma, float sigma_data, float* coef, int B, hipStream_t st);
int lift_launch(const float* x, const float* coef, const float* W, const float* bias, float* out, float* stats,
                int B, int N, int C, hipStream_t st);
int lower_edm_launch(const float* feat, const float* x, const float* coef, const float* W, const float* bias,
                     const float* gn_a, const float* gn_o, float* out, float* raw, int B, int N, int C, float eps,
                     hipStream_t st);
