// Launcher interface of the two EXPERIMENTAL split-bf16 kernels kept under tools/probe/kernels/ (not part of
// libgecco_hip.so): the persistent 8-phase GEMM on pre-split operand planes and the register-resident fused point MLP.
// Both are correct (tools/probe/x3g_probe.hip, mlpx3_probe.hip check them against fp64 host references) and neither
// beats the shipped 4-wave LDS-DMA GEMM end to end on MI355X — DESIGN.md section 6 has the measurements.
#pragma once
#include "../../../gecco_amd/csrc/kernels.h"

// gemm_x3_planes.hip — split-bf16 linear on PRE-SPLIT operands ("planes": per row K/32 blocks of [32 bf16 hi | 32 bf16 lo]),
// 8-wave / 8-phase LDS-DMA schedule.  Rows are flat (rows_total = B * rows_per_sample).
struct X3Args {
    const void* Y;          // activation planes (rows_total, K/32, 2, 32) bf16
    const void* Wimg;       // weight planes image (split_planes_image_launch), rows in the kernel's channel order
    const float* bias;      // (Nout) or null
    const float* alpha;     // GaussianActivation alpha (act 1 / 2)
    const float* residual;  // (rows_total, ldr) fp32 or null
    float* C;               // fp32 (rows_total, ldc); or, with c_planes, the output planes (rows_total, Nout/32, 2, 32)
    float* stats;           // (rows_total / gemm_x3_planes_row_tile(), 2, Nout) column sums / sums of squares, or null
    int rows_total, rows_per_sample, K, Nout, ldc, ldr, act, c_planes;
    float* C2;              // optional second output segment: channels [n_split, Nout) -> C2 (rows_total, ldc2), bias2
    const float* bias2;
    int n_split, ldc2;
    int skew;               // start-up skew per XCD index in s_sleep(127) units; < 0: the launcher's default
};
bool gemm_x3_planes_supported(const X3Args& g);
int gemm_x3_planes_row_tile(int Nout, int n_split);   // rows per statistics partial of the configuration picked (0: unsupported)
int gemm_x3_planes_launch(const X3Args& g, hipStream_t st);
size_t planes_image_bytes(int Nout, int K);           // ceil(Nout / 32) * 32 * K * 4
int split_planes_image_launch(const float* W, void* img, int Nout, int K, int ldw, hipStream_t st);
int split_planes_image_multi_launch(const SplitJobs& jobs, hipStream_t st);
// y planes = a[b, c] * x + o[b, c] (a == null: x itself), C % 32 == 0
int affine_split_planes_launch(const float* x, const float* a, const float* o, void* y, size_t rows_total, int rows_per_sample,
                               int C, hipStream_t st);

// mlp_x3_fused.hip — split-bf16 mode: x += mlp.2(act(mlp.0(a * x + o))) + GroupNorm partials in one launch; activations
// in registers (transposed products), only the weight stream image (mlp_x3_stream_launch) passes through LDS
struct MlpX3Args {
    float* x;                    // (rows_total, C) fp32, updated in place
    const float *pro_a, *pro_o;  // (B, C) AdaGN coefficients
    const void* w_stream;        // mlp_x3_stream_launch image of (W0, W2), mlp_x3_stream_bytes(C) bytes
    const float *b0, *b2, *alpha;
    int act;
    float* stats;                // (rows_total / mlp_x3_fused_row_tile(C), 2, C) or null
    int rows_total, rows_per_sample;
};
int mlp_x3_fused_row_tile(int C);
bool mlp_x3_fused_supported(int C, int Wd, int rows_per_sample);
size_t mlp_x3_stream_bytes(int C);
int mlp_x3_stream_launch(const float* W0, const float* W2, void* img, int C, hipStream_t st);
int mlp_x3_fused_launch(const MlpX3Args& g, int C, hipStream_t st);

