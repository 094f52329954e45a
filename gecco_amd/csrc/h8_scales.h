// Power-of-two scales of the fp8 operands of the "h8" product (fp16 main product + two fp8 cross terms):
//
//       a w  =  ah wh  +  fp8(ah 2^-3) fp8(2^16 wl)  +  fp8(2^11 al) fp8(2^5 w)        ah = fp16(a), al = a - ah, wh = fp16(w), wl = w - wh
//
// e4m3 holds |v| <= 448 with 3 mantissa bits (normal from 2^-6, subnormal steps of 2^-9); a cross term is ~2^-12 of the product, so
// its operands need 4 bits — the scales only have to keep them INSIDE the format:
//   * al = a - fp16(a), |al| <= 2^-11 |a|:  2^11 al stays inside 448 up to |a| = 448 and is a normal fp8 down to |a| ~ 0.03 (below
//     that its absolute error, 2^-21, is nothing against the 2^-16 relative target of the sum); beyond 448 the lo term saturates and
//     that element degrades towards fp16 accuracy (2^-11), gradually;
//   * fp8(ah 2^-3): finite up to |a| = 3584.  The scaled conversion instructions (v_cvt_scalef32_pk_fp8_*) return NaN, not the
//     largest finite value, beyond the format (tools/probe/cvt_scalef32_probe.hip) — so every producer of an h8 operand CLAMPS the
//     value to +-3584 before it is split (h8_clamp): an activation beyond that is saturated, finite, and never a NaN in the product;
//   * w 2^5 and wl 2^16 (|wl| <= 2^-11 |w|): inside 448 up to |w| = 14 — weights of a trained network, outliers included; beyond
//     it the weight images saturate (clamp448 at image build): that weight's cross terms are then short, its main fp16 product is not.
// Round 3 used 2^14 / 2^0 / 2^8 / 2^19 (|a| <= 56 before the lo term saturated, NaN from |a| > 448, |w| <= 1.75).
#pragma once

constexpr int H8_AL_EXP = 11;   // fp8(2^11 (a - fp16(a)))
constexpr int H8_AH_EXP = 3;    // fp8(fp16(a) 2^-3)
constexpr int H8_W8_EXP = 5;    // fp8(2^5 w)
constexpr int H8_WL_EXP = 16;   // fp8(2^16 (w - fp16(w)))
constexpr float H8_AL_SCALE = 2048.f, H8_AH_DIV = 8.f, H8_W8_SCALE = 32.f, H8_WL_SCALE = 65536.f;
constexpr float H8_A_MAX = 3584.f;
// E8M0 scale operands of v_mfma_scale_f32_32x32x64_f8f6f4 (127 = 2^0)
constexpr int H8_SC_AH = 127 + H8_AH_EXP, H8_SC_WL = 127 - H8_WL_EXP, H8_SC_AL = 127 - H8_AL_EXP, H8_SC_W8 = 127 - H8_W8_EXP;

__device__ __forceinline__ float h8_clamp(float v) { return __builtin_amdgcn_fmed3f(v, -H8_A_MAX, H8_A_MAX); }
