"""ctypes binding of libgecco_hip.so (C ABI declared in include/gecco_hip.h).

`load()` fails loudly when the library is absent — the product path never falls back to a CPU or
eager-PyTorch implementation.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GECCO_HIP_LIB") or os.path.join(_HERE, "libgecco_hip.so")   # override: A/B builds only
ABI_VERSION = 14

c_f = C.c_void_p  # device pointers travel as void*


class GeccoAdaGN(C.Structure):
    _fields_ = [("scale_w", c_f), ("scale_b", c_f), ("bias_w", c_f), ("bias_b", c_f)]


class GeccoMLP(C.Structure):
    _fields_ = [("w0", c_f), ("b0", c_f), ("alpha", c_f), ("w2", c_f), ("b2", c_f)]


class GeccoLayer(C.Structure):
    _fields_ = [("broadcast_norm", GeccoAdaGN), ("inducers", c_f), ("kv_proj_w", c_f), ("pool_out_w", c_f),
                ("norm_1", GeccoAdaGN), ("bmlp", GeccoMLP), ("norm_2", GeccoAdaGN), ("in_proj_w", c_f),
                ("in_proj_b", c_f), ("unpool_out_w", c_f), ("unpool_out_b", c_f), ("mlp_norm", GeccoAdaGN),
                ("mlp", GeccoMLP)]


class GeccoSetTransformer(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("C", C.c_int), ("H", C.c_int), ("I", C.c_int), ("ctx_dim", C.c_int),
                ("G", C.c_int), ("width", C.c_int), ("act", C.c_int), ("precision", C.c_int), ("images_ready", C.c_int),
                ("opt_mask", C.c_uint), ("opt_vals", C.c_uint), ("layers", C.POINTER(GeccoLayer))]


class GeccoLinearLift(C.Structure):
    _fields_ = [("inner", GeccoSetTransformer), ("lift_w", c_f), ("lift_b", c_f), ("lower_w", c_f),
                ("lower_b", c_f), ("sigma_data", C.c_float)]


class GeccoPyramid(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("C", C.c_int * 4), ("H", C.c_int * 4), ("W", C.c_int * 4), ("feat", c_f * 4),
                ("texel_f16", C.c_int)]


class GeccoReparam(C.Structure):
    _fields_ = [("kind", C.c_int), ("mean", c_f), ("std", c_f), ("logit_scale", C.c_float)]


class GeccoRayNetwork(C.Structure):
    _fields_ = [("backbone", GeccoSetTransformer), ("xyz_w", c_f), ("xyz_b", c_f), ("img_w", c_f), ("img_b", c_f),
                ("out_w", c_f), ("out_b", c_f), ("reparam", GeccoReparam), ("sigma_data", C.c_float)]


class GeccoGemm(C.Structure):
    _fields_ = [("A", c_f), ("B", c_f), ("bias", c_f), ("C", c_f), ("Z", C.c_int), ("zdiv", C.c_int), ("M", C.c_int),
                ("N", C.c_int), ("K", C.c_int), ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
                ("sA1", C.c_longlong), ("sA2", C.c_longlong), ("sB1", C.c_longlong), ("sB2", C.c_longlong),
                ("sC1", C.c_longlong), ("sC2", C.c_longlong), ("a_kmajor", C.c_int), ("b_kmajor", C.c_int),
                ("scale", C.c_float)]


class GeccoSplitJob(C.Structure):
    _fields_ = [("W", c_f), ("img", C.c_void_p), ("Nout", C.c_int), ("K", C.c_int), ("ldw", C.c_int), ("transposed", C.c_int)]


class GeccoAdamEma(C.Structure):
    _fields_ = [("p", c_f), ("g", c_f), ("m", c_f), ("v", c_f), ("ema", c_f), ("n", C.c_size_t), ("lr", C.c_double),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double),
                ("ema_decay", C.c_double), ("grad_scale", C.c_float), ("step", C.c_int), ("do_ema", C.c_int)]


i, sz, vp, fl, db = C.c_int, C.c_size_t, C.c_void_p, C.c_float, C.c_double
PP = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); every symbol include/gecco_hip.h declares
SIGNATURES = {
    "gecco_abi_version": (i, []),
    "gecco_build_arch": (C.c_char_p, []),
    "gecco_last_error": (C.c_char_p, []),
    "gecco_linear_f32": (i, [vp] * 9 + [i, i, i, i, i, vp]),
    "gecco_linear_row_tiles": (i, [i]),
    "gecco_linear_ex_f32": (i, [vp] * 9 + [i, i, i, i, i, i, vp, vp]),
    "gecco_linear_pair_f32": (i, [vp, vp, vp, i, vp, vp, vp, i, vp, vp, vp, i, i, i, i, vp, vp]),
    "gecco_split_bf16_images_f32": (i, [C.POINTER(GeccoSplitJob), i, vp]),
    "gecco_split_bf16_image_bytes": (sz, [i, i]),
    "gecco_linear_image_ok": (i, [i, i, i, i]),
    "gecco_split_f16_images_f32": (i, [C.POINTER(GeccoSplitJob), i, vp]),
    "gecco_split_f16_image_bytes": (sz, [i, i]),
    "gecco_linear_image_ok_f16": (i, [i, i, i, i]),
    "gecco_linear_actbwd_ok": (i, [i, i, i, i]),
    "gecco_linear_actbwd_tiles": (sz, [i, i, i]),
    "gecco_linear_actbwd_f32": (i, [vp, vp, vp, vp, i, vp, vp, vp, i, i, i, i, i, vp, vp]),
    "gecco_linear_act_keep_f32": (i, [vp, vp, vp, vp, i, vp, vp, i, i, i, i, i, vp, vp]),
    "gecco_linear_act_keep_pro_f32": (i, [vp, vp, vp, vp, vp, vp, i, vp, vp, i, i, i, i, i, vp, vp]),
    "gecco_linear_f16io": (i, [vp] * 7 + [i, i, i, i, i, i, i, vp, vp]),
    "gecco_linear_pair_f16io": (i, [vp, vp, vp, i, vp, vp, vp, i, vp, i, i, i, vp, vp]),
    "gecco_mlp_fused_f16": (i, [vp, vp, vp, vp, vp, vp, vp, vp, i, vp, i, i, i, i, vp, vp]),
    "gecco_unpool_outproj_f16": (i, [vp, vp, vp, vp, vp, vp, i, i, i, i, vp, vp]),
    "gecco_unpool_outproj_h8": (i, [vp, vp, vp, vp, vp, vp, i, i, i, i, vp, vp]),
    "gecco_mlp_fused_w": (i, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, vp, i, i, i, i, vp, vp, vp]),
    "gecco_mlp_fused_w_wsplit_bytes": (sz, [i, i]),
    "gecco_unpool_outproj_h8_wsplit_bytes": (sz, [i, i, i]),
    "gecco_unpool_attn_h8img": (i, [vp, vp, vp, i, i, i, i, vp]),
    "gecco_gemm_tn_x3_f32": (i, [vp, vp, vp, i, i, i, i, i, vp]),
    "gecco_gemm_tn_x3_bias_f32": (i, [vp, vp, vp, vp, i, i, i, i, i, vp]),
    "gecco_gemm_tn_x3_pro_f32": (i, [vp, vp, vp, vp, vp, vp, i, i, i, i, i, vp]),
    "gecco_gemm_tn_f16_f32": (i, [vp, vp, vp, vp, vp, vp, i, i, i, i, i, vp]),
    "gecco_gemm_tn_f16_tiles": (i, [i, i]),
    "gecco_gemm_tn_f16_b16_f32": (i, [vp, vp, vp, vp, i, i, i, i, i, vp]),
    "gecco_gemm_tn_f16_a16_f32": (i, [vp, vp, vp, vp, vp, vp, i, i, i, i, i, vp]),
    "gecco_gemm_tn_f16_ex_f32": (i, [vp, i, vp, i, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, vp]),
    "gecco_linear_dotstats_a16_f32": (i, [vp] * 6 + [i, i, i, i, vp, vp]),
    "gecco_linear_astat16_actbwd_h16": (i, [vp] * 4 + [i, vp, vp, i, i, i, i, vp, vp]),
    "gecco_linear_dotstats_f32": (i, [vp] * 5 + [i, i, i, i, i, vp, vp]),
    "gecco_h8_image_bytes": (sz, [i, i]),
    "gecco_linear_h8_train_ok": (i, [i, i, i]),
    "gecco_h8_images_f32": (i, [C.POINTER(GeccoSplitJob), i, vp]),
    "gecco_linear_h8_train_f32": (i, [vp] * 5 + [i, vp, vp, vp, i, vp, vp, i, vp, i, i, i, vp, vp]),
    "gecco_astat16_image_bytes": (sz, [i, i]),
    "gecco_linear_astat16_ok": (i, [i, i, i]),
    "gecco_astat16_images_f32": (i, [C.POINTER(GeccoSplitJob), i, vp]),
    "gecco_linear_astat16_f32": (i, [vp] * 5 + [i, vp, vp, vp, i, vp, vp, i, i, i, i, vp, vp]),
    "gecco_linear_astat16_keep": (i, [vp] * 6 + [i, vp, vp, i, i, i, i, vp, vp]),
    "gecco_linear_astat16_keep_y16": (i, [vp] * 6 + [i, vp, vp, vp, i, i, i, i, vp, vp]),
    "gecco_linear_astat16_actbwd": (i, [vp] * 4 + [i, vp, vp, i, i, i, i, vp, vp]),
    "gecco_linear_act_keep_h16": (i, [vp] * 6 + [i, vp, vp, i, i, i, i, vp, vp]),
    "gecco_set_option": (i, [C.c_char_p, i]),
    "gecco_option_index": (i, [C.c_char_p]),
    "gecco_cast_f16": (i, [c_f, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gecco_linear_astat_f16": (i, [vp, vp, vp, vp, vp, i, vp, vp, vp, i, vp, vp, i, i, i, i, i, vp, vp]),
    "gecco_linear_kvq_f16": (i, [vp, vp, vp, vp, vp, i, vp, vp, vp, i, vp, i, i, i, i, i, i, vp, vp]),
    "gecco_linear_kvq_y16_f16": (i, [vp, vp, vp, vp, vp, i, vp, vp, vp, i, vp, vp, i, i, i, i, i, i, vp, vp]),
    "gecco_linear_h8_img_f32": (i, [vp, vp, vp, vp, vp, vp, i, vp, i, i, i, i, i, vp, vp]),
    "gecco_linear_h8_areg_f32": (i, [vp, vp, vp, vp, vp, vp, i, i, i, i, vp, vp]),
    "gecco_affine_cast_f16": (i, [vp, vp, vp, vp, i, i, i, vp]),
    "gecco_pool_attn_f16in": (i, [vp, vp, vp, i, i, i, i, i, i, vp, sz, vp]),
    "gecco_unpool_attn_f16io": (i, [vp, vp, vp, i, i, i, i, i, i, vp]),
    "gecco_col_stats_f32": (i, [vp, vp, i, i, i, vp]),
    "gecco_stats_row_tiles": (i, [i]),
    "gecco_adagn_coeffs_f32": (i, [vp, i, i, vp, i, C.POINTER(GeccoAdaGN), vp, vp, i, i, i, fl, vp]),
    "gecco_affine_apply_f32": (i, [vp, vp, vp, vp, i, i, i, vp]),
    "gecco_adagn_f32": (i, [vp, vp, i, C.POINTER(GeccoAdaGN), vp, i, i, i, i, fl, vp, sz, vp]),
    "gecco_adagn_workspace_bytes": (sz, [i, i, i]),
    "gecco_pool_attn_f32": (i, [vp, vp, vp, i, i, i, i, i, vp, sz, vp]),
    "gecco_pool_attn_ex_f32": (i, [vp, vp, vp, i, i, i, i, i, i, vp, sz, vp]),
    "gecco_pool_attn_workspace_bytes": (sz, [i, i, i, i, i]),
    "gecco_unpool_attn_f32": (i, [vp, vp, vp, i, i, i, i, i, vp]),
    "gecco_unpool_attn_ex_f32": (i, [vp, vp, vp, i, i, i, i, i, i, vp]),
    "gecco_pool_attn_lse_f32": (i, [vp, sz, vp, i, i, i, i, i, vp]),
    "gecco_pool_attn_bwd_partials": (i, [i, i, i]),
    "gecco_pool_attn_bwd_f32": (i, [vp] * 7 + [i, i, i, i, i, vp]),
    "gecco_unpool_attn_bwd_partials": (i, [i, i, i]),
    "gecco_unpool_attn_bwd_f32": (i, [vp] * 5 + [i, i, i, i, i, vp]),
    "gecco_pool_attn_bwd_ex_f32": (i, [vp] * 7 + [i, i, i, i, i, i, vp]),
    "gecco_unpool_attn_bwd_ex_f32": (i, [vp] * 5 + [i, i, i, i, i, i, vp]),
    "gecco_edm_coeffs_f32": (i, [vp, fl, vp, i, vp]),
    "gecco_lift_f32": (i, [vp, vp, vp, vp, vp, vp, i, i, i, vp]),
    "gecco_lower_edm_f32": (i, [vp] * 9 + [i, i, i, fl, vp]),
    "gecco_set_transformer_fwd_f32": (i, [C.POINTER(GeccoSetTransformer), vp, vp, vp, i, PP, PP, vp, i, i, vp, sz, vp]),
    "gecco_set_transformer_workspace_bytes": (sz, [C.POINTER(GeccoSetTransformer), i, i]),
    "gecco_linear_lift_fwd_f32": (i, [C.POINTER(GeccoLinearLift), vp, vp, vp, vp, PP, PP, i, i, vp, sz, vp]),
    "gecco_linear_lift_workspace_bytes": (sz, [C.POINTER(GeccoLinearLift), i, i]),
    "gecco_nchw_to_nhwc_f32": (i, [vp, vp, i, i, i, i, vp]),
    "gecco_bilinear_taps_f32": (i, [vp, i, i, vp, vp, vp, vp, sz, vp]),
    "gecco_ray_lookup_taps_f32": (i, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp]),
    "gecco_ray_lookup_f32": (i, [vp, vp, vp, C.POINTER(GeccoReparam), C.POINTER(GeccoPyramid), vp, vp, i, i, vp]),
    "gecco_lookup_row_tiles": (i, [i]),
    "gecco_ray_lookup_bwd_f32": (i, [vp, vp, vp, C.POINTER(GeccoReparam), C.POINTER(GeccoPyramid), vp, PP, i, i, vp]),
    "gecco_ray_lookup_dgeom_f32": (i, [vp, vp, C.POINTER(GeccoReparam), C.POINTER(GeccoPyramid), vp, vp, vp, i, i, vp]),
    "gecco_ray_lookup_bwd_sorted_workspace_bytes": (C.c_size_t, [C.POINTER(GeccoPyramid), i, i]),
    "gecco_ray_lookup_bwd_sorted_f32": (i, [vp, vp, vp, C.POINTER(GeccoReparam), C.POINTER(GeccoPyramid), vp, PP, i, i, vp,
                                            C.c_size_t, vp]),
    "gecco_ray_network_fwd_f32": (i, [C.POINTER(GeccoRayNetwork), vp, vp, vp, C.POINTER(GeccoPyramid), vp, vp, PP, PP,
                                      i, i, vp, sz, vp]),
    "gecco_ray_network_workspace_bytes": (sz, [C.POINTER(GeccoRayNetwork), C.POINTER(GeccoPyramid), i, i]),
    "gecco_gaussian_reparam": (i, [vp, vp, vp, vp, sz, i, i, i, vp]),
    "gecco_uvl_reparam": (i, [vp, vp, vp, vp, db, vp, i, i, i, i, vp]),
    "gecco_gaussian_act_f32": (i, [vp, vp, vp, sz, i, vp]),
    "gecco_relu_f32": (i, [vp, vp, sz, vp]),
    "gecco_relu_bwd_f32": (i, [vp, vp, vp, sz, vp]),
    "gecco_sampler_add_noise_f64": (i, [vp, vp, sz, vp, vp, i, i, vp, vp, vp, sz, i, vp]),
    "gecco_sampler_add_noise_f32": (i, [vp, vp, sz, vp, vp, i, vp, vp, sz, i, vp]),
    "gecco_sampler_euler_f64": (i, [vp, vp, vp, vp, vp, vp, vp, vp, sz, i, vp]),
    "gecco_sampler_heun_f64": (i, [vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "gecco_sampler_advance": (i, [vp, i, vp]),
    "gecco_sampler_scale_f64": (i, [vp, db, vp, sz, vp]),
    "gecco_gemm_f32": (i, [C.POINTER(GeccoGemm), vp]),
    "gecco_reduce_batch_f32": (i, [vp, vp, sz, i, sz, i, vp]),
    "gecco_softmax_fwd_f32": (i, [vp, vp, sz, i, fl, vp]),
    "gecco_softmax_bwd_f32": (i, [vp, vp, vp, sz, i, fl, vp]),
    "gecco_gauss_act_bwd_f32": (i, [vp, vp, vp, vp, vp, sz, i, vp]),
    "gecco_gauss_act_bwd_blocks": (i, [sz]),
    "gecco_col_dot_stats_f32": (i, [vp, vp, vp, i, i, i, vp]),
    "gecco_adagn_bwd_coeffs_f32": (i, [vp, i, vp, i, i, vp, i, C.POINTER(GeccoAdaGN), vp, vp, vp, vp, vp, i, i, i, fl, vp]),
    "gecco_affine2_apply_f32": (i, [vp, vp, vp, vp, vp, vp, i, i, i, vp]),
    "gecco_affine2_apply_add_f32": (i, [vp, vp, vp, vp, vp, vp, vp, i, i, i, vp]),
    "gecco_adagn_param_grads_f32": (i, [vp, vp, vp, i, i, i, vp, vp, vp, vp, vp]),
    "gecco_lift_bwd_f32": (i, [vp, vp, vp, i, i, i, vp]),
    "gecco_lower_bwd_f32": (i, [vp, vp, vp, vp, vp, sz, i, fl, vp]),
    "gecco_lower_bwd_blocks": (i, [sz]),
    "gecco_sampler_refresh_known_f64": (i, [vp, vp, vp, vp, vp, i, i, i, i, vp]),
    "gecco_distance_matrix_f32": (i, [vp, vp, vp, i, i, i, i, vp]),
    "gecco_chamfer_f32": (i, [vp, vp, vp, vp, i, i, i, i, vp]),
    "gecco_set_chamfer_f32": (i, [vp, vp, vp, i, i, i, i, i, vp]),
    "gecco_set_metrics_f32": (i, [vp, vp, vp, i, vp, vp, vp]),
    "gecco_sinkhorn_f32": (i, [vp, vp, vp, vp, vp, i, i, i, fl, i, vp]),
    "gecco_convnext_stem_f32": (i, [vp] * 6 + [i, i, i, i, fl, vp]),
    "gecco_convnext_dwconv_ln_f32": (i, [vp] * 6 + [i, i, i, i, fl, vp]),
    "gecco_convnext_ln_patch2_f32": (i, [vp] * 4 + [i, i, i, i, fl, vp]),
    "gecco_convnext_fold_scale_f32": (i, [vp] * 5 + [i, i, vp]),
    "gecco_convnext_stem_train_f32": (i, [vp] * 7 + [i, i, i, i, fl, vp]),
    "gecco_convnext_dwconv_ln_train_f32": (i, [vp] * 7 + [i, i, i, i, fl, vp]),
    "gecco_convnext_dwconv_f32": (i, [vp] * 4 + [i, i, i, i, vp]),
    "gecco_convnext_dwconv_bwd_f32": (i, [vp] * 4 + [i, i, i, i, vp]),
    "gecco_convnext_fold_scale_bwd_f32": (i, [vp] * 8 + [i, i, vp]),
    "gecco_convnext_ln_bwd_blocks": (i, [i, i, i, i]),
    "gecco_convnext_ln_bwd_f32": (i, [vp] * 5 + [i, i, i, i, fl, i, vp]),
    "gecco_convnext_dwconv_dw_blocks": (i, [i, i, i, i]),
    "gecco_convnext_dwconv_dw_f32": (i, [vp] * 3 + [i, i, i, i, vp]),
    "gecco_gelu_f32": (i, [vp, vp, C.c_size_t, vp]),
    "gecco_gelu_bwd_f32": (i, [vp, vp, vp, C.c_size_t, vp]),
    "gecco_convnext_im2col4_f32": (i, [vp, vp, i, i, i, vp]),
    "gecco_adam_ema_step_f32": (i, [C.POINTER(GeccoAdamEma), vp]),
    "gecco_adam_ema_step_amp_f32": (i, [C.POINTER(GeccoAdamEma), vp, vp, vp, vp]),
    "gecco_ema_update_f32": (i, [vp, vp, sz, db, vp]),
}

_lib = None


class GeccoHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the library (after torch, so its libamdhip64.so.7 dependency resolves to the HIP
    runtime torch already loaded) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GeccoHipError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
            "gecco_amd has no CPU fallback.")
    import torch  # noqa: F401  (loads the HIP runtime first)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.gecco_abi_version() != ABI_VERSION:
        raise GeccoHipError(f"ABI mismatch: library {lib.gecco_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().gecco_last_error().decode()
        raise GeccoHipError(f"{what} failed with code {rc}: {msg}")
