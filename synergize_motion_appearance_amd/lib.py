"""ctypes binding of libsmx.so (the C ABI declared in include/smx.h).

There is NO fallback: if the library is missing or was not built, importing this module's
`load()` raises -- the product path never routes through a CPU implementation."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libsmx.so")

ACT_NONE, ACT_RELU, ACT_LRELU02, ACT_SWISH, ACT_GELU, ACT_SIGMOID = 0, 1, 2, 3, 4, 5

_p, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64


class GemmDesc(C.Structure):
    """mirror of `smx_gemm_desc` (include/smx.h)."""
    _fields_ = [
        ("a", _p), ("a_bs0", _i64), ("a_bs1", _i64),
        ("bt", _p), ("bt_bs0", _i64), ("bt_bs1", _i64),
        ("c", _p), ("c_bs0", _i64), ("c_bs1", _i64),
        ("bias", _p),
        ("res", _p), ("res_bs0", _i64), ("res_bs1", _i64),
        ("nb0", C.c_int32), ("nb1", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32), ("ldres", C.c_int32),
        ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("up2", C.c_int32),
        ("act", C.c_int32), ("alpha", C.c_float),
        ("bias_per_row", C.c_int32),
        ("d2s_p", C.c_int32), ("d2s_c", C.c_int32),
        ("tile", C.c_int32),
        ("ksplit", C.c_int32),
        ("ws", _p),
    ]


class Gemm16Desc(C.Structure):
    """mirror of `smx_gemm16_desc` (include/smx.h)."""
    _fields_ = [
        ("a", _p), ("a_bs0", _i64), ("a_bs1", _i64),
        ("bt", _p), ("bt_bs0", _i64), ("bt_bs1", _i64),
        ("c", _p), ("c_bs0", _i64), ("c_bs1", _i64),
        ("bias", _p),
        ("res", _p), ("res_bs0", _i64), ("res_bs1", _i64),
        ("in_ss", _p),
        ("nb0", C.c_int32), ("nb1", C.c_int32),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32), ("ldres", C.c_int32),
        ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("up2", C.c_int32),
        ("act", C.c_int32), ("alpha", C.c_float),
        ("bias_per_row", C.c_int32),
        ("d2s_p", C.c_int32), ("d2s_c", C.c_int32),
        ("tile", C.c_int32),
        ("ksplit", C.c_int32),
        ("ws", _p),
        ("a_f32", C.c_int32), ("c_f32", C.c_int32), ("res_f32", C.c_int32), ("in_swish", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/smx.h declares
SIGNATURES = {
    "smx_version": (C.c_char_p, []),
    "smx_set_tuning": (_i, [C.c_char_p, _i]),
    "smx_get_tuning": (_i, [C.c_char_p, C.POINTER(_i)]),
    "smx_gemm_conv_f32": (_i, [C.POINTER(GemmDesc), _p]),
    "smx_gemm_conv_bf16": (_i, [C.POINTER(Gemm16Desc), _p]),
    "smx_conv3x3_bf16_t32_pack_elems": (_i64, [_i, _i]),
    "smx_conv3x3_bf16_t32_pack": (_i, [_p, _i, _p, _i, _i, _p]),
    "smx_conv3x3_bf16_t32": (_i, [_p, _i, _p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p]),
    "smx_conv3x3_sft_bf16_t32": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _f, _p, _i, _i, _i, _i, _i, _i, _p]),
    "smx_winograd_conv3x3_f32": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p]),
    "smx_winograd43_conv3x3_f32": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p]),
    "smx_winograd_conv3x3_sft_f32": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _f, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "smx_groupnorm_stats_f32": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p]),
    "smx_conv3x3_smalln_f32": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "smx_resize_taps_gather_f32": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "smx_resize_taps_combine_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_groupnorm_finalize_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "smx_groupnorm_apply_f32": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "smx_groupnorm_ws_floats": (_i64, [_i, _i, _i]),
    "smx_groupnorm_swish_nhwc_f32": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p, _p]),
    "smx_layernorm_pos_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p]),
    "smx_attention_f32": (_i, [_p, _i, _i64, _p, _i, _i64, _p, _i, _i64, _p, _i, _i64, _p, _i, _i, _i, _i, _i, _f, _p]),
    "smx_attention_f32_uses_bf3": (_i, [_i, _i, _i, _i, _i]),
    "smx_softmax_rows_f32": (_i, [_p, _i, _i, _i, _f, _p, _i, _p]),
    "smx_warp_nhwc_f32": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "smx_resize_bilinear_ac_nhwc_f32": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_avgpool2_nhwc_f32": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "smx_antialias_down_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_kp_head_f32": (_i, [_p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _f, _p]),
    "smx_normalize_kp_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _i, _i, _p]),
    "smx_normalize_kp_dscale_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _i, _i, _p]),
    "smx_sparse_motion_f32": (_i, [_p, _i, _p, _p, _p, _p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _f, _p]),
    "smx_mask_deformation_f32": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "smx_flow_to_residual_f32": (_i, [_p, _p, _i, _i, _i, _p]),
    "smx_flow_occ_update_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "smx_motion_ignore_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "smx_sft_combine_f32": (_i, [_p, _i, _p, _p, _p, _f, _i64, _i, _p]),
    "smx_fingerprint_f32": (_i, [_p, _i64, _p, _p]),
    "smx_add_f32": (_i, [_p, _p, _p, _i64, _p]),
    "smx_copy_slice_f32": (_i, [_p, _i, _p, _i, _i64, _i, _p]),
    "smx_nchw_to_nhwc_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "smx_nhwc_to_nchw_f32": (_i, [_p, _i, _p, _i, _i, _i, _i, _p]),
    "smx_frames_u8_to_nchw_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _p]),
    "smx_to_uint8_f32": (_i, [_p, _p, _i64, _f, _f, _p]),
    "smx_conv3x3_bf16": (_i, [_p, _i, _p, _i, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p]),
    "smx_conv3x3_sft_bf16": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _i, _f, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_conv3x3_smalln_mfma_pack_elems": (_i64, [_i, _i]),
    "smx_conv3x3_smalln_mfma_pack": (_i, [_p, _p, _i, _i, _p]),
    "smx_conv3x3_smalln_mfma_bf16": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "smx_conv7_c2_f32_pack": (_i, [_p, _p, _i, _p]),
    "smx_conv7_c2_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "smx_conv7_c2_bf16_pack": (_i, [_p, _p, _i, _p]),
    "smx_conv7_c2_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "smx_conv7_f32_pack": (_i, [_p, _p, _i, _i, _p]),
    "smx_conv7_f32": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_conv7_bf16x3_pack_elems": (_i64, [_i, _i]),
    "smx_conv7_bf16x3_pack": (_i, [_p, _p, _i, _i, _p]),
    "smx_conv7_f16_pack": (_i, [_p, _p, _i, _i, _p]),
    "smx_conv7_f16_f32": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_conv7_bf16x3_f32": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_gemm_rp_f32_ok": (_i, [_i64, _i, _i]),
    "smx_gemm_rp_f32_pack": (_i, [_p, _i, _p, _i, _i, _p]),
    "smx_gemm_rp_d2s_f32": (_i, [_p, _i, _p, _p, _p, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_gemm_rp_f32": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _i64, _i, _i, _i, _p]),
    "smx_gemm_rp_bf16_ok": (_i, [_i64, _i, _i]),
    "smx_gemm_rp_bf16_pack": (_i, [_p, _i, _p, _i, _i, _p]),
    "smx_gemm_rp_d2s_bf16": (_i, [_p, _i, _p, _p, _p, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_gemm_rp_bf16": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _i64, _i, _i, _i, _p]),
    "smx_conv3x3_mfma16_f32": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _i] + [_i] * 8 + [_p]),
    "smx_groupnorm_swish_nhwc_bf16": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p, _p]),
    "smx_groupnorm_stats_bf16": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p]),
    "smx_groupnorm_apply_bf16": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p]),
    "smx_layernorm_pos_bf16": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p]),
    "smx_attention_bf16": (_i, [_p, _i, _i64, _p, _i, _i64, _p, _i, _i64, _p, _i, _i64, _p, _i, _i, _i, _i, _i, _f, _p]),
    "smx_attnblock_bf16": (_i, [_p, _i, _i64, _p, _i, _i64, _p, _i, _i64, _p, _i, _i64, _i, _i, _i, _i, _f, _p]),
    "smx_attnblock_f32": (_i, [_p, _i, _i64, _p, _i, _i64, _p, _i, _i64, _p, _i, _i64, _i, _i, _i, _i, _f, _p]),
    "smx_softmax_rows_bf16": (_i, [_p, _i, _i, _i, _f, _p, _i, _p]),
    "smx_warp_nhwc_bf16": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "smx_resize_bilinear_ac_nhwc_bf16": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_resize_taps_gather_bf16": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "smx_resize_taps_combine_bf16": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_conv3x3_smalln_bf16": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "smx_sft_combine_bf16": (_i, [_p, _i, _p, _p, _p, _f, _i64, _i, _p]),
    "smx_add_bf16": (_i, [_p, _p, _p, _i64, _p]),
    "smx_convert_slice": (_i, [_p, _i, _i, _p, _i, _i, _i64, _i, _p]),
    "smx_nchw_to_nhwc_bf16": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "smx_nhwc_to_nchw_bf16": (_i, [_p, _i, _p, _i, _i, _i, _i, _p]),
    "smx_vq_ws_floats": (_i64, [_i]),
    "smx_vq_nearest_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    # ---- training step ----
    "smx_wgrad_ws_floats": (_i64, [_i, _i, _i, _i, C.POINTER(_i)]),
    "smx_wgrad_reduce_describe": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p]),
    "smx_wgrad_reduce_batch": (_i, [_p, _i, _i, _p]),
    "smx_wgrad_conv_ws_floats": (_i64, [_i] * 14 + [C.POINTER(_i)]),
    "smx_wgrad_f32": (_i, [_p, _i, _i64, _p, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i64, _i, _i, _i, _f, _p, _p]),
    "smx_wgrad_mfma16_f32": (_i, [_p, _i, _i64, _p, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i64, _i, _i, _i, _f, _p, _p]),
    "smx_colsum_ws_floats": (_i64, [_i64, _i]),
    "smx_colsum_f32": (_i, [_p, _i, _i64, _i, _p, _p, _i, _f, _p]),
    "smx_partial_reduce_f32": (_i, [_p, _i, _i, _p, _i, _f, _p]),
    "smx_pack_weight_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "smx_pack_weight_bf16": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "smx_antialias_nhwc_f32": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_antialias_nhwc_bwd_f32": (_i, [_p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_maxpool2_f32": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "smx_maxpool2_bwd_f32": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "smx_chan_affine_f32": (_i, [_p, _i, _p, _p, _p, _i, _i64, _i, _p]),
    "smx_gemm_rp_f16_pack_bytes": (_i64, [_i, _i]),
    "smx_gemm_rp_f16_pack": (_i, [_p, _i, _p, _i, _i, _p]),
    "smx_gemm_rp_f16": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _i64, _i, _i, _i, _p]),
    "smx_gemm_rp_d2s_f16": (_i, [_p, _i, _p, _p, _p, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_gemm_rp_bf3_ok": (_i, [_i64, _i, _i]),
    "smx_gemm_rp_bf3_pack_bytes": (_i64, [_i, _i]),
    "smx_gemm_rp_bf3_pack": (_i, [_p, _i, _p, _i, _i, _p]),
    "smx_gemm_rp_bf3": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _i64, _i, _i, _i, _p]),
    "smx_gemm_rp_d2s_bf3": (_i, [_p, _i, _p, _p, _p, _i, _i64, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_png_unfilter_u8": (_i, [_p, _i, _i, _i, _p]),
    "smx_winograd_f16_u_bytes": (_i64, [_i, _i]),
    "smx_winograd_f16_pack": (_i, [_p, _p, _i, _i, _p]),
    "smx_winograd_bf3_u_bytes": (_i64, [_i, _i]),
    "smx_winograd_bf3_pack": (_i, [_p, _p, _i, _i, _p]),
    "smx_winograd_bf3_shape_ok": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "smx_winograd_bf3_conv3x3_f32": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p]),
    "smx_winograd_bf3_conv3x3_sft_f32": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _f, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "smx_winograd_u_floats": (_i64, [_i, _i]),
    "smx_pack_winograd_u_f32": (_i, [_p, _p, _i, _i, _i, _p]),
    "smx_pack_batch": (_i, [_p, _i, _i, _p]),
    "smx_transpose_f32": (_i, [_p, _i, _i64, _p, _i, _i64, _i, _i, _i, _p]),
    "smx_act_f32": (_i, [_p, _i, _p, _i, _i64, _i, _i, _p]),
    "smx_act_bwd_f32": (_i, [_p, _i, _p, _i, _p, _i, _i64, _i, _i, _p]),
    "smx_axpy_slice_f32": (_i, [_p, _i, _p, _i, _i64, _i, _f, _p]),
    "smx_groupnorm_stats_train_f32": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p]),
    "smx_groupnorm_bwd_f32": (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "smx_layernorm_bwd_ws_floats": (_i64, [_i, _i]),
    "smx_layernorm_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p]),
    "smx_batch_sum_f32": (_i, [_p, _p, _i, _i64, _p]),
    "smx_softmax_rows_bwd_f32": (_i, [_p, _p, _i64, _i, _f, _p]),
    "smx_attention_bwd_f32": (_i, [_p, _i, _i64, _p, _i, _i64, _p, _i, _i64, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "smx_warp_bwd_f32": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "smx_resize_ac_bwd_f32": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "smx_space_to_depth_f32": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "smx_vq_bwd_f32": (_i, [_p, _p, _p, _p, _p, _f, _p, _p, _i64, _i, _p]),
    "smx_flow_occ_update_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "smx_sft_combine_bwd_f32": (_i, [_p, _p, _i, _p, _p, _p, _p, _f, _i64, _i, _p]),
    "smx_l1_loss_f32": (_i, [_p, _p, _i64, _f, _p, _p, _p]),
    "smx_l1_loss_bwd_f32": (_i, [_p, _p, _p, _i64, _f, _p, _i, _p]),
    "smx_scale_f32": (_i, [_p, _p, _i64, _f, _i, _p]),
    "smx_adam_step_f32": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i, _f, _p]),
    "smx_ema_f32": (_i, [_p, _p, _i64, _f, _p]),
    "smx_batchnorm_ws_floats": (_i64, [_i64, _i]),
    "smx_batchnorm_train_f32": (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _p, _i64, _i, _f, _f, _i, _p, _p]),
    "smx_batchnorm_train_bwd_f32": (_i, [_p, _i, _p, _i, _p, _i, _p, _p, _p, _i, _p, _p, _i64, _i, _p, _p]),
    "smx_avgpool2_bwd_f32": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "smx_kp_head_bwd_f32": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _f, _p]),
    "smx_sparse_motion_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "smx_tps_transform_frame_f32": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "smx_mask_deformation_bwd_f32": (_i, [_p, _i, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _p]),
}

_lib = None


class SmxError(RuntimeError):
    pass


def load():
    """dlopen libsmx.so and bind every entry point; raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmxError(f"{LIB_PATH} not found: build it with `python -m synergize_motion_appearance_amd.build` "
                       "(__graft_entry__.build()). There is no CPU fallback for the HIP path.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the one
    # already resident when libsmx.so (NEEDED libamdhip64.so.7) is dlopen'ed, otherwise the system
    # runtime is pulled in first, torch then shares it with its bundled HSA and no device is found.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise SmxError(f"{what} failed with code {rc}")
