"""ctypes binding of libmasic_hip.so (include/masic_hip.h).

The product path has no CPU fallback: if the shared library is missing or a symbol declared in
the header is not exported, importing this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MASIC_HIP_LIB") or os.path.join(_HERE, "lib", "libmasic_hip.so")   # override: kernel experiments (tools/)

c_int, c_float, c_double, c_size_t, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_size_t, ctypes.c_void_p

EB_PARAMS_PER_CHANNEL = 58
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SOFTMAX_C = 0, 1, 2, 3
INOP_NONE, INOP_ABS, INOP_ROUND = 0, 1, 2
PREC_F32, PREC_BF16, PREC_FP8 = 0, 1, 2


class ConvDesc(ctypes.Structure):
    """masic_conv_desc_t"""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "B", "Cin", "Hi", "Wi", "in_ctot", "in_coff",
        "Cout", "Ho", "Wo", "out_ctot", "out_coff",
        "KH", "KW", "stride", "pad", "transposed", "masked", "in_op", "act",
        "gate_ctot", "gate_c", "prec")]


class GemmGroup(ctypes.Structure):
    """masic_gemm_group_t"""
    _fields_ = [("x", c_void_p), ("w_packed", c_void_p), ("wscale", c_void_p), ("bias", c_void_p),
                ("y_f16k", c_void_p), ("y_f8k", c_void_p), ("y_nchw", c_void_p), ("out_inv_scale", c_float),
                ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32), ("out_ctot", ctypes.c_int32), ("out_coff", ctypes.c_int32),
                ("act", ctypes.c_int32)]


_P = c_void_p
# name -> (restype, argtypes); every symbol of include/masic_hip.h
SIGNATURES = {
    "masic_version": (c_int, []),
    "masic_last_error": (ctypes.c_char_p, []),
    "masic_conv_packed_bytes": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "masic_conv_variant": (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_int)]),
    "masic_conv_kernel_name": (c_int, [ctypes.POINTER(ConvDesc), ctypes.c_char_p, c_size_t]),
    "masic_conv_pack_weight": (c_int, [_P, _P, ctypes.POINTER(ConvDesc), _P]),
    "masic_conv2d_fwd": (c_int, [_P, _P, _P, _P, _P, ctypes.POINTER(ConvDesc), _P]),
    "masic_conv2d_fwd_ex": (c_int, [_P, _P, _P, _P, _P, _P, _P, ctypes.POINTER(ConvDesc), _P]),
    "masic_f16k_bytes": (c_size_t, [c_int, c_int, c_int]),
    "masic_nchw_to_f16k": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_nchw_to_f16k_op": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_gemm_f16k_packed_bytes": (c_size_t, [c_int, c_int]),
    "masic_gemm_f16k_pack_weight": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "masic_gemm_f16k_pack_weights": (c_int, [ctypes.POINTER(_P), ctypes.POINTER(_P), ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int, _P]),
    "masic_gemm_f16k_fwd": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 7 + [_P]),
    "masic_gemm_f16k_group_fwd": (c_int, [ctypes.POINTER(GemmGroup), c_int, c_int, c_int, _P]),
    "masic_conv_f16k_supported": (c_int, [_P]),
    "masic_conv_f16k_kernel_name": (c_int, [_P, c_int, ctypes.c_char_p, c_size_t]),
    "masic_conv_f16k_packed_bytes": (c_size_t, [_P]),
    "masic_conv_f16k_pack_weight": (c_int, [_P, _P, _P, _P]),
    "masic_conv_f16k_pack_job_bytes": (c_size_t, []),
    "masic_conv_f16k_pack_job": (c_int, [_P, _P, ctypes.POINTER(ConvDesc), _P]),
    "masic_conv_f16k_pack_jobs_run": (c_int, [_P, c_int, c_int, _P]),
    "masic_conv_f16k_fwd": (c_int, [_P] * 8),
    "masic_conv_f16k_d2s_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P]),
    "masic_pmf_to_quantized_cdf": (c_int, [_P, c_int, c_int, _P]),
    "masic_rans_encode_bound": (c_size_t, [c_int]),
    "masic_rans_encode_with_indexes": (c_int, [_P, _P, c_int, _P, c_int, _P, _P, c_int, _P, c_size_t, _P]),
    "masic_rans_decode_with_indexes": (c_int, [_P, c_size_t, _P, c_int, _P, c_int, _P, _P, c_int, _P]),
    "masic_homography_from_corners": (c_int, [_P, _P, _P, c_int, c_float, c_float, _P]),
    "masic_pair_prep": (c_int, [_P] + [c_int] * 6 + [_P] + [c_int] * 4 + [_P, _P]),
    "masic_gmm_cdf_rows": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_int, _P, c_int, c_int, c_float, _P, _P, _P, _P, _P]),
    "masic_rans_encode_freqs": (c_int, [_P, c_size_t, _P, c_size_t, _P]),
    "masic_skinny_ctx_packed_bytes": (c_size_t, [c_int, c_int]),
    "masic_skinny_ctx_pack_weight": (c_int, [_P, _P, c_int, c_int, _P]),
    "masic_skinny_group_fwd": (c_int, [_P, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "masic_rans_encode_channels": (c_int, [_P, c_int, c_int, _P, c_size_t, _P, _P]),
    "masic_gmm_cdf_rows_at": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, c_int, c_float, _P, _P, _P]),
    "masic_rans_decode_step": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P]),
    "masic_rans_decoder_open": (c_int, [_P, c_size_t, _P]),
    "masic_rans_decoder_decode_rows": (c_int, [_P, _P, c_int, c_int, _P]),
    "masic_rans_decoder_close": (None, [_P]),
    "masic_rans_decoder_decode_indexes": (c_int, [_P, _P, c_int, _P, c_int, _P, _P, c_int, _P]),
    "masic_gdn1_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_double, _P]),
    "masic_conv5s1_pair_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, c_int, c_double, _P, c_int, c_int, c_int, _P]),
    "masic_conv_a_packed_bytes": (c_size_t, []),
    "masic_conv_a_pack_weight": (c_int, [_P, _P, _P]),
    "masic_conv_a_gdn_fwd": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_gdn_f16k_packed_bytes": (c_size_t, []),
    "masic_gdn_pack_f16k": (c_int, [_P, _P, _P, c_int, c_double, _P]),
    "masic_conv_f16k_gdn_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "masic_conv_f16k_gdn_dual_fwd": (c_int, [_P, _P, _P, _P, c_int, _P, _P, ctypes.POINTER(ConvDesc), _P]),
    "masic_conv_a_gdn_dual_fwd": (c_int, [_P, _P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_conv_a_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_deconv_s2_as_conv_weight": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "masic_conv3x3_wgrad_f16k_workspace_bytes": (c_size_t, [c_int, c_int]),
    "masic_conv3x3_wgrad_f16k": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_gemm_wgrad_f16k": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "masic_gemm_wgrad_bias_f16k": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_conv3x3_wgrad_f16k_ws": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_pic_wgrad_f16k_workspace_bytes": (c_size_t, []),
    "masic_pic_wgrad_f16k": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_conv5x5_wgrad_f16k_workspace_bytes": (c_size_t, [c_int, c_int]),
    "masic_conv5x5_wgrad_f16k_ws": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_conv_f16k_set_stamps": (None, [_P]),
    "masic_mask2weights_en_fwd": (c_int, [_P] * 10 + [c_int, c_int, c_int, _P]),
    "masic_conv3x3_resident_packed_bytes": (c_size_t, [c_int, c_int]),
    "masic_conv3x3_resident_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "masic_conv3x3_resident_pack_weight": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "masic_conv3x3_resident_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, _P, c_float, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_conv3x3_resident_sum_workspace_bytes": (ctypes.c_size_t, []),
    "masic_conv3x3_resident_ex_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, _P, c_float, _P, _P, _P, c_float, _P, c_int, _P, _P] + [c_int] * 10 + [_P]),
    "masic_set_warp_align_corners": (None, [c_int]),
    "masic_get_warp_align_corners": (c_int, []),
    "masic_conv_f16k_res_ex_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, _P, c_float, _P, _P, ctypes.POINTER(ConvDesc), _P]),
    "masic_f16k_act_bwd": (c_int, [_P, _P, _P, c_size_t, c_float, _P]),
    "masic_f16k_channel_sum_workspace_bytes": (c_size_t, [c_int, c_int]),
    "masic_f16k_channel_sum": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "masic_f16k_act_bwd_sum": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P]),
    "masic_conv_f16k_res_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, _P, ctypes.POINTER(ConvDesc), _P]),
    "masic_conv_f16k_few_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, ctypes.POINTER(ConvDesc), _P]),
    "masic_f16k_gate": (c_int, [_P, _P, _P, _P] + [c_int] * 8 + [_P]),
    "masic_nchw_to_f16k_view": (c_int, [_P, _P] + [c_int] * 7 + [_P]),
    "masic_nchw_to_f16k_view_op": (c_int, [_P, _P] + [c_int] * 8 + [_P, c_int, c_int, _P]),
    "masic_f16k_to_nchw": (c_int, [_P, _P] + [c_int] * 7 + [_P]),
    "masic_f16k_to_nchw_bf16": (c_int, [_P, _P] + [c_int] * 7 + [_P]),
    # fp8 operand path
    "masic_f8k_bytes": (c_size_t, [c_int, c_int, c_int]),
    "masic_nchw_to_f8k": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, _P]),
    "masic_absmax": (c_int, [_P, c_size_t, c_int, _P, _P]),
    "masic_conv_f8k_pack_weight": (c_int, [_P, _P, _P, ctypes.POINTER(ConvDesc), _P]),
    "masic_conv_f8k_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_float, ctypes.POINTER(ConvDesc), _P]),
    "masic_gemm_f8k_packed_bytes": (c_size_t, [c_int, c_int]),
    "masic_gemm_f8k_pack_weight": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "masic_gemm_f8k_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_float] + [c_int] * 7 + [_P]),
    "masic_conv_a_gdn_fwd_ex": (c_int, [_P, _P, _P, _P, c_int, _P, _P, c_float, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_gemm1x1_packed_bytes": (c_size_t, [c_int, c_int]),
    "masic_gemm1x1_pack_weight": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "masic_gemm1x1_bf16_fwd": (c_int, [_P, _P, _P, _P, _P] + [c_int] * 7 + [_P]),
    "masic_gdn_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_double, _P]),
    "masic_gdn_fwd_ex": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_double, c_int, _P]),
    "masic_gdn_fwd_f16k": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_double, _P]),
    "masic_quantize_fwd": (c_int, [_P, _P, _P, _P] + [c_int] * 9 + [_P]),
    "masic_symbols_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "masic_entropy_bottleneck_fwd": (c_int, [_P] * 6 + [c_int] * 5 + [c_float, _P]),
    "masic_entropy_bottleneck_auxloss": (c_int, [_P, _P, _P, c_int, c_double, _P]),
    "masic_gmm_likelihood_fwd": (c_int, [_P] * 8 + [c_int] * 7 + [c_float, c_float, _P]),
    "masic_softmax_k_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "masic_warp_matrix": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_warp_perspective_fwd": (c_int, [_P, _P, _P] + [c_int] * 8 + [_P]),
    "masic_mul_inplace": (c_int, [_P, _P, c_size_t, _P]),
    "masic_lower_bound_fwd": (c_int, [_P, _P, c_float, c_size_t, _P]),
    "masic_lower_bound_bwd": (c_int, [_P, _P, _P, c_float, c_size_t, _P]),
    "masic_copy_view": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_reduce_workspace_bytes": (c_size_t, []),
    "masic_sum_log": (c_int, [_P, c_size_t, _P, _P, _P]),
    "masic_sse": (c_int, [_P, _P, c_size_t, _P, _P, _P]),
    "masic_rd_loss_workspace_bytes": (c_size_t, []),
    "masic_rd_loss": (c_int, [_P, _P, _P, _P, c_size_t, _P, _P, c_int, c_double, c_double, _P, _P, _P, _P, _P, _P, _P]),
    "masic_rd_loss_bwd": (c_int, [_P, _P, _P, _P, c_size_t, _P, _P, c_int, ctypes.c_float, ctypes.c_float, _P, _P, _P, _P, _P]),
    # backward
    "masic_conv2d_wgrad_workspace_bytes": (c_size_t, [ctypes.POINTER(ConvDesc)]),
    "masic_conv2d_wgrad": (c_int, [_P, _P, _P, _P, ctypes.POINTER(ConvDesc), _P]),
    "masic_conv2d_wgrad_ws": (c_int, [_P, _P, _P, _P, ctypes.POINTER(ConvDesc), c_int, _P]),
    "masic_conv2d_wgrad_bf16in_supported": (c_int, [ctypes.POINTER(ConvDesc)]),
    "masic_conv2d_wgrad_bf16in": (c_int, [_P, _P, _P, _P, ctypes.POINTER(ConvDesc), c_int, _P]),
    "masic_elementwise": (c_int, [_P, _P, _P, c_size_t, c_int, c_float, c_float, _P]),
    "masic_channel_sum_workspace_bytes": (c_size_t, [c_int]),
    "masic_channel_sum": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_slice_copy": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_gate_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "masic_softmax_k_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "masic_gdn_bwd_pre": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_int, _P]),
    "masic_gdn_bwd_post": (c_int, [_P, _P, _P, _P, c_size_t, _P]),
    "masic_gdn_bwd_small_workspace_bytes": (c_size_t, []),
    "masic_gdn_bwd_small": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_double, _P]),
    "masic_gdn_bwd_fused_workspace_bytes": (c_size_t, []),
    "masic_gdn_bwd_fused": (c_int, [_P] * 8 + [c_int] * 5 + [c_double, _P]),
    "masic_gdn_bwd_fused_ex": (c_int, [_P] * 12 + [c_int] * 5 + [c_double, _P]),
    "masic_gdn_bwd_fused_ex2": (c_int, [_P] * 13 + [c_int] * 5 + [c_double, _P]),
    "masic_gmm_likelihood_bwd": (c_int, [_P] * 10 + [c_int] * 6 + [c_float, c_float, _P]),
    "masic_entropy_bottleneck_bwd": (c_int, [_P] * 6 + [c_int] * 4 + [c_float, _P]),
    "masic_eb_table_split": (c_int, [_P, _P, c_int, ctypes.POINTER(c_int), c_int, _P]),
    "masic_entropy_bottleneck_auxloss_bwd": (c_int, [_P, _P, _P, c_int, c_double, c_float, _P]),
    "masic_entropy_bottleneck_aux_step": (c_int, [_P, _P, _P, _P, _P, c_int, _P, _P, c_int, _P]),
    "masic_warp_perspective_bwd": (c_int, [_P, _P, _P] + [c_int] * 6 + [_P]),
    "masic_warp_perspective_bwd_gather": (c_int, [_P, _P, _P, _P] + [c_int] * 6 + [_P]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"masic_amd: {LIB_PATH} not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C masic_amd/csrc`). There is no CPU fallback for the HIP path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc, what=""):
    if rc != 0:
        msg = lib.masic_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"masic_hip {what}: error {rc}: {msg}")
