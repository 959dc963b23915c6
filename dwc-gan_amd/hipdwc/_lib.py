"""ctypes binding of libdwcgan_hip.so (C ABI declared in include/dwcgan_hip.h).

There is deliberately no fallback: if the shared library is missing or a kernel call
returns an error code, the caller gets an exception.  Nothing here touches the CPU oracle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DWC_HIP_LIB: another build of the SAME library (benchmarks: A/B runs of two builds inside one gpurun call); it must export every
# symbol of SIGNATURES like the default one, and a missing file is an error, never a fallback
LIB_PATH = os.environ.get("DWC_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libdwcgan_hip.so")

c_fp = ctypes.c_void_p   # device pointers travel as void*
c_int = ctypes.c_int
c_sz = ctypes.c_size_t
c_f = ctypes.c_float
c_u = ctypes.c_uint


class AdvSpec(ctypes.Structure):               # struct dwc_adv_spec (passed by value)
    _fields_ = [("target", c_f * 4), ("w_src", c_f * 4), ("w_cls", c_f * 4)]


# name -> (restype, argtypes): mirrors include/dwcgan_hip.h one to one
SIGNATURES = {
    "dwc_version": (c_int, []),
    "dwc_x3_gemm_mode": (c_int, [c_int]),
    "dwc_weight_prepared_elems": (c_sz, [c_int] * 8),
    "dwc_weight_prepare_fwd": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_weight_prepare_dgrad": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_conv2d_fwd_ws_bytes": (c_sz, [c_int] * 9),
    "dwc_conv2d_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp] + [c_int] * 10 + [c_fp, c_sz, c_fp]),
    "dwc_conv2d_bwd_data_ws_bytes": (c_sz, [c_int] * 9),
    "dwc_conv2d_bwd_data": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 9 + [c_fp, c_sz, c_fp]),
    "dwc_conv2d_bwd_data_fold": (c_int, [c_fp, c_fp, c_fp, c_fp] + [c_int] * 9 + [c_fp, c_sz, c_fp]),
    "dwc_reflect_pad_adjoint": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_conv2d_bwd_data_same_ws_bytes": (c_sz, [c_int] * 8),
    "dwc_conv2d_bwd_data_same": (c_int, [c_fp] * 4 + [c_int] * 8 + [c_fp, c_sz, c_fp]),
    "dwc_conv2d_bwd_data_ring": (c_int, [c_fp] * 4 + [c_int] * 8 + [c_fp, c_sz, c_fp]),
    "dwc_conv2d_bwd_data_image_ws_bytes": (c_sz, [c_int] * 7),
    "dwc_conv2d_bwd_data_image": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 7 + [c_fp, c_sz, c_fp]),
    "dwc_conv2d_fwd_zeropad": (c_int, [c_fp, c_fp, c_fp, c_fp] + [c_int] * 10 + [c_fp, c_sz, c_fp]),
    "dwc_conv2d_bwd_data_zeropad_ws_bytes": (c_sz, [c_int] * 8),
    "dwc_conv2d_bwd_data_zeropad": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 8 + [c_fp, c_sz, c_fp]),
    "dwc_maxpool2_fwd": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_maxpool2_bwd": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_conv2d_bwd_weight_ws_bytes": (c_sz, [c_int] * 9),
    "dwc_conv2d_bwd_weight": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 11 + [c_fp, c_sz, c_fp]),
    "dwc_conv2d_fwd_ex": (c_int, [c_fp, c_fp, c_fp, c_fp] + [c_int] * 12 + [c_fp]),
    "dwc_conv2d_bwd_weight_ex_ws_bytes": (c_sz, [c_int] * 11),
    "dwc_conv2d_bwd_weight_ex": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 13 + [c_fp, c_sz, c_fp]),
    "dwc_act_bwd_bias_ws_bytes": (c_sz, [c_int, c_int]),
    "dwc_act_bwd_bias": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "dwc_linear_small_ok": (c_int, [c_int] * 3),
    "dwc_linear_small_fwd": (c_int, [c_fp] * 4 + [c_int] * 4 + [c_fp]),
    "dwc_linear_small_bwd": (c_int, [c_fp] * 7 + [c_int] * 3 + [c_fp]),
    "dwc_instnorm_ws_bytes": (c_sz, [c_int, c_int, c_int]),
    "dwc_instnorm_fwd": (c_int, [c_fp] * 7 + [c_int, c_int, c_int, c_f, c_int, c_fp, c_sz, c_fp]),
    "dwc_instnorm_bwd": (c_int, [c_fp] * 9 + [c_int, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "dwc_layernorm_ws_bytes": (c_sz, [c_int, c_int, c_int]),
    "dwc_layernorm_fwd": (c_int, [c_fp] * 6 + [c_int, c_int, c_int, c_f, c_int, c_fp, c_sz, c_fp]),
    "dwc_layernorm_bwd": (c_int, [c_fp] * 9 + [c_int, c_int, c_int, c_f, c_int, c_fp, c_sz, c_fp]),
    "dwc_upsample2x_fwd": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_upsample2x_bwd": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_avgpool2_fwd": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_avgpool2_bwd": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_pack_nchw_to_nhwc4": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_unpack_nhwc4_to_nchw": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_blend_fwd": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp]),
    "dwc_blend_bwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_fp]),
    "dwc_l1_ws_bytes": (c_sz, [c_sz]),
    "dwc_gmm_kl_sp_fwd": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_f, c_fp, c_fp]),
    "dwc_gmm_kl_sp_bwd": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_f, c_fp, c_fp, c_fp, c_fp]),
    "dwc_l1_mean_fwd": (c_int, [c_fp, c_fp, c_fp, c_sz, c_int, c_fp, c_sz, c_fp]),
    "dwc_l1_mean_bwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_sz, c_int, c_fp]),
    "dwc_lstm_fwd": (c_int, [c_fp] * 6 + [c_int] * 4 + [c_fp]),
    "dwc_lstm_bwd": (c_int, [c_fp] * 8 + [c_int] * 4 + [c_fp]),
    "dwc_adam_step": (c_int, [c_fp, c_fp, c_fp, c_fp, c_sz, c_f, c_f, c_f, c_f, c_f, c_int, c_fp]),
    "dwc_ema_lerp": (c_int, [c_fp, c_fp, c_sz, c_f, c_fp]),
    "dwc_adam_multi": (c_int, [c_fp, c_fp, c_fp, c_int] + [ctypes.c_double] * 4 + [c_fp]),
    "dwc_ema_multi": (c_int, [c_fp, c_fp, c_fp, c_int, c_f, c_fp]),
    # ---- bf16-activation path (same argument lists as the fp32 entry points of the same name) ----
    "dwc_bf16_weight_prepared_elems": (c_sz, [c_int] * 8),
    "dwc_bf16_weight_prepare_fwd": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_bf16_weight_prepare_dgrad": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_bf16_conv2d_fwd_ws_bytes": (c_sz, [c_int] * 9),
    "dwc_bf16_conv2d_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp] + [c_int] * 10 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_conv2d_fwd_ex": (c_int, [c_fp, c_fp, c_fp, c_fp] + [c_int] * 12 + [c_fp]),
    "dwc_bf16_conv2d_bwd_data_ws_bytes": (c_sz, [c_int] * 9),
    "dwc_bf16_conv2d_bwd_data": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 9 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_conv2d_bwd_data_fold": (c_int, [c_fp, c_fp, c_fp, c_fp] + [c_int] * 9 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_reflect_pad_adjoint": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_bf16_reflect_pad_adjoint_band": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_bf16_conv2d_bwd_data_same_ws_bytes": (c_sz, [c_int] * 8),
    "dwc_bf16_conv2d_bwd_data_same": (c_int, [c_fp] * 4 + [c_int] * 8 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_conv2d_bwd_data_ring": (c_int, [c_fp] * 4 + [c_int] * 8 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_conv2d_same_halo_ok": (c_int, [c_int] * 6),
    "dwc_bf16_conv2d_same_halo": (c_int, [c_fp] * 4 + [c_int] * 8 + [c_fp]),
    "dwc_bf16_conv2d_same_halo_add": (c_int, [c_fp] * 5 + [c_int] * 8 + [c_fp]),
    "dwc_bf16_conv2d_bwd_data_same_fused_ok": (c_int, [c_int] * 6),
    "dwc_bf16_conv2d_bwd_data_same_fused": (c_int, [c_fp] * 4 + [c_int] * 6 + [c_fp]),
    "dwc_weight_refresh_multi": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_u, c_fp]),
    "dwc_bf16_conv2d_fwd_zeropad": (c_int, [c_fp, c_fp, c_fp, c_fp] + [c_int] * 10 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_conv2d_bwd_data_zeropad_ws_bytes": (c_sz, [c_int] * 8),
    "dwc_bf16_conv2d_bwd_data_zeropad": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 8 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_maxpool2_fwd": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_bf16_maxpool2_bwd": (c_int, [c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_lstm_seq_ws_bytes": (c_sz, [c_int, c_int]),
    "dwc_lstm_seq_fwd": (c_int, [c_fp] * 6 + [c_int] * 4 + [c_fp, c_sz, c_fp, c_int, c_fp]),
    "dwc_lstm_seq_bwd": (c_int, [c_fp] * 7 + [c_int] * 4 + [c_fp, c_sz, c_fp, c_int, c_fp]),
    "dwc_adv_tail_fwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, AdvSpec, c_fp]),
    "dwc_adv_tail_bwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_int, AdvSpec, c_fp]),
    "dwc_bf16_conv2d_s2_halo_ok": (c_int, [c_int] * 5),
    "dwc_bf16_conv2d_s2_halo": (c_int, [c_fp] * 4 + [c_int] * 6 + [c_fp]),
    "dwc_x3_conv2d_same_ok": (c_int, [c_int] * 6),
    "dwc_x3_weight_prepared_elems": (c_sz, [c_int] * 3),
    "dwc_x3_weight_prepare": (c_int, [c_fp, c_fp] + [c_int] * 5 + [c_fp]),
    "dwc_x3_conv2d_same": (c_int, [c_fp] * 4 + [c_int] * 9 + [c_fp]),
    "dwc_x3_conv2d_same_add": (c_int, [c_fp] * 5 + [c_int] * 9 + [c_fp]),
    "dwc_x3_conv2d_narrow_ok": (c_int, [c_int] * 8),
    "dwc_x3_conv2d_narrow_weight_elems": (c_sz, [c_int] * 2),
    "dwc_x3_conv2d_narrow": (c_int, [c_fp] * 4 + [c_int] * 12 + [c_fp]),
    "dwc_reflect_pad_adjoint_pitch": (c_int, [c_fp, c_fp] + [c_int] * 6 + [c_fp]),
    "dwc_x3_gather_split": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp]),
    "dwc_x3_conv2d_stem_ok": (c_int, [c_int] * 7),
    "dwc_x3_conv2d_stem_weight_elems": (c_sz, []),
    "dwc_x3_conv2d_stem": (c_int, [c_fp] * 4 + [c_int] * 9 + [c_fp]),
    "dwc_x3_conv2d_stem_crop": (c_int, [c_fp] * 5 + [c_int] * 10 + [c_fp]),
    "dwc_x3_conv2d_stem_amax": (c_int, [c_fp] * 5 + [c_u] + [c_int] * 9 + [c_fp]),
    "dwc_reflect_pad_adjoint_band": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_x3_conv2d_ksplit_ws_bytes": (c_sz, [c_int] * 7),
    "dwc_x3_conv2d_ksplit_ticket_words": (c_int, []),
    "dwc_x3_conv2d_same_add_ws": (c_int, [c_fp] * 5 + [c_int] * 9 + [c_fp, c_sz, c_fp, c_fp]),
    "dwc_x3_conv2d_s2_ws": (c_int, [c_fp] * 4 + [c_int] * 7 + [c_fp, c_sz, c_fp, c_fp]),
    "dwc_x3_conv2d_s2_ok": (c_int, [c_int] * 5),
    "dwc_x3_conv2d_s2": (c_int, [c_fp] * 4 + [c_int] * 7 + [c_fp]),
    "dwc_x3_conv2d_s2_bwd_data_ok": (c_int, [c_int] * 5),
    "dwc_x3_conv2d_s2_bwd_data": (c_int, [c_fp] * 3 + [c_int] * 6 + [c_fp]),
    "dwc_conv2d_bwd_data_s2_ring": (c_int, [c_fp] * 4 + [c_int] * 5 + [c_fp]),
    "dwc_bf16_conv2d_s2_halo_bwd_data_ok": (c_int, [c_int] * 5),
    "dwc_bf16_conv2d_s2_halo_bwd_data": (c_int, [c_fp] * 3 + [c_int] * 5 + [c_fp]),
    "dwc_bf16_conv2d_s2_halo_bwd_data_fused": (c_int, [c_fp] * 3 + [c_int] * 5 + [c_fp]),
    "dwc_bf16_conv2d_bwd_data_s2_ring": (c_int, [c_fp] * 4 + [c_int] * 5 + [c_fp]),
    "dwc_x3_conv2d_wgrad_ws_bytes": (c_sz, [c_int] * 6),
    "dwc_x3_conv2d_wgrad": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 8 + [c_fp, c_sz, c_fp]),
    "dwc_absmax": (c_int, [c_fp, c_sz, c_fp, c_u, c_fp]),
    "dwc_instnorm_fwd_amax": (c_int, [c_fp] * 7 + [c_int, c_int, c_int, c_f, c_int, c_fp, c_sz, c_fp, c_u, c_fp]),
    "dwc_instnorm_bwd_amax": (c_int, [c_fp] * 9 + [c_int, c_int, c_int, c_int, c_fp, c_sz, c_fp, c_u, c_fp]),
    "dwc_layernorm_fwd_amax": (c_int, [c_fp] * 6 + [c_int, c_int, c_int, c_f, c_int, c_fp, c_sz, c_fp, c_u, c_fp]),
    "dwc_layernorm_bwd_amax": (c_int, [c_fp] * 9 + [c_int, c_int, c_int, c_f, c_int, c_fp, c_sz, c_fp, c_u, c_fp]),
    "dwc_act_bwd_bias_amax": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_sz, c_fp, c_u, c_fp]),
    "dwc_h2_weight_prepared_elems": (c_sz, [c_int] * 3),
    "dwc_h2_weight_prepare": (c_int, [c_fp, c_fp] + [c_int] * 5 + [c_fp, c_u, c_fp]),
    "dwc_h2_conv2d_same_add_ws": (c_int, [c_fp, c_fp, c_u] + [c_fp] * 4 + [c_fp, c_u] + [c_int] * 9 + [c_fp, c_sz, c_fp, c_fp]),
    "dwc_h2_conv2d_bwd_data_same_fused": (c_int, [c_fp, c_fp, c_u] + [c_fp] * 3 + [c_int] * 7 + [c_fp, c_sz, c_fp, c_fp]),
    "dwc_h2_conv2d_s2_ws": (c_int, [c_fp, c_fp, c_u] + [c_fp] * 3 + [c_fp, c_u] + [c_int] * 7 + [c_fp, c_sz, c_fp, c_fp]),
    "dwc_h2_conv2d_s2_bwd_data": (c_int, [c_fp, c_fp, c_u, c_fp, c_fp] + [c_int] * 6 + [c_fp]),
    "dwc_h2_conv2d_s2_bwd_data_fused": (c_int, [c_fp, c_fp, c_u, c_fp, c_fp] + [c_int] * 6 + [c_fp]),
    "dwc_h2_conv2d_wgrad": (c_int, [c_fp, c_fp, c_u, c_fp, c_fp, c_u, c_fp] + [c_int] * 8 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_conv2d_wgrad_halo_ws_bytes": (c_sz, [c_int] * 6),
    "dwc_bf16_conv2d_wgrad_halo": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 8 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_conv7_smallk_wgrad_ws_bytes": (c_sz, [c_int] * 4),
    "dwc_bf16_conv7_smallk_wgrad": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 5 + [c_fp, c_sz, c_fp]),
    "dwc_x3_conv7_smallk_wgrad_ws_bytes": (c_sz, [c_int] * 4),
    "dwc_x3_conv7_smallk_wgrad": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 5 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_conv2d_stem_ok": (c_int, [c_int] * 7),
    "dwc_bf16_conv2d_stem": (c_int, [c_fp] * 4 + [c_int] * 9 + [c_fp]),
    "dwc_bf16_conv2d_stem_crop": (c_int, [c_fp] * 5 + [c_int] * 10 + [c_fp]),
    "dwc_bf16_conv2d_narrow_ok": (c_int, [c_int] * 8),
    "dwc_bf16_conv2d_narrow": (c_int, [c_fp] * 4 + [c_int] * 12 + [c_fp]),
    "dwc_bf16_conv2d_bwd_data_image_narrow": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 7 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_conv2d_bwd_data_image_ws_bytes": (c_sz, [c_int] * 7),
    "dwc_bf16_conv2d_bwd_data_image": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 7 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_conv2d_bwd_weight_ws_bytes": (c_sz, [c_int] * 9),
    "dwc_bf16_conv2d_bwd_weight": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 11 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_conv2d_bwd_weight_ex_ws_bytes": (c_sz, [c_int] * 11),
    "dwc_bf16_conv2d_bwd_weight_ex": (c_int, [c_fp, c_fp, c_fp] + [c_int] * 13 + [c_fp, c_sz, c_fp]),
    "dwc_bf16_act_bwd_bias": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "dwc_bf16_instnorm_fwd": (c_int, [c_fp] * 7 + [c_int, c_int, c_int, c_f, c_int, c_fp, c_sz, c_fp]),
    "dwc_bf16_instnorm_bwd": (c_int, [c_fp] * 9 + [c_int, c_int, c_int, c_int, c_fp, c_sz, c_fp]),
    "dwc_bf16_layernorm_fwd": (c_int, [c_fp] * 6 + [c_int, c_int, c_int, c_f, c_int, c_fp, c_sz, c_fp]),
    "dwc_bf16_layernorm_bwd": (c_int, [c_fp] * 9 + [c_int, c_int, c_int, c_f, c_int, c_fp, c_sz, c_fp]),
    "dwc_bf16_upsample2x_fwd": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_bf16_upsample2x_bwd": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_bf16_avgpool2_fwd": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_bf16_avgpool2_bwd": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_pack_nchw_to_nhwc8_bf16": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_unpack_nhwc8_bf16_to_nchw": (c_int, [c_fp, c_fp, c_int, c_int, c_int, c_int, c_fp]),
    "dwc_bf16_blend_fwd": (c_int, [c_fp, c_fp, c_fp, c_int, c_fp]),
    "dwc_bf16_blend_bwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_int, c_fp]),
    "dwc_bf16_l1_mean_fwd": (c_int, [c_fp, c_fp, c_fp, c_sz, c_int, c_fp, c_sz, c_fp]),
    "dwc_bf16_l1_mean_bwd": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_sz, c_int, c_fp]),
}

ABI_VERSION = 8                # DWC_ABI_VERSION of include/dwcgan_hip.h
EINVAL = -1
_ERRORS = {-1: "DWC_EINVAL (unsupported shape/argument)", -2: "DWC_EWORKSPACE (scratch too small)",
           -3: "DWC_ELAUNCH (kernel launch failed)"}

_lib = None


class HipKernelError(RuntimeError):
    pass


def load():
    """Load libdwcgan_hip.so and attach prototypes.  Raises loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libdwcgan_hip.so not found at %s — build it with `make -C dwc-gan_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    got = lib.dwc_version()
    if got != ABI_VERSION:         # (DWC_HIP_LIB may name an older build: same symbols, other argument lists -- undefined behaviour)
        raise ImportError("%s reports C-ABI version %d, this binding expects %d (include/dwcgan_hip.h DWC_ABI_VERSION): rebuild it with "
                          "`make -C dwc-gan_amd/csrc`" % (LIB_PATH, got, ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise HipKernelError("%s failed: %s" % (what, _ERRORS.get(rc, rc)))
