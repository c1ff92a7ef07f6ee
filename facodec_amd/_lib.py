"""ctypes binding of libfacodec_hip.so (C ABI declared in include/facodec_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, the product path
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FAC_LIB_PATH", os.path.join(_HERE, "libfacodec_hip.so"))

PAD_ZERO, PAD_REFLECT = 0, 1
ACT_NONE, ACT_TANH, ACT_MISH, ACT_LOG_MEL = 0, 1, 2, 3
ACT_GATE, ACT_WN_RES_SKIP = 4, 5          # epilogues of the few-column split-reduction launches only (facodec_hip.h)

_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64
_f = C.c_float


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", _p), ("w", _p), ("bias", _p), ("alpha_in", _p), ("alpha_out", _p), ("res", _p), ("y", _p), ("y2", _p), ("alpha_y2", _p), ("w_k1", _p), ("bias_k1", _p),
        ("x_bs", _i64), ("x_cs", _i64), ("y_bs", _i64), ("y_cs", _i64),
        ("B", C.c_int32), ("C_in", C.c_int32), ("T_in", C.c_int32), ("C_out", C.c_int32),
        ("C_out_pad", C.c_int32), ("T_out", C.c_int32),
        ("K", C.c_int32), ("stride", C.c_int32), ("dilation", C.c_int32), ("pad_left", C.c_int32),
        ("pad_mode", C.c_int32), ("n_phase", C.c_int32), ("y_tstride", C.c_int32), ("phase_shift", C.c_int32),
        ("act", C.c_int32),
        ("w_batched", C.c_int32), ("w_bs", _i64), ("ws", _p), ("ws_bytes", _i64), ("w_split", _p),
        ("K1", C.c_int32), ("dilation2", C.c_int32), ("row_phases", C.c_int32),
        ("x_p8", _p), ("x_p8_plane_bytes", _i64), ("y2_p8", _p), ("y2_p8_plane_bytes", _i64),
        ("pw_split", C.c_int32),
    ]


class VqDesc(C.Structure):
    _fields_ = [
        ("residual", _p), ("z_in", _p), ("zq_acc", _p), ("zq_out", _p), ("w_in", _p), ("b_in", _p),
        ("codebook", _p), ("w_out", _p), ("w_out_scale", _p), ("b_out", _p), ("mask", _p), ("codes", _p), ("z_e", _p),
        ("loss_part", _p), ("codes_bs", _i64),
        ("B", C.c_int32), ("D", C.c_int32), ("T", C.c_int32), ("Kc", C.c_int32),
    ]


# name -> (restype, argtypes); must list every symbol include/facodec_hip.h declares
SIGNATURES = {
    "fac_version": (_i, []),
    "fac_last_error": (C.c_char_p, []),
    "fac_wn_scale": (_i, [_p, _p, _p, _i, _i, _p]),
    "fac_pack_conv_w": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_pack_convtr_w_rows": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fac_conv_w_split_bytes": (_i64, [_i, _i, _i]),
    "fac_conv_w_split2_bytes": (_i64, [_i, _i, _i, _i]),
    "fac_pack_conv_w_split2": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_gemm_w_split_bytes": (_i64, [_i, _i, _i, _i]),
    "fac_pack_gemm_w_split": (_i, [_p, _i64, _i64, _i64, _p, _p, _i, _i, _i, _i, _p]),
    "fac_pack_conv_w_split": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fac_flip_transpose_w": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fac_prep_begin": (_i, []),
    "fac_prep_set_phase": (_i, [_i]),
    "fac_prep_abort": (_i, []),
    "fac_prep_end": (_i, []),
    "fac_prep_replay": (_i, [_i, _p]),
    "fac_prep_info": (_i, [_i, _p, _p]),
    "fac_prep_free": (_i, [_i]),
    "fac_pack_conv_w_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_pad_fold_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "fac_conv1d_bwd_weight_ws_bytes": (_i64, [_i, _i, _i, _i, _i]),
    "fac_conv1d_bwd_weight": (_i, [_p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fac_conv1d_bwd_weight_split_ws_bytes": (_i64, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "fac_conv1d_bwd_weight_k1_ws_bytes": (_i64, [_i, _i, _i, _i]),
    "fac_conv1d_bwd_weight_taps_tx": (_i64, [_i, _i, _i, _i, _i]),
    "fac_conv1d_bwd_weight_taps_ws_bytes": (_i64, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "fac_conv1d_bwd_weight_taps": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fac_conv1d_bwd_weight_k1": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _p]),
    "fac_conv1d_bwd_weight_split": (_i, [_p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fac_conv1d_bwd_weight_split_db_ok": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "fac_conv1d_bwd_weight_split_db": (_i, [_p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fac_weight_norm_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _p]),
    "fac_snake_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "fac_pack_lstm_whh_t": (_i, [_p, _p, _i, _p]),
    "fac_lstm_layer_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "fac_to_p8": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fac_lstm_persist_ok": (_i, [_i, _i]),
    "fac_lstm_persist_stream_ok": (_i, [_p]),
    "fac_pack_lstm_whh16": (_i, [_p, _p, _i, _i, _p]),
    "fac_lstm_layer_fwd_persist": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_lstm_layer_bwd_persist": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_lstm_persist_split_ok": (_i, [_i, _i]),
    "fac_lstm_persist_timeouts": (_i, []),
    "fac_lstm_persist_arm": (_i, [_p]),
    "fac_lstm_abort_flag": (_i, [_p, _p]),
    "fac_mask_flags_if": (_i, [_p, _i, _p, _p]),
    "fac_pack_lstm_whh_split": (_i, [_p, _p, _i, _p]),
    "fac_lstm_layer_fwd_persist_split": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_snake_bwd_fused": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "fac_bias_grad": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fac_snake_bwd_fused_rs": (_i, [_p, _p, _p, _i64, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "fac_pad_fold_edges": (_i, [_p, _i, _i, _i, _i, _i, _p]),
    "fac_pack_convtr_w": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_conv1d_fwd": (_i, [C.POINTER(ConvDesc), _p]),
    "fac_conv1d_variant": (_i, [C.POINTER(ConvDesc), C.c_char_p, _i]),
    "fac_snake_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fac_lstm_to_time_major": (_i, [_p, _p, _i, _i, _i, _p]),
    "fac_lstm_from_time_major": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "fac_pack_lstm_whh": (_i, [_p, _p, _i, _p]),
    "fac_lstm_layer_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "fac_lstm_layer_fwd_from": (_i, [_p, _p, _p, _p, _i, _i, _i, _i64, _p]),
    "fac_lstm_layer_fwd_train": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i64, _p]),
    "fac_lstm_gate_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _p]),
    "fac_tanh_bwd": (_i, [_p, _p, _p, _i64, _p]),
    "fac_pair_bwd": (_i, [_p, _p, _p, _i64, _i, C.c_float, C.c_float, _i, _p]),
    "fac_spec_power_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_stft_frames_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fac_vq_latent_bwd": (_i, [_p, _p, _p, _i64, _p, _p, _p, _p, _i, _i, _p]),
    "fac_vq_codebook_grad": (_i, [_p, _p, _p, _i64, _p, _p, _i, _i, _i, _i, _p]),
    "fac_layernorm_c_affine_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "fac_rows_fma": (_i, [_p, _p, _p, _p, _i, _i64, _f, _p]),
    "fac_grad_norm_clip": (_i, [_p, _i64, _f, _p, _p, _p]),
    "fac_adamw_step": (_i, [_p, _p, _p, _p, _i64, _f, _f, _f, _f, _f, _i64, _p, _p]),
    "fac_logdiff_rms_bwd": (_i, [_p, _p, _p, _i, _i, _i, _f, _f, _i, _p]),
    "fac_gather_copy": (_i, [_p, _p, _p, _i, _p]),
    "fac_adamw_step_masked": (_i, [_p, _p, _p, _p, _i64, _p, _i, _p, _p, _p, _f, _f, _f, _f, _f, _p, _p]),
    "fac_gate_bwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fac_mish_fwd": (_i, [_p, _p, _i64, _p]),
    "fac_mish_bwd": (_i, [_p, _p, _p, _i64, _p]),
    "fac_glu_bwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fac_mul_scaled": (_i, [_p, _p, _p, _f, _i64, _p]),
    "fac_masked_mean_bwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fac_attention_probs": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_attention_pv": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_attention_bwd_pv": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_attention_bwd_qk": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_aa_snakebeta_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "fac_leaky_relu": (_i, [_p, _p, _p, _i64, _f, _i, _i, _i, _i, _i, _p]),
    "fac_spec_to_cat": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fac_period_fold": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "fac_zero_insert": (_i, [_p, _p, _i64, _i, _i, _p]),
    "fac_row_stack3": (_i, [_p, _p, _i64, _i, _i, _i, _i, _p]),
    "fac_spec_to_rows": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "fac_pad_reflect": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "fac_disc_preprocess": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "fac_cross_entropy": (_i, [_p, _p, _p, _p, _p, _i64, _i, _f, _p]),
    "fac_focal_scalar": (_i, [_p, _p, _f, _p]),
    "fac_crop_rows": (_i, [_p, _p, _p, _i, _i, _i64, _i, _i, _p]),
    "fac_stream_push": (_i, [_p, _p, _i64, _i64, _i, _i, _i, _p]),
    "fac_vq_fwd": (_i, [C.POINTER(VqDesc), _p]),
    "fac_vq_loss_tiles": (_i, [_i]),
    "fac_rccl_available": (_i, []),
    "fac_rccl_unique_id": (_i, [_p]),
    "fac_rccl_comm_init": (_i, [C.POINTER(_p), _p, _i, _i]),
    "fac_rccl_comm_destroy": (_i, [_p]),
    "fac_allreduce_arena": (_i, [_p, _p, _i64, _i, _p]),
    "fac_vq_search": (_i, [_p, _p, _p, _i64, _i, _p]),
    "fac_gate_tanh_sigmoid": (_i, [_p, _p, _i64, _p, _i, _i, _i, _p]),
    "fac_embed_sum": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fac_glu_residual": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fac_add": (_i, [_p, _p, _p, _i64, _p]),
    "fac_sub2": (_i, [_p, _p, _p, _p, _i64, _p]),
    "fac_mul_mask": (_i, [_p, _p, _i, _i, _i, _p]),
    "fac_wn_res_skip": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_attention": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "fac_masked_mean": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fac_layernorm_c_affine": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "fac_stft_frames": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "fac_spec_power": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "fac_reduce_pair": (_i, [_p, _p, _p, _p, _i64, _i, _f, _f, _i, _p]),
    "fac_aa_snakebeta_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "fac_logdiff_rms": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _f, _i, _p]),
}

_lib = None


class FacodecHipError(RuntimeError):
    pass


def load():
    """Loads the shared library (once).  Raises FacodecHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FacodecHipError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -m facodec_amd.build` (needs hipcc, cross-compiles gfx950).")
    # One HIP runtime per process: torch bundles its own libamdhip64, and a second copy (the system one, pulled
    # in if this library were loaded first) initialises with "no ROCm-capable device".  Import torch first so
    # that our DT_NEEDED entry resolves to the copy already in the process.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().fac_last_error().decode("utf-8", "replace")
        raise FacodecHipError(f"{what} failed (rc={rc}): {msg}")
