"""ctypes binding of libdvae_hip.so (the C ABI declared in include/dvae_hip.h).

There is no fallback: if the library has not been built, importing this module raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_void_p

# torch must be imported first: it brings its own HIP runtime (libamdhip64), and the kernels have to
# run on that same runtime instance to share torch's device context, streams and allocations.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# DVAE_HIP_LIB names another build of the same library (A/B of two builds in one session); the default is the
# in-tree one
LIB_PATH = os.environ.get('DVAE_HIP_LIB') or os.path.join(os.path.dirname(_HERE), 'libdvae_hip.so')

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the HIP kernels first "
        f"(`make -C {os.path.join(os.path.dirname(_HERE), 'csrc')}` or __graft_entry__.build()). "
        "There is no CPU fallback for this package.")

lib = ctypes.CDLL(LIB_PATH)

P = c_void_p
I = c_int
L = c_int64
F = c_float

# name -> (restype, argtypes); mirrors include/dvae_hip.h one to one
SIGNATURES = {
    'dvae_version': (I, []),
    'dvae_last_error': (c_char_p, []),
    'dvae_split_mean_diff_f32': (I, [P, P, P, L, L, L, P]),
    'dvae_recon_loss_ws_floats': (L, [L, L, L]),
    'dvae_recon_loss_fwd_f32': (I, [P, P, P, P, L, L, L, I, I, P]),
    'dvae_recon_metrics_f32': (I, [P, P, P, P, L, L, L, F, P]),
    'dvae_recon_metrics_noalloc_max_elements': (L, []),
    'dvae_recon_metrics_noalloc_init': (I, []),
    'dvae_recon_metrics_noalloc_f32': (I, [P, P, P, L, L, L, F, P]),
    'dvae_recon_loss_bwd_f32': (I, [P, P, P, P, L, L, L, I, F, P]),
    'dvae_recon_loss_bwd_logit_ws_floats': (L, [L, L]),
    'dvae_recon_loss_bwd_logit_f32': (I, [P, P, P, P, P, P, L, L, L, I, F, P]),
    'dvae_reparam_fwd_f32': (I, [P, P, P, P, L, P]),
    'dvae_reparam_bwd_f32': (I, [P, P, P, P, P, L, P]),
    'dvae_kl_fwd_f32': (I, [P, P, P, L, L, F, P]),
    'dvae_render_clips_f32': (I, [P, L, P, P, P, L, L, L, L, L, P]),
    'dvae_set_mfma_precision': (I, [I]),
    'dvae_get_mfma_precision': (I, []),
    'dvae_kl_bwd_f32': (I, [P, P, P, F, P, P, P, L, L, F, P]),
    'dvae_vae_loss_fwd_f32': (I, [P, P, L, L, L, I, I, P, P, L, F, P, P, L, F, P, I, F, F, P, F, P, P, P, P]),
    'dvae_kl_bwd2_f32': (I, [P, P, F, P, P, L, P, P, F, P, P, L, P, I, P, L, F, P]),
    'dvae_indep_ws_floats': (L, [L, L]),
    'dvae_indep_fwd_f32': (I, [P, P, P, P, L, L, L, P]),
    'dvae_indep_bwd_f32': (I, [P, P, P, P, F, P, P, L, L, L, P]),
    'dvae_bn_finalize_f32': (I, [P, L, L, L, P, P, P, P, P, F, F, P, P, P, P, P]),
    'dvae_bn_finalize_rep_f32': (I, [P, L, L, L, L, P, P, P, P, P, F, F, P, P, P, P, P]),
    'dvae_bn_eval_affine_f32': (I, [P, P, P, P, F, P, P, L, P]),
    'dvae_affine_act_f32': (I, [P, P, P, I, P, L, L, P]),
    'dvae_colsum_ws_floats': (L, [L, L]),
    'dvae_colsum_tickets': (L, []),
    'dvae_bn_bwd_f32': (I, [P, P, P, P, P, P, P, P, P, P, P, L, L, P]),
    'dvae_bn_bwd_reduce_coeffs_f32': (I, [P, P, P, P, P, P, P, P, P, P, P, P, L, L, P]),
    'dvae_bias_act_bwd_f32': (I, [P, P, I, P, P, P, L, L, P]),
    'dvae_bias_act_bwd_to_f32': (I, [P, P, P, I, P, P, P, L, L, P]),
    'dvae_sigmoid_bwd_f32': (I, [P, P, P, P, P, L, P]),
    'dvae_permute3_f32': (I, [P, P, L, L, L, I, I, I, P]),
    'dvae_weighted_sum_f32': (I, [I, POINTER(c_void_p), POINTER(c_float), P, P]),
    'dvae_gemm_ws_floats': (L, [L, L, L]),
    'dvae_gemm_nt_f32': (I, [P, L, P, L, P, I, P, L, L, L, L, P, P]),
    'dvae_gemm_nn_f32': (I, [P, L, P, L, P, L, L, L, L, I, P, P]),
    'dvae_gemm_tn_f32': (I, [P, L, P, L, P, L, L, L, L, I, P, P]),
    'dvae_pack_conv_w_f32': (I, [P, P, L, L, P]),
    'dvae_pack_deconv_w_f32': (I, [P, P, L, L, P]),
    'dvae_pack_w_many_f32': (I, [I, P, P, P, P, P, P]),
    'dvae_conv_stat_parts': (L, [L]),
    'dvae_conv_stat_parts_written': (L, [L, I]),
    'dvae_conv4s2_fwd_f32': (I, [P, P, P, I, P, P, P, L, L, L, L, L, P]),
    'dvae_deconv4s2_fwd_f32': (I, [P, P, P, I, P, P, P, L, L, L, L, L, P]),
    'dvae_conv_fwd_ws_floats': (L, [I, L, L, L, L, L]),
    'dvae_conv4s2_fwd_ws_f32': (I, [P, P, P, I, P, P, P, L, L, L, L, L, P, L, I, P]),
    'dvae_deconv4s2_fwd_ws_f32': (I, [P, P, P, I, P, P, P, L, L, L, L, L, P, L, I, P]),
    'dvae_conv4s2_dgrad_bnstat_f32': (I, [P, P, P, P, L, L, L, L, L, P, P, P, P, P, I, P]),
    'dvae_deconv4s2_dgrad_bnstat_f32': (I, [P, P, P, P, L, L, L, L, L, P, P, P, P, P, P]),
    'dvae_bn_bwd_from_partials_f32': (I, [P, P, P, P, P, P, P, L, P, P, L, L, P]),
    'dvae_bn_bwd_coeffs_f32': (I, [P, L, P, P, P, P, P, P, P, L, L, P]),
    'dvae_bn_bwd_reduce_rows': (L, [L, L]),
    'dvae_bn_bwd_reduce_f32': (I, [P, P, P, P, P, P, P, L, L, P]),
    'dvae_deconv4s2_bnin_supported': (I, [L, L, L, L]),
    'dvae_deconv4s2_dgrad_bnstat_bnin_f32': (I, [P, P, P, P, P, P, P, P, P, P, L, L, L, L, L, P, P, P, P, P, I, P]),
    'dvae_conv4s2_wgrad_bnp_f32': (I, [P, P, P, P, P, P, P, P, P, L, L, L, L, I, P]),
    'dvae_conv4s2_wgrad_bnq_supported': (I, [L, L, L, L, L]),
    'dvae_conv4s2_wgrad_bnq_f32': (I, [P, P, P, I, P, P, P, P, P, P, P, P, P, L, L, L, L, L, I, P]),
    'dvae_conv4s2_bnin_supported': (I, [L, L, L, L]),
    'dvae_conv4s2_dgrad_bnstat_bnin_f32': (I, [P, P, P, P, P, P, P, P, P, P, L, L, L, L, L, P, P, P, P, P, I, P]),
    'dvae_conv_wgrad_ws_floats': (L, [L, L, L, L, L]),
    'dvae_conv4s2_wgrad_f32': (I, [P, P, P, I, P, P, P, I, P, P, L, L, L, L, L, I, P]),
    'dvae_deconv4s2_final_fwd_f32': (I, [P, P, P, I, P, P, I, P, L, L, L, L, I, P]),
    'dvae_lstm_step_fwd_f32': (I, [P, L, P, L, P, L, P, L, P, L, P, P, P, P, P, L, P, L, L, L, P]),
    'dvae_lstm_step_bwd_pointwise_f32': (I, [P, L, P, L, P, P, L, P, L, P, P, L, L, L, P]),
    'dvae_skinny_gemm_nt_f32': (I, [P, L, P, L, P, L, L, L, L, I, P]),
    'dvae_lstm_bwd_step_f32': (I, [P, L, P, L, P, L, P, L, P, P, L, P, L, P, P, L, L, L, L, P]),
    'dvae_lstm_bwd_step2_f32': (I, [P, L, P, L, L, P, L, P, L, L, P, L, P, L, P, P, L, P, L, P, P, L, L, L, P]),
    'dvae_lstm_seq_supported': (I, [L, L, L, L]),
    'dvae_lstm_seq_ws_floats': (L, [L, L, L]),
    'dvae_lstm_seq_fwd_f32': (I, [L] + [POINTER(c_void_p)] * 5 + [L, POINTER(c_void_p), L, POINTER(c_void_p),
                                  POINTER(c_void_p), L, L, POINTER(c_void_p), POINTER(c_int), L, L, L, P, P]),
    'dvae_lstm_seq_bwd_ws_floats': (L, [L, L, L]),
    'dvae_lstm_seq_bwd_f32': (I, [L] + [POINTER(c_void_p)] * 4 + [L, POINTER(c_void_p), L, L, POINTER(c_void_p),
                                  POINTER(c_void_p), L, POINTER(c_void_p), L, I, POINTER(c_int), L, L, L, P, P]),
    'dvae_lstm_seq_timeouts': (L, []),
    'dvae_lstm_seq_timeouts_clear': (I, []),
    'dvae_lstm_resident_workgroups': (L, [I, L, L]),
    'dvae_lstm_dec2_supported': (I, [L, L, L]),
    'dvae_lstm_dec2_ws_floats': (L, [L, L]),
    'dvae_lstm_dec2_fwd_f32': (I, [POINTER(c_void_p)] * 4 + [P, L, P, L, P, P, P, P, L, L, L, P, P]),
    'dvae_lstm_dec2_bwd_f32': (I, [POINTER(c_void_p)] * 2 + [P, P, P, L, P, L, L, P, P, L, P, L, P, L, L, L, P, P]),
    'dvae_grad_norm_ws_floats': (L, [L, POINTER(c_int64)]),
    'dvae_grad_norm_f32': (I, [L, POINTER(c_void_p), POINTER(c_int64), F, P, P, P]),
    'dvae_scale_tensors_f32': (I, [L, POINTER(c_void_p), POINTER(c_int64), P, P]),
    'dvae_copy_tensors_f32': (I, [L, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), P]),
    'dvae_adam_step_f32': (I, [L, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                               POINTER(c_int64), F, F, F, F, F, F, F, P, P, P]),
    'dvae_trace_begin': (I, [L]),
    'dvae_trace_end': (L, []),
    'dvae_trace_read': (I, [L, ctypes.c_char_p, L, POINTER(c_float), POINTER(c_double), POINTER(c_int)]),
    'dvae_adam_advance_f32': (I, [P, c_double, c_double, P]),
}

MISSING = []
for _name, (_res, _args) in SIGNATURES.items():
    try:
        _fn = getattr(lib, _name)
    except AttributeError:
        MISSING.append(_name)
        continue
    _fn.restype = _res
    _fn.argtypes = _args
if MISSING:
    raise ImportError(f"{LIB_PATH} does not export: {', '.join(MISSING)} (stale build?)")


class DvaeError(RuntimeError):
    pass


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = lib.dvae_last_error()
        raise DvaeError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
