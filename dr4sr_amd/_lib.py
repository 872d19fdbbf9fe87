"""ctypes binding of libdr4sr_hip.so (the C ABI declared in include/dr4sr_hip.h).

There is NO fallback: if the HIP library is missing or fails to load, importing any compute path
of dr4sr_amd raises.  PyTorch is only used for device memory and streams (tensor.data_ptr(),
torch.cuda.current_stream().cuda_stream).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DR4SR_LIB_PATH") or os.path.join(_HERE, "csrc", "libdr4sr_hip.so")     # override: A/B runs of two builds on one box

ABI_VERSION = 8
COMM_ID_BYTES = 128           # DR4SR_COMM_ID_BYTES (the RCCL unique id, a host buffer)
GRAD_TAIL = 4
STATE_WORDS = 16
STATE_STEP, STATE_T, STATE_NVALID, STATE_RNGSTEP = 0, 1, 2, 3
POOL_NONE, POOL_ORIGIN, POOL_LAST, POOL_MEAN = 0, 1, 2, 3
SITE_EMB, SITE_ATTN, SITE_PROJ, SITE_ACT, SITE_FFN = 0, 1, 2, 3, 4
OPT_ADAM, OPT_SGD, OPT_ADAGRAD, OPT_RMSPROP = 0, 1, 2, 3        # DR4SR_OPT_* (include/dr4sr_hip.h)


def check_ids(idx, n_items: int, what: str = "index"):
    """raise what torch.nn.Embedding raises for ids outside [0, n_items) — the HIP gathers clamp them (csrc/embed.hip).  One small
    launch + one host read: for dataset tensors (once) and the dense API op, never inside the fused training step"""
    import torch
    if idx is None or idx.numel() == 0:
        return
    bad = torch.zeros(1, dtype=torch.int32, device=idx.device)
    t = idx.contiguous()
    check(load().dr4sr_check_ids(ptr(t), t.numel(), int(n_items), ptr(bad), cur_stream()), "dr4sr_check_ids")
    if int(bad):
        raise IndexError(f"index out of range in self ({what}: {int(bad)} ids outside [0, {int(n_items)}))")


def set_env(name: str, value=None):
    """set (value=None: unset) a DR4SR_* switch in os.environ AND make the loaded library read it: libdr4sr_hip.so caches every switch
    per call site until dr4sr_reload_env() bumps its generation (csrc/common.h), so a plain os.environ change after the first call of an
    entry point would be silently ignored"""
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    if _lib is not None:
        _lib.dr4sr_reload_env()


def optimizer_settings(name: str, weight_decay: float):
    """/root/reference model/basemodel.py:79-98 -> (DR4SR_OPT_* kind, (beta1, beta2), eps, weight_decay): each optimizer with torch's
    defaults as the reference constructs it — Adam(lr, weight_decay), SGD(lr, weight_decay) (no momentum), Adagrad(lr, weight_decay)
    (eps 1e-10, lr_decay 0, accumulator 0), RMSprop(lr, weight_decay) (alpha 0.99 carried in beta2, eps 1e-8, no momentum, not
    centered); an UNKNOWN name falls back to Adam(lr) without weight decay, as the reference's else branch does.  'sparse_adam': the
    reference builds torch.optim.SparseAdam over dense nn.Embedding gradients, which raises at its first step — the same RuntimeError
    is raised here, when the optimizer is built."""
    n = str(name).lower()
    if n == "adam":
        return OPT_ADAM, (0.9, 0.999), 1e-8, float(weight_decay)
    if n == "sgd":
        return OPT_SGD, (0.9, 0.999), 1e-8, float(weight_decay)
    if n == "adagrad":
        return OPT_ADAGRAD, (0.9, 0.999), 1e-10, float(weight_decay)
    if n == "rmsprop":
        return OPT_RMSPROP, (0.9, 0.99), 1e-8, float(weight_decay)
    if n == "sparse_adam":
        raise RuntimeError("SparseAdam does not support dense gradients, please consider Adam instead")
    return OPT_ADAM, (0.9, 0.999), 1e-8, 0.0

KERNEL_IDS = {"prep": 0, "embed_fwd": 1, "qkv_fwd": 2, "attn_fwd": 3, "post_fwd": 4, "score": 5, "transpose": 6,
              "post_bwd": 7, "attn_bwd": 8, "qkv_bwd": 9, "embed_bwd": 10, "wgrad": 11, "adam": 12, "zero_grads": 13,
              "embqkv_fwd": 14, "post_mid": 15, "qkv_embed_bwd": 16, "wgrad_fused": 17}

FMLP_KERNEL_IDS = {"filter_fwd": 0, "ffn_fwd": 1, "ffn_bwd": 2, "filter_bwd": 3, "wgrad": 4}      # DR4SR_FK_* (include/dr4sr_hip_hooks.h)

_f32p = C.c_void_p
_i64p = C.c_void_p


class SasrecPlan(C.Structure):
    """mirror of `dr4sr_sasrec_plan` (include/dr4sr_hip.h)"""
    _fields_ = [
        ("abi_version", C.c_int32),
        ("B", C.c_int32), ("L", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("F", C.c_int32),
        ("n_layer", C.c_int32), ("n_items", C.c_int32),
        ("ln_eps", C.c_float), ("p_drop", C.c_float),
        ("seed", C.c_uint64),
        ("params", _f32p), ("grads", _f32p), ("adam_m", _f32p), ("adam_v", _f32p),
        ("n_params", C.c_int64),
        ("in_item_id", _i64p), ("item_id", _i64p), ("seqlen", _i64p), ("rows", _i64p), ("neg_item", _i64p),
        ("sample_neg", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("state", C.c_void_p),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("weight_decay", C.c_float),
        ("perm", _i64p), ("n_perm", C.c_int64), ("perm_stride", C.c_int64), ("perm_offset", C.c_int64),
        ("perm_counter", C.c_void_p),
        ("loss_log", _f32p),
        ("expected_tokens", C.c_int32),
        ("optimizer", C.c_int32),
    ]


class MetaWeighting(C.Structure):
    """mirror of `dr4sr_meta_weighting` (include/dr4sr_hip.h)"""
    _fields_ = [("phi", _f32p), ("gumbel", _f32p), ("user_id", _i64p), ("gate_in", C.c_void_p), ("gate_out", C.c_void_p),
                ("weight_out", _f32p), ("tau", C.c_float)]


class FmlpPlan(C.Structure):
    """mirror of `dr4sr_fmlp_plan` (include/dr4sr_hip.h)"""
    _fields_ = [
        ("abi_version", C.c_int32),
        ("B", C.c_int32), ("L", C.c_int32), ("D", C.c_int32), ("F", C.c_int32), ("n_layer", C.c_int32), ("n_items", C.c_int32),
        ("ln_eps", C.c_float), ("p_drop", C.c_float),
        ("seed", C.c_uint64),
        ("params", _f32p), ("grads", _f32p), ("adam_m", _f32p), ("adam_v", _f32p),
        ("n_params", C.c_int64),
        ("in_item_id", _i64p), ("item_id", _i64p), ("rows", _i64p), ("neg_item", _i64p),
        ("sample_neg", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("state", C.c_void_p),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("weight_decay", C.c_float),
        ("optimizer", C.c_int32),
        ("perm", _i64p), ("n_perm", C.c_int64), ("perm_stride", C.c_int64), ("perm_offset", C.c_int64),
        ("perm_counter", C.c_void_p),
        ("loss_log", _f32p),
    ]


class GruPlan(C.Structure):
    """mirror of `dr4sr_gru4rec_plan` (include/dr4sr_hip.h)"""
    _fields_ = [
        ("abi_version", C.c_int32),
        ("B", C.c_int32), ("L", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("n_layer", C.c_int32), ("n_items", C.c_int32),
        ("p_drop", C.c_float),
        ("seed", C.c_uint64),
        ("params", _f32p), ("grads", _f32p), ("adam_m", _f32p), ("adam_v", _f32p),
        ("n_params", C.c_int64),
        ("in_item_id", _i64p), ("item_id", _i64p), ("seqlen", _i64p), ("rows", _i64p), ("neg_item", _i64p),
        ("sample_neg", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("state", C.c_void_p),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("weight_decay", C.c_float),
        ("optimizer", C.c_int32),
        ("perm", _i64p), ("n_perm", C.c_int64), ("perm_stride", C.c_int64), ("perm_offset", C.c_int64),
        ("perm_counter", C.c_void_p),
        ("loss_log", _f32p),
    ]


_PLANP = C.POINTER(SasrecPlan)
_FPLANP = C.POINTER(FmlpPlan)
_GPLANP = C.POINTER(GruPlan)

# name -> (restype, argtypes); every symbol include/dr4sr_hip.h and include/dr4sr_hip_hooks.h (test / measurement hooks) declare
SYMBOLS = {
    "dr4sr_abi_version": (C.c_int, []),
    "dr4sr_sasrec_plan_sizeof": (C.c_int, []),
    "dr4sr_sasrec_param_layout": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_sasrec_workspace_bytes": (C.c_int64, [_PLANP]),
    "dr4sr_sasrec_at_scale": (C.c_int, [C.c_void_p]),
    "dr4sr_sasrec_fwd_bwd": (C.c_int, [_PLANP, C.c_void_p]),
    "dr4sr_adam_step": (C.c_int, [_PLANP, C.c_void_p]),
    "dr4sr_sasrec_fwd_bwd_weighted": (C.c_int, [_PLANP, C.c_void_p, C.c_void_p]),
    "dr4sr_sasrec_fwd_bwd_weighted_prepared": (C.c_int, [_PLANP, C.c_void_p, C.c_void_p]),
    "dr4sr_sasrec_train_step": (C.c_int, [_PLANP, C.c_void_p]),
    "dr4sr_sasrec_train_steps": (C.c_int, [_PLANP, C.c_int32, C.c_void_p]),
    "dr4sr_sasrec_fwd_bwd_prepared": (C.c_int, [_PLANP, C.c_void_p]),
    "dr4sr_adam_step_prepare_next": (C.c_int, [_PLANP, C.c_void_p]),
    "dr4sr_sasrec_grad_buckets": (C.c_int, [_PLANP, C.POINTER(C.c_int64)]),
    "dr4sr_sasrec_fwd_bwd_phase": (C.c_int, [_PLANP, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_sasrec_encode": (C.c_int, [_PLANP, C.c_int32, C.c_int32, _f32p, C.c_void_p]),
    "dr4sr_sasrec_encode_bwd": (C.c_int, [_PLANP, C.c_int32, C.c_int32, _f32p, C.c_void_p]),
    "dr4sr_select_rows": (C.c_int, [_i64p, C.c_int64, _i64p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "dr4sr_embed_gather_posadd": (C.c_int, [_f32p, _f32p, _i64p, _f32p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_score_bce_fwd": (C.c_int, [_f32p, _f32p, _i64p, _i64p, _f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_score_bce_bwd": (C.c_int, [_f32p, _f32p, _i64p, _i64p, _f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_score_bpr_fwd": (C.c_int, [_f32p, _f32p, _i64p, _i64p, _f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_score_bpr_bwd": (C.c_int, [_f32p, _f32p, _i64p, _i64p, _f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_loss_from_scores_fwd": (C.c_int, [_f32p, _f32p, C.c_int64, C.c_int32, C.c_int32, _f32p, _f32p, C.c_void_p]),
    "dr4sr_loss_from_scores_bwd": (C.c_int, [_f32p, _f32p, C.c_int64, C.c_int32, C.c_int32, _f32p, _f32p, _f32p, _f32p, C.c_void_p]),
    "dr4sr_neg_sample": (C.c_int, [_i64p, C.c_int64, C.c_int32, C.c_uint64, C.c_uint32, C.c_void_p]),
    "dr4sr_neg_sample_dev": (C.c_int, [_i64p, C.c_int64, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p]),
    "dr4sr_dropout_mask": (C.c_int, [_f32p, C.c_int64, C.c_float, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]),
    "dr4sr_fmlp_plan_sizeof": (C.c_int, []),
    "dr4sr_fmlp_param_layout": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_fmlp_workspace_bytes": (C.c_int64, [_FPLANP]),
    "dr4sr_fmlp_fwd_bwd": (C.c_int, [_FPLANP, C.c_void_p]),
    "dr4sr_fmlp_train_step": (C.c_int, [_FPLANP, C.c_void_p]),
    "dr4sr_fmlp_encode": (C.c_int, [_FPLANP, C.c_int32, _f32p, C.c_void_p]),
    "dr4sr_fmlp_encode_bwd": (C.c_int, [_FPLANP, C.c_int32, _f32p, C.c_void_p]),
    "dr4sr_gru4rec_plan_sizeof": (C.c_int, []),
    "dr4sr_gru4rec_param_layout": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_gru4rec_workspace_bytes": (C.c_int64, [_GPLANP]),
    "dr4sr_gru4rec_uses_cooperative": (C.c_int, [C.c_int32, C.c_int32]),
    "dr4sr_gru4rec_uses_wavefront": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "dr4sr_gru4rec_fwd_bwd": (C.c_int, [_GPLANP, C.c_void_p]),
    "dr4sr_gru4rec_train_step": (C.c_int, [_GPLANP, C.c_void_p]),
    "dr4sr_gru4rec_encode": (C.c_int, [_GPLANP, C.c_int32, C.c_int32, _f32p, C.c_void_p]),
    "dr4sr_gru4rec_encode_bwd": (C.c_int, [_GPLANP, C.c_int32, C.c_int32, _f32p, C.c_void_p]),
    "dr4sr_adam_flat": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "dr4sr_reload_env": (C.c_int, []),
    "dr4sr_build_flags": (C.c_int, []),
    "dr4sr_sasrec_launch_kernel": (C.c_int, [_PLANP, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_sasrec_launch_kernel_weighted": (C.c_int, [_PLANP, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_gru4rec_launch_kernel": (C.c_int, [_GPLANP, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_fmlp_launch_kernel": (C.c_int, [_FPLANP, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_meta_param_count": (C.c_int64, [C.c_int32]),
    "dr4sr_meta_select_workspace_floats": (C.c_int64, [C.c_int64]),
    "dr4sr_meta_select_fwd": (C.c_int, [_f32p, _f32p, _f32p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_float, _i64p, _i64p, C.c_int64, C.c_int32,
                                        C.c_int32, C.c_void_p, C.c_void_p, _f32p, C.c_void_p]),
    "dr4sr_meta_select_bwd": (C.c_int, [_f32p, _f32p, _f32p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_float, _i64p, _i64p, C.c_int64, C.c_int32,
                                        C.c_int32, C.c_void_p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_void_p]),
    "dr4sr_fd_step_size": (C.c_int, [_f32p, _f32p, C.c_int64, C.c_float, _f32p, C.c_void_p]),
    "dr4sr_fd_step_size_scratch_floats": (C.c_int64, []),
    "dr4sr_fd_step_size_ws": (C.c_int, [_f32p, _f32p, C.c_int64, C.c_float, _f32p, _f32p, C.c_void_p]),
    "dr4sr_fd_shift": (C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_float, C.c_int64, C.c_void_p]),
    "dr4sr_fd_neumann": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_int64, C.c_void_p]),
    "dr4sr_fd_diff": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_int64, C.c_void_p]),
    "dr4sr_fd_diff4": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_int64, C.c_void_p]),
    "dr4sr_scale_by": (C.c_int, [_f32p, _f32p, _f32p, C.c_int64, C.c_void_p]),
    "dr4sr_meta_opt_step": (C.c_int, [C.c_int32, _f32p, _f32p, _f32p, _f32p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                      C.c_float, C.c_void_p, _f32p, C.c_void_p]),
    "dr4sr_meta_sgd_step": (C.c_int, [_f32p, _f32p, _f32p, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                      _f32p, C.c_void_p]),
    "dr4sr_cl_augment": (C.c_int, [_i64p, _i64p, _i64p, _i64p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double,
                                   C.c_int64, C.c_uint64, C.c_uint32, C.c_void_p]),
    "dr4sr_cl_augment_dev": (C.c_int, [_i64p, _i64p, _i64p, _i64p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double,
                                       C.c_int64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "dr4sr_cl_augment2_dev": (C.c_int, [_i64p, _i64p, _i64p, _i64p, _i64p, _i64p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double,
                                        C.c_double, C.c_int64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "dr4sr_cl_augment2_rows_dev": (C.c_int, [_i64p, _i64p, _i64p, _i64p, _i64p, _i64p, _i64p, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                             C.c_double, C.c_double, C.c_int64, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    "dr4sr_cl_prepare": (C.c_int, [_i64p, C.c_int32, C.c_void_p, _f32p, _f32p, C.c_int64, C.c_void_p]),
    "dr4sr_cl_prepare_rows": (C.c_int, [_i64p, _i64p, C.c_int32, C.c_void_p, _f32p, _f32p, C.c_int64, C.c_void_p]),
    "dr4sr_cl_prepare_rows_step": (C.c_int, [_i64p, _i64p, C.c_int32, C.c_void_p, _f32p, _f32p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    "dr4sr_infonce_bwd_scaled": (C.c_int, [_f32p, _f32p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, _f32p, _f32p, _f32p, C.c_float, C.c_int32,
                                           _f32p, _f32p, C.c_void_p]),
    "dr4sr_cl_scalars": (C.c_int, [_f32p, _f32p, C.c_float, _f32p, _f32p, C.c_void_p]),
    "dr4sr_fmlp_adam_step": (C.c_int, [_FPLANP, C.c_void_p]),
    "dr4sr_gru4rec_adam_step": (C.c_int, [_GPLANP, C.c_void_p]),
    "dr4sr_gru4rec_train_steps": (C.c_int, [_GPLANP, C.c_int32, C.c_void_p]),
    "dr4sr_check_ids": (C.c_int, [_i64p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "dr4sr_optimizer_flat": (C.c_int, [C.c_int32, _f32p, _f32p, _f32p, _f32p, C.c_int64, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                       C.c_float, C.c_void_p]),
    "dr4sr_cl_scalars_dp": (C.c_int, [_f32p, C.c_int32, C.c_int64, _f32p, C.c_float, _f32p, _f32p, C.c_void_p]),
    "dr4sr_infonce_fwd": (C.c_int, [_f32p, _f32p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, _f32p, _f32p, _f32p, C.c_void_p]),
    "dr4sr_infonce_bwd": (C.c_int, [_f32p, _f32p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, _f32p, _f32p, _f32p, _f32p, C.c_void_p]),
    "dr4sr_full_score_topk": (C.c_int, [_f32p, _f32p, _i64p, _f32p, _i64p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "dr4sr_full_score_topk_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int32]),
    "dr4sr_full_score_topk_masked_ws": (C.c_int, [_f32p, _f32p, _i64p, C.c_void_p, _f32p, _i64p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                                  C.c_int32, _f32p, C.c_int64, C.c_void_p]),
    "dr4sr_full_score_topk_ws": (C.c_int, [_f32p, _f32p, _i64p, _f32p, _i64p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           _f32p, C.c_int64, C.c_void_p]),
    # ABI 8: the data-parallel transport (csrc/comm.hip, RCCL on the caller's stream); comm handles are opaque pointers
    "dr4sr_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dr4sr_comm_init_rank": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "dr4sr_comm_destroy": (C.c_int, [C.c_void_p]),
    "dr4sr_comm_rank": (C.c_int, [C.c_void_p]),
    "dr4sr_comm_world": (C.c_int, [C.c_void_p]),
    "dr4sr_comm_async_error": (C.c_int, [C.c_void_p]),
    "dr4sr_comm_error_string": (C.c_char_p, [C.c_int]),
    "dr4sr_allreduce_f32": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_void_p]),
    "dr4sr_allreduce_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "dr4sr_allreduce_f32_async": (C.c_int, [C.c_void_p, _f32p, C.c_int64, C.c_void_p]),
    "dr4sr_comm_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dr4sr_allgather_bytes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "dr4sr_broadcast_bytes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "dr4sr_crash_line_set": (C.c_int, [C.c_char_p, C.c_int32, C.c_int32]),           # measurement hook (include/dr4sr_hip_hooks.h)
}

_lib = None


class Dr4srError(RuntimeError):
    pass


def load():
    """dlopen the library once and type every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Dr4srError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C dr4sr_amd/csrc`).  dr4sr_amd has no CPU / PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    v = lib.dr4sr_abi_version()
    if v != ABI_VERSION:
        raise Dr4srError(f"libdr4sr_hip.so ABI {v} != binding ABI {ABI_VERSION}")
    if lib.dr4sr_sasrec_plan_sizeof() != C.sizeof(SasrecPlan):
        raise Dr4srError("ctypes mirror of dr4sr_sasrec_plan does not match the compiled struct")
    if lib.dr4sr_fmlp_plan_sizeof() != C.sizeof(FmlpPlan):
        raise Dr4srError("ctypes mirror of dr4sr_fmlp_plan does not match the compiled struct")
    if lib.dr4sr_gru4rec_plan_sizeof() != C.sizeof(GruPlan):
        raise Dr4srError("ctypes mirror of dr4sr_gru4rec_plan does not match the compiled struct")
    _lib = lib
    return lib


_ERR = {-1: "DR4SR_E_ARG (null pointer / bad size)", -2: "DR4SR_E_SHAPE (unsupported D/H/F/L)",
        -3: "DR4SR_E_WS (workspace too small)"}


def check(rc: int, what: str):
    if rc != 0:
        raise Dr4srError(f"{what} failed: {_ERR.get(rc, 'hipError_t ' + str(rc))}")


def ptr(t):
    """device pointer of a contiguous torch tensor (or None)"""
    if t is None:
        return None
    assert t.is_contiguous(), "dr4sr_amd kernels need contiguous tensors"
    return C.c_void_p(t.data_ptr())


def cur_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
