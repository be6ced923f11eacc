"""ctypes binding of libgdn_hip.so (C ABI declared in include/gdn_hip.h).

There is NO CPU fallback: if the shared library has not been built, or a tensor is not on
a HIP device, every op raises.  Build with `python -c "import __graft_entry__ as g; g.build()"`.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GDN_HIP_LIB", os.path.join(_HERE, "libgdn_hip.so"))   # override: diagnostic builds
ABI_VERSION = 21

_c_int, _c_float, _p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p

# name -> argtypes (restype is always int); mirrors include/gdn_hip.h one to one
SIGNATURES = {
    "gdn_abi_version": [],
    "gdn_nbr_pitch": [_c_int],
    "gdn_topk_graph": [_p, _c_int, _c_int, _c_int, _p, _p, _p, _p, _p],
    "gdn_graph_from_topk": [_p, _c_int, _c_int, _p, _p, _p],
    "gdn_node_terms": [_p, _p, _p, _p, _p, _p, _c_int, _c_int, _c_int, _p, _p],
    "gdn_bn_fold": [_p, _p, _p, _p, _c_float, _c_int, _p, _p],
    "gdn_project_fwd": [_p, _p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p],
    "gdn_attn_aggregate_fwd": [_p, _p, _p, _p, _p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p],
    "gdn_head_fwd": [_p, _p, _p, _p, _p, _p, _c_int, _c_int, _c_int, _p, _p, _p],
    "gdn_head_train_fwd": [_p] * 10 + [_c_float] + [_c_int] * 3 + [_c_float] * 4 + [_p] * 9,
    "gdn_head_train_workspace_bytes": [_c_int, _c_int],
    "gdn_head_train_stats_bytes": [_c_int],
    "gdn_exact_sum_workspace_bytes": [],
    "gdn_exact_sum": [_p, _c_int, _p, _p, _p],
    "gdn_head_train_bwd": [_p] * 10 + [_c_float, _p] + [_c_int] * 3 + [_c_float] * 2 + [_p] * 10,
    "gdn_head_train_fwd_rng": [_p] * 9 + [_c_float] + [_c_int] * 3 + [_c_float] * 4 + [_p] * 8 + [_c_int, _p],
    "gdn_head_train_bwd_rng": [_p] * 9 + [_c_float, _p] + [_c_int] * 3 + [_c_float] * 2 + [_p] * 9 + [_c_int, _p],
    "gdn_head_train_fwd_act": [_p] * 8 + [_c_float, _p, _c_float] + [_c_int] * 3 + [_c_float] * 4 + [_p] * 8 + [_c_int, _p],
    "gdn_head_train_bwd_act": [_p] * 9 + [_c_float, _p, _c_float, _p] + [_c_int] * 3 + [_c_float] * 2 + [_p] * 7 + [_c_int, _p],
    "gdn_mlp_train_saved_bytes": [_c_int] * 4,
    "gdn_mlp_train_workspace_bytes": [_c_int] * 4,
    "gdn_mlp_train_fwd": [_p] * 8 + [_c_int] * 4 + [_p] * 4,
    "gdn_mlp_train_bwd": [_p] * 4 + [_c_int] * 4 + [_p] * 7,
    "gdn_mlp_eval_workspace_bytes": [_c_int] * 4,
    "gdn_mlp_eval_fwd": [_p] * 6 + [_c_int] * 4 + [_p] * 3,
    "gdn_topk_graph_terms": [_p, _c_int, _c_int, _c_int] + [_p] * 8 + [_c_int, _p, _p],
    "gdn_forward_fused_plan_keys": [_p] * 4 + [_c_int] * 7 + [_p, _p],
    "gdn_forward_fused_series_plan_keys": [_p, _c_int, _c_int] + [_p] * 3 + [_c_int] * 6 + [_p, _p],
    "gdn_attn_aggregate_bwd_uses_reverse": [_c_int] * 3,
    "gdn_head_mse_workspace_bytes": [],
    "gdn_head_train_fwd_rng_mse": [_p] * 9 + [_c_float] + [_c_int] * 3 + [_c_float] * 4 + [_p] * 12 + [_c_int, _p],
    "gdn_project_bwd_partials": [_p] * 4 + [_c_int] * 4 + [_p, _p, _p],
    "gdn_train_finish": [_p, _p] + [_c_int] * 4 + [_p] * 8 + [_c_int, _c_int] + [_p] * 4,
    "gdn_adam_step": [_p] * 5 + [_c_int] + [ctypes.c_double] * 6 + [_c_int, _c_int, _p],
    "gdn_terms_bwd_acc": [_p] * 8 + [_c_int] * 3 + [_p] * 6 + [_c_int, _p],
    "gdn_mse_workspace_bytes": [],
    "gdn_mse_loss_grad": [_p, _p, ctypes.c_longlong, _p, _p, _p, _p],
    "gdn_forward_fused": [_p] * 11 + [_c_int] * 5 + [_p, _p],
    "gdn_forward_fused_series": [_p, _c_int, _c_int] + [_p] * 10 + [_c_int] * 5 + [_p, _p],
    "gdn_mlp_plan_bytes": [_c_int] * 3,
    "gdn_mlp_plan_layer": [_p] * 6 + [_c_float] + [_c_int] * 4 + [_p, _p],
    "gdn_mlp_plan_out": [_p, _p] + [_c_int] * 3 + [_p, _p],
    "gdn_mlp_fwd": [_p, _p] + [_c_int] * 4 + [_p, _p],
    "gdn_fused_plan_bytes": [_c_int] * 5,
    "gdn_fused_plan_build": [_p] * 10 + [_c_int] * 5 + [_p, _p],
    "gdn_forward_fused_plan": [_p, _p] + [_c_int] * 6 + [_p, _p, _p],
    "gdn_forward_fused_series_plan": [_p, _c_int, _c_int, _p] + [_c_int] * 5 + [_p, _p, _p],
    "gdn_fused_plan_limit_offset": [_c_int] * 5,
    "gdn_graph_bank_order": [_p, _c_int, _c_int, _p, _p],
    "gdn_forward_fused_gated": [_p] * 12 + [_c_int] * 5 + [_p, _p],
    "gdn_forward_fused_series_gated": [_p, _p, _c_int, _c_int] + [_p] * 10 + [_c_int] * 5 + [_p, _p],
    "gdn_project_fwd_wide": [_p, _p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p],
    "gdn_attn_aggregate_fwd_wide": [_p, _p, _p, _p, _p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p],
    "gdn_attn_aggregate_bwd_wide": [_p] * 8 + [_c_int] * 4 + [_p] * 6,
    "gdn_project_fwd_bf16": [_p, _p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p, _p],
    "gdn_attn_aggregate_fwd_bf16": [_p, _p, _p, _p, _p, _p, _c_int, _c_int, _c_int, _c_int, _p, _p, _p],
    "gdn_head_fwd_bf16": [_p, _p, _p, _p, _p, _p, _c_int, _c_int, _c_int, _p, _p, _p],
    "gdn_forward_fused_bf16": [_p] * 11 + [_c_int] * 5 + [_p, _p],
    "gdn_attn_aggregate_bwd": [_p] * 8 + [_c_int] * 4 + [_p] * 6,
    "gdn_attn_aggregate_bwd_workspace_bytes": [_c_int] * 4,
    "gdn_train_supported": [_c_int] * 4,
    "gdn_rev_pitch": [_c_int],
    "gdn_graph_reverse": [_p, _p, _c_int, _c_int, _p, _p, _p],
    "gdn_project_bwd_workspace_bytes": [_c_int, _c_int, _c_int],
    "gdn_project_bwd": [_p] * 4 + [_c_int] * 4 + [_p] * 5,
    "gdn_terms_bwd": [_p] * 8 + [_c_int] * 3 + [_p] * 7,
    "gdn_score_workspace_bytes": [_c_int, _c_int],
    "gdn_score_select_workspace_bytes": [_c_int, _c_int, _c_int],
    "gdn_score_keys": [_p, _p, _c_int, _c_int, _c_int, _p, _p],
    "gdn_score_select": [_p, _c_int, _c_int, _c_int, ctypes.c_longlong, _p, _p, _p],
    "gdn_score_quantiles": [_p, _p, _c_int, _c_int, _p, _p, _p],
    "gdn_score_smooth_max": [_p, _p, _p, _c_int, _c_int, _c_int, _p, _p, _p, _p, _p],
}

ERRORS = {-1: "GDN_ERR_ARG (null pointer or non-positive dimension)",
          -2: "GDN_ERR_LAUNCH (HIP launch failed)",
          -3: "GDN_ERR_UNSUPPORTED (shape outside the supported set, see include/gdn_hip.h)"}

_lib = None


class GdnHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the library once; raise loudly when it is missing or from another ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GdnHipError(
            f"{LIB_PATH} not found: the HIP extension is not built and gdn_amd has no CPU "
            "fallback. Run `python -c \"import __graft_entry__ as g; g.build()\"` (needs hipcc).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = ctypes.c_longlong if name.endswith(("_bytes", "_offset")) else _c_int
    if lib.gdn_abi_version() != ABI_VERSION:
        raise GdnHipError(f"ABI mismatch: library {lib.gdn_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def call(name: str, *args) -> None:
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise GdnHipError(f"{name} failed: {ERRORS.get(rc, rc)}")
