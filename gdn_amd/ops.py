"""Tensor-level wrappers over the C ABI (include/gdn_hip.h): PyTorch-ROCm tensors in,
tensors out.  torch is used for device memory and the current HIP stream only; all
arithmetic happens in libgdn_hip.so.  Every function raises if a tensor is not on a HIP
device — there is no CPU path.
"""
from __future__ import annotations

import os

import ctypes

import torch

from . import _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype=torch.float32, name="tensor") -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.GdnHipError(f"{name} is on {t.device}: gdn_amd ops need a HIP device (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def nbr_pitch(k: int) -> int:
    return ((k + 1) + 15) & ~15


class SensorGraph:
    """Neighbour lists shared by every window (reference models/GDN.py:148-165 and
    models/graph_layer.py:61-63 in list form).  topk: [n,k] int64 = `learned_graph`."""

    def __init__(self, topk, nbr, deg, cos=None):
        self.topk, self.nbr, self.deg, self.cos = topk, nbr, deg, cos
        self.n, self.k = topk.shape
        self.pitch = nbr.shape[1]
        self._reverse = None
        self._ordered = False       # not built yet; None = shape outside the matrix-core kernels

    def nbr_ordered(self):
        """The neighbour lists with every row permuted inside its two halves for LDS bank spread
        (gdn_graph_bank_order), or `nbr` itself for shapes the matrix-core kernels do not take.  For launches
        that do not hand alpha out in rank order.  Built on first use, once per graph."""
        if self._ordered is False:
            self._ordered = None
            if self.n <= 127 and nbr_pitch(self.k) <= 64:
                out = torch.empty_like(self.nbr)
                out.copy_(self.nbr)                         # (padding rows / slots as in the original)
                _lib.call("gdn_graph_bank_order", _ptr(self.nbr), self.n, self.k, _ptr(out), _stream())
                self._ordered = out
        return self.nbr if self._ordered is None else self._ordered

    def reverse(self):
        """(rent[n, rpitch] u32, rlen[n] i32): per source, the (target << 16 | slot) entries that
        name it — the backward pass gathers through these.  Built on first use."""
        if self._reverse is None:
            rpitch = (self.n + 15) & ~15
            rent = torch.empty((self.n, rpitch), dtype=torch.int32, device=self.nbr.device)
            rlen = torch.empty((self.n,), dtype=torch.int32, device=self.nbr.device)
            _lib.call("gdn_graph_reverse", _ptr(self.nbr), _ptr(self.deg), self.n, self.k, _ptr(rent), _ptr(rlen),
                      _stream())
            self._reverse = (rent, rlen)
        return self._reverse


def topk_graph(emb: torch.Tensor, k: int, want_cos: bool = False) -> SensorGraph:
    emb = _chk(emb.detach(), name="embedding")            # graph is built on a detached copy (GDN.py:145)
    n, d = emb.shape
    dev = emb.device
    topk = torch.empty((n, k), dtype=torch.int64, device=dev)
    nbr = torch.empty((n, nbr_pitch(k)), dtype=torch.uint16, device=dev)
    deg = torch.empty((n,), dtype=torch.int32, device=dev)
    cos = torch.empty((n, n), dtype=torch.float32, device=dev) if want_cos else None
    _lib.call("gdn_topk_graph", _ptr(emb), n, d, k, _ptr(topk), _ptr(nbr), _ptr(deg), _ptr(cos), _stream())
    return SensorGraph(topk, nbr, deg, cos)


def graph_from_topk(topk: torch.Tensor) -> SensorGraph:
    topk = _chk(topk, torch.int64, "topk")
    n, k = topk.shape
    if k > 0 and (int(topk.min()) < 0 or int(topk.max()) >= n):     # caller data: one sync, once per graph
        raise ValueError(f"injected top-k table has entries outside [0, {n})")
    nbr = torch.empty((n, nbr_pitch(k)), dtype=torch.uint16, device=topk.device)
    deg = torch.empty((n,), dtype=torch.int32, device=topk.device)
    _lib.call("gdn_graph_from_topk", _ptr(topk), n, k, _ptr(nbr), _ptr(deg), _stream())
    return SensorGraph(topk, nbr, deg)


def node_terms(lin_w, att_i, att_j, att_em_i, att_em_j, emb) -> torch.Tensor:
    lin_w = _chk(lin_w.detach(), name="lin.weight")
    d, w = lin_w.shape
    emb = _chk(emb.detach(), name="embedding")
    n = emb.shape[0]
    vs = [_chk(v.detach().reshape(-1), name="att") for v in (att_i, att_j, att_em_i, att_em_j)]
    out = torch.empty((128 + 2 * n,), dtype=torch.float32, device=emb.device)
    _lib.call("gdn_node_terms", _ptr(lin_w), *[_ptr(v) for v in vs], _ptr(emb), n, d, w, _ptr(out), _stream())
    return out


def bn_fold(bn: torch.nn.BatchNorm1d) -> torch.Tensor:
    c = bn.num_features
    out = torch.empty((2 * c,), dtype=torch.float32, device=bn.weight.device)
    _lib.call("gdn_bn_fold", _ptr(_chk(bn.weight.detach())), _ptr(_chk(bn.bias.detach())),
              _ptr(_chk(bn.running_mean)), _ptr(_chk(bn.running_var)), float(bn.eps), c, _ptr(out), _stream())
    return out


def _storage(t: torch.Tensor, name: str) -> str:
    """"" for fp32 tensors, "_bf16" for bfloat16 ones: the suffix of the C entry point (bf16 STORAGE of
    x / xlin / z, fp32 arithmetic — include/gdn_hip.h)."""
    if t.dtype == torch.bfloat16:
        return "_bf16"
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32 or bfloat16, got {t.dtype}")
    return ""


def project_fwd(x, lin_w, terms, wide: bool = False):
    """x[B,n,w] -> xlin[B*n,d], s_i[B*n], s_j[B*n]  (models/graph_layer.py:56 + logit scalars).
    bfloat16 x gives bfloat16 xlin (bf16 storage).  `wide`: the fp32 row-gather kernels at every shape (inputs
    beyond the 16-bit operand range of the matrix-core kernels, include/gdn_hip.h "range guard")."""
    sfx = _storage(x, "x")
    if wide and not sfx:
        sfx = "_wide"
    x = _chk(x, x.dtype, name="x")
    lin_w = _chk(lin_w.detach(), name="lin.weight")
    b, n, w = x.shape
    d = lin_w.shape[0]
    xlin = torch.empty((b * n, d), dtype=x.dtype, device=x.device)
    s_i = torch.empty((b * n,), dtype=torch.float32, device=x.device)
    s_j = torch.empty_like(s_i)
    _lib.call("gdn_project_fwd" + sfx, _ptr(x), _ptr(lin_w), _ptr(terms), b, n, w, d,
              _ptr(xlin), _ptr(s_i), _ptr(s_j), _stream())
    return xlin, s_i, s_j


def attn_aggregate_fwd(xlin, s_i, s_j, graph: SensorGraph, bias, batch: int, want_alpha: bool, wide: bool = False):
    """models/graph_layer.py:65-74,82-117 -> z[B*n,d] (+ dense alpha[B*n,pitch]).  Without alpha the lists come
    from the bank-ordered copy of the graph (same z; alpha, when asked for, is in rank order as documented)."""
    sfx = _storage(xlin, "xlin")
    if wide and not sfx:
        sfx = "_wide"
    xlin = _chk(xlin, xlin.dtype, name="xlin")
    bn, d = xlin.shape
    n = bn // batch
    z = torch.empty_like(xlin)
    alpha = torch.empty((bn, graph.pitch), dtype=torch.float32, device=xlin.device) if want_alpha else None
    nbr = graph.nbr if (want_alpha or wide or d != 64) else graph.nbr_ordered()
    _lib.call("gdn_attn_aggregate_fwd" + sfx, _ptr(xlin), _ptr(_chk(s_i)), _ptr(_chk(s_j)), _ptr(nbr),
              _ptr(graph.deg), _ptr(_chk(bias.detach())), batch, n, d, graph.k, _ptr(z), _ptr(alpha), _stream())
    return z, alpha


def head_fwd(z, emb, bn1_affine, bn2_affine, out_w, out_b, batch: int, want_h2: bool = False):
    """Eval head: models/GDN.py:77-79,175-184 with out_layer_num == 1."""
    sfx = _storage(z, "z")
    z = _chk(z, z.dtype, name="z")
    bn, d = z.shape
    n = bn // batch
    out = torch.empty((batch, n), dtype=torch.float32, device=z.device)
    h2 = torch.empty((bn, d), dtype=torch.float32, device=z.device) if want_h2 else None
    _lib.call("gdn_head_fwd" + sfx, _ptr(z), _ptr(_chk(emb.detach())), _ptr(bn1_affine), _ptr(bn2_affine),
              _ptr(_chk(out_w.detach().reshape(-1))), _ptr(_chk(out_b.detach().reshape(-1))),
              batch, n, d, _ptr(out), _ptr(h2), _stream())
    return out, h2


def mlp_plan(out_layer, d_in: int):
    """Plan of the eval-mode OutLayer MLP (include/gdn_hip.h "OutLayer MLP"): (plan, hidden, layers), or None
    when the configuration is outside what gdn_mlp_fwd takes (hidden > 256)."""
    mods = list(out_layer.mlp)
    linears = [m for m in mods if isinstance(m, torch.nn.Linear)]
    bns = [m for m in mods if isinstance(m, torch.nn.BatchNorm1d)]
    layers = len(linears)
    if layers < 2 or any(not b.track_running_stats or not b.affine for b in bns):
        return None
    hidden = linears[0].out_features
    nbytes = _lib.load().gdn_mlp_plan_bytes(d_in, hidden, layers)
    if nbytes == 0:
        return None
    dev = linears[0].weight.device
    plan = torch.empty(((nbytes + 3) // 4,), dtype=torch.int32, device=dev)
    for i, (lin, bn) in enumerate(zip(linears[:-1], bns)):
        _lib.call("gdn_mlp_plan_layer", _ptr(_chk(lin.weight.detach())), _ptr(_chk(lin.bias.detach())),
                  _ptr(_chk(bn.weight.detach())), _ptr(_chk(bn.bias.detach())), _ptr(_chk(bn.running_mean)),
                  _ptr(_chk(bn.running_var)), float(bn.eps), d_in, hidden, layers, i, _ptr(plan), _stream())
    last = linears[-1]
    _lib.call("gdn_mlp_plan_out", _ptr(_chk(last.weight.detach().reshape(-1))), _ptr(_chk(last.bias.detach().reshape(-1))),
              d_in, hidden, layers, _ptr(plan), _stream())
    return plan, hidden, layers


def mlp_fwd(h2, plan_info, out: torch.Tensor | None = None):
    """h2[rows, d] -> out[rows]: OutLayer.forward (models/GDN.py:45-56) in eval mode."""
    plan, hidden, layers = plan_info
    h2 = _chk(h2, name="h2")
    rows, d_in = h2.shape
    if out is None:
        out = torch.empty((rows,), dtype=torch.float32, device=h2.device)
    _lib.call("gdn_mlp_fwd", _ptr(h2), _ptr(plan), rows, d_in, hidden, layers, _ptr(out), _stream())
    return out


def _bn_running(bn):
    """(momentum, running_mean, running_var, num_batches_tracked) pointers for the train-mode head."""
    if not bn.track_running_stats or bn.running_mean is None:
        return 0.0, None, None, None
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm momentum=None (cumulative average) is not implemented")
    return float(bn.momentum), bn.running_mean, bn.running_var, bn.num_batches_tracked


def _mask_args(mask, numel: int, scale: float):
    """(fp32 multiplier pointer, byte keep-mask pointer, scale) for the train-mode head."""
    if mask is None:
        return None, None, 1.0
    if mask.numel() != numel:
        raise ValueError("dropout mask must have batch*n*d elements")
    if mask.dtype == torch.uint8:
        return None, _ptr(_chk(mask, dtype=torch.uint8, name="dropout keep mask")), float(scale)
    return _ptr(_chk(mask, name="dropout mask")), None, 1.0


def head_train_fwd(z, emb, bn1, bn2, lin_w, lin_b, mask, batch: int, mask_scale: float = 1.0):
    """Train-mode head (models/GDN.py:77-79,175-184, out_layer_num == 1): returns (out[B,n], stats).
    `mask`: fp32 multipliers [B*n, d] (0 or 1/(1-p)), or a uint8 keep mask (1/0) with `mask_scale` =
    1/(1-p), or None.  Updates the running statistics of the BatchNorm modules `bn1`, `bn2` in place."""
    z = _chk(z, name="z")
    bn, d = z.shape
    n = bn // batch
    out = torch.empty((batch, n), dtype=torch.float32, device=z.device)
    stats = torch.empty((_lib.load().gdn_head_train_stats_bytes(d) // 8,), dtype=torch.float64, device=z.device)
    mptr, kptr, kscale = _mask_args(mask, z.numel(), mask_scale)
    m1, rm1, rv1, nb1 = _bn_running(bn1)
    m2, rm2, rv2, nb2 = _bn_running(bn2)
    _lib.call("gdn_head_train_fwd", _ptr(z), _ptr(_chk(emb.detach())), _ptr(_chk(bn1.weight.detach())),
              _ptr(_chk(bn1.bias.detach())), _ptr(_chk(bn2.weight.detach())), _ptr(_chk(bn2.bias.detach())),
              _ptr(_chk(lin_w.detach().reshape(-1))), _ptr(_chk(lin_b.detach().reshape(-1))), mptr, kptr, kscale,
              batch, n, d, float(bn1.eps), float(bn2.eps), m1, m2, _ptr(rm1), _ptr(rv1), _ptr(nb1),
              _ptr(rm2), _ptr(rv2), _ptr(nb2), _ptr(stats), _ptr(out), _stream())
    return out, stats


def head_train_bwd(d_out, z, emb, bn1_w, bn1_b, bn2_w, bn2_b, lin_w, mask, stats, eps1: float, eps2: float,
                   batch: int, mask_scale: float = 1.0):
    """Gradients of head_train_fwd: (d_z, d_emb, d_bn1_w, d_bn1_b, d_bn2_w, d_bn2_b, d_lin_w, d_lin_b)."""
    d_out = _chk(d_out, name="d_out")
    bn, d = z.shape
    n = bn // batch
    dev = z.device
    ws = torch.empty((_lib.load().gdn_head_train_workspace_bytes(n, d) // 8,), dtype=torch.float64, device=dev)
    d_z = torch.empty_like(z)
    d_emb = torch.empty((n, d), dtype=torch.float32, device=dev)
    small = torch.empty((5 * d + 1,), dtype=torch.float32, device=dev)
    g1w, g1b, g2w, g2b, glw = (small[i * d:(i + 1) * d] for i in range(5))
    glb = small[5 * d:]
    mptr, kptr, kscale = _mask_args(mask, z.numel(), mask_scale)
    _lib.call("gdn_head_train_bwd", _ptr(d_out), _ptr(z), _ptr(_chk(emb)), _ptr(_chk(bn1_w)), _ptr(_chk(bn1_b)),
              _ptr(_chk(bn2_w)), _ptr(_chk(bn2_b)), _ptr(_chk(lin_w.reshape(-1))), mptr, kptr, kscale, _ptr(stats),
              batch, n, d, eps1, eps2, _ptr(ws), _ptr(d_z), _ptr(d_emb), _ptr(g1w), _ptr(g1b), _ptr(g2w),
              _ptr(g2b), _ptr(glw), _ptr(glb), _stream())
    return d_z, d_emb, g1w, g1b, g2w, g2b, glw, glb


def head_train_fwd_act(z, emb, bn1, bn2, mask, batch: int, mask_scale: float = 1.0):
    """Train-mode head in front of an OutLayer MLP (out_layer_num > 1): the passes of head_train_fwd ending at
    the [B*n, d] activation after dropout (models/GDN.py:182) instead of the fused Linear(d->1).
    Returns (act, stats)."""
    z = _chk(z, name="z")
    bn, d = z.shape
    n = bn // batch
    act = torch.empty_like(z)
    stats = torch.empty((_lib.load().gdn_head_train_stats_bytes(d) // 8,), dtype=torch.float64, device=z.device)
    mptr, kptr, kscale = _mask_args(mask, z.numel(), mask_scale)
    m1, rm1, rv1, nb1 = _bn_running(bn1)
    m2, rm2, rv2, nb2 = _bn_running(bn2)
    _lib.call("gdn_head_train_fwd_act", _ptr(z), _ptr(_chk(emb.detach())), _ptr(_chk(bn1.weight.detach())),
              _ptr(_chk(bn1.bias.detach())), _ptr(_chk(bn2.weight.detach())), _ptr(_chk(bn2.bias.detach())),
              mptr, kptr, kscale, None, 0.0, batch, n, d, float(bn1.eps), float(bn2.eps), m1, m2, _ptr(rm1), _ptr(rv1),
              _ptr(nb1), _ptr(rm2), _ptr(rv2), _ptr(nb2), _ptr(stats), _ptr(act), 0, _stream())
    return act, stats


def head_train_bwd_act(d_act, z, emb, bn1_w, bn1_b, bn2_w, bn2_b, mask, stats, eps1: float, eps2: float,
                       batch: int, mask_scale: float = 1.0):
    """Gradients of head_train_fwd_act: (d_z, d_emb, d_bn1_w, d_bn1_b, d_bn2_w, d_bn2_b)."""
    d_act = _chk(d_act, name="d_act")
    bn, d = z.shape
    n = bn // batch
    dev = z.device
    ws = torch.empty((_lib.load().gdn_head_train_workspace_bytes(n, d) // 8,), dtype=torch.float64, device=dev)
    d_z = torch.empty_like(z)
    d_emb = torch.empty((n, d), dtype=torch.float32, device=dev)
    small = torch.empty((4 * d,), dtype=torch.float32, device=dev)
    g1w, g1b, g2w, g2b = (small[i * d:(i + 1) * d] for i in range(4))
    mptr, kptr, kscale = _mask_args(mask, z.numel(), mask_scale)
    _lib.call("gdn_head_train_bwd_act", _ptr(d_act), _ptr(z), _ptr(_chk(emb)), _ptr(_chk(bn1_w)), _ptr(_chk(bn1_b)),
              _ptr(_chk(bn2_w)), _ptr(_chk(bn2_b)), mptr, kptr, kscale, None, 0.0, _ptr(stats), batch, n, d, eps1, eps2,
              _ptr(ws), _ptr(d_z), _ptr(d_emb), _ptr(g1w), _ptr(g1b), _ptr(g2w), _ptr(g2b), 0, _stream())
    return d_z, d_emb, g1w, g1b, g2w, g2b


def _ptr_array(tensors):
    """Host array of device pointers (None -> null) for the entry points that take one per layer."""
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def mlp_train_layers(out_layer):
    """[(Linear, BatchNorm1d), ...] of the hidden layers and the final Linear of a reference OutLayer
    (models/GDN.py:27-45), or None when its structure is not that."""
    mods = list(out_layer.mlp)
    if len(mods) < 4 or (len(mods) - 1) % 3 != 0:
        return None
    hidden = []
    for i in range(0, len(mods) - 1, 3):
        lin, bn, act = mods[i], mods[i + 1], mods[i + 2]
        if not (isinstance(lin, torch.nn.Linear) and isinstance(bn, torch.nn.BatchNorm1d) and
                isinstance(act, torch.nn.ReLU) and lin.bias is not None and bn.affine):
            return None
        if bn.track_running_stats and bn.momentum is None:
            return None
        hidden.append((lin, bn))
    last = mods[-1]
    if not isinstance(last, torch.nn.Linear) or last.out_features != 1 or last.bias is None:
        return None
    return hidden, last


def mlp_train_supported(out_layer, d_in: int, rows: int) -> bool:
    """True when gdn_mlp_train_fwd/bwd take this OutLayer: 2..8 layers, d_in (<= 256) a multiple of 4, hidden
    1..512, every hidden layer of the same width."""
    parts = mlp_train_layers(out_layer)
    if parts is None:
        return False
    hidden, _last = parts
    h = hidden[0][0].out_features
    if any(lin.out_features != h for lin, _ in hidden) or hidden[0][0].in_features != d_in:
        return False
    return _lib.load().gdn_mlp_train_saved_bytes(rows, d_in, h, len(hidden) + 1) > 0


def mlp_eval_wide_supported(out_layer, d_in: int) -> bool:
    """True when gdn_mlp_eval_fwd takes this OutLayer (eval mode, hidden up to 512, running
    statistics tracked): the path for widths beyond the one-launch chain (mlp_plan / mlp_fwd: hidden <= 256)."""
    parts = mlp_train_layers(out_layer)
    if parts is None:
        return False
    hidden, _last = parts
    h = hidden[0][0].out_features
    if any(lin.out_features != h for lin, _ in hidden) or hidden[0][0].in_features != d_in:
        return False
    if any(bn.running_mean is None or not bn.track_running_stats for _lin, bn in hidden):
        return False
    return _lib.load().gdn_mlp_eval_workspace_bytes(2, d_in, h, len(hidden) + 1) > 0


def mlp_eval_wide(h2, out_layer, out: torch.Tensor | None = None):
    """h2[rows, d] -> out[rows]: OutLayer.forward (models/GDN.py:45-56) in eval mode on the fp32 matrix-core GEMM
    kernels (gdn_mlp_eval_fwd)."""
    h2 = _chk(h2, name="h2")
    rows, d_in = h2.shape
    hidden, last = mlp_train_layers(out_layer)
    h, layers = hidden[0][0].out_features, len(hidden) + 1
    ws = torch.empty((_lib.load().gdn_mlp_eval_workspace_bytes(rows, d_in, h, layers),), dtype=torch.uint8, device=h2.device)
    if out is None:
        out = torch.empty((rows,), dtype=torch.float32, device=h2.device)
    params, running, eps = [], [], []
    for lin, bn in hidden:
        params += [_chk(lin.weight.detach()), _chk(lin.bias.detach()), _chk(bn.weight.detach()), _chk(bn.bias.detach())]
        running += [_chk(bn.running_mean), _chk(bn.running_var)]
        eps.append(float(bn.eps))
    _lib.call("gdn_mlp_eval_fwd", _ptr(h2), _ptr_array(params), _ptr_array(running), (ctypes.c_float * len(eps))(*eps),
              _ptr(_chk(last.weight.detach().reshape(-1))), _ptr(_chk(last.bias.detach().reshape(-1))),
              rows, d_in, h, layers, _ptr(ws), _ptr(out), _stream())
    return out


def mlp_train_fwd(act, out_layer):
    """Train-mode OutLayer MLP forward (models/GDN.py:47-56 under model.train(): Linear, batch-statistics
    BatchNorm, ReLU per hidden layer, then Linear(hidden->1)).  Returns (out[rows], saved); updates the
    BatchNorm running statistics in place."""
    act = _chk(act, name="act")
    rows, d_in = act.shape
    hidden, last = mlp_train_layers(out_layer)
    h, layers = hidden[0][0].out_features, len(hidden) + 1
    lib = _lib.load()
    saved = torch.empty((lib.gdn_mlp_train_saved_bytes(rows, d_in, h, layers),), dtype=torch.uint8, device=act.device)
    ws = torch.empty((lib.gdn_mlp_train_workspace_bytes(rows, d_in, h, layers),), dtype=torch.uint8, device=act.device)
    out = torch.empty((rows,), dtype=torch.float32, device=act.device)
    params, running, batches, eps, mom = [], [], [], [], []
    for lin, bn in hidden:
        params += [_chk(lin.weight.detach()), _chk(lin.bias.detach()), _chk(bn.weight.detach()), _chk(bn.bias.detach())]
        m, rm, rv, nb = _bn_running(bn)
        running += [rm, rv]
        batches.append(nb)
        eps.append(float(bn.eps))
        mom.append(m)
    _lib.call("gdn_mlp_train_fwd", _ptr(act), _ptr_array(params), _ptr_array(running), _ptr_array(batches),
              (ctypes.c_float * len(eps))(*eps), (ctypes.c_float * len(mom))(*mom),
              _ptr(_chk(last.weight.detach().reshape(-1))), _ptr(_chk(last.bias.detach().reshape(-1))),
              rows, d_in, h, layers, _ptr(saved), _ptr(ws), _ptr(out), _stream())
    return out, saved


def mlp_train_bwd(d_out, act, params, out_w, saved, d_in: int, hidden: int, layers: int):
    """Gradients of mlp_train_fwd.  `params` = [W, b, gamma, beta] * (layers-1).  Returns
    (d_act, [dW, db, dgamma, dbeta] * (layers-1), d_out_w[hidden], d_out_b[1])."""
    d_out = _chk(d_out.reshape(-1), name="d_out")
    rows = act.shape[0]
    dev = act.device
    ws = torch.empty((_lib.load().gdn_mlp_train_workspace_bytes(rows, d_in, hidden, layers),), dtype=torch.uint8, device=dev)
    grads = []
    for l in range(layers - 1):
        k = d_in if l == 0 else hidden
        grads += [torch.empty((hidden, k), dtype=torch.float32, device=dev)] + \
                 [torch.empty((hidden,), dtype=torch.float32, device=dev) for _ in range(3)]
    d_ow = torch.empty((hidden,), dtype=torch.float32, device=dev)
    d_ob = torch.empty((1,), dtype=torch.float32, device=dev)
    d_act = torch.empty_like(act)
    _lib.call("gdn_mlp_train_bwd", _ptr(d_out), _ptr(act), _ptr_array([_chk(p) for p in params]),
              _ptr(_chk(out_w.reshape(-1))), rows, d_in, hidden, layers, _ptr(saved), _ptr(ws), _ptr_array(grads),
              _ptr(d_ow), _ptr(d_ob), _ptr(d_act), _stream())
    return d_act, grads, d_ow, d_ob


def forward_fused(x, lin_w, terms, graph: SensorGraph, gnn_bias, emb, bn1_affine, bn2_affine, out_w, out_b,
                  out: torch.Tensor | None = None):
    """One launch from x[B,n,w] to out[B,n]: models/GDN.py:122-187 under model.eval().
    bfloat16 x selects the bf16-storage kernel."""
    sfx = _storage(x, "x")
    x = _chk(x, x.dtype, name="x")
    b, n, w = x.shape
    lin_w = _chk(lin_w.detach(), name="lin.weight")
    d = lin_w.shape[0]
    if out is None:
        out = torch.empty((b, n), dtype=torch.float32, device=x.device)
    _lib.call("gdn_forward_fused" + sfx, _ptr(x), _ptr(lin_w), _ptr(terms), _ptr(graph.nbr), _ptr(graph.deg),
              _ptr(_chk(gnn_bias.detach())), _ptr(_chk(emb.detach())), _ptr(bn1_affine), _ptr(bn2_affine),
              _ptr(_chk(out_w.detach().reshape(-1))), _ptr(_chk(out_b.detach().reshape(-1))),
              b, n, w, d, graph.k, _ptr(out), _stream())
    return out


def fused_plan(lin_w, terms, graph: SensorGraph, gnn_bias, emb, bn1_affine, bn2_affine, out_w, out_b,
               bf16_storage: bool = False):
    """Per-launch constants of the fused forward, precomputed (include/gdn_hip.h "plans"); None when the
    shape is not on the matrix-core path.  Rebuild after every parameter update."""
    if os.environ.get("GDN_FUSED_PATH", "").startswith("v"):      # diagnostic: keep the fp32 VALU kernels
        return None
    lin_w = _chk(lin_w.detach(), name="lin.weight")
    d, w = lin_w.shape
    n = emb.shape[0]
    nbytes = _lib.load().gdn_fused_plan_bytes(n, w, d, graph.k, int(bf16_storage))
    if nbytes == 0:
        return None
    plan = torch.empty(((nbytes + 3) // 4,), dtype=torch.int32, device=emb.device)
    _lib.call("gdn_fused_plan_build", _ptr(lin_w), _ptr(terms), _ptr(graph.nbr), _ptr(graph.deg),
              _ptr(_chk(gnn_bias.detach())), _ptr(_chk(emb.detach())), _ptr(bn1_affine), _ptr(bn2_affine),
              _ptr(_chk(out_w.detach().reshape(-1))), _ptr(_chk(out_b.detach().reshape(-1))),
              n, w, d, graph.k, int(bf16_storage), _ptr(plan), _stream())
    return plan


def fused_plan_limit(plan: torch.Tensor, n: int, w: int, d: int, k: int, bf16_storage: bool = False) -> torch.Tensor:
    """The plan's x limit (include/gdn_hip.h "range guard") as a 0-d float32 DEVICE tensor (a view of the plan:
    reading it on the host is a synchronisation, done once per resident series)."""
    off = _lib.load().gdn_fused_plan_limit_offset(n, w, d, k, int(bf16_storage))
    return plan[off // 4].view(torch.float32)


def forward_fused_series(series, first: int, batch: int, w: int, lin_w, terms, graph: SensorGraph, gnn_bias, emb,
                         bn1_affine, bn2_affine, out_w, out_b, out: torch.Tensor | None = None):
    """Fused eval forward of `batch` stride-1 windows read straight from series[n, T]
    (datasets/TimeDataset.py:46-49 without the w-fold copy)."""
    series = _chk(series, name="series")
    n, t_len = series.shape
    lin_w = _chk(lin_w.detach(), name="lin.weight")
    d = lin_w.shape[0]
    if out is None:
        out = torch.empty((batch, n), dtype=torch.float32, device=series.device)
    _lib.call("gdn_forward_fused_series", _ptr(series), t_len, first, _ptr(lin_w), _ptr(terms), _ptr(graph.nbr),
              _ptr(graph.deg), _ptr(_chk(gnn_bias.detach())), _ptr(_chk(emb.detach())), _ptr(bn1_affine),
              _ptr(bn2_affine), _ptr(_chk(out_w.detach().reshape(-1))), _ptr(_chk(out_b.detach().reshape(-1))),
              batch, n, w, d, graph.k, _ptr(out), _stream())
    return out


def attn_aggregate_bwd(d_z, xlin, alpha, s_i, s_j, graph: SensorGraph, batch: int, wide: bool = False):
    """Gradient of attn_aggregate_fwd w.r.t. xlin, s_i, s_j and bias."""
    d_z = _chk(d_z, name="d_z")
    bn, d = d_z.shape
    n = bn // batch
    d_xlin = torch.empty_like(d_z)
    d_si = torch.empty((bn,), dtype=torch.float32, device=d_z.device)
    d_sj = torch.empty_like(d_si)
    d_bias = torch.empty((d,), dtype=torch.float32, device=d_z.device)
    # the matrix-core backward (n <= 127, d = 64) does not read the reverse lists: they are not even built then
    rent, rlen = graph.reverse() if wide or _lib.load().gdn_attn_aggregate_bwd_uses_reverse(n, d, graph.k) else (None, None)
    # [ticket, zero on entry | d_bias partial rows | d_pi tables when they exceed LDS]: a fresh one per call, so
    # calls on different streams never share a ticket (NativeTrainStep owns one for its stream instead)
    nbytes = _lib.load().gdn_attn_aggregate_bwd_workspace_bytes(batch, n, d, graph.k)
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=d_z.device)
    ws[:4].zero_()
    _lib.call("gdn_attn_aggregate_bwd_wide" if wide else "gdn_attn_aggregate_bwd", _ptr(d_z), _ptr(_chk(xlin)),
              _ptr(_chk(alpha)), _ptr(_chk(s_i)),
              _ptr(_chk(s_j)), _ptr(graph.nbr), _ptr(rent), _ptr(rlen), batch, n, d, graph.k,
              _ptr(d_xlin), _ptr(d_si), _ptr(d_sj), _ptr(d_bias), _ptr(ws), _stream())
    return d_xlin, d_si, d_sj, d_bias


def project_bwd(x, d_xlin, d_si, d_sj, d: int):
    """Gradients of project_fwd: d_lin_w[d,w] (direct term), d_a[2,64], d_c[2,n]."""
    x = _chk(x, name="x")
    b, n, w = x.shape
    ws = torch.empty((_lib.load().gdn_project_bwd_workspace_bytes(n, w, d) // 4,), dtype=torch.float32,
                     device=x.device)
    flat = torch.empty((d * w + 128 + 2 * n,), dtype=torch.float32, device=x.device)
    d_lin_w, d_a, d_c = flat[:d * w].view(d, w), flat[d * w:d * w + 128].view(2, 64), flat[d * w + 128:].view(2, n)
    _lib.call("gdn_project_bwd", _ptr(x), _ptr(_chk(d_xlin)), _ptr(_chk(d_si)), _ptr(_chk(d_sj)),
              b, n, w, d, _ptr(ws), _ptr(d_lin_w), _ptr(d_a), _ptr(d_c), _stream())
    return d_lin_w, d_a, d_c


def mse_workspace(device) -> torch.Tensor:
    """Zeroed scratch for mse_loss_grad; allocate once and reuse (every call leaves it zeroed)."""
    return torch.zeros((_lib.load().gdn_mse_workspace_bytes() // 8,), dtype=torch.float64, device=device)


def mse_loss_grad(out, y, workspace, loss=None, d_out=None):
    """(loss, d_out): F.mse_loss(out, y, reduction='mean') (train.py:20-23) and d loss / d out."""
    out, y = _chk(out, name="out"), _chk(y, name="y")
    if out.shape != y.shape:
        raise ValueError(f"shape mismatch: {tuple(out.shape)} vs {tuple(y.shape)}")
    if loss is None:
        loss = torch.empty((), dtype=torch.float32, device=out.device)
    if d_out is None:
        d_out = torch.empty_like(out)
    _lib.call("gdn_mse_loss_grad", _ptr(out), _ptr(y), out.numel(), _ptr(workspace), _ptr(loss), _ptr(d_out),
              _stream())
    return loss, d_out


def terms_bwd(lin_w, att_i, att_j, att_em_i, att_em_j, emb, d_lin_w, d_a, d_c):
    """Chain rule through node_terms; d_lin_w (the direct term from project_bwd) is completed in place.
    Returns (d_lin_w, d_att_i, d_att_j, d_att_em_i, d_att_em_j, d_emb)."""
    d, w = lin_w.shape
    n = emb.shape[0]
    small = torch.empty((4 * d,), dtype=torch.float32, device=emb.device)
    d_emb = torch.empty((n, d), dtype=torch.float32, device=emb.device)
    _lib.call("gdn_terms_bwd", _ptr(_chk(lin_w)), _ptr(_chk(att_i)), _ptr(_chk(att_j)), _ptr(_chk(att_em_i)),
              _ptr(_chk(att_em_j)), _ptr(_chk(emb)), _ptr(d_a), _ptr(d_c), n, d, w, _ptr(d_lin_w),
              _ptr(small), _ptr(small[d:]), _ptr(small[2 * d:]), _ptr(small[3 * d:]), _ptr(d_emb), _stream())
    return (d_lin_w, small[:d].view_as(att_i), small[d:2 * d].view_as(att_j), small[2 * d:3 * d].view_as(att_em_i),
            small[3 * d:].view_as(att_em_j), d_emb)


def score_workspace(t: int, n: int, device) -> torch.Tensor:
    nbytes = _lib.load().gdn_score_workspace_bytes(t, n)
    return torch.empty(((nbytes + 7) // 8,), dtype=torch.float64, device=device)


def score_keys(pred, gt, pitch: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """keys[n, pitch] float64 = |pred-gt| transposed (radix keys); slots t..pitch-1 = filler."""
    pred, gt = _chk(pred, name="pred"), _chk(gt, name="gt")
    t, n = pred.shape
    keys = out if out is not None else torch.empty((n, pitch), dtype=torch.float64, device=pred.device)
    if keys.shape != (n, pitch) or keys.dtype != torch.float64 or not keys.is_contiguous():
        raise ValueError("keys buffer must be a contiguous float64 [n, pitch] tensor")
    _lib.call("gdn_score_keys", _ptr(pred), _ptr(gt), t, n, pitch, _ptr(keys), _stream())
    return keys


def score_select_workspace(blocks: int, n: int, pitch: int, device) -> torch.Tensor:
    nbytes = _lib.load().gdn_score_select_workspace_bytes(blocks, n, pitch)
    return torch.empty(((nbytes + 7) // 8,), dtype=torch.float64, device=device)


def score_select(keys, blocks: int, n: int, pitch: int, total: int, ws: torch.Tensor | None = None,
                 out: torch.Tensor | None = None) -> torch.Tensor:
    """Median / IQR per sensor over keys[blocks, n, pitch] holding `total` real keys per sensor.
    `ws` (score_select_workspace) and `out` [n, 2] float64 may be preallocated (no allocation per call)."""
    keys = _chk(keys, torch.float64, "keys")
    if ws is None:
        ws = score_select_workspace(blocks, n, pitch, keys.device)
    if out is None:
        out = torch.empty((n, 2), dtype=torch.float64, device=keys.device)
    _lib.call("gdn_score_select", _ptr(keys), blocks, n, pitch, total, _ptr(ws), _ptr(out), _stream())
    return out


def score_quantiles(pred, gt):
    """Per-sensor median and IQR of |pred-gt| over all ticks (util/data.py:75-82), float64.
    pred, gt: fp32 [t, n].  Returns med_iqr[n, 2]."""
    pred, gt = _chk(pred, name="pred"), _chk(gt, name="gt")
    t, n = pred.shape
    ws = score_workspace(t, n, pred.device)
    out = torch.empty((n, 2), dtype=torch.float64, device=pred.device)
    _lib.call("gdn_score_quantiles", _ptr(pred), _ptr(gt), t, n, _ptr(ws), _ptr(out), _stream())
    return out


def score_smooth_max(pred, gt, med_iqr, want_scores: bool = True, first_tick: int = 0,
                     halo_pred=None, halo_gt=None, anomaly: torch.Tensor | None = None):
    """evaluate.py:54-68 + the max over sensors of :131-139.  Returns (scores[n,t] | None, anomaly[t]);
    `anomaly` may be a preallocated float64 [t] buffer."""
    pred, gt = _chk(pred, name="pred"), _chk(gt, name="gt")
    t, n = pred.shape
    scores = torch.empty((n, t), dtype=torch.float64, device=pred.device) if want_scores else None
    if anomaly is None:
        anomaly = torch.empty((t,), dtype=torch.float64, device=pred.device)
    hp = None if halo_pred is None else _chk(halo_pred, name="halo_pred")
    hg = None if halo_gt is None else _chk(halo_gt, name="halo_gt")
    _lib.call("gdn_score_smooth_max", _ptr(pred), _ptr(gt), _ptr(_chk(med_iqr, torch.float64)), t, n,
              first_tick, _ptr(hp), _ptr(hg), _ptr(scores), _ptr(anomaly), _stream())
    return scores, anomaly
