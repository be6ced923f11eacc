"""Host-side mirror of the reference's model interface (models/GDN.py, models/graph_layer.py).

Same class names, constructor arguments, `forward(data, org_edge_index) -> [batch, node_num]`
signature, side attributes (`learned_graph`, `att_weight_1`, `edge_index_1`) and
`state_dict()` keys / shapes, so a reference checkpoint loads here and vice versa, and
`train.py` / `test.py`-style loops run unchanged.

The torch modules below are PARAMETER CONTAINERS (they give the reference's state_dict
layout and initialisation stream); the arithmetic of the hot path runs in hand-written HIP
kernels reached through gdn_amd.ops -> libgdn_hip.so.  There is no CPU path: calling
forward on CPU tensors raises.

  eval, out_layer_num == 1 : one fused launch  x[B,N,W] -> out[B,N]  (the planned matrix-core kernel, with the
                             range guard: inputs beyond the 16-bit operand range are detected on the device
                             and the launch is redone by the fp32 row-gather kernel — include/gdn_hip.h)
  eval, otherwise          : project -> attention/aggregate -> head kernels, then the OutLayer MLP as ONE
                             matrix-core launch (gdn_mlp_fwd; hidden 257..512: gdn_mlp_eval_fwd, fp32 matrix-core
                             GEMMs; beyond 512: torch's library GEMMs)
  train, out_layer_num == 1: project + attention/aggregate AND the BatchNorm/ReLU/embedding/dropout/Linear
                             head run as HIP kernels forward and backward (the autograd.Functions below)
  train, otherwise         : the same, with the head passes ending at the dropped-out activation and the
                             OutLayer MLP on the fp32 matrix cores (gdn_mlp_train_fwd/bwd, any hidden width
                             up to 512; beyond: torch ops)
                             (`loss.backward()` reaches every parameter exactly as in the reference)
  harness.NativeTrainStep is the same arithmetic without autograd (flat buffers, in-kernel dropout, gdn_adam_step).
"""
from __future__ import annotations

import contextlib
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops


def _glorot(t: torch.Tensor) -> None:
    """torch_geometric.nn.inits.glorot (1.5.0): U(-a, a), a = sqrt(6 / (size(-2) + size(-1)))."""
    bound = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    t.data.uniform_(-bound, bound)


class GraphLayer(nn.Module):
    """Parameters of the reference GraphLayer (models/graph_layer.py:12-49), heads = 1,
    concat = False as hard-wired by GNNLayer (models/GDN.py:65)."""

    def __init__(self, in_channels, out_channels, heads=1, concat=False, negative_slope=0.2,
                 dropout=0, bias=True, inter_dim=-1, **kwargs):
        super().__init__()
        if heads != 1 or concat or dropout != 0 or not bias or negative_slope != 0.2:
            raise NotImplementedError("the GDN hot path uses heads=1, concat=False, dropout=0, "
                                      "negative_slope=0.2, bias=True (models/GDN.py:65)")
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.concat, self.negative_slope, self.dropout = concat, negative_slope, dropout
        self.lin = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.att_i = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_j = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_em_i = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_em_j = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        # same order and RNG draws as models/graph_layer.py:41-49
        _glorot(self.lin.weight)
        _glorot(self.att_i)
        _glorot(self.att_j)
        self.att_em_i.data.fill_(0)
        self.att_em_j.data.fill_(0)
        self.bias.data.fill_(0)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, heads={self.heads})"


class _GraphAttentionFn(torch.autograd.Function):
    """project + attention/aggregate with hand-written forward and backward kernels
    (reference models/graph_layer.py:53-117 and the autograd graph behind train.py:72)."""

    @staticmethod
    def forward(ctx, x, lin_w, att_i, att_j, att_em_i, att_em_j, emb, bias, graph, batch, terms, wide=False):
        xlin, s_i, s_j = ops.project_fwd(x, lin_w, terms, wide=wide)
        z, alpha = ops.attn_aggregate_fwd(xlin, s_i, s_j, graph, bias, batch, want_alpha=True, wide=wide)
        ctx.save_for_backward(x, lin_w, att_i, att_j, att_em_i, att_em_j, emb, xlin, s_i, s_j, alpha)
        ctx.graph, ctx.batch, ctx.wide = graph, batch, wide
        ctx.mark_non_differentiable(alpha)
        return z, alpha

    @staticmethod
    def backward(ctx, d_z, _d_alpha):
        x, lin_w, att_i, att_j, att_em_i, att_em_j, emb, xlin, s_i, s_j, alpha = ctx.saved_tensors
        d_xlin, d_si, d_sj, d_bias = ops.attn_aggregate_bwd(d_z.contiguous(), xlin, alpha, s_i, s_j,
                                                            ctx.graph, ctx.batch, wide=ctx.wide)
        d_lin_w, d_a, d_c = ops.project_bwd(x, d_xlin, d_si, d_sj, lin_w.shape[0])
        d_lin_w, d_att_i, d_att_j, d_att_em_i, d_att_em_j, d_emb = ops.terms_bwd(
            lin_w, att_i, att_j, att_em_i, att_em_j, emb, d_lin_w, d_a, d_c)
        return None, d_lin_w, d_att_i, d_att_j, d_att_em_i, d_att_em_j, d_emb, d_bias, None, None, None, None


class _HeadTrainFn(torch.autograd.Function):
    """Train-mode BN+ReLU, x embedding, BN+ReLU, dropout, Linear(d->1) (models/GDN.py:77-79,:175-184)
    as three streaming HIP passes forward and three backward; batch statistics in fp64."""

    @staticmethod
    def forward(ctx, z, emb, bn1_w, bn1_b, bn2_w, bn2_b, lin_w, lin_b, mask, mask_scale, bn1, bn2, batch):
        out, stats = ops.head_train_fwd(z, emb, bn1, bn2, lin_w, lin_b, mask, batch, mask_scale)
        ctx.save_for_backward(z, emb, bn1_w, bn1_b, bn2_w, bn2_b, lin_w, stats)
        ctx.mask, ctx.mask_scale, ctx.eps, ctx.batch = mask, mask_scale, (float(bn1.eps), float(bn2.eps)), batch
        ctx.shapes = (lin_w.shape, lin_b.shape)
        return out

    @staticmethod
    def backward(ctx, d_out):
        z, emb, bn1_w, bn1_b, bn2_w, bn2_b, lin_w, stats = ctx.saved_tensors
        d_z, d_emb, g1w, g1b, g2w, g2b, glw, glb = ops.head_train_bwd(
            d_out.contiguous(), z, emb, bn1_w, bn1_b, bn2_w, bn2_b, lin_w, ctx.mask, stats,
            ctx.eps[0], ctx.eps[1], ctx.batch, ctx.mask_scale)
        return (d_z, d_emb, g1w, g1b, g2w, g2b, glw.view(ctx.shapes[0]), glb.view(ctx.shapes[1]),
                None, None, None, None, None)


class _MlpHeadTrainFn(torch.autograd.Function):
    """out_layer_num > 1 in train mode: the head passes up to the dropped-out activation, then the OutLayer
    MLP (Linear, batch-statistics BatchNorm, ReLU per hidden layer, Linear(hidden->1)) on the fp32 matrix
    cores, forward and backward (models/GDN.py:27-56,:77-79,:175-184)."""

    @staticmethod
    def forward(ctx, z, emb, bn1_w, bn1_b, bn2_w, bn2_b, mask, mask_scale, bn1, bn2, out_layer, batch, *mlp_params):
        act, stats = ops.head_train_fwd_act(z, emb, bn1, bn2, mask, batch, mask_scale)
        out, saved = ops.mlp_train_fwd(act, out_layer)
        hidden, last = ops.mlp_train_layers(out_layer)
        ctx.save_for_backward(z, emb, bn1_w, bn1_b, bn2_w, bn2_b, stats, act, saved, *mlp_params)
        ctx.mask, ctx.mask_scale, ctx.eps, ctx.batch = mask, mask_scale, (float(bn1.eps), float(bn2.eps)), batch
        ctx.geom = (act.shape[1], hidden[0][0].out_features, len(hidden) + 1)
        ctx.shapes = [p.shape for p in mlp_params]
        return out

    @staticmethod
    def backward(ctx, d_out):
        z, emb, bn1_w, bn1_b, bn2_w, bn2_b, stats, act, saved, *mlp_params = ctx.saved_tensors
        d_in, hidden, layers = ctx.geom
        d_act, grads, d_ow, d_ob = ops.mlp_train_bwd(d_out.contiguous(), act, mlp_params[:-2], mlp_params[-2], saved,
                                                      d_in, hidden, layers)
        d_z, d_emb, g1w, g1b, g2w, g2b = ops.head_train_bwd_act(
            d_act, z, emb, bn1_w, bn1_b, bn2_w, bn2_b, ctx.mask, stats, ctx.eps[0], ctx.eps[1], ctx.batch,
            ctx.mask_scale)
        mlp_grads = [g.view(shape) for g, shape in zip(grads + [d_ow, d_ob], ctx.shapes)]
        return (d_z, d_emb, g1w, g1b, g2w, g2b, None, None, None, None, None, None, *mlp_grads)


class GNNLayer(nn.Module):
    """models/GDN.py:60-79.  `att_weight_1` / `edge_index_1` are materialised lazily in the
    reference's edge-list format from the dense per-target attention table."""

    def __init__(self, in_channel, out_channel, inter_dim=0, heads=1, node_num=100):
        super().__init__()
        self.gnn = GraphLayer(in_channel, out_channel, inter_dim=inter_dim, heads=heads, concat=False)
        self.bn = nn.BatchNorm1d(out_channel)
        self.relu = nn.ReLU()
        self.leaky_relu = nn.LeakyReLU()
        self._dense = None      # (alpha[B*n, pitch], graph, batch) of the last forward, or a thunk

    def _set_dense(self, value):
        self._dense = value

    def _materialise(self):
        if self._dense is None:
            raise AttributeError("att_weight_1 / edge_index_1 exist only after a forward pass")
        if callable(self._dense):
            self._dense = self._dense()
        return self._dense

    @property
    def edge_index_1(self) -> torch.Tensor:
        """[2, E'] int64: non-self edges window-major / target-major / rank order, then one
        self-loop per node (models/graph_layer.py:61-63 applied to models/GDN.py:161-165)."""
        _, graph, batch = self._materialise()
        n, k = graph.topk.shape
        dev = graph.topk.device
        src = graph.topk.reshape(-1)
        tgt = torch.arange(n, device=dev).repeat_interleave(k)
        keep = src != tgt
        src, tgt = src[keep], tgt[keep]
        shift = (torch.arange(batch, device=dev) * n).view(-1, 1)
        loops = torch.arange(batch * n, device=dev)
        return torch.stack((torch.cat(((src.view(1, -1) + shift).reshape(-1), loops)),
                            torch.cat(((tgt.view(1, -1) + shift).reshape(-1), loops))))

    @property
    def att_weight_1(self) -> torch.Tensor:
        """[E', 1, 1] attention weights in `edge_index_1` order."""
        alpha, graph, batch = self._materialise()
        n = graph.n
        a = alpha.view(batch, n, graph.pitch)
        deg = graph.deg.long()
        slots = torch.arange(graph.pitch, device=alpha.device).view(1, -1)
        nonself = slots < (deg.view(-1, 1) - 1)                       # [n, pitch]
        self_w = torch.gather(a, 2, (deg - 1).view(1, n, 1).expand(batch, n, 1)).reshape(-1)
        return torch.cat((a[:, nonself].reshape(-1), self_w)).view(-1, 1, 1)


class OutLayer(nn.Module):
    """models/GDN.py:27-56."""

    def __init__(self, in_num, node_num, layer_num, inter_num=512):
        super().__init__()
        modules = []
        for i in range(layer_num):
            if i == layer_num - 1:
                modules.append(nn.Linear(in_num if layer_num == 1 else inter_num, 1))
            else:
                modules.append(nn.Linear(in_num if i == 0 else inter_num, inter_num))
                modules.append(nn.BatchNorm1d(inter_num))
                modules.append(nn.ReLU())
        self.mlp = nn.ModuleList(modules)

    def forward(self, x):
        out = x
        for mod in self.mlp:
            if isinstance(mod, nn.BatchNorm1d):
                out = mod(out.permute(0, 2, 1)).permute(0, 2, 1)
            else:
                out = mod(out)
        return out


class _EvalConstants:
    __slots__ = ("key", "graph", "terms", "bn1", "bn2", "fused_args", "plans", "mlp", "stream", "ready", "guards",
                 "limits")


class GDN(nn.Module):
    """Drop-in for the reference `GDN` (models/GDN.py:82-187)."""

    def __init__(self, edge_index_sets, node_num, dim=64, out_layer_inter_dim=256, input_dim=10,
                 out_layer_num=1, topk=20):
        super().__init__()
        if len(edge_index_sets) != 1:
            # models/GDN.py:176 multiplies a [B,N,dim*sets] tensor by a [N,dim] one: only 1 set works
            raise NotImplementedError("the reference forward only runs with exactly one edge-index set")
        self.edge_index_sets = edge_index_sets          # kept for API parity; forward ignores it (GDN.py:122)
        embed_dim = dim
        # construction order = the reference's, so a given torch seed yields identical parameters
        self.embedding = nn.Embedding(node_num, embed_dim)
        self.bn_outlayer_in = nn.BatchNorm1d(embed_dim)
        edge_set_num = len(edge_index_sets)
        self.gnn_layers = nn.ModuleList([
            GNNLayer(input_dim, dim, inter_dim=dim + embed_dim, heads=1) for _ in range(edge_set_num)])
        self.node_embedding = None
        self.topk = topk
        self.learned_graph = None
        self.out_layer_num = out_layer_num
        self.out_layer = OutLayer(dim * edge_set_num, node_num, out_layer_num, inter_num=out_layer_inter_dim)
        self.cache_edge_index_sets = [None] * edge_set_num
        self.cache_embed_index = None
        self.dp = nn.Dropout(0.2)
        self.injected_graph = None      # optional [N,K] int64 table overriding the learned top-k
        # Range of the inputs (include/gdn_hip.h "range guard"): the matrix-core kernels carry fp32 inputs as two
        # f16 terms, which end at 65504.  "auto": the eval fast path detects out-of-range windows ON THE DEVICE and
        # recomputes the launch in fp32 (no synchronisation); the staged / training paths compare max|x| with the
        # limit on the host (one synchronisation per call).  "narrow" / "wide": the caller knows (harness.train,
        # SeriesEvaluator and python -m gdn_amd.main look at the data they hold once): no check, matrix-core /
        # fp32 row-gather kernels respectively.
        self.operand_range = "auto"
        self._consts = None
        self.init_params()

    def init_params(self):
        nn.init.kaiming_uniform_(self.embedding.weight, a=math.sqrt(5))

    # ------------------------------------------------------------------ constants
    def _apply(self, fn, *args, **kwargs):
        # .to(device) / .float() replace the parameter storage: forget cached tensors and constants
        self._key_tensors = None
        self._consts = None
        self._ones = None
        return super()._apply(fn, *args, **kwargs)

    def invalidate_constants(self):
        """Forget the cached sensor graph / folded constants (see `_constants`).  The generation counter is
        part of the cache key, so holders of captured HIP graphs (harness.SeriesEvaluator) see the change too."""
        self._consts = None
        self._generation = getattr(self, "_generation", 0) + 1

    def _param_key(self):
        ps = getattr(self, "_key_tensors", None)
        if ps is None:
            ps = [self.embedding.weight, *self.gnn_layers[0].parameters(), *self.bn_outlayer_in.parameters(),
                  *self.gnn_layers[0].bn.buffers(), *self.bn_outlayer_in.buffers(), *self.out_layer.parameters()]
            self._key_tensors = ps
        inj = None if self.injected_graph is None else (self.injected_graph.data_ptr(), self.injected_graph._version)
        return tuple((p.data_ptr(), p._version) for p in ps) + (self.training, inj, getattr(self, "_generation", 0))

    def _constants(self) -> _EvalConstants:
        """Sensor graph + folded per-forward constants.  In eval they are cached and rebuilt when a
        parameter's (storage, version) changed; in training they are rebuilt every forward — an
        optimizer step changes the embedding, and fused optimizers (`Adam(fused=True)`) and writes
        through `.data` do not bump the version counter.  After such a write in eval mode call
        `invalidate_constants()`."""
        key = self._param_key()
        c = self._consts
        if c is not None and c.key == key and not self.training:
            return c
        gnn = self.gnn_layers[0].gnn
        c = _EvalConstants()
        c.key = key
        emb = self.embedding.weight
        if self.injected_graph is not None:
            c.graph = ops.graph_from_topk(self.injected_graph.to(emb.device))
        else:
            c.graph = ops.topk_graph(emb, self.topk)                       # GDN.py:145-159
        c.terms = ops.node_terms(gnn.lin.weight, gnn.att_i, gnn.att_j, gnn.att_em_i, gnn.att_em_j, emb)
        c.bn1 = c.bn2 = None
        c.fused_args = None
        c.plans = {}                 # bf16_storage -> plan tensor (or None: shape not on the matrix-core path)
        c.guards = {}                # stream -> int32[2] range guard of the eval fast path
        c.limits = {}                # bf16_storage -> the plan's x limit as a host float (read on first use)
        c.mlp = False                # eval-mode OutLayer MLP plan: False = not built yet, None = unsupported
        if not self.training:
            c.bn1 = ops.bn_fold(self.gnn_layers[0].bn)
            c.bn2 = ops.bn_fold(self.bn_outlayer_in)
            if self.out_layer_num == 1:
                # raw argument tuple of gdn_forward_fused (validated once here, not per call)
                lin = self.out_layer.mlp[0]
                ts = (gnn.lin.weight, c.terms, c.graph.nbr, c.graph.deg, gnn.bias, emb, c.bn1, c.bn2,
                      lin.weight, lin.bias)
                assert all(t.is_cuda and t.is_contiguous() for t in ts)
                d, w = gnn.lin.weight.shape
                c.fused_args = (tuple(t.data_ptr() for t in ts), emb.shape[0], w, d, c.graph.k)
        if emb.is_cuda:     # built on this stream: launches on another stream order themselves behind it
            c.stream = torch.cuda.current_stream()
            c.ready = torch.cuda.Event()
            c.ready.record(c.stream)
        else:
            c.stream = c.ready = None
        self._consts = c
        return c

    def _wait_ready(self, c):
        """Launches from another stream order themselves behind the constants / plan, which were built on
        `c.stream` (a capture is preceded by a warm-up + synchronize: nothing to wait for)."""
        cur = torch.cuda.current_stream()
        if c.ready is not None and cur != c.stream and not torch.cuda.is_current_stream_capturing():
            cur.wait_event(c.ready)
        return cur

    def _guard(self, c, stream):
        """The int32[2] range guard of this stream (zero: the gated launch leaves it zeroed)."""
        g = c.guards.get(stream.cuda_stream)
        if g is None and torch.cuda.is_current_stream_capturing():
            # first use inside somebody's capture: no stream switch, no event — the zero fill becomes a node of that
            # graph (every replay starts from a lowered guard, which is what the gated launch leaves anyway)
            g = torch.zeros((2,), dtype=torch.int32, device=self.embedding.weight.device)
            c.guards[stream.cuda_stream] = g
        if g is None:
            with torch.cuda.stream(c.stream) if c.stream is not None else contextlib.nullcontext():
                g = torch.zeros((2,), dtype=torch.int32, device=self.embedding.weight.device)
                if c.ready is not None:
                    c.ready = torch.cuda.Event()
                    c.ready.record(c.stream)
            c.guards[stream.cuda_stream] = g
        return g

    def _launch_fused(self, x, c, out, keys=None, guard: bool = False, wide: bool = False):
        """One ctypes call (two with the range guard); every argument except x / out comes from the constants
        cache.  `keys` = (gt[B, n] fp32, device pointer of the float64 key rows, row pitch): the launch also
        leaves the scoring keys |out - gt| (planned matrix-core path only).  `wide`: the inputs are known to
        exceed the 16-bit operand range — fp32 row-gather kernel.  `guard`: they are not known — the planned
        launch flags out-of-range windows and a gated row-gather launch on the same stream redoes them."""
        if not x.is_cuda:
            raise _lib.GdnHipError(f"input is on {x.device}: gdn_amd needs a HIP device (no CPU fallback)")
        ptrs, n, w, d, k = c.fused_args
        b = x.shape[0]
        if x.shape[1] != n or x.shape[2] != w:
            raise ValueError(f"expected data of shape [B, {n}, {w}], got {tuple(x.shape)}")
        bf16 = x.dtype == torch.bfloat16
        plan = None if (wide and not bf16) else self._plan(c, bf16)
        g = self._guard(c, torch.cuda.current_stream()) if (guard and plan is not None and not bf16) else None
        cur = self._wait_ready(c)
        st = cur.cuda_stream
        if keys is not None:
            if plan is None:
                raise _lib.GdnHipError("scoring keys from the forward launch need the planned matrix-core path")
            gt, key_ptr, key_pitch = keys
            _lib.call("gdn_forward_fused_plan_keys", x.data_ptr(), plan.data_ptr(), ops._chk(gt, name="gt").data_ptr(),
                      key_ptr, key_pitch, b, n, w, d, k, int(bf16), out.data_ptr(), st)
        elif plan is not None:
            _lib.call("gdn_forward_fused_plan", x.data_ptr(), plan.data_ptr(), b, n, w, d, k, int(bf16),
                      out.data_ptr(), None if g is None else g.data_ptr(), st)
            if g is not None:       # no-op unless the launch above met a value outside the operand range
                _lib.call("gdn_forward_fused_gated", g.data_ptr(), x.data_ptr(), *ptrs, b, n, w, d, k, out.data_ptr(), st)
        elif wide and not bf16:
            _lib.call("gdn_forward_fused_gated", None, x.data_ptr(), *ptrs, b, n, w, d, k, out.data_ptr(), st)
        else:
            _lib.call("gdn_forward_fused_bf16" if bf16 else "gdn_forward_fused", x.data_ptr(), *ptrs, b, n, w, d, k,
                      out.data_ptr(), st)
        return out

    def operand_limit(self, bf16: bool = False) -> float:
        """Largest |x| the eval fast path's matrix-core kernel represents with the current parameters (inf: no
        limit — bf16 storage, or a shape that runs on the fp32 row-gather kernels anyway).  Reads one float of
        the plan: a host synchronisation, cached until the parameters change."""
        c = self._constants()
        if bf16 not in c.limits:
            plan = self._plan(c, bf16) if c.fused_args is not None else None
            if plan is None:
                c.limits[bf16] = float("inf")
            else:
                _, n, w, d, k = c.fused_args
                c.limits[bf16] = float(ops.fused_plan_limit(plan, n, w, d, k, bf16))
        return c.limits[bf16]

    def input_exceeds_limit(self, data: torch.Tensor, margin: float = 1.0) -> bool:
        """True when `data` (windows or a raw series, already on the device) holds a value the matrix-core
        kernels cannot represent (|x| >= the limit, or NaN): callers that keep their data resident ask ONCE and
        pass `wide=` afterwards.  Three small reductions + ONE synchronisation.  The limit covers every kernel
        family: the fused kernel's plan limit (eval, out_layer_num == 1), and for the staged / training kernels
        |x| < 60000 and |xlin| <= |x| * max_c sum_w |lin[c, w]| < 60000.  `margin` > 1 tightens the limit by that
        factor (training: the weights the limit was computed from move)."""
        if data.dtype == torch.bfloat16:
            return False
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("operand_range='auto' needs a host check: set model.operand_range to 'narrow' or "
                               "'wide' before capturing (harness.GraphedTrainStep does)")
        l1 = self.gnn_layers[0].gnn.lin.weight.detach().abs().sum(dim=1).amax().clamp_min(1e-30)
        limit = torch.clamp(60000.0 / l1, max=60000.0)
        if not self.training and self.out_layer_num == 1:
            c = self._constants()
            plan = self._plan(c, False)
            if plan is not None:
                _, n, w, d, k = c.fused_args
                limit = torch.minimum(limit, ops.fused_plan_limit(plan, n, w, d, k, False))
        return not bool(data.detach().abs().amax() * margin < limit)            # (NaN compares false: wide)

    def _wide_for(self, x) -> bool:
        if self.operand_range == "wide":
            return x.dtype != torch.bfloat16
        if self.operand_range == "narrow":
            return False
        return self.input_exceeds_limit(x)

    def _plan(self, c, bf16: bool):
        """The fused kernel's precomputed per-launch constants for these parameters (built on first use,
        dropped with the constants cache)."""
        if bf16 not in c.plans:
            gnn = self.gnn_layers[0].gnn
            lin = self.out_layer.mlp[0]
            with torch.cuda.stream(c.stream) if c.stream is not None else contextlib.nullcontext():
                c.plans[bf16] = ops.fused_plan(gnn.lin.weight, c.terms, c.graph, gnn.bias, self.embedding.weight,
                                               c.bn1, c.bn2, lin.weight, lin.bias, bf16_storage=bf16)
                if c.ready is not None:
                    c.ready = torch.cuda.Event()
                    c.ready.record(c.stream)
        return c.plans[bf16]

    # ------------------------------------------------------------------ forward
    def forward(self, data, org_edge_index=None):
        x = data.detach()                                                   # GDN.py:124
        # bfloat16 windows select bf16 STORAGE of x / xlin / z (eval only; fp32 arithmetic, fp32 output)
        bf16 = x.dtype == torch.bfloat16 and not self.training
        if x.dtype != torch.float32 and not bf16:
            x = x.float()
        x = x.contiguous()
        batch, node_num, _ = x.shape
        layer = self.gnn_layers[0]
        gnn = layer.gnn
        c = self._constants()
        self.learned_graph = c.graph.topk                                   # GDN.py:159
        if batch == 0 and not self.training:
            # an empty minibatch: the reference's ops run on empty tensors and return [0, N]
            return torch.empty((0, node_num), dtype=torch.float32, device=x.device)
        emb = self.embedding.weight

        if not self.training:
            if self.out_layer_num == 1:
                out = torch.empty((batch, node_num), dtype=torch.float32, device=x.device)
                mode = self.operand_range
                self._launch_fused(x, c, out, guard=mode == "auto", wide=mode == "wide")
                layer._set_dense(lambda: self._dense_attention(x, c, batch))
                return out
            wide = self._wide_for(x)
            xlin, s_i, s_j = ops.project_fwd(x, gnn.lin.weight, c.terms, wide=wide)
            # (alpha only on request — att_weight_1 — like the fused path: the launch then reads the bank-ordered lists)
            z, _ = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, batch, want_alpha=False, wide=wide)
            layer._set_dense(lambda: self._dense_attention(x, c, batch))
            # out_layer_num > 1: the head kernel hands its [BN,d] activation to the OutLayer MLP
            # (plain library GEMMs through torch); its own Linear(d->1) result is unused here
            zero_w = torch.zeros((emb.shape[1],), device=x.device)
            zero_b = torch.zeros((1,), device=x.device)
            _, h2 = ops.head_fwd(z, emb, c.bn1, c.bn2, zero_w, zero_b, batch, want_h2=True)
            if c.mlp is False:
                c.mlp = ops.mlp_plan(self.out_layer, emb.shape[1])
            if c.mlp is not None:                                           # GDN.py:183 on the matrix cores
                return ops.mlp_fwd(h2, c.mlp).view(batch, node_num)
            if ops.mlp_eval_wide_supported(self.out_layer, emb.shape[1]):   # hidden 257 .. 512: fp32 matrix-core GEMMs
                return ops.mlp_eval_wide(h2, self.out_layer).view(batch, node_num)
            with torch.no_grad():       # hidden > 512: library GEMMs through torch
                out = self.out_layer(h2.view(batch, node_num, -1))
            return out.view(-1, node_num)

        # ---- training: HIP forward/backward for the graph layer and (out_layer_num == 1) the head
        z, alpha = _GraphAttentionFn.apply(x, gnn.lin.weight, gnn.att_i, gnn.att_j, gnn.att_em_i,
                                           gnn.att_em_j, emb, gnn.bias, c.graph, batch, c.terms, self._wide_for(x))
        layer._set_dense((alpha, c.graph, batch))
        if self.out_layer_num == 1 and self._hip_train_head_ok():
            lin = self.out_layer.mlp[0]
            mask, mask_scale = self._dropout_mask(batch, node_num, emb.shape[1], x.device)
            return _HeadTrainFn.apply(z, emb, layer.bn.weight, layer.bn.bias, self.bn_outlayer_in.weight,
                                      self.bn_outlayer_in.bias, lin.weight, lin.bias, mask, mask_scale, layer.bn,
                                      self.bn_outlayer_in, batch)
        if (self.out_layer_num > 1 and self._hip_train_head_ok() and
                ops.mlp_train_supported(self.out_layer, emb.shape[1], batch * node_num)):
            hidden, last = ops.mlp_train_layers(self.out_layer)
            mlp_params = [t for lin, bn in hidden for t in (lin.weight, lin.bias, bn.weight, bn.bias)]
            mlp_params += [last.weight, last.bias]
            mask, mask_scale = self._dropout_mask(batch, node_num, emb.shape[1], x.device)
            out = _MlpHeadTrainFn.apply(z, emb, layer.bn.weight, layer.bn.bias, self.bn_outlayer_in.weight,
                                        self.bn_outlayer_in.bias, mask, mask_scale, layer.bn, self.bn_outlayer_in,
                                        self.out_layer, batch, *mlp_params)
            return out.view(batch, node_num)
        # MLP head outside gdn_mlp_train_fwd's shapes (hidden > 512, or hidden layers of different widths): torch for the BN
        # statistics, dropout and the library GEMMs
        h = layer.relu(layer.bn(z))                                         # GDN.py:77-79
        h = h.view(batch, node_num, -1)                                     # GDN.py:171-172
        h = torch.mul(h, emb)                                               # GDN.py:175-176
        h = F.relu(self.bn_outlayer_in(h.permute(0, 2, 1))).permute(0, 2, 1)   # GDN.py:178-180
        h = self.dp(h)                                                      # GDN.py:182
        return self.out_layer(h).view(-1, node_num)                         # GDN.py:183-184

    def _hip_train_head_ok(self):
        bns = (self.gnn_layers[0].bn, self.bn_outlayer_in)
        return all(b.affine and (not b.track_running_stats or b.momentum is not None) for b in bns)

    def _dropout_mask(self, batch, node_num, d, device):
        """(mask, scale) of models/GDN.py:182, or (None, 1).  A plain `nn.Dropout` is drawn as ONE byte
        per element from torch's generator (`bernoulli_`; multiplier = keep * 1/(1-p)) — a quarter of the
        bytes of an fp32 mask, which four head passes re-read; any other `dp` module (tests install
        fixed masks) is asked for its fp32 multipliers by feeding it ones."""
        dp = self.dp
        if type(dp) is nn.Dropout:
            if not dp.training or dp.p == 0:
                return None, 1.0
            if dp.p >= 1:
                return torch.zeros((batch * node_num, d), dtype=torch.uint8, device=device), 0.0
            keep = torch.empty((batch * node_num, d), dtype=torch.uint8, device=device).bernoulli_(1.0 - dp.p)
            return keep, 1.0 / (1.0 - dp.p)
        if not dp.training:
            return None, 1.0
        ones = getattr(self, "_ones", None)
        if ones is None or ones.shape != (batch, node_num, d) or ones.device != device or getattr(dp, "inplace", False):
            ones = torch.ones((batch, node_num, d), dtype=torch.float32, device=device)
            self._ones = None if getattr(dp, "inplace", False) else ones
        return dp(ones).reshape(batch * node_num, d), 1.0

    def forward_into(self, data, out, keys=None, wide: bool = False):
        """Eval fast path writing into a caller-owned [B, N] slice (no allocation, HIP-graph
        capturable once `_constants()` is warm): used by harness.SeriesEvaluator.  `keys`: see _launch_fused.
        The caller vouches for the range of `data` (`input_exceeds_limit`, asked once per resident tensor):
        `wide=True` runs the fp32 row-gather kernel, False the matrix-core one WITHOUT the range guard."""
        if self.training or self.out_layer_num != 1:
            raise RuntimeError("forward_into is the eval / out_layer_num == 1 fast path")
        c = self._constants()
        self.learned_graph = c.graph.topk
        if data.dtype not in (torch.float32, torch.bfloat16):
            data = data.float()
        return self._launch_fused(data.contiguous(), c, out, keys, wide=wide)

    def fused_keys_supported(self, bf16: bool = False) -> bool:
        """True when the eval forward of this model can leave the scoring keys itself (`keys=` of forward_into /
        forward_series): out_layer_num == 1 on the planned matrix-core path."""
        return (not self.training) and self.out_layer_num == 1 and self._plan(self._constants(), bf16) is not None

    def forward_series(self, series, first: int, batch: int, out=None, keys=None, wide: bool = False):
        """Eval forward of `batch` consecutive stride-1 windows taken directly from the raw series
        [node_num, T] (the layout `TimeDataset` slices, datasets/TimeDataset.py:42-49): window b is
        series[:, first+b : first+b+W] and predicts column first+b+W.  No [T, N, W] tensor exists."""
        if self.training or self.out_layer_num != 1:
            raise RuntimeError("forward_series is the eval / out_layer_num == 1 fast path")
        c = self._constants()
        gnn = self.gnn_layers[0].gnn
        lin = self.out_layer.mlp[0]
        self.learned_graph = c.graph.topk
        plan = None if wide else self._plan(c, False)
        st = self._wait_ready(c).cuda_stream
        series = ops._chk(series, name="series")
        n, t_len = series.shape
        d, w = gnn.lin.weight.shape
        if out is None:
            out = torch.empty((batch, n), dtype=torch.float32, device=series.device)
        if plan is not None:
            if keys is not None:
                gt, key_ptr, key_pitch = keys
                _lib.call("gdn_forward_fused_series_plan_keys", series.data_ptr(), t_len, first, plan.data_ptr(),
                          ops._chk(gt, name="gt").data_ptr(), key_ptr, key_pitch, batch, n, w, d, c.graph.k,
                          out.data_ptr(), st)
                return out
            _lib.call("gdn_forward_fused_series_plan", series.data_ptr(), t_len, first, plan.data_ptr(), batch, n, w, d,
                      c.graph.k, out.data_ptr(), None, st)
            return out
        if keys is not None:
            raise _lib.GdnHipError("scoring keys from the forward launch need the planned matrix-core path")
        if wide:     # the fp32 row-gather kernel on the raw series (inputs beyond the 16-bit operand range)
            ptrs = c.fused_args[0]
            _lib.call("gdn_forward_fused_series_gated", None, series.data_ptr(), t_len, first, *ptrs, batch, n, w, d,
                      c.graph.k, out.data_ptr(), st)
            return out
        return ops.forward_fused_series(series, first, batch, w, gnn.lin.weight, c.terms,
                                        c.graph, gnn.bias, self.embedding.weight, c.bn1, c.bn2, lin.weight,
                                        lin.bias, out=out)

    def _dense_attention(self, x, c, batch):
        gnn = self.gnn_layers[0].gnn
        wide = self._wide_for(x)
        xlin, s_i, s_j = ops.project_fwd(x, gnn.lin.weight, c.terms, wide=wide)
        _, alpha = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, batch, want_alpha=True, wide=wide)
        return alpha, c.graph, batch
