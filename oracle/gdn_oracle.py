"""TEST INFRASTRUCTURE — not product code.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import this; nothing under gdn_amd/ does.

CPU restatement (torch-CPU, fp32) of the GDN forward hot path, OP-FAITHFUL to the
reference: it materialises the same per-edge tensors in the same order, so that
(a) it is the checker for the HIP kernels and (b) timing it is a fair stand-in for
"the reference's CPU path" (`cpu_baseline.kind = "port"`).

Pinning: `tests/test_oracle_golden.py` checks every function here against
tests/golden/*.npz, which were produced by running the reference's own
models/GDN.py + models/graph_layer.py (loaded by path, unmodified) on top of
oracle/pyg_restatement.py — see tests/golden/make_golden.py.
**PARITY UNPINNED at the torch-geometric 1.5.0 boundary** (library absent from the
image and from /root/reference; see oracle/pyg_restatement.py header).

All `file:line` citations are relative to /root/reference.
State is passed as a plain dict with the reference's state_dict key names
(SURVEY.md §8b), so a reference checkpoint can be fed in directly.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

NEG_SLOPE = 0.2      # models/graph_layer.py:13  (LeakyReLU slope)
SOFTMAX_EPS = 1e-16  # torch_geometric.utils.softmax 1.5.0 (restated)
BN_EPS = 1e-5        # nn.BatchNorm1d default, models/GDN.py:67,96
BN_MOMENTUM = 0.1
DROPOUT_P = 0.2      # models/GDN.py:114


# ------------------------------------------------------------------ graph build
def cosine_matrix(emb: torch.Tensor) -> torch.Tensor:
    """models/GDN.py:148-152 — V Vᵀ divided by the outer product of row norms (no ε)."""
    w = emb.detach().clone().view(emb.shape[0], -1)
    dots = torch.matmul(w, w.T)
    nrm = w.norm(dim=-1)
    return dots / torch.matmul(nrm.view(-1, 1), nrm.view(1, -1))


def learned_graph(emb: torch.Tensor, topk: int) -> torch.Tensor:
    """models/GDN.py:157-159 — per-row top-k column indices, descending cosine, self included."""
    return torch.topk(cosine_matrix(emb), topk, dim=-1)[1]


def batched_edge_index(graph: torch.Tensor, batch: int) -> torch.Tensor:
    """models/GDN.py:161-165 + get_batch_edge_index :15-24.
    Row 0 = source j (the top-k entry), row 1 = target i (the row it came from);
    window b's copy is shifted by b*N."""
    n, k = graph.shape
    tgt = torch.arange(n).unsqueeze(1).repeat(1, k).flatten()
    src = graph.flatten()
    one = torch.stack((src, tgt), dim=0)                       # [2, N*K]
    shift = (torch.arange(batch) * n).repeat_interleave(n * k)  # same result as the :21-22 loop
    return (one.repeat(1, batch) + shift.unsqueeze(0)).long()


def strip_and_append_self_loops(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """models/graph_layer.py:61-63 via PyG remove_self_loops / add_self_loops."""
    keep = edge_index[0] != edge_index[1]
    loops = torch.arange(num_nodes, dtype=edge_index.dtype).unsqueeze(0).repeat(2, 1)
    return torch.cat((edge_index[:, keep], loops), dim=1)


# ------------------------------------------------------------------ graph layer
def round_bf16(t: torch.Tensor) -> torch.Tensor:
    """Round to the nearest bfloat16 (ties to even) and return in the input dtype: what storing a
    tensor in bf16 does to its values."""
    return t.to(torch.bfloat16).to(t.dtype)


def graph_layer(p: dict, prefix: str, x: torch.Tensor, edge_index: torch.Tensor,
                emb_rep: torch.Tensor, storage: str = "fp32"):
    """models/graph_layer.py:53-117 with heads=1, concat=False, dropout=0
    (hard-wired by models/GDN.py:65).  Returns (out, edge_index', alpha, xlin).

    storage="bf16" (BASELINE configs[2]/[4]: bf16 storage, fp32 logits / softmax / accumulate): the
    projected features are ROUNDED TO bf16 WHERE THEY ARE STORED, i.e. as the rows the messages gather
    (x_j * alpha, :117); the attention logits (:94-104) are computed in fp32 from the unrounded
    projection, as the HIP kernels compute them from x directly.  Returned xlin = the stored one."""
    lin_w = p[prefix + "lin.weight"]                      # [D, W]
    d = lin_w.shape[0]
    xlin = F.linear(x, lin_w)                             # :56
    xlin_msg = round_bf16(xlin) if storage == "bf16" else xlin
    ei = strip_and_append_self_loops(edge_index, xlin.shape[0])  # :61-63
    src, tgt = ei[0], ei[1]
    # PyG propagate: _j <- edge_index[0], _i <- edge_index[1]
    x_i = xlin.index_select(0, tgt).view(-1, 1, d)        # :87
    x_j = xlin.index_select(0, src).view(-1, 1, d)        # :88
    emb_i = emb_rep[tgt].unsqueeze(1)                     # :91-92 (repeat over heads=1)
    emb_j = emb_rep[src].unsqueeze(1)                     # :91,93
    key_i = torch.cat((x_i, emb_i), dim=-1)               # :95
    key_j = torch.cat((x_j, emb_j), dim=-1)               # :96
    cat_i = torch.cat((p[prefix + "att_i"], p[prefix + "att_em_i"]), dim=-1)  # :100
    cat_j = torch.cat((p[prefix + "att_j"], p[prefix + "att_em_j"]), dim=-1)  # :101
    logit = (key_i * cat_i).sum(-1) + (key_j * cat_j).sum(-1)                # :103
    logit = F.leaky_relu(logit.view(-1, 1, 1), NEG_SLOPE)                    # :106-109
    # PyG utils.softmax(alpha, edge_index_i, size_i)                          # :110
    n = xlin.shape[0]
    idx = tgt.view(-1, 1, 1)
    gmax = torch.full((n, 1, 1), float("-inf"), dtype=logit.dtype).scatter_reduce_(
        0, idx, logit, reduce="amax", include_self=True)
    e = (logit - gmax[tgt]).exp()
    gsum = torch.zeros((n, 1, 1), dtype=logit.dtype).scatter_add_(0, idx, e)
    alpha = e / (gsum[tgt] + SOFTMAX_EPS)
    if storage == "bf16":
        x_j = xlin_msg.index_select(0, src).view(-1, 1, d)
    msg = x_j * alpha.view(-1, 1, 1)                      # :117 (dropout p=0 at :115 is identity)
    agg = torch.zeros((n, 1, d), dtype=msg.dtype).scatter_add_(
        0, tgt.view(-1, 1, 1).expand_as(msg), msg)        # PyG aggregate, aggr='add'
    out = agg.mean(dim=1) + p[prefix + "bias"]            # :71-74
    return out, ei, alpha, xlin_msg


# ------------------------------------------------------------------ batch norm
def batch_norm(p: dict, prefix: str, x: torch.Tensor, training: bool, new_stats: dict | None):
    """nn.BatchNorm1d over dim 1 of [rows, C] or [B, C, L] (models/GDN.py:77,179; :49-52)."""
    rm, rv = p[prefix + "running_mean"], p[prefix + "running_var"]
    if training:
        rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, p[prefix + "weight"], p[prefix + "bias"],
                     training, BN_MOMENTUM, BN_EPS)
    if training and new_stats is not None:
        new_stats[prefix + "running_mean"] = rm
        new_stats[prefix + "running_var"] = rv
    return y


def out_layer(p: dict, x: torch.Tensor, layer_num: int, training: bool, new_stats):
    """models/GDN.py:27-56 — `layer_num-1` × (Linear, BN over the feature axis, ReLU) then Linear(→1)."""
    h = x
    for l in range(layer_num):
        key = f"out_layer.mlp.{3 * l}."
        h = F.linear(h, p[key + "weight"], p[key + "bias"])
        if l != layer_num - 1:
            bn = f"out_layer.mlp.{3 * l + 1}."
            h = batch_norm(p, bn, h.permute(0, 2, 1), training, new_stats).permute(0, 2, 1)
            h = F.relu(h)
    return h


# ------------------------------------------------------------------ full forward
def forward(p: dict, data: torch.Tensor, topk: int, out_layer_num: int = 1, *,
            training: bool = False, dropout_mask: torch.Tensor | None = None,
            graph: torch.Tensor | None = None, storage: str = "fp32", round_agg: bool = False) -> dict:
    """models/GDN.py:122-187.  `graph` overrides the top-k result (kernel-level parity
    with an injected neighbour list); `dropout_mask` ([B,N,D], already scaled by 1/(1-p))
    replaces nn.Dropout's RNG when training.  Returns every intermediate the tests pin.
    `storage="bf16"`: the input windows and the projected features are stored in bf16 (see
    graph_layer); `round_agg` additionally rounds the aggregate z to bf16 (the staged pipeline stores it,
    the fused kernel keeps it on chip in fp32).  Everything else stays fp32."""
    x = data.clone().detach()                               # :124
    if storage == "bf16":
        x = round_bf16(x)
    b, n, w = x.shape
    x = x.view(-1, w).contiguous()                          # :130
    emb = p["embedding.weight"]                             # :143 (arange lookup = the table)
    g = learned_graph(emb, topk) if graph is None else graph   # :145-159 (detached)
    emb_rep = emb.repeat(b, 1)                              # :146
    ei = batched_edge_index(g, b)                           # :161-165
    new_stats: dict = {}
    pre = "gnn_layers.0.gnn."
    agg, ei1, alpha, xlin = graph_layer(p, pre, x, ei, emb_rep, storage)  # :73
    if round_agg:
        agg = round_bf16(agg)
    h = F.relu(batch_norm(p, "gnn_layers.0.bn.", agg, training, new_stats))   # :77-79
    h = h.view(b, n, -1)                                    # :171-172
    h = h * emb                                             # :175-176
    h = batch_norm(p, "bn_outlayer_in.", h.permute(0, 2, 1), training, new_stats)  # :178-179
    h = F.relu(h).permute(0, 2, 1)                          # :179-180
    if training:                                            # :182
        if dropout_mask is not None:
            h = h * dropout_mask
        else:
            h = F.dropout(h, DROPOUT_P, True)
    out = out_layer(p, h, out_layer_num, training, new_stats).view(-1, n)   # :183-184
    return {"out": out, "learned_graph": g, "edge_index_1": ei1, "att_weight_1": alpha,
            "xlin": xlin, "agg": agg, "new_stats": new_stats}


# ------------------------------------------------------------------ readable spec
def forward_separable(p: dict, data: torch.Tensor, graph: torch.Tensor, out_layer_num: int = 1):
    """Same eval-mode arithmetic written the way the HIP kernels compute it: per-node
    attention scalars (the logit at models/graph_layer.py:103 is a sum of a target-only and
    a source-only term), dense [N, K+1] neighbour lists shared by every window, no edge
    tensors.  Checked against `forward` in tests/test_oracle_golden.py."""
    b, n, w = data.shape
    pre = "gnn_layers.0.gnn."
    emb = p["embedding.weight"]
    xlin = data.reshape(-1, w) @ p[pre + "lin.weight"].T
    d = xlin.shape[1]
    xl = xlin.view(b, n, d)
    s_i = xl @ p[pre + "att_i"].view(d) + emb @ p[pre + "att_em_i"].view(d)     # [B,N]
    s_j = xl @ p[pre + "att_j"].view(d) + emb @ p[pre + "att_em_j"].view(d)
    agg = torch.zeros_like(xl)
    for i in range(n):
        srcs = [int(j) for j in graph[i] if int(j) != i] + [i]   # strip self, append self
        srcs = torch.tensor(srcs)
        e = F.leaky_relu(s_i[:, i:i + 1] + s_j[:, srcs], NEG_SLOPE)      # [B, deg]
        e = (e - e.max(dim=1, keepdim=True).values).exp()
        a = e / (e.sum(dim=1, keepdim=True) + SOFTMAX_EPS)
        agg[:, i] = (a.unsqueeze(-1) * xl[:, srcs]).sum(dim=1)
    agg = agg + p[pre + "bias"]

    def affine(prefix):
        scale = p[prefix + "weight"] / torch.sqrt(p[prefix + "running_var"] + BN_EPS)
        return scale, p[prefix + "bias"] - p[prefix + "running_mean"] * scale

    s1, t1 = affine("gnn_layers.0.bn.")
    s2, t2 = affine("bn_outlayer_in.")
    h = F.relu(agg * s1 + t1) * emb
    h = F.relu(h * s2 + t2)
    return {"out": out_layer(p, h, out_layer_num, False, None).view(-1, n),
            "agg": agg.view(-1, d), "xlin": xlin}
