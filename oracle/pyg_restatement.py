"""TEST INFRASTRUCTURE — not product code.  Nothing under gdn_amd/ may import this.

Restatement of the six torch-geometric **1.5.0** symbols the reference's hot
path calls.  torch-geometric (pinned `torch-geometric==1.5.0`, reference
install.sh:5 / README.md:10-11) is a third-party dependency that is NOT
vendored in /root/reference and NOT installed in this image, and there is no
network to fetch it.  The semantics below are therefore restated from the
library's published 1.5.0 algorithm; no reference-authored test pins them.

    **PARITY UNPINNED at the PyG boundary** — every golden vector under
    tests/golden/ is "reference model files + this restatement", not
    "reference model files + real torch-geometric".

Reference call sites (file:line relative to /root/reference):
  MessagePassing.__init__(aggr='add')     models/graph_layer.py:14
  MessagePassing.propagate(...)           models/graph_layer.py:65
  self.node_dim                           models/graph_layer.py:63
  remove_self_loops(edge_index)           models/graph_layer.py:61
  add_self_loops(edge_index, num_nodes=)  models/graph_layer.py:62-63
  softmax(alpha, edge_index_i, size_i)    models/graph_layer.py:110
  glorot / zeros                          models/graph_layer.py:42-49
  GCNConv, GATConv, EdgeConv (imported, never used)  models/GDN.py:8
"""
import inspect
import math

import torch


# --------------------------------------------------------------------- utils
def remove_self_loops(edge_index, edge_attr=None):
    """Drop every column whose source equals its target; order of the rest kept."""
    keep = edge_index[0] != edge_index[1]
    if edge_attr is not None:
        edge_attr = edge_attr[keep]
    return edge_index[:, keep], edge_attr


def add_self_loops(edge_index, edge_weight=None, fill_value=1, num_nodes=None):
    """Append one (n, n) column per node, n = 0..num_nodes-1, AFTER the existing columns."""
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1 if edge_index.numel() else 0
    loops = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    loops = loops.unsqueeze(0).expand(2, -1)
    if edge_weight is not None:
        extra = edge_weight.new_full((num_nodes,), fill_value)
        edge_weight = torch.cat([edge_weight, extra], dim=0)
    return torch.cat([edge_index, loops], dim=1), edge_weight


def _segment_reduce(src, index, num_segments, op):
    shape = (num_segments,) + tuple(src.shape[1:])
    idx = index.view((-1,) + (1,) * (src.dim() - 1)).expand_as(src)
    if op == "sum":
        return src.new_zeros(shape).scatter_add_(0, idx, src)
    if op == "max":
        out = src.new_full(shape, float("-inf"))
        return out.scatter_reduce_(0, idx, src, reduce="amax", include_self=True)
    raise ValueError(op)


def softmax(src, index, num_nodes=None):
    """Softmax over the entries of `src` that share a value of `index` (PyG 1.x,
    third positional argument is the NODE COUNT, not `ptr` as in PyG 2.x):
    subtract the per-group max, exponentiate, divide by (group sum + 1e-16)."""
    if num_nodes is None:
        num_nodes = int(index.max()) + 1
    shifted = src - _segment_reduce(src, index, num_nodes, "max")[index]
    e = shifted.exp()
    return e / (_segment_reduce(e, index, num_nodes, "sum")[index] + 1e-16)


# --------------------------------------------------------------------- inits
def glorot(tensor):
    if tensor is not None:
        bound = math.sqrt(6.0 / (tensor.size(-2) + tensor.size(-1)))
        tensor.data.uniform_(-bound, bound)


def zeros(tensor):
    if tensor is not None:
        tensor.data.fill_(0)


# ------------------------------------------------------------ MessagePassing
class MessagePassing(torch.nn.Module):
    """Gather / message / scatter-add skeleton, flow source→target:
    `foo_j` = foo[edge_index[0]] (sources), `foo_i` = foo[edge_index[1]] (targets),
    `edge_index_i` = edge_index[1], `size_i` = number of target nodes; every other
    keyword is handed to `message` untouched; messages are summed onto their target
    along `node_dim`; `update` is the identity."""

    def __init__(self, aggr="add", flow="source_to_target", node_dim=0):
        super().__init__()
        assert aggr == "add" and flow == "source_to_target"
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim
        self._msg_args = [n for n in inspect.signature(self.message).parameters]

    def propagate(self, edge_index, size=None, **kwargs):
        side = {"_i": 1, "_j": 0}
        sizes = [None, None]
        call = {}
        for name in self._msg_args:
            suffix = name[-2:]
            if suffix not in side:
                call[name] = kwargs.get(name)
                continue
            which = side[suffix]
            data = kwargs.get(name[:-2])
            if isinstance(data, (tuple, list)):
                sizes[1 - which] = data[1 - which].size(self.node_dim)
                data = data[which]
            if torch.is_tensor(data):
                sizes[which] = data.size(self.node_dim)
                data = data.index_select(self.node_dim, edge_index[which])
            call[name] = data
        sizes[0] = sizes[1] if sizes[0] is None else sizes[0]
        sizes[1] = sizes[0] if sizes[1] is None else sizes[1]
        if "edge_index_i" in self._msg_args:
            call["edge_index_i"] = edge_index[1]
        if "edge_index_j" in self._msg_args:
            call["edge_index_j"] = edge_index[0]
        if "size_i" in self._msg_args:
            call["size_i"] = sizes[1]
        if "size_j" in self._msg_args:
            call["size_j"] = sizes[0]
        msg = self.message(**call)
        return _segment_reduce(msg, edge_index[1], sizes[1], "sum")

    def message(self, x_j):  # pragma: no cover - always overridden
        return x_j


class _NeverUsed(torch.nn.Module):
    """GCNConv / GATConv / EdgeConv are imported at models/GDN.py:8 and never referenced."""

    def __init__(self, *a, **k):
        raise RuntimeError("not part of the GDN hot path")


def install_as_torch_geometric():
    """Bind this restatement under the module names the reference imports.
    Used ONLY by tests/golden/make_golden.py inside the build container."""
    import sys
    import types

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    inits = mod("torch_geometric.nn.inits", glorot=glorot, zeros=zeros)
    conv = mod("torch_geometric.nn.conv", MessagePassing=MessagePassing)
    utils = mod("torch_geometric.utils", remove_self_loops=remove_self_loops,
                add_self_loops=add_self_loops, softmax=softmax)
    nn = mod("torch_geometric.nn", GCNConv=_NeverUsed, GATConv=_NeverUsed, EdgeConv=_NeverUsed,
             conv=conv, inits=inits)
    mod("torch_geometric", nn=nn, utils=utils)
