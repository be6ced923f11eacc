"""Backward kernels alone, chained launches between two events: python3 tools/probe_bwd.py [B ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_forward_parity import random_params
from gdn_amd import ops
dev = torch.device("cuda:0")
N, W, K, D = 127, 15, 30, 64
model = random_params(N, W, K, D, seed=0).to(dev).train()
gnn = model.gnn_layers[0].gnn
c = model._constants()
c.graph.reverse()


def timed(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for B in [int(v) for v in sys.argv[1:]] or [512, 4096]:
    x = torch.rand((B, N, W), device=dev)
    xlin, s_i, s_j = ops.project_fwd(x, gnn.lin.weight, c.terms)
    z, alpha = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, B, want_alpha=True)
    d_z = torch.randn_like(z)
    t_f = timed(lambda: ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, B, want_alpha=True))
    t_b = timed(lambda: ops.attn_aggregate_bwd(d_z, xlin, alpha, s_i, s_j, c.graph, B))
    d_xlin, d_si, d_sj, _ = ops.attn_aggregate_bwd(d_z, xlin, alpha, s_i, s_j, c.graph, B)
    t_p = timed(lambda: ops.project_bwd(x, d_xlin, d_si, d_sj, D))
    print(f"B={B:6d}: attn fwd(+alpha) {t_f:8.1f} us   attn bwd {t_b:8.1f} us   project bwd {t_p:8.1f} us")
