"""K8 (and the staged / fused forward) at B=262144 windows: correctness against 512-window launches of
the same inputs (64-bit addressing: xlin is 8.5 GB) and the launch duration."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_forward_parity import random_params
from gdn_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
model = random_params(127, 15, 30, 64, seed=0).to(dev).eval()
gnn = model.gnn_layers[0].gnn
c = model._constants()
x = torch.rand((B, 127, 15), device=dev)
xlin, s_i, s_j = ops.project_fwd(x, gnn.lin.weight, c.terms)
z, _ = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, B, want_alpha=False)
lin = model.out_layer.mlp[0]
out, _ = ops.head_fwd(z, model.embedding.weight, c.bn1, c.bn2, lin.weight, lin.bias, B)
with torch.no_grad():
    fused = model(x, None)
torch.cuda.synchronize()
bad = 0
for s in (0, 512 * 100, B - 512, B // 2 + 37):
    with torch.no_grad():
        ref = model(x[s:s + 512].contiguous(), None)
    xl2, si2, sj2 = ops.project_fwd(x[s:s + 512].contiguous(), gnn.lin.weight, c.terms)
    z2, _ = ops.attn_aggregate_fwd(xl2, si2, sj2, c.graph, gnn.bias, 512, want_alpha=False)
    ok = torch.equal(ref, fused[s:s + 512]) and torch.equal(z2, z[s * 127:(s + 512) * 127]) and \
        torch.allclose(out[s:s + 512], ref, atol=2e-6, rtol=0)
    bad += 0 if ok else 1
    print(f"slice {s}: fused equal {torch.equal(ref, fused[s:s + 512])}, K8 equal {torch.equal(z2, z[s * 127:(s + 512) * 127])}, "
          f"staged-vs-fused max diff {(out[s:s + 512] - ref).abs().max().item():.2e}")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(8): ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, B, want_alpha=False)
a.record()
for _ in range(8): ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, B, want_alpha=False)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / 8 * 1e3
alg = 2 * B * 127 * 64 * 4
print(f"B={B}: K8 {us:.1f} us/launch = {alg / us / 1e3:.1f} GB/s ({alg / us / 1e3 / 8000:.3f} of 8 TB/s); mismatching slices: {bad}")
