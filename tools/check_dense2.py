"""Calibration run for the dense staged kernels and the bf16-storage variants (prints error statistics)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gdn_amd import ops  # noqa: E402
from oracle import gdn_oracle  # noqa: E402
from test_gpu_forward_parity import random_params  # noqa: E402

dev = torch.device("cuda:0")


def stats(name, got, want):
    err = (got.double().cpu() - want.double()).abs()
    print(f"   {name}: max {err.max():.3e} mean {err.mean():.3e} p99.9 {err.flatten().kthvalue(max(1, int(err.numel() * 0.999))).values:.3e}"
          f"  |ref|max {want.abs().max():.3f}", flush=True)


for (n, w, k, d, b) in [(127, 15, 30, 64, 16), (27, 5, 5, 64, 64), (64, 15, 63, 64, 8), (100, 30, 40, 64, 5), (127, 15, 30, 64, 512)]:
    print(f"n={n} w={w} k={k} b={b}")
    model = random_params(n, w, k, d, seed=5)
    p = {key: v.detach().clone() for key, v in model.state_dict().items()}
    model = model.to(dev).eval()
    x = torch.rand((b, n, w), generator=torch.Generator().manual_seed(1))
    c = model._constants()
    gnn, lin = model.gnn_layers[0].gnn, model.out_layer.mlp[0]
    graph = c.graph.topk.cpu()
    f64 = torch.float64
    p64 = {key: (v.to(f64) if v.is_floating_point() else v) for key, v in p.items()}
    # fp32 staged dense
    ref = gdn_oracle.forward(p64, x.to(f64), k, graph=graph)
    xlin, s_i, s_j = ops.project_fwd(x.to(dev), gnn.lin.weight, c.terms)
    stats("fp32 project xlin", xlin, ref["xlin"])
    z, alpha = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, b, want_alpha=True)
    stats("fp32 K8 z", z, ref["agg"])
    z2, _ = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, b, want_alpha=False)
    print("   z same with/without alpha:", bool(torch.equal(z, z2)), " alpha rowsum dev", float((alpha.sum(1) - 1).abs().max()))
    out, _ = ops.head_fwd(z, model.embedding.weight, c.bn1, c.bn2, lin.weight, lin.bias, b)
    stats("fp32 staged out", out, ref["out"])
    # bf16 storage
    xb = x.bfloat16()
    refb = gdn_oracle.forward(p64, xb.to(f64), k, graph=graph, storage="bf16")
    refb2 = gdn_oracle.forward(p64, xb.to(f64), k, graph=graph, storage="bf16", round_agg=True)
    with torch.no_grad():
        outb = model(xb.to(dev), None)
    stats("bf16 fused out vs bf16 oracle", outb, refb["out"])
    stats("bf16 fused out vs fp32 oracle", outb, ref["out"])
    xlb, sib, sjb = ops.project_fwd(xb.to(dev), gnn.lin.weight, c.terms)
    stats("bf16 project xlin", xlb.float(), refb["xlin"])
    stats("bf16 project s_i", sib, (ref["xlin"].view(b, n, d) @ p64["gnn_layers.0.gnn.att_i"].view(d)
                                    + p64["embedding.weight"] @ p64["gnn_layers.0.gnn.att_em_i"].view(d)).reshape(-1))
    zb, _ = ops.attn_aggregate_fwd(xlb, sib, sjb, c.graph, gnn.bias, b, want_alpha=False)
    stats("bf16 K8 z", zb.float(), refb2["agg"])
    outs, _ = ops.head_fwd(zb, model.embedding.weight, c.bn1, c.bn2, lin.weight, lin.bias, b)
    stats("bf16 staged out vs bf16 oracle", outs, refb2["out"])
print("done")
