import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gdn_amd import ops
from oracle import gdn_oracle
from test_gpu_forward_parity import random_params
dev = torch.device("cuda:0")
for (n, w, k, b) in [(127, 15, 30, 16), (64, 15, 63, 8), (27, 5, 5, 64)]:
    model = random_params(n, w, k, 64, seed=5)
    p = {key: v.detach().clone().double() if v.is_floating_point() else v for key, v in model.state_dict().items()}
    model = model.to(dev).eval()
    x = torch.rand((b, n, w), generator=torch.Generator().manual_seed(6))
    c = model._constants()
    gnn = model.gnn_layers[0].gnn
    ref = gdn_oracle.forward(p, x.double(), k, graph=c.graph.topk.cpu())
    xlin, s_i, s_j = ops.project_fwd(x.to(dev), gnn.lin.weight, c.terms)
    z0, _ = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, b, want_alpha=False)
    z2, al = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, b, want_alpha=True)
    e0 = (z0.cpu().double() - ref["agg"]).abs().view(b, n, 64)
    e2 = (z2.cpu().double() - ref["agg"]).abs().view(b, n, 64)
    print(n, w, k, "no-alpha err", float(e0.max()), "alpha err", float(e2.max()))
    d = (z0 - z2).abs().cpu().view(b, n, 64)
    idx = torch.nonzero(d > 1e-6)
    print("   differing elements:", idx.shape[0], "of", d.numel(), "; rows (sensor) involved:", sorted(set(idx[:, 1].tolist()))[:40])
    print("   per-window counts:", [int((d[i] > 1e-6).sum()) for i in range(min(b, 8))])
    print("   e0 bad rows:", sorted(set(torch.nonzero(e0 > 2e-6)[:, 1].tolist()))[:40], " e2 bad rows:", sorted(set(torch.nonzero(e2 > 2e-6)[:, 1].tolist()))[:40])
