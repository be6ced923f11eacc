"""Time of the staged gather-aggregate (K8) at one batch size, fp32 and bf16 storage (same-box A/B runs of alternate
builds: GDN_HIP_LIB=...).  python3 tools/probe_k8_time.py [batch]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gdn_amd import ops  # noqa: E402
from test_gpu_forward_parity import random_params  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = torch.device("cuda:0")
model = random_params(127, 15, 30, 64, seed=0).to(dev).eval()
x = torch.rand((B, 127, 15), generator=torch.Generator().manual_seed(0)).to(dev)
c = model._constants()
gnn = model.gnn_layers[0].gnn
res = []
for name, xin in (("fp32", x), ("bf16", x.bfloat16())):
    xlin, s_i, s_j = ops.project_fwd(xin, gnn.lin.weight, c.terms)
    fn = lambda: ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, B, False)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
    best, tot = 1e9, 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        best, tot = min(best, us), tot + us
    res.append(f"{name} avg {tot / 5:.1f} best {best:.1f} us")
print(f"K8 B={B} {os.path.basename(os.environ.get('GDN_HIP_LIB', 'libgdn_hip.so'))}: " + "; ".join(res))
