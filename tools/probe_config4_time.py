"""Time of the fused forward at BASELINE configs[4] (512 sensors, top-k 64, W=30) for same-box A/B runs of alternate
builds (GDN_HIP_LIB=...).  python3 tools/probe_config4_time.py [batch] [d]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_forward_parity import random_params  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
model = random_params(512, 30, 64, d, seed=0).to(dev).eval()
x = torch.rand((B, 512, 30), generator=torch.Generator().manual_seed(0)).to(dev)
out = torch.empty((B, 512), device=dev)
res = []
for name, xin in (("fp32", x), ("bf16", x.bfloat16())):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        model.forward_into(xin, out)
        torch.cuda.synchronize()
    best, tot = 1e9, 0.0
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            model.forward_into(xin, out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        best, tot = min(best, us), tot + us
    res.append(f"{name} avg {tot / 4:.1f} best {best:.1f} us, sum {float(out.sum()):.4f}")
print(f"config4 d={d} B={B} {os.path.basename(os.environ.get('GDN_HIP_LIB', 'libgdn_hip.so'))}: " + "; ".join(res))
