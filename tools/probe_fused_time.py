"""Time of the planned fused forward at one batch size (A/B runs of environment knobs, one process per setting):
python3 tools/probe_fused_time.py [batch] [bf16]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_forward_parity import random_params  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
bf16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
dev = torch.device("cuda:0")
model = random_params(127, 15, 30, 64, seed=0).to(dev).eval()
x = torch.rand((B, 127, 15), generator=torch.Generator().manual_seed(0)).to(dev)
if bf16:
    x = x.bfloat16()
out = torch.empty((B, 127), device=dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:        # steady clocks
    for _ in range(8):
        model.forward_into(x, out)
    torch.cuda.synchronize()
best, tot, reps = 1e9, 0.0, 5
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        model.forward_into(x, out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    best, tot = min(best, us), tot + us
knobs = {k: os.path.basename(v) for k, v in os.environ.items() if k.startswith("GDN_")}
print(f"B={B} {'bf16' if bf16 else 'fp32'} {knobs}: avg {tot / reps:.1f} us, best {best:.1f} us, checksum {float(out.sum()):.6f}")
