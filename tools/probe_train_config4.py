"""Captured training step at BASELINE configs[4] (512 sensors, top-k 64, W=30, 512 windows), d = 64 and 128: for
same-box A/B runs of alternate builds (GDN_HIP_LIB=...).  python3 tools/probe_train_config4.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_forward_parity import random_params
from gdn_amd.harness import GraphedTrainStep
dev = torch.device("cuda:0")
res = []
for d in (64, 128):
    model = random_params(512, 30, 64, d, seed=0).to(dev)
    st = GraphedTrainStep(model, 512)
    st.x.copy_(torch.rand_like(st.x)); st.y.copy_(torch.rand_like(st.y))
    for _ in range(5): st.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): st.step()
    torch.cuda.synchronize()
    res.append(f"d={d}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms")
print(f"train config4 {os.path.basename(os.environ.get('GDN_HIP_LIB', 'libgdn_hip.so'))}: " + "; ".join(res))
