"""Training step at the paper's WADI width (d = 128, 127 sensors) against d = 64: python3 tools/probe_train_d128.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_forward_parity import random_params
from gdn_amd.harness import GraphedTrainStep
dev = torch.device("cuda:0")
for d in (64, 128):
    model = random_params(127, 15, 30, d, seed=0).to(dev)
    st = GraphedTrainStep(model, 512)
    st.x.copy_(torch.rand_like(st.x)); st.y.copy_(torch.rand_like(st.y))
    for _ in range(5): st.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): st.step()
    torch.cuda.synchronize()
    print(f"d={d}: {type(st).__name__} {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per 512-window step")
