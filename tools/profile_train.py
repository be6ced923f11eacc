"""One-GPU training steps at the SWaT shape (for rocprofv3): python3 tools/profile_train.py [B] [steps] [graph|eager]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_forward_parity import random_params
from gdn_amd.harness import GraphedTrainStep
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else "graph"
dev = torch.device("cuda:0")
model = random_params(127, 15, 30, 64, seed=0).to(dev)
st = GraphedTrainStep(model, B, use_graph=(mode == "graph"))
st.x.copy_(torch.rand_like(st.x)); st.y.copy_(torch.rand_like(st.y))
for _ in range(3): st.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): st.step()
torch.cuda.synchronize()
print(f"train step B={B} ({mode}): {(time.perf_counter() - t0) / steps * 1e3:.3f} ms  loss {st.loss.item():.5f}")
