"""One-GPU training steps at the SWaT shape (for rocprofv3): python3 tools/profile_train.py [B] [steps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_forward_parity import random_params
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
model = random_params(127, 15, 30, 64, seed=0).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
x = torch.rand((B, 127, 15), device=dev); y = torch.rand((B, 127), device=dev)
def step():
    opt.zero_grad(); loss = torch.nn.functional.mse_loss(model(x, None), y); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps): step()
torch.cuda.synchronize()
print(f"train step B={B}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms")
