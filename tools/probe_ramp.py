"""Per-replay duration of the headline step after a synchronize (events between replays)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_forward_parity import random_params
from gdn_amd import harness
dev = torch.device("cuda:0")
model = random_params(127, 15, 30, 64, seed=0).to(dev).eval()
t = 32768
x = torch.rand((t, 127, 15), device=dev); y = torch.rand((t, 127), device=dev)
ev = harness.SeriesEvaluator(model, x, y, batch=512, coalesce=8, streams=4)
for _ in range(3): ev.step()
torch.cuda.synchronize()
for trial in range(2):
    n = 60
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    evs[0].record()
    for i in range(n):
        ev.step(); evs[i + 1].record()
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    d = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
    print(f"trial {trial}: wall {tot*1e3:.2f} ms for {n} steps (host issue {host*1e3:.2f}); per-step ms:", " ".join(f"{v:.2f}" for v in d[:24]), "... last", " ".join(f"{v:.2f}" for v in d[-4:]))
    time.sleep(0.5)
