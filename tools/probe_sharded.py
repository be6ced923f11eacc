"""ShardedEvaluator step time vs exchange chunk, 1-rank RCCL group: python3 tools/probe_sharded.py"""
import os, sys, time
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_forward_parity import random_params
from gdn_amd import harness
dev = torch.device("cuda:0")
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
fd = os.dup(1); os.dup2(2, 1)
dist.init_process_group("nccl", device_id=dev)
model = random_params(127, 15, 30, 64, seed=0).to(dev).eval()
t = 32768
x = torch.rand((t, 127, 15), device=dev); y = torch.rand((t, 127), device=dev)
res = []
for chunk in (4096, 8192, 16384, 32768, 16384, 8192):
    sev = harness.ShardedEvaluator(model, x, y, t, chunk=chunk)
    for _ in range(5): sev.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): sev.step()
    host = (time.perf_counter() - t0) / 30
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 30
    res.append(f"chunk {chunk:6d}: {tot * 1e3:.3f} ms/step (host issue {host * 1e3:.3f} ms)  chunks {sev.nchunks}")
dist.destroy_process_group()
os.dup2(fd, 1)
print("\n".join(res))
