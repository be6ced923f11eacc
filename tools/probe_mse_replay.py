"""Round-1 finding "F.mse_loss inside the captured training step gives wrong losses at 4096 windows when the
replays are issued back to back": reproduce it and find the first tensor that diverges from the eager loop.
usage: GDN_TORCH_MSE=1 python3 tools/probe_mse_replay.py [sync]   ("sync" = synchronize after every replay)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gdn_amd.harness import AutogradTrainStep  # noqa: E402
from test_gpu_forward_parity import random_params  # noqa: E402

dev = torch.device("cuda:0")
sync_each = len(sys.argv) > 1 and sys.argv[1] == "sync"
b, steps = 4096, 10
g = torch.Generator().manual_seed(0)
x = torch.rand((b, 127, 15), generator=g).to(dev)
y = torch.rand((b, 127), generator=g).to(dev)


def make():
    m = random_params(127, 15, 30, 64, seed=0).to(dev).train()
    m.dp.p = 0.0
    return m


def dbg_buffers(model):
    d = {"out": torch.zeros((b, 127), device=dev), "loss_raw": torch.zeros((), device=dev)}
    for name, prm in model.named_parameters():
        d["g/" + name] = torch.zeros_like(prm)
    return d


runs = {}
for mode in ("eager", "graph"):
    model = make()
    step = AutogradTrainStep(model, b, use_graph=(mode == "graph"))
    step._dbg = dbg_buffers(model)
    step.x.copy_(x)
    step.y.copy_(y)
    snaps = []
    for k in range(steps):
        step.step()
        snaps.append({key: t.clone() for key, t in step._dbg.items()} | {"loss": step.loss.clone()})   # same stream, no host sync
        if sync_each or mode == "eager":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    runs[mode] = snaps
print(f"torch_mse={bool(os.environ.get('GDN_TORCH_MSE'))} sync_each={sync_each}")
for k in range(steps):
    e, r = runs["eager"][k], runs["graph"][k]
    bad = [(key, float((e[key] - r[key]).abs().max())) for key in e if float((e[key] - r[key]).abs().max()) > 1e-5]
    print(f"step {k}: loss eager {float(e['loss']):.6f} graph {float(r['loss']):.6f}  diverging: {bad[:6]}")
