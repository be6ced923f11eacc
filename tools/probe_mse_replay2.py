"""As probe_mse_replay.py but with NOTHING between the replays (no snapshots): final loss / parameters only.
usage: GDN_TORCH_MSE=1 python3 tools/probe_mse_replay2.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gdn_amd.harness import AutogradTrainStep  # noqa: E402
from test_gpu_forward_parity import random_params  # noqa: E402

dev = torch.device("cuda:0")
for b in (512, 4096, 16384):
    steps = 20
    g = torch.Generator().manual_seed(0)
    x = torch.rand((b, 127, 15), generator=g).to(dev)
    y = torch.rand((b, 127), generator=g).to(dev)
    res = {}
    for mode in ("eager", "graph", "graph_sync"):
        m = random_params(127, 15, 30, 64, seed=0).to(dev).train()
        m.dp.p = 0.0
        step = AutogradTrainStep(m, b, use_graph=(mode != "eager"))
        step.x.copy_(x)
        step.y.copy_(y)
        losses = []
        for k in range(steps):
            step.step()
            if mode != "graph":
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        res[mode] = (float(step.loss), [p.detach().clone() for p in m.parameters()])
    names = [n for n, _ in m.named_parameters()]
    for mode in ("graph", "graph_sync"):
        dmax = max((float((a - c).abs().max()), n) for a, c, n in zip(res["eager"][1], res[mode][1], names) if not n.endswith("gnn.bias"))
        print(f"torch_mse={bool(os.environ.get('GDN_TORCH_MSE'))} b={b} {mode}: final loss {res[mode][0]:.6f} (eager {res['eager'][0]:.6f}); "
              f"largest parameter difference {dmax[0]:.2e} ({dmax[1]})")
