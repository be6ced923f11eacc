"""Headline step split: forward-only graph vs forward+score graph, and the scoring kernels alone in a graph."""
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


def timed(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for coalesce, streams in ((8, 4), (64, 1)):
    ev = harness.SeriesEvaluator(model, x, y, batch=512, coalesce=coalesce, streams=streams)
    f = timed(ev.forward_only); s = timed(ev.step)
    g = torch.cuda.CUDAGraph()
    ev._launch_score(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        ev._launch_score()
    sc = timed(g.replay)
    print(f"coalesce {coalesce} streams {streams}: forward graph {f:.3f} ms, forward+score graph {s:.3f} ms, score-only graph {sc:.3f} ms")
