"""Dense (matrix-core) fused forward vs the fp64 oracle and vs the VALU kernel; launch timings.
usage: python3 tools/check_dense.py [batch]"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import gdn_oracle  # noqa: E402
from test_gpu_forward_parity import random_params  # noqa: E402

dev = torch.device("cuda:0")


def check(n, w, k, d, b, seed=3):
    model = random_params(n, w, k, d, seed=seed)
    p = {key: v.detach().clone() for key, v in model.state_dict().items()}
    model = model.to(dev).eval()
    x = torch.rand((b, n, w), generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        out = model(x.to(dev), None)
    torch.cuda.synchronize()
    f64 = torch.float64
    p64 = {key: (v.to(f64) if v.is_floating_point() else v) for key, v in p.items()}
    ref = gdn_oracle.forward(p64, x.to(f64), k, graph=model.learned_graph.cpu())["out"]
    err = (out.cpu().double() - ref).abs()
    print(f"n={n} w={w} k={k} d={d} b={b}: max|hip-f64 oracle| = {err.max():.3e}  mean {err.mean():.3e}  "
          f"|ref|max {ref.abs().max():.3f}  nan={int(torch.isnan(out).sum())}", flush=True)
    return float(err.max())


def timing(b):
    from gdn_amd import _lib
    n, w, k, d = 127, 15, 30, 64
    model = random_params(n, w, k, d, seed=0).to(dev).eval()
    x = torch.rand((b, n, w), device=dev)
    out = torch.empty((b, n), device=dev)
    c = model._constants()
    ptrs = c.fused_args[0]
    plan = model._plan(c, False)
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        if plan is not None:
            _lib.call("gdn_forward_fused_plan", x.data_ptr(), plan.data_ptr(), b, n, w, d, k, 0, out.data_ptr(), None, st)
        else:
            _lib.call("gdn_forward_fused", x.data_ptr(), *ptrs, b, n, w, d, k, out.data_ptr(), st)
    for _ in range(200):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 100
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"path={os.environ.get('GDN_FUSED_PATH', 'dense')} batch {b}: {us:.1f} us/launch = {b / us:.2f} M windows/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        timing(int(sys.argv[2]))
        sys.exit(0)
    worst = 0.0
    for shape in [(127, 15, 30, 64, 8), (27, 5, 5, 64, 16), (64, 15, 63, 64, 4), (51, 15, 15, 64, 700), (100, 30, 40, 64, 5),
                  (127, 15, 30, 64, 2000), (33, 12, 1, 64, 3)]:
        worst = max(worst, check(*shape))
    print("worst", worst)
    for path in ("dense", "valu"):
        for b in (512, 4096, 32768):
            subprocess.run([sys.executable, __file__, "time", str(b)], env=dict(os.environ, GDN_FUSED_PATH=path))
