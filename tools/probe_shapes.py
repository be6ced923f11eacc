"""Throughput of other BASELINE shapes (not bench lines): fused eval forward and one training step."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gdn_amd import ops
from test_gpu_forward_parity import random_params
dev = torch.device("cuda:0")

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

for name, (n, w, k, d, B) in {"C2 fc64": (64, 15, 64, 64, 128), "C3 swat": (127, 15, 30, 64, 4096),
                               "C5 wadi-stress d64": (512, 30, 64, 64, 512), "C5 wadi-stress d128": (512, 30, 64, 128, 512),
                               "msl demo": (27, 5, 5, 64, 4096)}.items():
    try:
        model = random_params(n, w, k, d, seed=0).to(dev).eval()
        x = torch.rand((B, n, w), device=dev)
        with torch.no_grad():
            ms = timeit(lambda: model(x, None))
        print(f"{name:22s} N={n} W={w} K={k} D={d} B={B}: eval forward {ms*1e3:9.1f} us  {B/ms/1e3:8.3f} Mwin/s")
    except Exception as e:
        print(f"{name}: eval failed: {type(e).__name__}: {str(e)[:120]}")
    try:
        Bt = min(B, 512)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        xt = torch.rand((Bt, n, w), device=dev); yt = torch.rand((Bt, n), device=dev)
        def step():
            opt.zero_grad(); loss = torch.nn.functional.mse_loss(model(xt, None), yt); loss.backward(); opt.step()
        ms = timeit(step, iters=5, warm=2)
        print(f"{'':22s} train step B={Bt}: {ms:8.3f} ms  {Bt/ms:8.1f} kwin/s")
    except Exception as e:
        print(f"{'':22s} train failed: {type(e).__name__}: {str(e)[:120]}")
