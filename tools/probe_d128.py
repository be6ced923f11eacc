"""d = 128 fused forward: matrix-core kernel vs the fp32 row-gather kernel (GDN_FUSED_PATH=valu).
python3 tools/probe_d128.py [batch]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_forward_parity import random_params
b = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for (n, w, k, d) in ((127, 15, 30, 128), (127, 15, 30, 64), (127, 30, 30, 128)):
    m = random_params(n, w, k, d, seed=1).cuda().eval()
    for dt in (torch.float32, torch.bfloat16):
        x = torch.rand((b, n, w), device="cuda").to(dt)
        with torch.no_grad():
            for _ in range(30): m(x, None)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50): m(x, None)
            torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 50
        print(f"path={os.environ.get('GDN_FUSED_PATH','dense')} n={n} w={w} k={k} d={d} {str(dt)[6:]}: {dtm*1e6:.1f} us per {b} windows = {b/dtm/1e6:.1f} M windows/s", flush=True)
