"""Compare the lane words of a fused plan (bl: lin' operand fragments, bs: a_i/a_j fragments, cin, e2) with a host
emulation of the same arithmetic (fp32 products, f16 round-to-nearest split)."""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from test_gpu_forward_parity import random_params
dev = torch.device("cuda:0")
n, w, k, d, b = 127, 15, 30, 64, 300
model = random_params(n, w, k, d, seed=5).to(dev).eval()
c = model._constants()
plan = model._plan(c, False).cpu().numpy().view(np.uint32)
NT, DC, WK, SL, NTL = 4, 2, 1, 16, 2
T = 64 * NT
TABLE_WORDS = 4 * 32 * DC + 3 * 32 * NT
lanes = plan[TABLE_WORDS:]
def word(i, tid): return lanes[i * T + tid]
lin = model.gnn_layers[0].gnn.lin.weight.detach().cpu().numpy().astype(np.float32)
bn1 = c.bn1.cpu().numpy().astype(np.float32)
terms = c.terms.cpu().numpy().astype(np.float32)
def split(v):
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi, lo
bad = {"bl_hi": 0, "bl_lo": 0, "bs_hi": 0, "bs_lo": 0}
tot = 0
worst = 0.0
for tid in range(T):
    lane = tid & 63; l32 = lane & 31; h = lane >> 5
    i = 2 * SL
    for cb in range(DC):
        cc = cb * 32 + l32
        sc = np.float32(bn1[cc] * np.float32(8.0))
        for wk in range(WK):
            v = np.zeros(8, np.float32)
            for j in range(8):
                kk = wk * 16 + 8 * h + j
                v[j] = np.float32(lin[cc, kk] * sc) if kk < w else 0.0
            hi, lo = split(v)
            for t, ref in ((0, hi), (1, lo)):
                for e in range(4):
                    wd = int(word(i, tid)); i += 1
                    got = np.array([wd & 0xffff, wd >> 16], dtype=np.uint16).view(np.float16)
                    want = ref[2 * e: 2 * e + 2]
                    tot += 2
                    if not np.array_equal(got.view(np.uint16), want.view(np.uint16)):
                        bad["bl_hi" if t == 0 else "bl_lo"] += 1
                        worst = max(worst, float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max()))
    for wk in range(WK):
        v = np.zeros(8, np.float32)
        for j in range(8):
            kk = wk * 16 + 8 * h + j
            v[j] = np.float32(terms[l32 * 64 + kk] * np.float32(1.44269504088896340736)) if l32 < 2 else 0.0
        hi, lo = split(v)
        for t, ref in ((0, hi), (1, lo)):
            for e in range(4):
                wd = int(word(i, tid)); i += 1
                got = np.array([wd & 0xffff, wd >> 16], dtype=np.uint16).view(np.float16)
                if not np.array_equal(got.view(np.uint16), ref[2 * e: 2 * e + 2].view(np.uint16)):
                    bad["bs_hi" if t == 0 else "bs_lo"] += 1
print("mismatching words:", bad, "of", tot, "bl halves; worst |bl diff|", worst)
# one sample
tid = 5; i = 2 * SL
print("lane 5 bl words:", [hex(int(word(i + q, tid))) for q in range(8)])
import os
tag = os.environ.get("PLAN_TAG", "x")
os.makedirs("gpurun_out", exist_ok=True)
np.save(f"gpurun_out/plan_{tag}.npy", plan)
x = torch.rand((b, n, w), device=dev, generator=torch.Generator(device=dev).manual_seed(1))
with torch.no_grad():
    out = model(x, None)
np.save(f"gpurun_out/out_{tag}.npy", out.cpu().numpy())
