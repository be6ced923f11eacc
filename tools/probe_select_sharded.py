"""Cost of the per-sensor median / IQR select in the shape a rank of an N-rank evaluation sees: its n/N sensors,
N x T ticks each, as N row blocks of pitch T (what ShardedEvaluator hands to gdn_score_select).
python3 tools/probe_select_sharded.py [ranks] [ticks_per_rank]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gdn_amd import ops
from gdn_amd.harness import HipScoreBackend, sensor_range
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
t = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
dev = torch.device("cuda:0")
n = 127
for r in (1, ranks):
    a, b = sensor_range(n, 0, r)
    mine = b - a
    keys = torch.rand((r, mine, t), dtype=torch.float64, device=dev, generator=None)
    ws = HipScoreBackend.select_workspace(r, mine, t, dev)
    out = torch.empty((mine, 2), dtype=torch.float64, device=dev)
    fn = lambda: HipScoreBackend.select(keys.reshape(-1), r, mine, t, r * t, ws=ws, out=out)
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ref = torch.quantile(keys.permute(1, 0, 2).reshape(mine, -1)[:2], torch.tensor([0.5], dtype=torch.float64, device=dev), dim=1)
    print(f"select: {r} block(s) x {mine} sensors x {t} ticks: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us per call; median[0] {float(out[0, 0]):.6f} (torch {float(ref[0, 0]):.6f})")
