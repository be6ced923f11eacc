"""Scoring kernels only (for rocprofv3): python3 tools/profile_score.py [T] [N] [reps]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gdn_amd import ops
t = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
n = int(sys.argv[2]) if len(sys.argv) > 2 else 127
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
g = torch.Generator().manual_seed(0)
pred = torch.rand((t, n), generator=g).cuda()
gt = torch.rand((t, n), generator=g).cuda()
for _ in range(reps):
    mi = ops.score_quantiles(pred, gt)
    ops.score_smooth_max(pred, gt, mi, want_scores=False)
torch.cuda.synchronize()
print("done")
