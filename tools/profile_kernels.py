"""Launch each forward kernel a few times at one batch size (for rocprofv3 runs).
usage: python3 tools/profile_kernels.py [batch] [reps] [n w k d]      (default shape 127 15 30 64; n > 127 =
BASELINE configs[4]: the fp32 row-gather family, staged bf16 storage does not exist there)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gdn_amd import ops  # noqa: E402
from test_gpu_forward_parity import random_params  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
n, w, k, d = (int(v) for v in sys.argv[3:7]) if len(sys.argv) >= 7 else (127, 15, 30, 64)
model = random_params(n, w, k, d, seed=0).to(dev).eval()
g = torch.Generator().manual_seed(0)
x = torch.rand((B, n, w), generator=g).to(dev)
c = model._constants()
gnn = model.gnn_layers[0].gnn
lin = model.out_layer.mlp[0]
out = torch.empty((B, n), device=dev)
xb = x.bfloat16()
# Every kernel is launched `reps` times BACK TO BACK (the way a throughput kernel runs), after ~100 ms of sustained
# launches: this GPU needs ~50 ms of load after an idle period to reach its steady clocks (tools/probe_ramp.py), and a
# roofline fraction is a statement about steady state — rocprofv3's per-kernel average then agrees with the HIP-event
# timing inside bench.py (interleaved, cold: the round-2 form of this script, K8 read 442 us instead of 413).
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.1:
    for _ in range(8):
        model.forward_into(x, out)
    torch.cuda.synchronize()
dense16 = n <= 127 and d == 64
xlin, s_i, s_j = ops.project_fwd(x, gnn.lin.weight, c.terms)
z, _ = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, B, False)
if dense16:
    xl16, si16, sj16 = ops.project_fwd(xb, gnn.lin.weight, c.terms)
    z16, _ = ops.attn_aggregate_fwd(xl16, si16, sj16, c.graph, gnn.bias, B, False)
legs = [lambda: model.forward_into(x, out),              # gdn_forward_fused_plan (what GDN.forward launches)
        lambda: model.forward_into(xb, out),             # bf16 storage
        lambda: ops.project_fwd(x, gnn.lin.weight, c.terms),
        lambda: ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, B, False),
        lambda: ops.head_fwd(z, model.embedding.weight, c.bn1, c.bn2, lin.weight, lin.bias, B)]
if dense16:
    legs += [lambda: ops.project_fwd(xb, gnn.lin.weight, c.terms),
             lambda: ops.attn_aggregate_fwd(xl16, si16, sj16, c.graph, gnn.bias, B, False),
             lambda: ops.head_fwd(z16, model.embedding.weight, c.bn1, c.bn2, lin.weight, lin.bias, B)]
for leg in legs:
    for _ in range(reps):
        leg()
torch.cuda.synchronize()
print("done")
