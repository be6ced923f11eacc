"""Launch each forward kernel a few times at one batch size (for rocprofv3 runs).
usage: python3 tools/profile_kernels.py [batch] [reps]"""
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
n, w, k, d = 127, 15, 30, 64
model = random_params(n, w, k, d, seed=0).to(dev).eval()
g = torch.Generator().manual_seed(0)
x = torch.rand((B, n, w), generator=g).to(dev)
c = model._constants()
gnn = model.gnn_layers[0].gnn
lin = model.out_layer.mlp[0]
out = torch.empty((B, n), device=dev)
xb = x.bfloat16()
for _ in range(reps):
    model.forward_into(x, out)              # gdn_forward_fused_plan (what GDN.forward launches)
    model.forward_into(xb, out)             # bf16 storage
    xl16, si16, sj16 = ops.project_fwd(xb, gnn.lin.weight, c.terms)
    z16, _ = ops.attn_aggregate_fwd(xl16, si16, sj16, c.graph, gnn.bias, B, False)
    ops.head_fwd(z16, model.embedding.weight, c.bn1, c.bn2, lin.weight, lin.bias, B)
    xlin, s_i, s_j = ops.project_fwd(x, gnn.lin.weight, c.terms)
    z, _ = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, B, False)
    ops.head_fwd(z, model.embedding.weight, c.bn1, c.bn2, lin.weight, lin.bias, B)
torch.cuda.synchronize()
print("done")
