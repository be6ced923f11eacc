"""Diagnostic: phases of gdn_dense_attn_bwd_kernel (workgroup 0, wave 0, its FIRST window; 100 MHz clock).
Needs a library built with -DGDN_STAMPS:  GDN_HIP_LIB=/path/libgdn_stamps.so python3 tools/probe_stamps_bwd.py"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gdn_amd import _lib, ops  # noqa: E402
from test_gpu_forward_parity import random_params  # noqa: E402
dev = torch.device("cuda:0")
model = random_params(127, 15, 30, 64, seed=0).to(dev)
c = model._constants()
gnn = model.gnn_layers[0].gnn
names = {20: "prologue done", 21: "alpha loads issued", 22: "max + B0", 23: "tiles written + B1", 24: "G products", 25: "B2 + G stored",
         26: "softmax backward", 27: "B3 + zero + B4", 28: "dl scatter + image scatter + B5", 29: "row sums d_sj", 30: "dX hi products",
         31: "B6 + lo scatter + B7 + lo products", 32: "d_xlin stored", 33: "all windows done", 34: "kernel end"}
for b in (512, 2048):
    x = torch.rand((b, 127, 15), device=dev)
    xlin, s_i, s_j = ops.project_fwd(x, gnn.lin.weight, c.terms)
    z, alpha = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, b, want_alpha=True)
    d_z = torch.randn_like(z) * 1e-5
    for _ in range(20):
        ops.attn_aggregate_bwd(d_z, xlin, alpha, s_i, s_j, c.graph, b)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    lib = _lib.load()
    lib.gdn_debug_read_stamps.argtypes = [ctypes.c_void_p]
    assert lib.gdn_debug_read_stamps(buf) == 0
    t = [buf[i] for i in range(64)]
    print(f"batch {b}: (ns since stamp 20)")
    for i in range(21, 35):
        print(f"   {names[i]:40s} +{(t[i] - t[i - 1]) * 10:7d} ns   at {(t[i] - t[20]) * 10:7d} ns")
