import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from test_gpu_forward_parity import random_params
from gdn_amd import _lib, ops
dev = torch.device("cuda:0")
n, w, k, d, b = 127, 15, 30, 64, 300
model = random_params(n, w, k, d, seed=5).to(dev).eval()
x = torch.rand((b, n, w), device=dev)
c = model._constants()
plan = model._plan(c, False)
print("limit", model.operand_limit(), "plan words", plan.numel())
g = torch.zeros((2,), dtype=torch.int32, device=dev)
out = torch.empty((b, n), device=dev)
st = torch.cuda.current_stream().cuda_stream
_lib.call("gdn_forward_fused_plan", x.data_ptr(), plan.data_ptr(), b, n, w, d, k, 0, out.data_ptr(), g.data_ptr(), st)
torch.cuda.synchronize()
print("guard after planned launch on x in [0,1):", g.tolist())
ptrs = c.fused_args[0]
out2 = torch.full((b, n), 7.0, device=dev)
_lib.call("gdn_forward_fused_gated", g.data_ptr(), x.data_ptr(), *ptrs, b, n, w, d, k, out2.data_ptr(), st)
torch.cuda.synchronize()
print("guard after gated:", g.tolist(), "gated wrote:", bool((out2 != 7.0).any()))
outs = {}
def run(name, fn):
    o = torch.empty((b, n), device=dev); fn(o); torch.cuda.synchronize(); outs[name] = o
run("plan_noguard", lambda o: _lib.call("gdn_forward_fused_plan", x.data_ptr(), plan.data_ptr(), b, n, w, d, k, 0, o.data_ptr(), None, st))
run("plan_guard", lambda o: _lib.call("gdn_forward_fused_plan", x.data_ptr(), plan.data_ptr(), b, n, w, d, k, 0, o.data_ptr(), g.data_ptr(), st))
run("plain", lambda o: _lib.call("gdn_forward_fused", x.data_ptr(), *ptrs, b, n, w, d, k, o.data_ptr(), st))
run("valu", lambda o: _lib.call("gdn_forward_fused_gated", None, x.data_ptr(), *ptrs, b, n, w, d, k, o.data_ptr(), st))
with torch.no_grad():
    outs["model"] = model(x, None)
for a_ in outs:
    print(a_, {b_: float((outs[a_] - outs[b_]).abs().max()) for b_ in outs})
