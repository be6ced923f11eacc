import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from test_gpu_forward_parity import random_params
from gdn_amd import _lib
from oracle import gdn_oracle
dev = torch.device("cuda:0")
n, w, k, d, b = 127, 15, 30, 64, 300
model = random_params(n, w, k, d, seed=5).to(dev).eval()
x = torch.rand((b, n, w), device=dev)
c = model._constants()
ptrs = c.fused_args[0]
st = torch.cuda.current_stream().cuda_stream
def plain():
    o = torch.empty((b, n), device=dev)
    _lib.call("gdn_forward_fused", x.data_ptr(), *ptrs, b, n, w, d, k, o.data_ptr(), st)
    torch.cuda.synchronize()
    return o
first = plain()
with torch.no_grad():
    planned = model(x, None)
second = plain()
junk = torch.randn((64, 1024, 1024), device=dev).sum()      # other kernels in between
third = plain()
p = {key: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for key, v in model.state_dict().items()}
ref = gdn_oracle.forward(p, x.cpu().double(), k, 1, graph=model.learned_graph.cpu())["out"]
for name, t in (("plain before the plan exists", first), ("planned", planned), ("plain after planned", second), ("plain after other work", third)):
    print(name, "max |err vs float64|", float((t.cpu().double() - ref).abs().max()))
