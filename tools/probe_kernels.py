import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gdn_amd import GDN, ops, harness
from tests.test_gpu_forward_parity import random_params
dev = torch.device("cuda:0")
n, w, k, d = 127, 15, 30, 64
model = random_params(n, w, k, d, seed=0).to(dev).eval()
T = 32768
g = torch.Generator().manual_seed(0)
x = torch.rand((T, n, w), generator=g).to(dev); y = torch.rand((T, n), generator=g).to(dev)

def _bw():
    a = torch.empty((1 << 28,), dtype=torch.float32, device=dev); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    print(f"calibration: torch copy of 1 GiB: {2 * a.numel() * 4 * 10 / e0.elapsed_time(e1) / 1e6:.0f} GB/s (read+write)")
_bw()

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters   # ms

c = model._constants()
gnn = model.gnn_layers[0].gnn; lin = model.out_layer.mlp[0]
for B in (512, 2048, 8192, 32768):
    out = torch.empty((B, n), device=dev)
    xs = x[:B]
    ms = timeit(lambda: ops.forward_fused(xs, gnn.lin.weight, c.terms, c.graph, gnn.bias, model.embedding.weight, c.bn1, c.bn2, lin.weight, lin.bias, out=out), iters=50)
    print(f"fused  B={B:6d}: {ms*1e3:8.1f} us/launch  {B/ms/1e3:8.2f} Mwin/s   x+out {B*(n*w*4+n*4)/ms/1e6:7.1f} GB/s")
    xlin, s_i, s_j = ops.project_fwd(xs, gnn.lin.weight, c.terms)
    ms_p = timeit(lambda: ops.project_fwd(xs, gnn.lin.weight, c.terms), iters=30)
    z, _ = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, B, False)
    zz = torch.empty_like(xlin)
    from gdn_amd import _lib
    st = torch.cuda.current_stream().cuda_stream
    def k8():
        _lib.call("gdn_attn_aggregate_fwd", xlin.data_ptr(), s_i.data_ptr(), s_j.data_ptr(), c.graph.nbr.data_ptr(), c.graph.deg.data_ptr(), gnn.bias.data_ptr(), B, n, d, k, zz.data_ptr(), None, st)
    ms_a = timeit(k8, iters=50)
    ms_h = timeit(lambda: ops.head_fwd(z, model.embedding.weight, c.bn1, c.bn2, lin.weight, lin.bias, B), iters=30)
    bytes_k8 = 2*B*n*d*4 + n*32*2
    print(f"staged B={B:6d}: project {ms_p*1e3:7.1f} us  attn {ms_a*1e3:7.1f} us ({bytes_k8/ms_a/1e6:7.1f} GB/s)  head {ms_h*1e3:7.1f} us  total {B/(ms_p+ms_a+ms_h)/1e3:7.2f} Mwin/s")
for batch in (512, 4096, 32768):
    for use_graph in (False, True):
        ev = harness.SeriesEvaluator(model, x, y, batch=batch, use_graph=use_graph)
        ms = timeit(ev.step, iters=10, warm=2)
        print(f"series T={T} batch={batch} graph={use_graph}: {ms:8.3f} ms/step  {T/ms/1e3:8.2f} Mwin/s")
# score only
pred = torch.rand((T, n), device=dev)
ms = timeit(lambda: ops.score_smooth_max(pred, y, ops.score_quantiles(pred, y), want_scores=False), iters=10)
print(f"score T={T}: {ms*1e3:.1f} us")
