import sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_gpu_forward_parity import random_params
from gdn_amd import harness
dev = torch.device("cuda:0")
model = random_params(127, 15, 30, 64, seed=0).to(dev).eval()
t = 32768
x = torch.rand((t, 127, 15), device=dev); y = torch.rand((t, 127), device=dev)
ev = harness.SeriesEvaluator(model, x, y, batch=512, coalesce=8, streams=2)
first = ev.step().clone(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20000):
    ev.step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"eval soak: 20000 replays in {dt:.1f} s = {20000 * t / dt / 1e6:.1f} M windows/s; result unchanged: {torch.equal(first, ev.anomaly)}")
m2 = random_params(127, 15, 30, 64, seed=1).to(dev)
st = harness.GraphedTrainStep(m2, 512)
st.x.copy_(torch.rand_like(st.x)); st.y.copy_(torch.rand_like(st.y))
l0 = float(st.step()); torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20000):
    st.step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
l1 = float(st.loss)
ok = all(torch.isfinite(p).all().item() for p in m2.parameters())
print(f"train soak: 20000 graphed steps in {dt:.1f} s = {dt / 20000 * 1e3:.3f} ms/step; loss {l0:.4f} -> {l1:.4f}; parameters finite: {ok}")
