"""GPU suite: HIP forward kernels (through the C ABI) against the golden vectors captured
from the reference and against the oracle on seeded inputs.  fp32 tolerance from
BASELINE.json's north_star: 1e-4 (we assert 2e-5, observed ~1e-6)."""
import numpy as np
import pytest
import torch

from conftest import MODEL_CASES, load_golden, meta
from oracle import gdn_oracle

pytestmark = pytest.mark.gpu
TOL = 2e-5


def build_model(params, m, device):
    from gdn_amd import GDN
    model = GDN([torch.zeros((2, 1), dtype=torch.long)], m["n"], dim=m["d"], out_layer_inter_dim=m["inter"],
                input_dim=m["w"], out_layer_num=m["out_layer_num"], topk=m["k"])
    missing = model.load_state_dict(params, strict=True)          # same keys/shapes as the reference
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.to(device).eval()


@pytest.mark.parametrize("case", MODEL_CASES)
def test_learned_graph_matches_reference_topk(case, gpu_device):
    data, p = load_golden(case)
    m = meta(data)
    from gdn_amd import ops
    g = ops.topk_graph(p["embedding.weight"].to(gpu_device), m["k"], want_cos=True)
    cos_ref = gdn_oracle.cosine_matrix(p["embedding.weight"])
    np.testing.assert_allclose(g.cos.cpu().numpy(), cos_ref.numpy(), atol=1e-6, rtol=0)
    if float(data["cos_gap"]) > 1e-5:
        # descending order is only defined up to ties; fc64 (k = n) has none above the gap either
        got, want = g.topk.cpu().numpy(), data["learned_graph"]
        if m["k"] < m["n"]:
            np.testing.assert_array_equal(got, want)
        else:
            np.testing.assert_array_equal(np.sort(got, axis=1), np.sort(want, axis=1))
    deg = g.deg.cpu().numpy()
    topk = g.topk.cpu().numpy()
    for i in range(m["n"]):
        assert deg[i] == (m["k"] if i in topk[i] else m["k"] + 1)


@pytest.mark.parametrize("case", MODEL_CASES)
def test_staged_kernels_with_injected_graph(case, gpu_device):
    """Kernel-level parity: the reference's learned_graph is injected, every intermediate the
    fixture pins is compared (xlin via the oracle, agg / att_weight_1 / edge_index_1 / out)."""
    data, p = load_golden(case)
    m = meta(data)
    from gdn_amd import ops
    model = build_model(p, m, gpu_device)
    model.injected_graph = torch.from_numpy(data["learned_graph"]).to(gpu_device)
    x = torch.from_numpy(data["x"]).to(gpu_device)
    c = model._constants()
    gnn = model.gnn_layers[0].gnn
    xlin, s_i, s_j = ops.project_fwd(x, gnn.lin.weight, c.terms)
    ref = gdn_oracle.forward(p, torch.from_numpy(data["x"]), m["k"], m["out_layer_num"],
                             graph=torch.from_numpy(data["learned_graph"]))
    np.testing.assert_allclose(xlin.cpu().numpy(), ref["xlin"].numpy(), atol=TOL, rtol=0)
    z, alpha = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, m["b"], want_alpha=True)
    np.testing.assert_allclose(z.cpu().numpy(), data["agg"], atol=TOL, rtol=0)
    layer = model.gnn_layers[0]
    layer._set_dense((alpha, c.graph, m["b"]))
    np.testing.assert_array_equal(layer.edge_index_1.cpu().numpy(), data["edge_index_1"])
    np.testing.assert_allclose(layer.att_weight_1.cpu().numpy(), data["att_weight_1"], atol=TOL, rtol=0)
    # alpha rows sum to 1 and padding slots are exactly 0
    a = alpha.view(m["b"] * m["n"], -1).cpu().numpy()
    np.testing.assert_allclose(a.sum(axis=1), 1.0, atol=1e-5)
    with torch.no_grad():
        out = model(x, None)
    np.testing.assert_allclose(out.cpu().numpy(), data["eval_out"], atol=TOL, rtol=0)
    if m["out_layer_num"] == 1:
        bn1, bn2 = c.bn1, c.bn2
        lin = model.out_layer.mlp[0]
        out2, _ = ops.head_fwd(z, model.embedding.weight, bn1, bn2, lin.weight, lin.bias, m["b"])
        np.testing.assert_allclose(out2.cpu().numpy(), data["eval_out"], atol=TOL, rtol=0)


@pytest.mark.parametrize("case", [c for c in MODEL_CASES if c != "dupemb_n10_k3"])
def test_end_to_end_forward_learns_its_own_graph(case, gpu_device):
    """Drop-in path: GDN.forward(data, org_edge_index) with the top-k built on the GPU."""
    data, p = load_golden(case)
    m = meta(data)
    model = build_model(p, m, gpu_device)
    x = torch.from_numpy(data["x"]).to(gpu_device)
    ignored = torch.zeros((m["b"], 2, 4), device=gpu_device)      # callers pass a float tensor here
    with torch.no_grad():
        out = model(x, ignored)
    assert out.shape == (m["b"], m["n"]) and out.dtype == torch.float32
    np.testing.assert_allclose(out.cpu().numpy(), data["eval_out"], atol=TOL, rtol=0)
    layer = model.gnn_layers[0]
    if m["k"] < m["n"]:
        np.testing.assert_array_equal(model.learned_graph.cpu().numpy(), data["learned_graph"])
        np.testing.assert_array_equal(layer.edge_index_1.cpu().numpy(), data["edge_index_1"])
        np.testing.assert_allclose(layer.att_weight_1.cpu().numpy(), data["att_weight_1"], atol=TOL, rtol=0)


def random_params(n, w, k, d, seed, out_layer_num=1, inter=256):
    from gdn_amd import GDN
    torch.manual_seed(seed)
    model = GDN([torch.zeros((2, 1), dtype=torch.long)], n, dim=d, input_dim=w, topk=k, out_layer_num=out_layer_num,
                out_layer_inter_dim=inter)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        gnn = model.gnn_layers[0].gnn
        for t in (gnn.att_em_i, gnn.att_em_j, gnn.bias):
            t.copy_(torch.rand(t.shape, generator=g) * 0.2 - 0.1)
        for bn in (model.gnn_layers[0].bn, model.bn_outlayer_in):
            bn.weight.copy_(torch.rand(bn.weight.shape, generator=g) + 0.5)
            bn.bias.copy_(torch.rand(bn.bias.shape, generator=g) * 0.4 - 0.2)
            bn.running_mean.copy_(torch.randn(bn.running_mean.shape, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(bn.running_var.shape, generator=g) + 0.5)
    return model


@pytest.mark.parametrize("shape", [
    dict(b=16, n=127, w=15, k=30, d=64),      # BASELINE config 3 shape, small batch
    dict(b=128, n=64, w=15, k=64, d=64),      # BASELINE config 2 (fully connected), full size
    dict(b=3, n=512, w=30, k=64, d=64),       # BASELINE config 5 shape (big LDS tile, x chunking)
    dict(b=2, n=300, w=30, k=64, d=128),      # d = 128: two 16-lane rows per target
    dict(b=3, n=512, w=30, k=64, d=128),      # BASELINE config 5 at d=128: tile > LDS -> 2 column slices
    dict(b=5, n=33, w=5, k=1, d=32),          # k = 1: every list is {self} or {other, self}
    dict(b=700, n=27, w=5, k=5, d=64),        # more windows than resident workgroups
    dict(b=3, n=200, w=12, k=100, d=32),      # lists longer than 5 rounds: generic (recompute) variant
    dict(b=4, n=90, w=40, k=20, d=64),        # w > 32: VALU projection in w-chunks of 16
    dict(b=6, n=50, w=7, k=8, d=16),          # d = 16: VALU projection, one float per lane
], ids=lambda s: "b{b}_n{n}_w{w}_k{k}_d{d}".format(**s))
def test_seeded_shapes_against_oracle(shape, gpu_device):
    model = random_params(shape["n"], shape["w"], shape["k"], shape["d"], seed=123)
    p = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(gpu_device).eval()
    g = torch.Generator().manual_seed(9)
    x = torch.rand((shape["b"], shape["n"], shape["w"]), generator=g)
    with torch.no_grad():
        out = model(x.to(gpu_device), None)
    graph = model.learned_graph.cpu()
    ref = gdn_oracle.forward(p, x, shape["k"], graph=graph)
    np.testing.assert_allclose(out.cpu().numpy(), ref["out"].numpy(), atol=TOL, rtol=0)
    # the GPU graph must be a valid descending top-k of the oracle's cosine matrix
    cos = gdn_oracle.cosine_matrix(p["embedding.weight"])
    picked = torch.gather(cos, 1, graph)
    assert bool((picked[:, :-1] >= picked[:, 1:] - 1e-6).all())
    kth = picked[:, -1:]
    mask = torch.ones_like(cos, dtype=torch.bool).scatter_(1, graph, False)
    assert bool((cos[mask].view(shape["n"], -1) <= kth + 1e-6).all())


def test_cpu_tensors_are_refused_loudly():
    from gdn_amd import GDN
    from gdn_amd._lib import GdnHipError
    model = GDN([torch.zeros((2, 1), dtype=torch.long)], 8, dim=16, input_dim=4, topk=3).eval()
    with pytest.raises(GdnHipError):
        model(torch.rand((2, 8, 4)), None)


def test_zero_norm_embedding_row_reproduces_reference_nan_behaviour(gpu_device):
    """models/GDN.py:152 divides by the norm product without an epsilon: a zero embedding row makes
    its cosines NaN, and torch.topk ranks NaN above every number.  Reproduced, not "fixed"."""
    from gdn_amd import ops
    g = torch.Generator().manual_seed(3)
    n, d, k = 20, 32, 5
    emb = torch.randn((n, d), generator=g)
    emb[7] = 0.0
    graph = ops.topk_graph(emb.to(gpu_device), k, want_cos=True)
    cos = graph.cos.cpu()
    assert torch.isnan(cos[7]).all() and torch.isnan(cos[:, 7]).all()
    ref = gdn_oracle.learned_graph(emb, k)
    got = graph.topk.cpu()
    for i in range(n):
        if i == 7:
            continue            # an all-NaN row: torch's pick among equal keys is unspecified
        assert got[i, 0] == 7   # the NaN column ranks first
        assert got[i].tolist() == ref[i].tolist()


@pytest.mark.parametrize("n,w,k,d,t_len", [(127, 15, 30, 64, 700), (27, 5, 5, 64, 300), (40, 30, 16, 128, 200)])
def test_windows_read_from_the_raw_series_equal_materialised_windows(n, w, k, d, t_len, gpu_device):
    """x[b] = series[:, b : b+w] (datasets/TimeDataset.py:46-49, test mode) built inside
    the kernel must give bit-identical predictions to the host-built [T', N, W] tensor."""
    model = random_params(n, w, k, d, seed=11).to(gpu_device).eval()
    g = torch.Generator().manual_seed(2)
    series = torch.rand((n, t_len), generator=g).to(gpu_device)
    windows = torch.stack([series[:, i - w:i] for i in range(w, t_len)]).contiguous()    # TimeDataset.process
    with torch.no_grad():
        want = model(windows, None)
    got = model.forward_series(series, 0, t_len - w)
    assert torch.equal(got, want)
    part = model.forward_series(series, 37, 101)
    assert torch.equal(part, want[37:138])


@pytest.mark.parametrize("case", ["cfg0_msl27_w15_k20_b128", "cfg1_fc64_w15_k64_b128", "cfg2_swat127_w15_k30_b512"])
def test_baseline_configs_as_worded_at_full_batch(case, gpu_device):
    """BASELINE.json configs[0] (msl, 27 sensors, slide_win=15, k=20, batch 128), configs[1] (64 sensors fully
    connected, batch 128) and configs[2] (127 sensors, k=30, batch 512): GDN.forward at the stated batch
    against the output of the reference's own model code (fixture), graph learned on the GPU."""
    from test_oracle_golden import full_batch_input
    data, p = load_golden(case)
    m = meta(data)
    model = build_model(p, m, gpu_device)
    x = full_batch_input(data).to(gpu_device)
    with torch.no_grad():
        out = model(x, torch.zeros((m["b"], 2, 4), device=gpu_device))
    if m["k"] < m["n"]:
        np.testing.assert_array_equal(model.learned_graph.cpu().numpy(), data["learned_graph"])
    np.testing.assert_allclose(out.cpu().numpy(), data["eval_out"], atol=TOL, rtol=0)
    # the staged pipeline (what training and out_layer_num > 1 use) at the same batch
    from gdn_amd import ops
    c = model._constants()
    gnn, lin = model.gnn_layers[0].gnn, model.out_layer.mlp[0]
    xlin, s_i, s_j = ops.project_fwd(x, gnn.lin.weight, c.terms)
    z, _ = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, m["b"], want_alpha=False)
    out2, _ = ops.head_fwd(z, model.embedding.weight, c.bn1, c.bn2, lin.weight, lin.bias, m["b"])
    np.testing.assert_allclose(out2.cpu().numpy(), data["eval_out"], atol=TOL, rtol=0)


@pytest.mark.parametrize("case,raw_case", [("msl_demo_w5_k5", "msl_raw_slice"),
                                           ("cfg0_msl27_w15_k20_b128", None)])
def test_forward_series_on_the_reference_demo_series_against_golden(case, raw_case, gpu_device):
    """SURVEY §8f-1 against GOLDEN data (not HIP-vs-HIP): the windows the reference's TimeDataset builds
    from data/msl/test.csv are built inside the kernel from the raw [N, T] slice, and the predictions are
    compared with what the reference's model produced on the host-built windows."""
    data, p = load_golden(case)
    m = meta(data)
    raw = load_golden(raw_case)[0]["raw"] if raw_case else data["raw"]
    series = torch.from_numpy(raw.astype(np.float32)).to(gpu_device)
    assert series.shape == (m["n"], m["w"] + m["b"])
    model = build_model(p, m, gpu_device)
    got = model.forward_series(series, 0, m["b"])
    np.testing.assert_allclose(got.cpu().numpy(), data["eval_out"], atol=TOL, rtol=0)
    part = model.forward_series(series, 3, m["b"] - 3)
    np.testing.assert_allclose(part.cpu().numpy(), data["eval_out"][3:], atol=TOL, rtol=0)


def test_262144_window_launch_equals_small_launches(gpu_device):
    """SURVEY §8d's largest launch (xlin alone is 8.5 GB: every row offset needs 64-bit addressing): the
    staged kernels and the fused forward on 262144 windows give, bit for bit, what 512-window launches of
    the same rows give — including the last rows of the buffers."""
    from gdn_amd import ops
    b = 262144
    model = random_params(127, 15, 30, 64, seed=0).to(gpu_device).eval()
    gnn = model.gnn_layers[0].gnn
    c = model._constants()
    x = torch.rand((b, 127, 15), device=gpu_device)
    xlin, s_i, s_j = ops.project_fwd(x, gnn.lin.weight, c.terms)
    z, _ = ops.attn_aggregate_fwd(xlin, s_i, s_j, c.graph, gnn.bias, b, want_alpha=False)
    with torch.no_grad():
        fused = model(x, None)
    for s in (0, 51200, b // 2 + 37, b - 512):
        xs = x[s:s + 512].contiguous()
        with torch.no_grad():
            assert torch.equal(model(xs, None), fused[s:s + 512]), s
        xl2, si2, sj2 = ops.project_fwd(xs, gnn.lin.weight, c.terms)
        z2, _ = ops.attn_aggregate_fwd(xl2, si2, sj2, c.graph, gnn.bias, 512, want_alpha=False)
        assert torch.equal(xl2, xlin[s * 127:(s + 512) * 127]) and torch.equal(z2, z[s * 127:(s + 512) * 127]), s


@pytest.mark.parametrize("d,bf16", [(64, False), (64, True), (128, False)])
def test_large_launch_at_the_512_sensor_shape_runs_the_16_wave_kernel(d, bf16, gpu_device):
    """BASELINE configs[4] (512 sensors, top-k 64, W=30) in a launch of more than 4 windows per CU: the row-gather
    kernel then runs 16-wave workgroups (gdn_forward.hip, make_plan) — a different kernel instance from the 8-wave
    one every small test exercises.  Same per-target arithmetic, so: bit for bit what 300-window launches of the
    same rows give, the ragged tail included, and a few windows against the float64 oracle."""
    n, w, k = 512, 30, 64
    b = 4 * torch.cuda.get_device_properties(gpu_device).multi_processor_count + 37
    model = random_params(n, w, k, d, seed=31)
    p = {key: v.detach().clone() for key, v in model.state_dict().items()}
    model = model.to(gpu_device).eval()
    x = torch.rand((b, n, w), generator=torch.Generator().manual_seed(32))
    xin = (x.bfloat16() if bf16 else x).to(gpu_device)
    with torch.no_grad():
        big = model(xin, None)
        for s in (0, 300, b - 300):
            assert torch.equal(model(xin[s:s + 300].contiguous(), None), big[s:s + 300]), s
    pick = [0, 1, b // 2, b - 1]
    p64 = {key: (v.double() if v.is_floating_point() else v) for key, v in p.items()}
    ref = gdn_oracle.forward(p64, xin[pick].cpu().double(), k, graph=model.learned_graph.cpu(),
                             storage="bf16" if bf16 else "fp32")
    np.testing.assert_allclose(big[pick].cpu().double().numpy(), ref["out"].numpy(), atol=2e-4 if bf16 else 2e-6, rtol=0)


@pytest.mark.parametrize("case", MODEL_CASES)
def test_eval_forward_against_float64_oracle(case, gpu_device):
    """Accuracy rather than parity: the eval forward (fused, or staged + MLP for out_layer_num > 1) against
    the oracle run in float64 on the same parameters, inputs and sensor graph: 2e-6 (+1e-5 relative),
    i.e. 50x inside north_star's 1e-4 bar."""
    data, p = load_golden(case)
    m = meta(data)
    model = build_model(p, m, gpu_device)
    model.injected_graph = torch.from_numpy(data["learned_graph"]).to(gpu_device)
    x = torch.from_numpy(data["x"])
    with torch.no_grad():
        out = model(x.to(gpu_device), None)
    f64 = torch.float64
    p64 = {k: (v.to(f64) if v.is_floating_point() else v) for k, v in p.items()}
    ref = gdn_oracle.forward(p64, x.to(f64), m["k"], m["out_layer_num"], graph=torch.from_numpy(data["learned_graph"]))
    np.testing.assert_allclose(out.cpu().numpy().astype(np.float64), ref["out"].numpy(), atol=2e-6, rtol=1e-5)


def test_empty_minibatch_returns_empty_prediction(gpu_device):
    """Edge case: an eval forward of zero windows (a ragged last batch of size 0) gives a [0, N] tensor, as
    the reference's op sequence does on empty tensors; the sensor graph is still published."""
    model = random_params(27, 5, 5, 64, seed=1).to(gpu_device).eval()
    with torch.no_grad():
        out = model(torch.empty((0, 27, 5), device=gpu_device), None)
    assert out.shape == (0, 27) and out.dtype == torch.float32
    assert model.learned_graph.shape == (27, 5)


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_seeded_models_against_float64_oracle_with_own_graph(seed, gpu_device):
    """Light fuzz at the bench shape: random parameters (non-trivial BatchNorm statistics, att_em), the
    sensor graph learned by the HIP top-k itself.  (1) wherever the k-th / (k+1)-th cosine gap of a sensor
    exceeds 1e-6 the HIP neighbour set equals the float64 oracle's; (2) with that graph the eval forward is
    within 2e-6 (+1e-5 relative) of the float64 oracle."""
    n, w, k, d, b = 127, 15, 30, 64, 4
    model = random_params(n, w, k, d, seed=seed).to(gpu_device).eval()
    p = {key: v.detach().cpu() for key, v in model.state_dict().items()}
    x = torch.rand((b, n, w), generator=torch.Generator().manual_seed(seed))
    with torch.no_grad():
        out = model(x.to(gpu_device), None)
    graph = model.learned_graph.cpu()
    f64 = torch.float64
    p64 = {key: (v.to(f64) if v.is_floating_point() else v) for key, v in p.items()}
    cos = gdn_oracle.cosine_matrix(p64["embedding.weight"])
    srt = torch.sort(cos, dim=1, descending=True)
    clear = (srt.values[:, k - 1] - srt.values[:, k]) > 1e-6
    assert clear.float().mean() > 0.9
    want_sets = srt.indices[:, :k]
    for i in torch.nonzero(clear).flatten().tolist():
        assert set(graph[i].tolist()) == set(want_sets[i].tolist()), i
    ref = gdn_oracle.forward(p64, x.to(f64), k, graph=graph)
    np.testing.assert_allclose(out.cpu().numpy().astype(np.float64), ref["out"].numpy(), atol=2e-6, rtol=1e-5)


def test_integration_md_ctypes_stub_runs(gpu_device):
    """INTEGRATION.md §B shows the ctypes binding a reference maintainer would paste into models/GDN.py.
    Execute that very snippet (library path made absolute) as the forward of a model object and compare
    with the packaged forward: the documentation must stay runnable."""
    import os
    import re
    import types
    from gdn_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    section = text[text.index("## B."):]
    code = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    code = code.replace('ctypes.CDLL("libgdn_hip.so")', f'ctypes.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    model = random_params(51, 15, 15, 64, seed=21).to(gpu_device).eval()     # SWaT's 51 sensors
    x = torch.rand((16, 51, 15), device=gpu_device)
    with torch.no_grad():
        want = model(x, None)
        got = types.MethodType(ns["forward"], model)(x, None)
    torch.cuda.synchronize()
    # (the packaged forward goes through a plan, which reorders each lane's list slots: same numbers to rounding)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), atol=2e-7, rtol=0)
    assert torch.equal(model.learned_graph, model._constants().graph.topk)


# ---------------------------------------------------------------- inputs beyond the 16-bit operand range
def _oracle64(p, x, k, graph, layers=1):
    f64 = torch.float64
    p64 = {key: (v.to(f64) if v.is_floating_point() else v) for key, v in p.items()}
    return gdn_oracle.forward(p64, x.to(f64), k, layers, graph=graph)["out"]


def _assert_fp32_grade(got, p, x, k, graph, what=""):
    """`got` against float64, as close as fp32 arithmetic gets: at large input scales the logits are O(1e3) and
    the softmax is nearly one-hot, so an fp32 forward — the reference's own — is off by up to 1e-4 .. 1e-2 of the
    output scale on the targets whose two largest logits nearly tie.  The bound is therefore the op-faithful fp32
    oracle's own worst deviation from float64 (x4) plus the usual 2e-5 of the output scale."""
    ref = _oracle64(p, x, k, graph)
    ref32 = gdn_oracle.forward(p, x, k, 1, graph=graph)["out"].double()
    scale = max(1.0, float(ref.abs().max()))
    bound = 4.0 * float((ref32 - ref).abs().max()) + 2e-5 * scale
    err = float((got.cpu().double() - ref).abs().max())
    assert err <= bound, (what, err, bound, scale)


@pytest.mark.parametrize("scale", [1.0, 3.0e3, 1.0e5, 3.0e7])
def test_raw_unit_inputs_equal_the_float64_oracle_without_any_switch(scale, gpu_device):
    """The reference takes any fp32 input (models/graph_layer.py:56); the matrix-core kernels carry x as two f16
    terms, which end at 65504.  `model(x)` must still equal float64 on inputs in raw engineering units — no
    environment variable, no flag, no host synchronisation: the planned launch raises its range guard, the gated
    fp32 row-gather launch behind it recomputes the batch (include/gdn_hip.h "range guard").  Batches inside the
    range (scale 1, 3e3) must come from the matrix-core kernel alone, and a guard raised by one call must not
    leak into the next."""
    n, w, k, d, b = 127, 15, 30, 64, 37
    model = random_params(n, w, k, d, seed=31)
    p = {key: v.detach().clone() for key, v in model.state_dict().items()}
    model = model.to(gpu_device).eval()
    g = torch.Generator().manual_seed(32)
    x = torch.rand((b, n, w), generator=g) * scale
    x[3, 5, 2] = -x[3, 5, 2]
    with torch.no_grad():
        out = model(x.to(gpu_device), None)
        graph = model.learned_graph.cpu()
        _assert_fp32_grade(out, p, x, k, graph, what=f"scale {scale}")
        c = model._constants()
        guard = next(iter(c.guards.values()))
        assert guard.tolist() == [0, 0]                       # whatever was raised has been consumed
        limit = model.operand_limit()
        assert 1e3 < limit <= 60000.0
        assert model.input_exceeds_limit(x.to(gpu_device)) == (scale * 1.0 >= limit)
        # a normal batch right after: the guard is down again, same result as a fresh model
        x2 = torch.rand((b, n, w), generator=g)
        out2 = model(x2.to(gpu_device), None)
        np.testing.assert_allclose(out2.cpu().double().numpy(), _oracle64(p, x2, k, graph).numpy(), atol=2e-5, rtol=0)


def test_evaluator_and_training_on_raw_unit_series(gpu_device):
    """The resident-series callers look at their data ONCE: harness.SeriesEvaluator (windows or raw series) and
    harness.train pick the fp32 row-gather kernels for a series in raw units (x 1e5) and the matrix-core ones for
    the same series normalised; predictions equal float64 either way, and a training step on raw units produces
    finite, oracle-accurate gradients through the captured native step."""
    from gdn_amd import harness
    n, w, k, d, t = 27, 10, 8, 64, 300
    model = random_params(n, w, k, d, seed=33)
    p = {key: v.detach().clone() for key, v in model.state_dict().items()}
    model = model.to(gpu_device).eval()
    g = torch.Generator().manual_seed(34)
    series = torch.rand((n, t + w), generator=g)
    idx = torch.arange(t).view(-1, 1) + torch.arange(w).view(1, -1)
    for scale, expect_wide in ((1.0, False), (1.0e5, True)):
        raw = (series * scale).to(gpu_device)
        xs = (series * scale)[:, idx].permute(1, 0, 2).contiguous()
        y = raw[:, w:].t().contiguous()
        ev = harness.SeriesEvaluator(model, None, y, batch=64, use_graph=True, series=raw)
        assert ev.wide == expect_wide
        ev.step()
        with torch.no_grad():
            model(xs[:1].to(gpu_device), None)
        _assert_fp32_grade(ev.pred, p, xs, k, model.learned_graph.cpu(), what=f"series evaluator, scale {scale}")
        ev2 = harness.SeriesEvaluator(model, xs.to(gpu_device), y, batch=64, use_graph=False)
        assert ev2.wide == expect_wide
        ev2.step()
        _assert_fp32_grade(ev2.pred, p, xs, k, model.learned_graph.cpu(), what=f"window evaluator, scale {scale}")
    # training on raw units: harness.train decides from the first batch and pins the model to the wide kernels
    from _grad_check import assert_grads_close, oracle_step
    xs = (series * 1.0e5)[:, idx].permute(1, 0, 2).contiguous()[:64]
    ys = (series * 1.0e5)[:, w:].t().contiguous()[:64]
    model.train()
    model.dp.p = 0.0
    assert model.input_exceeds_limit(xs.to(gpu_device), margin=16.0)
    step = harness.GraphedTrainStep(model, 64, wide=True)
    assert isinstance(step, harness.NativeTrainStep) and step.wide
    step.x.copy_(xs.to(gpu_device)); step.y.copy_(ys.to(gpu_device))
    start = {key: v.detach().clone() for key, v in model.state_dict().items()}
    loss = float(step.step())
    got = {name: step.flat_g[off:off + cnt].view(prm.shape)
           for (name, prm), (off, cnt) in zip(model.named_parameters(), step.slices)}
    ref_loss, want, _kink = oracle_step(start, xs, ys, step.ws["topk"].cpu(), 1, None)
    assert abs(loss - ref_loss) <= 1e-5 * ref_loss
    assert all(bool(torch.isfinite(v).all()) for v in got.values())
    # (logits are O(1e5) here: the softmax is one-hot and its gradient is what survives cancellation, so fp32 —
    # any fp32 — keeps 3 digits of the attention gradients, not 5: bounds widened accordingly)
    assert_grads_close(got, want, tol_max=1e-3, tol_elem=0.5, what="raw-unit training step")


def test_range_guard_inside_a_user_captured_graph(gpu_device):
    """A caller that captures `model(x)` in a HIP graph of their own (static input buffer) gets the guarded pair of
    launches captured with it: replays on in-range and on raw-unit contents of the buffer both equal float64 — the
    flag is raised and consumed on the device, nothing was decided at capture time."""
    n, w, k, d, b = 127, 15, 30, 64, 16
    model = random_params(n, w, k, d, seed=41)
    p = {key: v.detach().clone() for key, v in model.state_dict().items()}
    model = model.to(gpu_device).eval()
    g = torch.Generator().manual_seed(42)
    static_x = torch.rand((b, n, w), generator=g).to(gpu_device)
    with torch.no_grad():
        model._constants()
        model._plan(model._constants(), False)           # constants and plan exist before the capture
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = model(static_x, None)
    graph_nbr = model.learned_graph.cpu()
    for scale in (1.0, 2.0e5, 1.0, 7.0e6):
        x = torch.rand((b, n, w), generator=g) * scale
        static_x.copy_(x.to(gpu_device))
        graph.replay()
        torch.cuda.synchronize()
        _assert_fp32_grade(static_out, p, x, k, graph_nbr, what=f"captured, scale {scale}")
