"""GPU suite: device scoring against the reference's evaluate.py outputs (golden) and against the
oracle at full size; eval loop (test.py mirror) against the golden eval-loop fixture."""
import numpy as np
import pytest
import torch

from conftest import SCORE_CASES, load_golden, meta
from oracle import score_oracle
from test_gpu_forward_parity import build_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", SCORE_CASES)
def test_scores_match_reference_evaluate(case, gpu_device):
    from gdn_amd import evaluate
    data, _ = load_golden(case)
    scores, anomaly, med_iqr = evaluate.anomaly_scores(data["pred"], data["gt"], device=gpu_device)
    np.testing.assert_allclose(med_iqr.cpu().numpy(), data["med_iqr"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(scores.cpu().numpy(), data["scores"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(anomaly.cpu().numpy(), data["scores"].max(axis=0), rtol=1e-12, atol=1e-13)
    # the reference-named entry points
    s2, _ = evaluate.get_full_err_scores([data["pred"].tolist(), data["gt"].tolist(), None], device=gpu_device)
    np.testing.assert_allclose(s2, data["scores"], rtol=1e-12, atol=1e-13)
    one = evaluate.get_err_scores((data["pred"][:, 0].tolist(), data["gt"][:, 0].tolist()), device=gpu_device)
    np.testing.assert_allclose(one, data["scores"][0], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("t,n", [(32768, 127), (40001, 5), (8192, 9), (4097, 51), (4, 3), (3, 2), (1, 1)])
def test_scores_full_size_against_oracle(t, n, gpu_device):
    from gdn_amd import evaluate
    g = torch.Generator().manual_seed(t + n)
    pred = torch.rand((t, n), generator=g)
    gt = (pred + 0.05 * torch.randn((t, n), generator=g)).float()
    gt[:, 0] = pred[:, 0]                     # a sensor with zero error everywhere: IQR = 0 -> eps only
    if t > 100:
        gt[50:60, 1] = pred[50:60, 1] + 3.0   # ties and a burst
    scores, anomaly, med_iqr = evaluate.anomaly_scores(pred, gt, device=gpu_device)
    want = score_oracle.full_err_scores(pred.numpy(), gt.numpy()) if t <= 5000 else None
    for i in sorted({0, min(1, n - 1), n - 1}):
        med, rng = score_oracle.err_median_and_iqr(pred[:, i].numpy(), gt[:, i].numpy())
        np.testing.assert_allclose(med_iqr[i].cpu().numpy(), [med, rng], rtol=1e-14, atol=0)
    if want is not None:
        np.testing.assert_allclose(scores.cpu().numpy(), want, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(anomaly.cpu().numpy(), want.max(axis=0), rtol=1e-12, atol=1e-13)
    else:
        # size-independent properties at BASELINE's T: first 3 ticks are 0, anomaly is the max over
        # sensors, and a few sensors re-scored by the oracle agree
        s = scores.cpu().numpy()
        assert (s[:, :3] == 0).all()
        np.testing.assert_array_equal(anomaly.cpu().numpy(), s.max(axis=0))
        for i in (0, 1, n // 2):
            np.testing.assert_allclose(s[i], score_oracle.err_scores(pred[:, i].numpy(), gt[:, i].numpy()),
                                       rtol=1e-12, atol=1e-13)


def test_eval_loop_matches_reference_test_py(gpu_device):
    """harness.test() on the batches the reference's test() was given."""
    from gdn_amd import evaluate, harness
    data, p = load_golden("eval_loop_msl_shape")
    m = meta(data)
    model = build_model(p, m, gpu_device)
    x, y, lab = (torch.from_numpy(data[k]) for k in ("x", "y", "labels"))
    bsz = m["b"]
    batches = [(x[s:s + bsz].double(), y[s:s + bsz].double(), lab[s:s + bsz].double(), torch.zeros((bsz, 2, 4)))
               for s in range(0, x.shape[0], bsz)]
    avg_loss, (pred, gt, labels) = harness.test(model, batches)
    np.testing.assert_allclose(np.array(pred), data["pred"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(np.array(gt), data["gt"], atol=0, rtol=0)
    np.testing.assert_allclose(np.array(labels), data["lab"], atol=0, rtol=0)
    np.testing.assert_allclose(avg_loss, float(data["avg_loss"]), atol=2e-6)
    # scoring the REFERENCE's predictions reproduces the reference's scores to float64 round-off
    scores, _ = evaluate.get_full_err_scores([data["pred"], data["gt"], None], device=gpu_device)
    np.testing.assert_allclose(scores, data["scores"], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("use_graph", [False, True])
def test_series_evaluator_equals_per_batch_loop(use_graph, gpu_device):
    from gdn_amd import evaluate, harness
    from test_gpu_forward_parity import random_params
    model = random_params(127, 15, 30, 64, seed=5).to(gpu_device).eval()
    g = torch.Generator().manual_seed(1)
    t = 1000                                          # not a multiple of the batch: ragged last launch
    x = torch.rand((t, 127, 15), generator=g).to(gpu_device)
    y = torch.rand((t, 127), generator=g).to(gpu_device)
    ev = harness.SeriesEvaluator(model, x, y, batch=128, use_graph=use_graph, want_scores=True)
    a1 = ev.step().clone()
    a2 = ev.step().clone()                             # replay must be idempotent
    torch.cuda.synchronize()
    assert torch.equal(a1, a2)
    with torch.no_grad():
        pred = torch.cat([model(x[s:s + 128], None) for s in range(0, t, 128)])
    assert torch.equal(pred, ev.pred)
    _, anomaly, _ = evaluate.anomaly_scores(pred, y, device=gpu_device)
    assert torch.equal(anomaly, a1)


@pytest.mark.parametrize("t,n,cuts", [(1000, 27, (333, 700)), (40, 5, (1, 2, 3, 5, 38)), (4100, 130, (2049,))])
def test_sharded_smoothing_with_halo_equals_whole_series(t, n, cuts, gpu_device):
    """The per-rank half of harness.distributed_anomaly on the GPU (first_tick > 0, 3-row halo that can
    span several short shards), emulated in one process: shard-wise smooth/max == whole series."""
    from gdn_amd import ops
    g = torch.Generator().manual_seed(t)
    pred = torch.rand((t, n), generator=g).to(gpu_device)
    gt = (pred + 0.1 * torch.randn((t, n), generator=g).to(gpu_device)).contiguous()
    med_iqr = ops.score_quantiles(pred, gt)
    _, whole = ops.score_smooth_max(pred, gt, med_iqr, want_scores=False)
    bounds = [0, *cuts, t]
    parts = []
    for s0, s1 in zip(bounds[:-1], bounds[1:]):
        hp = hg = None
        if s0 > 0:
            lo = max(0, s0 - 3)
            hp = torch.zeros((3, n), device=gpu_device)
            hg = torch.zeros((3, n), device=gpu_device)
            hp[3 - (s0 - lo):] = pred[lo:s0]
            hg[3 - (s0 - lo):] = gt[lo:s0]
        sc, an = ops.score_smooth_max(pred[s0:s1].contiguous(), gt[s0:s1].contiguous(), med_iqr,
                                      want_scores=True, first_tick=s0, halo_pred=hp, halo_gt=hg)
        parts.append(an)
        assert torch.equal(sc.max(dim=0).values, an)
    assert torch.equal(torch.cat(parts), whole)


def test_series_evaluator_from_raw_series_equals_window_tensor(gpu_device):
    """SURVEY §8f-1 through the evaluator: windows built in-kernel from the raw [N, T] series give the
    same predictions and anomaly scores as the host-built [T', N, W] tensor."""
    from gdn_amd import harness
    from test_gpu_forward_parity import random_params
    n, w = 127, 15
    model = random_params(n, w, 30, 64, seed=5).to(gpu_device).eval()
    g = torch.Generator().manual_seed(4)
    raw = torch.rand((n, 1200), generator=g).to(gpu_device)
    x = torch.stack([raw[:, i - w:i] for i in range(w, raw.shape[1])]).contiguous()     # TimeDataset.process
    y = raw[:, w:].t().contiguous()
    a = harness.SeriesEvaluator(model, x, y, batch=256, coalesce=2, use_graph=True)
    b = harness.SeriesEvaluator(model, None, y, batch=256, coalesce=2, use_graph=True, series=raw)
    ra, rb = a.step().clone(), b.step().clone()
    assert torch.equal(a.pred, b.pred) and torch.equal(ra, rb)


def test_captured_graph_is_dropped_when_a_parameter_changes(gpu_device):
    from gdn_amd import harness
    from test_gpu_forward_parity import random_params
    model = random_params(27, 5, 5, 64, seed=9).to(gpu_device).eval()
    g = torch.Generator().manual_seed(6)
    x = torch.rand((300, 27, 5), generator=g).to(gpu_device)
    y = torch.rand((300, 27), generator=g).to(gpu_device)
    ev = harness.SeriesEvaluator(model, x, y, batch=100, use_graph=True)
    ev.step()
    before = ev.pred.clone()
    with torch.no_grad():
        model.out_layer.mlp[0].bias.add_(1.0)          # in-place update, as an optimizer would do
    ev.step()
    torch.cuda.synchronize()
    assert torch.allclose(ev.pred, before + 1.0, atol=1e-6)


@pytest.mark.parametrize("t,n,ranks", [(5000, 23, 3), (32768 * 2, 16, 2), (2500, 7, 4),
                                        (32768 * 8, 127, 8)])      # the shape of an 8-rank bench step: ~24 k survivors per sensor
def test_blocked_key_select_equals_whole_series_quantiles(t, n, ranks, gpu_device):
    """The owner-side half of the multi-GPU exchange on one GPU: keys built shard by shard (padded
    pitch, filler slots), laid out [rank, sensor, pitch] as all_to_all_single delivers them, selected
    in place — must equal the single-shot median / IQR bit for bit."""
    from gdn_amd import harness, ops
    g = torch.Generator().manual_seed(t)
    pred = torch.rand((t, n), generator=g).to(gpu_device)
    gt = (pred + 0.1 * torch.randn((t, n), generator=g).to(gpu_device)).contiguous()
    want = ops.score_quantiles(pred, gt)
    bounds = [harness.shard_range(t, r, ranks) for r in range(ranks)]
    longest = max(e - s for s, e in bounds)
    pitch = (longest + harness.KEY_SLICE - 1) // harness.KEY_SLICE * harness.KEY_SLICE
    blocks = [ops.score_keys(pred[s:e].contiguous(), gt[s:e].contiguous(), pitch) for s, e in bounds]
    for a, b in [harness.sensor_range(n, r, ranks) for r in range(ranks)]:
        if b == a:
            continue
        recv = torch.stack([blk[a:b] for blk in blocks]).contiguous()        # [ranks, sensors, pitch]
        keep = recv.clone()
        got = ops.score_select(recv.reshape(-1), ranks, b - a, pitch, t)
        assert torch.equal(got, want[a:b])
        assert torch.equal(recv.view(torch.int64), keep.view(torch.int64))   # the input is not modified


def test_evaluator_replays_back_to_back_at_bench_size(gpu_device):
    """What bench.py's timed region does — graph replays with no host sync in between, 4 side streams,
    8 minibatches per launch, T=32768 — must end with the same predictions and anomaly scores as the
    un-graphed, per-launch path."""
    from gdn_amd import harness
    from test_gpu_forward_parity import random_params
    model = random_params(127, 15, 30, 64, seed=2).to(gpu_device).eval()
    g = torch.Generator().manual_seed(8)
    t = 32768
    x = torch.rand((t, 127, 15), generator=g).to(gpu_device)
    y = torch.rand((t, 127), generator=g).to(gpu_device)
    ref = harness.SeriesEvaluator(model, x, y, batch=512, coalesce=1, use_graph=False, streams=1)
    want = ref.step().clone()
    ev = harness.SeriesEvaluator(model, x, y, batch=512, coalesce=8, use_graph=True, streams=4)
    for _ in range(6):
        ev.step()                                          # no synchronisation between replays
    torch.cuda.synchronize()
    assert torch.equal(ev.pred, ref.pred)
    assert torch.equal(ev.anomaly, want)


def test_sharded_evaluator_one_rank_rccl_equals_series_evaluator(gpu_device):
    """harness.ShardedEvaluator (the N>1 eval step: per-chunk async all-to-all of the radix keys, blocked
    select over chunks*ranks row blocks, one all-gather) run with a 1-rank RCCL group on this GPU: same
    predictions and anomaly scores as the single-GPU evaluator, also across repeated steps, eagerly and with
    the compute segments captured in HIP graphs (the collectives stay eager between the replays).  The
    multi-rank exchange logic itself is covered with gloo in tests/test_cpu_distributed.py."""
    import socket
    import torch.distributed as dist
    from gdn_amd import harness
    from test_gpu_forward_parity import random_params
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=gpu_device)
    try:
        model = random_params(127, 15, 30, 64, seed=3).to(gpu_device).eval()
        g = torch.Generator().manual_seed(12)
        t = 9000                                           # 3 chunks of 4096 slots, the last partly filled
        x = torch.rand((t, 127, 15), generator=g).to(gpu_device)
        y = torch.rand((t, 127), generator=g).to(gpu_device)
        ref = harness.SeriesEvaluator(model, x, y, batch=512, use_graph=False, streams=1)
        want = ref.step().clone()
        for use_graph in (False, True):
            sev = harness.ShardedEvaluator(model, x, y, t, chunk=4096, use_graph=use_graph)
            assert sev.nchunks == 3 and sev.pitch == 4096
            for _ in range(4):                  # (use_graph: eager, capture + replay, replay, replay)
                got = sev.step()
            torch.cuda.synchronize()
            # the compute segments replay from HIP graphs around the eager collectives
            assert (sev._graphs is not None) == use_graph
            assert torch.equal(sev.pred, ref.pred)
            assert torch.equal(got, want)
            sev.pred.zero_()                    # a replay really recomputes
            got = sev.step()
            torch.cuda.synchronize()
            assert torch.equal(sev.pred, ref.pred) and torch.equal(got, want)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["perf_T1000_N27", "perf_T777_N5_ties"])
def test_threshold_sweep_matches_reference_evaluate(case, gpu_device):
    """SURVEY §8f-4 on the device: gdn_amd.evaluate.get_best_performance_data / get_val_performance_data /
    eval_scores against what the reference's own functions (scipy rankdata, sklearn metrics) returned."""
    import os
    from gdn_amd import evaluate
    data = np.load(os.path.join(os.path.dirname(__file__), "golden", case + ".npz"))
    scores, labels = data["scores"], data["labels"]
    dev_scores, _ = evaluate.get_full_err_scores([data["pred"], data["gt"], None], device=gpu_device)
    np.testing.assert_allclose(dev_scores, scores, rtol=1e-12, atol=1e-13)
    for topk in (1, 3):
        total = np.sort(scores, axis=0)[scores.shape[0] - topk:].sum(axis=0)
        fmeas, ths = evaluate.eval_scores(total.tolist(), labels.tolist(), 400, return_thresold=True, device=gpu_device)
        np.testing.assert_allclose(fmeas, data[f"fmeas_top{topk}"], rtol=1e-13, atol=0)
        np.testing.assert_allclose(ths, data[f"thresholds_top{topk}"], rtol=1e-15, atol=0)
        best = evaluate.get_best_performance_data(scores, labels.tolist(), topk=topk, device=gpu_device)
        np.testing.assert_allclose(best, data[f"best_top{topk}"], rtol=1e-12, atol=0)
        val = evaluate.get_val_performance_data(scores, scores[:, : scores.shape[1] // 6], labels.tolist(), topk=topk,
                                                device=gpu_device)
        np.testing.assert_allclose(val, data[f"val_top{topk}"], rtol=1e-12, atol=0)


def test_threshold_sweep_full_size_against_oracle(gpu_device):
    """The sweep at T=32768 (the bench's series) against the oracle's numpy restatement, ties included."""
    from gdn_amd import evaluate
    from oracle import score_oracle
    g = np.random.default_rng(3)
    t, n = 32768, 127
    scores = np.round(g.gamma(2.0, 0.6, size=(n, t)), 3)                      # rounded: many tied scores
    labels = np.zeros(t)
    for a in range(2000, t, 4096):
        labels[a:a + 150] = 1
        scores[:7, a:a + 150] += 6.0
    for topk in (1, 2):
        got = evaluate.get_best_performance_data(scores, labels, topk=topk, device=gpu_device)
        np.testing.assert_allclose(got, score_oracle.best_performance(scores, labels, topk), rtol=1e-12, atol=0)


@pytest.mark.parametrize("t", [32768, 20000, 5000, 40000, 262144])
def test_quantile_select_on_adversarial_error_distributions(t, gpu_device):
    """Every sensor's median / IQR must equal numpy's bit for bit whatever the shape of the error distribution:
    the one-workgroup select (t <= 32768) decides digit 0 by COUNTING around the top byte of tick 0's key and
    falls back to histograms when the ranks leave that bin, compacts by 16- and 24-bit prefixes and falls back
    when the compaction does not fit; the multi-launch select (t > 32768) has its own hand-offs."""
    from gdn_amd import evaluate
    g = torch.Generator().manual_seed(t)
    n = 12
    pred = torch.zeros((t, n))
    err = torch.empty((t, n), dtype=torch.float64)
    err[:, 0] = torch.exp(12.0 * torch.randn((t,), generator=g, dtype=torch.float64)).clamp(1e-30, 1e30)   # ~100 binades: ranks in different top bytes
    err[:, 1] = torch.rand((t,), generator=g, dtype=torch.float64) * 1e-3
    err[0, 1] = 5e4                                                                       # tick 0 is an outlier: wrong reference bin
    err[:, 2] = 0.25                                                                      # constant: every key equal
    err[:, 3] = torch.randint(0, 3, (t,), generator=g).double() * 0.5                     # three values: huge tie groups
    err[:, 4] = torch.rand((t,), generator=g, dtype=torch.float64) * 2.0 ** -20 + 1.0     # all keys share 20+ leading bits
    err[:, 5] = torch.where(torch.rand((t,), generator=g) < 0.5, 1e-30, 1e30).double()    # two far-apart clusters
    err[:, 6] = 0.0
    err[: t // 4, 6] = 7.0                                                                # q75 on the boundary of a tie group
    err[:, 7] = torch.rand((t,), generator=g, dtype=torch.float64)
    err[:, 8] = torch.rand((t,), generator=g, dtype=torch.float64) ** 8                   # mass near zero, many exponents
    err[:, 9] = torch.arange(t, dtype=torch.float64) * 1e-6                               # sorted input
    err[:, 10] = torch.arange(t, 0, -1, dtype=torch.float64)                              # reverse sorted
    err[:, 11] = torch.rand((t,), generator=g, dtype=torch.float64) * 1e-38               # fp32 denormal range
    gt = err.float()                  # pred = 0, gt >= 0  ->  |pred - gt| = gt exactly (float32 values)
    _scores, _anomaly, med_iqr = evaluate.anomaly_scores(pred, gt, device=gpu_device)
    for i in range(n):
        med, rng = score_oracle.err_median_and_iqr(pred[:, i].numpy(), gt[:, i].numpy())
        np.testing.assert_array_equal(med_iqr[i].cpu().numpy(), np.array([med, rng]), err_msg=f"sensor {i}")


@pytest.mark.parametrize("mode", ["windows", "raw_series", "bf16", "coalesced"])
def test_scoring_keys_written_by_the_forward_equal_the_keys_kernel(mode, gpu_device):
    """gdn_forward_fused_plan_keys / _series_plan_keys: the forward's epilogue leaves |pred - y| as float64
    radix keys (what gdn_score_keys computes in a launch of its own) — same predictions, the same key block bit
    for bit, the same anomaly scores, for ragged multi-launch series, the raw-series form and bf16 windows."""
    from gdn_amd import harness, ops
    from test_gpu_forward_parity import random_params
    model = random_params(127, 15, 30, 64, seed=8).to(gpu_device).eval()
    g = torch.Generator().manual_seed(4)
    t, w = 1500, 15
    raw = torch.rand((127, t + w), generator=g).to(gpu_device)
    y = raw[:, w:].t().contiguous()
    x = raw.unfold(1, w, 1)[:, :t].permute(1, 0, 2).contiguous()          # x[b] = raw[:, b : b+w]
    kw = dict(batch=256, use_graph=True, coalesce=4 if mode == "coalesced" else 1)
    def make():
        if mode == "raw_series":
            return harness.SeriesEvaluator(model, None, y, series=raw, **kw)
        return harness.SeriesEvaluator(model, x.bfloat16() if mode == "bf16" else x, y, **kw)
    fused, plain = make(), make()
    fused.fuse_keys, plain.fuse_keys = True, False          # (off by default: measured slower, harness.py)
    assert model.fused_keys_supported(mode == "bf16")
    a_f, a_p = fused.step().clone(), plain.step().clone()
    torch.cuda.synchronize()
    assert torch.equal(fused.pred, plain.pred)
    keys_f = fused.ws[: 127 * t].view(127, t)
    assert torch.equal(keys_f, ops.score_keys(fused.pred, y, t))
    assert torch.equal(keys_f, plain.ws[: 127 * t].view(127, t))
    assert torch.equal(fused.med_iqr, plain.med_iqr) and torch.equal(a_f, a_p)
    assert torch.equal(fused.step(), a_f)                                  # replay is idempotent
