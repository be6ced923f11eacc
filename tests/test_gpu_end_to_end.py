"""GPU suite: the whole path used the way the reference uses it (main.py:103-147): train on a normal
series, predict a test series, score it, and find the injected fault.  A functional check — the
numbers are not the reference's (synthetic data, no fixture); parity is pinned elsewhere."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N, W, K, D = 12, 8, 4, 32


def _series(t, seed, fault=None):
    """Three groups of phase-coupled sensors + small noise; `fault` = (sensor, start, stop) decouples one."""
    g = np.random.default_rng(seed)
    base = np.arange(t)[None, :] * (2 * np.pi / np.array([37.0, 53.0, 71.0]))[:, None]
    grp = np.repeat(np.arange(3), N // 3)
    phase = g.uniform(0, 0.3, size=N)[:, None]
    x = 0.5 + 0.4 * np.sin(base[grp] + phase) + 0.01 * g.standard_normal((N, t))
    if fault is not None:
        s, a, b = fault
        x[s, a:b] = 1.6 + 0.02 * g.standard_normal(b - a)       # sensor stuck far outside its range
    return x.astype(np.float32)


def _windows(series):
    """datasets/TimeDataset.py:42-58 with stride 1: x[b] = series[:, b:b+W], y[b] = series[:, b+W]."""
    t = series.shape[1] - W
    idx = np.arange(t)[:, None] + np.arange(W)[None, :]
    return torch.from_numpy(series[:, idx].transpose(1, 0, 2).copy()), torch.from_numpy(series[:, W:].T.copy())


def test_train_eval_score_finds_injected_fault(gpu_device):
    from gdn_amd import GDN, harness
    from gdn_amd.evaluate import get_best_performance_data, get_full_err_scores, get_top1_anomaly
    torch.manual_seed(0)
    model = GDN([torch.zeros((2, 1), dtype=torch.long)], N, dim=D, input_dim=W, topk=K).to(gpu_device)
    xtr, ytr = _windows(_series(2048 + W, seed=1))
    loader = [(xtr[s:s + 128], ytr[s:s + 128], torch.zeros(len(xtr[s:s + 128])), None)
              for s in range(0, len(xtr), 128)]
    losses = harness.train(model, "", {"epoch": 20}, loader, None, use_graph=True)
    assert np.mean(losses[-16:]) < 0.25 * np.mean(losses[:16])          # it learns the coupling

    fault = (5, 600, 680)
    xte, yte = _windows(_series(1024 + W, seed=2, fault=fault))
    labels = torch.zeros(len(xte))
    labels[fault[1] - W:fault[2] - W] = 1
    tloader = [(xte[s:s + 256], yte[s:s + 256], labels[s:s + 256], None) for s in range(0, len(xte), 256)]
    _loss, result = harness.test(model, tloader)
    scores, _ = get_full_err_scores(result)
    _loss_t, result_t = harness.test(model, tloader, as_tensors=True)          # device tensors, no .tolist()
    scores_t, _ = get_full_err_scores(result_t)
    assert _loss_t == _loss and np.array_equal(scores_t, scores)
    assert scores.shape == (N, len(xte))
    anomaly = get_top1_anomaly(scores)
    lab = labels.numpy().astype(bool)
    # ticks inside the fault carry clearly higher scores than normal ticks, and most of them clear the
    # 99th percentile of the normal ticks (the sensor that peaks need not be the faulty one: its
    # neighbours' predictions are corrupted through the attention, as in the reference)
    assert np.median(anomaly[lab]) > 3.0 * np.median(anomaly[~lab])
    assert (anomaly[lab] > np.quantile(anomaly[~lab], 0.99)).mean() > 0.5

    # the reference's report (main.py:139-147 -> evaluate.py:129-158): best-F1 threshold sweep, AUC
    f1, pre, rec, auc, thr = get_best_performance_data(scores, labels.numpy(), topk=1)
    assert f1 > 0.6 and auc > 0.9 and 0 < pre <= 1 and 0 < rec <= 1 and thr > np.median(anomaly)

    # the resident-series evaluator gives the same anomaly score as the loop above
    ev = harness.SeriesEvaluator(model, xte.to(gpu_device), yte.to(gpu_device), batch=256, use_graph=True)
    np.testing.assert_allclose(ev.step().cpu().numpy(), anomaly, rtol=1e-9, atol=1e-9)
