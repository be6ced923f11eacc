"""GPU suite: the whole path used the way the reference uses it (main.py:103-147): train on a normal
series, predict a test series, score it, and find the injected fault.  A functional check — the
numbers are not the reference's (synthetic data, no fixture); parity is pinned elsewhere."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

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


def _write_cli_dataset(data, root):
    import pandas as pd
    os.makedirs(os.path.join(root, "msl"), exist_ok=True)
    pd.DataFrame(data["train_raw"], columns=[str(c) for c in data["columns_train"]]).to_csv(os.path.join(root, "msl", "train.csv"))
    pd.DataFrame(data["test_raw"], columns=[str(c) for c in data["columns_test"]]).to_csv(os.path.join(root, "msl", "test.csv"))
    with open(os.path.join(root, "msl", "list.txt"), "w") as f:
        f.write("\n".join(str(c) for c in data["features"]) + "\n")


@pytest.mark.parametrize("report", ["best", "val"])
def test_command_line_reproduces_the_reference_report(report, gpu_device, tmp_path, capsys):
    """`python -m gdn_amd.main` with the reference's flags on a slice of the reference's demo data
    and a checkpoint written by the reference's model: same validation block (same RNG draws), same predictions,
    same printed F1 / precision / recall as the reference's main.py (fixture: tests/golden/make_golden.py cli_case)."""
    from gdn_amd import main as cli
    data, p = load_golden("cli_msl_slice")
    batch, w, dim, stride, topk, seed, inter = (int(v) for v in data["meta_cfg"])
    root = str(tmp_path / "data")
    _write_cli_dataset(data, root)
    ckpt = str(tmp_path / "ckpt.pt")
    torch.save(p, ckpt)
    argv = ["-dataset", "msl", "-data_root", root, "-device", "cuda", "-batch", str(batch), "-slide_win", str(w),
            "-dim", str(dim), "-slide_stride", str(stride), "-topk", str(topk), "-random_seed", str(seed),
            "-out_layer_inter_dim", str(inter), "-val_ratio", str(float(data["val_ratio"])), "-report", report,
            "-load_model_path", ckpt]
    args = cli.build_parser().parse_args(argv)
    import random
    random.seed(args.random_seed)
    torch.manual_seed(args.random_seed)
    m = cli.Main({"batch": batch, "epoch": 1, "slide_win": w, "dim": dim, "slide_stride": stride, "comment": "",
                  "seed": seed, "out_layer_num": 1, "out_layer_inter_dim": inter, "decay": 0,
                  "val_ratio": float(data["val_ratio"]), "topk": topk},
                 {"save_path": "msl", "dataset": "msl", "report": report, "device": "cuda", "load_model_path": ckpt,
                  "data_root": root})
    val_idx = m.val_dataloader.loader.dataset.tensors[0]
    np.testing.assert_array_equal(val_idx.numpy(), data["val_indices"])          # the reference's random block
    info = m.run()
    printed = capsys.readouterr().out
    assert "F1 score:" in printed and "precision:" in printed and "recall:" in printed
    np.testing.assert_allclose(m.test_result[0].cpu().numpy(), data["test_pred"], atol=2e-5, rtol=0)
    want = data["info_" + report]
    # rank-based thresholds: a 1e-7 difference in two near-equal scores can move one tick across the threshold
    np.testing.assert_allclose(info[:3], want[:3], atol=5e-3, rtol=0)
    np.testing.assert_allclose(info[3], want[3], atol=2e-3, rtol=0)              # AUC
    np.testing.assert_allclose(info[4], want[4], rtol=1e-4)                      # threshold
    # and the whole command line through main(): prints the same three lines
    cli.main(argv)
    out2 = capsys.readouterr().out
    f1 = float([ln for ln in out2.splitlines() if ln.startswith("F1 score:")][0].split(":")[1])
    assert abs(f1 - info[0]) < 1e-12


def test_command_line_trains_and_reports(gpu_device, tmp_path, capsys, monkeypatch):
    """The training branch of the command line (no -load_model_path): 2 epochs on the demo slice with the native
    HIP-graph step, best-validation checkpoint written under ./pretrained/<pattern>/, report printed."""
    from gdn_amd import main as cli
    data, _p = load_golden("cli_msl_slice")
    root = str(tmp_path / "data")
    _write_cli_dataset(data, root)
    monkeypatch.chdir(tmp_path)
    info = cli.main(["-dataset", "msl", "-data_root", root, "-batch", "32", "-slide_win", "5", "-dim", "64", "-slide_stride", "1",
                     "-topk", "5", "-random_seed", "5", "-epoch", "2", "-val_ratio", "0.2", "-save_path_pattern", "msl",
                     "-out_layer_inter_dim", "128"])
    assert 0.0 <= info[0] <= 1.0 and len(os.listdir(tmp_path / "pretrained" / "msl")) == 1
    assert "F1 score:" in capsys.readouterr().out


@pytest.mark.parametrize("units", ["normalised", "raw"])
def test_command_line_at_the_512_sensor_stress_shape(units, gpu_device, tmp_path, capsys, monkeypatch):
    """BASELINE configs[4]'s width through the command line (HIP graph on by default: round 2 raised
    GDN_ERR_UNSUPPORTED at the first batch): a synthetic 512-sensor series, top-k 64, W = 30, one epoch of the
    native captured step, best-validation checkpoint, report printed.  `raw`: the same series in engineering units
    (x 3e3: the reference's train() only checkpoints below a loss of 1e8, train.py:47) — the command line looks at its
    resident series once and trains / evaluates on the fp32 row-gather kernels (include/gdn_hip.h "range guard")."""
    import pandas as pd
    from gdn_amd import main as cli
    n, t_train, t_test = 512, 230, 120
    rng = np.random.default_rng(3)
    phase = rng.uniform(0, 6.28, size=n)
    scale = 3.0e3 if units == "raw" else 1.0
    def series(t0, t):
        tt = np.arange(t0, t0 + t)[:, None]
        return (0.5 + 0.4 * np.sin(0.07 * tt + phase[None, :]) + 0.02 * rng.standard_normal((t, n))) * scale
    cols = [f"s{i}" for i in range(n)]
    root = tmp_path / "data" / "wadi512"
    os.makedirs(root)
    pd.DataFrame(series(0, t_train), columns=cols).to_csv(root / "train.csv")
    test = pd.DataFrame(series(t_train, t_test), columns=cols)
    attack = np.zeros(t_test, dtype=int)
    attack[60:80] = 1
    test.iloc[60:80, :8] += 0.8 * scale
    test["attack"] = attack
    test.to_csv(root / "test.csv")
    (root / "list.txt").write_text("\n".join(cols) + "\n")
    monkeypatch.chdir(tmp_path)
    from gdn_amd import harness
    seen = []
    real = harness.GraphedTrainStep
    monkeypatch.setattr(harness, "GraphedTrainStep", lambda *a, **k: (seen.append(k.get("wide")), real(*a, **k))[1])
    info = cli.main(["-dataset", "wadi512", "-data_root", str(tmp_path / "data"), "-batch", "32", "-slide_win", "30",
                     "-dim", "64", "-slide_stride", "1", "-topk", "64", "-random_seed", "5", "-epoch", "1",
                     "-val_ratio", "0.2", "-save_path_pattern", "wadi512"])
    assert all(np.isfinite(v) for v in info[:3]) and 0.0 <= info[0] <= 1.0
    assert len(os.listdir(tmp_path / "pretrained" / "wadi512")) == 1
    assert "F1 score:" in capsys.readouterr().out
    assert seen == [units == "raw"]              # the captured native step ran, on the kernels the data's range asks for
