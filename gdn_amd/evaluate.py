"""Host-side mirror of the scoring half of the reference's evaluate.py (the part on the
hot path: `get_err_scores`, `get_full_err_scores`, and the max over sensors inside
`get_best_performance_data`).  Same function names and argument meaning; the arithmetic runs
in float64 HIP kernels (gdn_amd/csrc/gdn_score.hip).  The reporting half — threshold sweep, F1,
precision, recall, AUC (evaluate.py:99-158, util/data.py:28-51) — runs on the device too, as
float64 torch ops (one sort, prefix sums; five numbers come back), with no sklearn/scipy."""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def _to_device_tn(a, device) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.asarray(a, dtype=np.float32), device=device).contiguous()


def anomaly_scores(pred, gt, want_scores: bool = True, device="cuda"):
    """Device-native entry: pred, gt [T, N] -> (scores [N, T] float64 | None, anomaly [T] float64,
    med_iqr [N, 2]).  `scores` = evaluate.py:48-68 per sensor; `anomaly` = max over sensors."""
    pred = _to_device_tn(pred, device)
    gt = _to_device_tn(gt, device)
    med_iqr = ops.score_quantiles(pred, gt)
    scores, anomaly = ops.score_smooth_max(pred, gt, med_iqr, want_scores=want_scores)
    return scores, anomaly, med_iqr


def get_err_scores(test_res, val_res=None, device="cuda") -> np.ndarray:
    """evaluate.py:48-68 for ONE sensor: test_res = (predicted, ground_truth) sequences."""
    pred, gt = test_res
    p = _to_device_tn(pred, device).view(-1, 1)
    g = _to_device_tn(gt, device).view(-1, 1)
    scores, _, _ = anomaly_scores(p, g, device=device)
    return scores[0].cpu().numpy()


def get_full_err_scores(test_result, val_result=None, device="cuda"):
    """evaluate.py:6-36.  test_result = [predictions, ground truth, labels], each [T][N] (the
    structure test() returns).  Returns (all_scores [N,T], all_normals [N,T'] or None) as float64
    numpy arrays like the reference."""
    scores, _, _ = anomaly_scores(test_result[0], test_result[1], device=device)
    normals = None
    if val_result is not None:
        normals, _, _ = anomaly_scores(val_result[0], val_result[1], device=device)
        normals = normals.cpu().numpy()
    return scores.cpu().numpy(), normals


def get_top1_anomaly(total_err_scores) -> np.ndarray:
    """The topk=1 reduction at evaluate.py:131-139: max over sensors at each tick."""
    return np.max(np.asarray(total_err_scores), axis=0)


# ---------------------------------------------------------------------------- threshold sweep / F1
TH_STEPS = 400      # evaluate.py:146


def _to_device_f64(a, device) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float64)
    return torch.as_tensor(np.asarray(a, dtype=np.float64), device=device)


def _topk_total(scores_nt: torch.Tensor, topk: int) -> torch.Tensor:
    """evaluate.py:131-139: per tick, the sum of the `topk` largest sensor scores, added smallest first
    (the order np.sum sees after argpartition)."""
    if topk == 1:
        return scores_nt.max(dim=0).values
    vals = torch.topk(scores_nt, topk, dim=0).values.flip(0)
    acc = vals[0]
    for r in range(1, topk):
        acc = acc + vals[r]
    return acc


def _f1(tp, fp, fn):
    den = 2.0 * tp + fp + fn
    return torch.where(den > 0, 2.0 * tp / torch.where(den > 0, den, torch.ones_like(den)), torch.zeros_like(den))


def _sweep(total: torch.Tensor, lab: torch.Tensor, th_steps: int):
    """util/data.py:28-51 on the device: ordinal ranks = one stable sort; the F1 of `rank > i/th_steps*T`
    for every i from a suffix sum of the labels in rank order."""
    t = total.numel()
    order = torch.argsort(total, stable=True)
    lab_sorted = lab[order].to(torch.int64)
    suffix = torch.cat([lab_sorted.flip(0).cumsum(0).flip(0), lab_sorted.new_zeros(1)])
    v = (torch.arange(th_steps, dtype=torch.float64, device=total.device) * 1.0 / th_steps) * t
    cut = v.floor().to(torch.int64)
    tp = suffix[cut].to(torch.float64)
    n_pred = (t - cut).to(torch.float64)
    n_pos = lab.sum().to(torch.float64)
    fmeas = _f1(tp, n_pred - tp, n_pos - tp)
    thresholds = total[order[(v + 1).to(torch.int64) - 1]]
    return fmeas, thresholds, order


def _prf(pred: torch.Tensor, lab: torch.Tensor):
    tp = (pred & lab).sum().to(torch.float64)
    fp = (pred & ~lab).sum().to(torch.float64)
    fn = (~pred & lab).sum().to(torch.float64)
    one = torch.ones_like(tp)
    pre = torch.where(tp + fp > 0, tp / torch.where(tp + fp > 0, tp + fp, one), torch.zeros_like(tp))
    rec = torch.where(tp + fn > 0, tp / torch.where(tp + fn > 0, tp + fn, one), torch.zeros_like(tp))
    return _f1(tp, fp, fn), pre, rec


def _roc_auc(total: torch.Tensor, lab: torch.Tensor, order: torch.Tensor) -> torch.Tensor:
    """sklearn roc_auc_score for binary labels: Mann-Whitney U with tied scores counted 1/2."""
    s = total[order]
    _vals, inverse, counts = torch.unique_consecutive(s, return_inverse=True, return_counts=True)
    ends = counts.cumsum(0).to(torch.float64)
    avg_rank = (ends - (counts.to(torch.float64) - 1.0) / 2.0)[inverse]      # mean of the 1-based ranks of a tie
    n_pos = lab.sum().to(torch.float64)
    n_neg = float(total.numel()) - n_pos
    return (avg_rank[lab[order]].sum() - n_pos * (n_pos + 1.0) / 2.0) / (n_pos * n_neg)


def eval_scores(scores, true_scores, th_steps, return_thresold=False, device="cuda"):
    """util/data.py:28-51 (same name, arguments and list results)."""
    total = _to_device_f64(scores, device).reshape(-1)
    lab = _to_device_f64(true_scores, device).reshape(-1) > 0
    if lab.numel() > total.numel():                                           # util/data.py:29-33 padding
        total = torch.cat([total.new_zeros(lab.numel() - total.numel()), total])
    fmeas, ths, _ = _sweep(total, lab, th_steps)
    return (fmeas.tolist(), ths.tolist()) if return_thresold else fmeas.tolist()


def get_best_performance_data(total_err_scores, gt_labels, topk=1, device="cuda"):
    """evaluate.py:129-158: (best F1 of the 400-step sweep, precision, recall, AUC, threshold)."""
    scores = _to_device_f64(total_err_scores, device)
    lab = _to_device_f64(gt_labels, device).reshape(-1) > 0
    total = _topk_total(scores, topk)
    fmeas, ths, order = _sweep(total, lab, TH_STEPS)
    th_i = torch.argmax(fmeas)                                                # first maximum, as list.index(max)
    thr = ths[th_i]
    _f, pre, rec = _prf(total > thr, lab)
    out = torch.stack([fmeas[th_i], pre, rec, _roc_auc(total, lab, order), thr]).tolist()
    return tuple(out)


def get_val_performance_data(total_err_scores, normal_scores, gt_labels, topk=1, device="cuda"):
    """evaluate.py:99-127: threshold = the largest validation score."""
    scores = _to_device_f64(total_err_scores, device)
    lab = _to_device_f64(gt_labels, device).reshape(-1) > 0
    total = _topk_total(scores, topk)
    thr = _to_device_f64(normal_scores, device).max()
    f1, pre, rec = _prf(total > thr, lab)
    order = torch.argsort(total, stable=True)
    return tuple(torch.stack([f1, pre, rec, _roc_auc(total, lab, order), thr]).tolist())
