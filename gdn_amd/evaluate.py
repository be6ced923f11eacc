"""Host-side mirror of the scoring half of the reference's evaluate.py (the part on the
hot path: `get_err_scores`, `get_full_err_scores`, and the max over sensors inside
`get_best_performance_data`).  Same function names and argument meaning; the arithmetic runs
in float64 HIP kernels (gdn_amd/csrc/gdn_score.hip).  Threshold search / F1 / AUC reporting
(evaluate.py:99-158, util/data.py:28-51) is host-side reporting and out of scope."""
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
