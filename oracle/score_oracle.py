"""TEST INFRASTRUCTURE — not product code (same import rule as gdn_oracle.py).

numpy/scipy restatement of the reference's anomaly scoring (float64 throughout, as
the reference: its inputs are python-float lists, evaluate.py:7-8, 54-57).
Pinned by tests/golden/score_*.npz, produced by calling the reference's own
evaluate.py functions (numpy/scipy/sklearn are present in the image, so this part of
the oracle IS pinned against the real reference code).

Citations relative to /root/reference.
"""
from __future__ import annotations

import numpy as np
from scipy.stats import iqr

SCORE_EPS = 1e-2   # evaluate.py:58
SMOOTH_BEFORE = 3  # evaluate.py:63


def err_median_and_iqr(pred: np.ndarray, gt: np.ndarray):
    """util/data.py:75-82 — median and inter-quartile range of |pred-gt| for one sensor.
    The reference always arrives here with float64 data (python-float lists from
    test.py:73-75 turned into arrays at evaluate.py:7), so fp32 inputs are widened first."""
    delta = np.abs(np.subtract(np.asarray(pred, dtype=np.float64), np.asarray(gt, dtype=np.float64)))
    return np.median(delta), iqr(delta)


def err_scores(pred: np.ndarray, gt: np.ndarray) -> np.ndarray:
    """evaluate.py:48-68 for one sensor: robust-normalised |error|, then the mean of the
    current and the 3 previous ticks; the first 3 ticks stay 0."""
    med, rng = err_median_and_iqr(pred, gt)
    delta = np.abs(np.asarray(pred, dtype=np.float64) - np.asarray(gt, dtype=np.float64))
    a = (delta - med) / (np.abs(rng) + SCORE_EPS)
    out = np.zeros(a.shape)
    for t in range(SMOOTH_BEFORE, len(a)):
        out[t] = np.mean(a[t - SMOOTH_BEFORE:t + 1])
    return out


def full_err_scores(pred: np.ndarray, gt: np.ndarray) -> np.ndarray:
    """evaluate.py:6-36 (test half): per-sensor err_scores stacked to [N, T].
    pred, gt are [T, N]."""
    return np.vstack([err_scores(pred[:, i], gt[:, i]) for i in range(pred.shape[1])])


def anomaly_score(scores_nt: np.ndarray, topk: int = 1) -> np.ndarray:
    """evaluate.py:131-139 — sum of the `topk` largest per-sensor scores at each tick
    (topk=1: the max over sensors)."""
    n = scores_nt.shape[0]
    idx = np.argpartition(scores_nt, range(n - topk - 1, n), axis=0)[-topk:]
    return np.sum(np.take_along_axis(scores_nt, idx, axis=0), axis=0)
