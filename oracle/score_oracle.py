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


# ---------------------------------------------------------------------------- threshold sweep / F1
# SURVEY §8f-4.  Pinned by tests/golden/perf_*.npz (the reference's evaluate.py / util/data.py run with
# the real scipy.rankdata and sklearn metrics).

TH_STEPS = 400   # evaluate.py:146 `eval_scores(..., 400, ...)`


def topk_total(scores_nt: np.ndarray, topk: int = 1) -> np.ndarray:
    """evaluate.py:131-139: per tick, the SUM of the `topk` largest sensor scores."""
    s = np.asarray(scores_nt, dtype=np.float64)
    return np.sort(s, axis=0)[s.shape[0] - topk:].sum(axis=0) if topk > 1 else s.max(axis=0)


def _f1(tp, fp, fn):
    den = 2.0 * tp + fp + fn
    return np.where(den > 0, 2.0 * tp / np.where(den > 0, den, 1.0), 0.0)   # sklearn: 0 when undefined


def eval_scores(scores, true_labels, th_steps: int = TH_STEPS):
    """util/data.py:28-51: ordinal ranks (ties by position), `th_steps` rank thresholds i/th_steps*T;
    prediction = rank > threshold; F1 per threshold and the score value sitting at rank int(thr + 1)."""
    scores = np.asarray(scores, dtype=np.float64)
    lab = np.asarray(true_labels) > 0
    t = len(scores)
    order = np.argsort(scores, kind="stable")                 # rankdata(method='ordinal') - 1
    lab_sorted = lab[order].astype(np.int64)
    suffix = np.concatenate([np.cumsum(lab_sorted[::-1])[::-1], [0]])   # positives among ranks > r
    v = (np.arange(th_steps) * 1.0 / th_steps) * t            # th_vals[i] * len(scores), float64 as there
    cut = np.floor(v).astype(np.int64)                        # rank > v  <=>  (rank - 1) >= floor(v)
    tp = suffix[cut].astype(np.float64)
    n_pred = (t - cut).astype(np.float64)
    n_pos = float(lab.sum())
    fmeas = _f1(tp, n_pred - tp, n_pos - tp)
    thresholds = scores[order[(v + 1).astype(np.int64) - 1]]  # scores[index of rank int(v + 1)]
    return fmeas, thresholds


def _prf(pred, lab):
    tp = float(np.sum(pred & lab)); fp = float(np.sum(pred & ~lab)); fn = float(np.sum(~pred & lab))
    pre = tp / (tp + fp) if tp + fp > 0 else 0.0
    rec = tp / (tp + fn) if tp + fn > 0 else 0.0
    return float(_f1(tp, fp, fn)), pre, rec


def roc_auc(labels, scores) -> float:
    """sklearn.metrics.roc_auc_score for binary labels = the Mann-Whitney statistic with ties counted 1/2."""
    scores = np.asarray(scores, dtype=np.float64)
    lab = np.asarray(labels) > 0
    order = np.argsort(scores, kind="stable")
    s = scores[order]
    ranks = np.empty(len(s))
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = (i + j) / 2.0 + 1.0
        i = j + 1
    n_pos = float(lab.sum()); n_neg = float(len(s) - lab.sum())
    return float((ranks[lab[order]].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def best_performance(scores_nt, gt_labels, topk: int = 1):
    """evaluate.py:129-158: (max F1 of the sweep, precision, recall, AUC, threshold); precision/recall are
    those of `score > threshold` at the FIRST best threshold."""
    total = topk_total(scores_nt, topk)
    lab = np.asarray(gt_labels) > 0
    fmeas, ths = eval_scores(total, lab)
    th_i = int(np.argmax(fmeas))                              # list.index(max(...)): first maximum
    _f, pre, rec = _prf(total > ths[th_i], lab)
    return float(fmeas[th_i]), pre, rec, roc_auc(lab, total), float(ths[th_i])


def val_performance(scores_nt, normal_scores_nt, gt_labels, topk: int = 1):
    """evaluate.py:99-127: threshold = max of the validation ("normal") scores."""
    total = topk_total(scores_nt, topk)
    lab = np.asarray(gt_labels) > 0
    thr = float(np.max(normal_scores_nt))
    f1, pre, rec = _prf(total > thr, lab)
    return f1, pre, rec, roc_auc(lab, total), thr
