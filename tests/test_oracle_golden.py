"""CPU suite, part 1: the oracle (oracle/*.py) against the golden vectors captured from the
reference's own model / evaluate files (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import MODEL_CASES, SCORE_CASES, load_golden, meta
from oracle import gdn_oracle, score_oracle

FP32_TOL = 1e-5   # oracle vs reference run: same ops, same library -> only summation-order noise


@pytest.mark.parametrize("case", MODEL_CASES)
def test_eval_forward_matches_reference(case):
    data, p = load_golden(case)
    m = meta(data)
    x = torch.from_numpy(data["x"])
    r = gdn_oracle.forward(p, x, m["k"], m["out_layer_num"])
    assert torch.equal(r["learned_graph"], torch.from_numpy(data["learned_graph"]))
    assert torch.equal(r["edge_index_1"], torch.from_numpy(data["edge_index_1"]))
    np.testing.assert_allclose(r["att_weight_1"].numpy(), data["att_weight_1"], atol=FP32_TOL, rtol=0)
    np.testing.assert_allclose(r["agg"].numpy(), data["agg"], atol=FP32_TOL, rtol=0)
    np.testing.assert_allclose(r["out"].numpy(), data["eval_out"], atol=FP32_TOL, rtol=0)


@pytest.mark.parametrize("case", MODEL_CASES)
def test_separable_spec_matches_op_faithful(case):
    data, p = load_golden(case)
    m = meta(data)
    x = torch.from_numpy(data["x"])
    g = torch.from_numpy(data["learned_graph"])
    r = gdn_oracle.forward_separable(p, x, g, m["out_layer_num"])
    np.testing.assert_allclose(r["agg"].numpy(), data["agg"], atol=FP32_TOL, rtol=0)
    np.testing.assert_allclose(r["out"].numpy(), data["eval_out"], atol=FP32_TOL, rtol=0)


@pytest.mark.parametrize("case", MODEL_CASES)
def test_train_step_loss_grads_and_bn_stats(case):
    data, p = load_golden(case)
    m = meta(data)
    leaf = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v)
            for k, v in p.items()}
    x = torch.from_numpy(data["x"])
    y = torch.from_numpy(data["y"])
    r = gdn_oracle.forward(leaf, x, m["k"], m["out_layer_num"], training=True,
                           dropout_mask=torch.from_numpy(data["dropout_mask"]))
    loss = torch.nn.functional.mse_loss(r["out"], y, reduction="mean")   # train.py:20-23
    loss.backward()
    np.testing.assert_allclose(r["out"].detach().numpy(), data["train_out"], atol=FP32_TOL, rtol=0)
    np.testing.assert_allclose(loss.item(), float(data["train_loss"]), atol=FP32_TOL, rtol=0)
    for key in data:
        if key.startswith("g/"):
            got = leaf[key[2:]].grad
            assert got is not None, key
            np.testing.assert_allclose(got.numpy(), data[key], atol=2e-5, rtol=1e-4, err_msg=key)
    for key, val in r["new_stats"].items():
        np.testing.assert_allclose(val.numpy(), data["p_after_train_fwd/" + key], atol=FP32_TOL, rtol=0,
                                   err_msg=key)


def test_degree_k_plus_one_case_is_really_in_the_fixture():
    data, _ = load_golden("dupemb_n10_k3")
    m = meta(data)
    ei = data["edge_index_1"]
    deg = np.bincount(ei[1], minlength=m["b"] * m["n"])
    assert deg.min() == m["k"] and deg.max() == m["k"] + 1


@pytest.mark.parametrize("case", SCORE_CASES)
def test_scoring_matches_reference_evaluate(case):
    data, _ = load_golden(case)
    s = score_oracle.full_err_scores(data["pred"], data["gt"])
    np.testing.assert_allclose(s, data["scores"], atol=1e-12, rtol=1e-12)
    for i in range(data["pred"].shape[1]):
        # the reference feeds python-float lists of the fp32 values: identical numbers in float64
        med, rng = score_oracle.err_median_and_iqr(data["pred"][:, i], data["gt"][:, i])
        np.testing.assert_allclose([med, rng], data["med_iqr"][i], atol=0, rtol=1e-15)
    a = score_oracle.anomaly_score(s, topk=1)
    np.testing.assert_array_equal(a, s.max(axis=0))


def test_eval_loop_fixture_consistent_with_oracle():
    data, p = load_golden("eval_loop_msl_shape")
    m = meta(data)
    x = torch.from_numpy(data["x"])
    y = torch.from_numpy(data["y"])
    preds, losses = [], []
    for s in range(0, x.shape[0], m["b"]):                       # test.py:43-65
        out = gdn_oracle.forward(p, x[s:s + m["b"]], m["k"])["out"]
        preds.append(out)
        losses.append(torch.nn.functional.mse_loss(out, y[s:s + m["b"]]).item())
    pred = torch.cat(preds)
    np.testing.assert_allclose(pred.numpy(), data["pred"], atol=FP32_TOL, rtol=0)
    np.testing.assert_allclose(sum(losses) / len(losses), float(data["avg_loss"]), atol=1e-6)
    s = score_oracle.full_err_scores(data["pred"].astype(np.float64), data["gt"].astype(np.float64))
    np.testing.assert_allclose(s, data["scores"], atol=1e-12, rtol=1e-12)


@pytest.mark.parametrize("case", ["perf_T1000_N27", "perf_T777_N5_ties"])
def test_threshold_sweep_oracle_matches_reference_evaluate(case):
    """the oracle's sweep / F1 / precision / recall / AUC against the reference's own
    evaluate.get_best_performance_data / get_val_performance_data / util.data.eval_scores."""
    import os
    from oracle import score_oracle
    data = np.load(os.path.join(os.path.dirname(__file__), "golden", case + ".npz"))
    scores, labels = data["scores"], data["labels"]
    np.testing.assert_allclose(score_oracle.full_err_scores(data["pred"], data["gt"]), scores, rtol=1e-12, atol=1e-13)
    for topk in (1, 3):
        fmeas, ths = score_oracle.eval_scores(score_oracle.topk_total(scores, topk), labels)
        np.testing.assert_allclose(fmeas, data[f"fmeas_top{topk}"], rtol=1e-13, atol=0)
        np.testing.assert_array_equal(ths, data[f"thresholds_top{topk}"])
        np.testing.assert_allclose(score_oracle.best_performance(scores, labels, topk), data[f"best_top{topk}"],
                                   rtol=1e-12, atol=0)
        normal = scores[:, : scores.shape[1] // 6]
        np.testing.assert_allclose(score_oracle.val_performance(scores, normal, labels, topk), data[f"val_top{topk}"],
                                   rtol=1e-12, atol=0)


FULL_BATCH_CASES = ["cfg0_msl27_w15_k20_b128", "cfg1_fc64_w15_k64_b128", "cfg2_swat127_w15_k30_b512"]


def full_batch_input(data):
    """Input of a full-batch fixture (tests/golden/make_golden.py: run_eval_only_case): windows rebuilt from
    the stored raw series slice (datasets/TimeDataset.py:46-57), or torch.rand under the stored seed."""
    b, n, w = (int(v) for v in data["meta_bnwkd"][:3])
    if "raw" in data:
        raw = torch.from_numpy(data["raw"].astype(np.float32))
        return torch.stack([raw[:, t - w:t] for t in range(w, w + b)])
    x = torch.rand((b, n, w), generator=torch.Generator().manual_seed(int(data["x_seed"])))
    assert float(x.double().sum()) == float(data["x_sum"]), "torch.rand stream changed: regenerate the fixture"
    return x


@pytest.mark.parametrize("case", FULL_BATCH_CASES)
def test_full_batch_configs_oracle_matches_reference(case):
    """BASELINE configs[0..2] as worded, at their stated batch: oracle vs the reference's eval output."""
    data, p = load_golden(case)
    m = meta(data)
    x = full_batch_input(data)
    with torch.no_grad():
        r = gdn_oracle.forward(p, x, m["k"])
    if m["k"] < m["n"]:
        assert torch.equal(r["learned_graph"], torch.from_numpy(data["learned_graph"]))
    np.testing.assert_allclose(r["out"].numpy(), data["eval_out"], atol=FP32_TOL, rtol=0)


@pytest.mark.parametrize("case", MODEL_CASES)
def test_staged_float64_restatement_equals_the_op_faithful_oracle(case):
    """tests/_grad_check.py::staged_f64 — the training step in the decomposition the HIP kernels use, in float64,
    the checker of every gradient test — against gdn_oracle.forward (the op-faithful restatement the fixtures pin)
    run in float64 on the fixture's train-mode inputs: loss and every parameter gradient to 1e-10 relative; and
    against the gradients the REFERENCE produced in fp32 (fixture), to the fp32 level."""
    from _grad_check import f64_leaves, oracle_step
    data, p = load_golden(case)
    m = meta(data)
    x, y = torch.from_numpy(data["x"]), torch.from_numpy(data["y"])
    mask, graph = torch.from_numpy(data["dropout_mask"]), torch.from_numpy(data["learned_graph"])
    loss, grads, kink = oracle_step(p, x, y, graph, m["out_layer_num"], mask)
    leaf = f64_leaves(p)
    r = gdn_oracle.forward(leaf, x.double(), m["k"], m["out_layer_num"], training=True, dropout_mask=mask.double(),
                           graph=graph)
    ref_loss = torch.nn.functional.mse_loss(r["out"], y.double())
    ref_loss.backward()
    assert abs(loss - float(ref_loss.detach())) < 1e-12
    for name, g in grads.items():
        want = leaf[name].grad
        top = float(want.abs().max())
        if top < 1e-12:           # a bias in front of a train-mode BatchNorm: both sides hold float64 rounding noise
            assert float(g.abs().max()) < 1e-12, name
            continue
        assert float((g - want).abs().max()) <= 1e-10 * top, name
        np.testing.assert_allclose(g.numpy(), data["g/" + name].astype(np.float64), atol=5e-5, rtol=1e-3, err_msg=name)
    assert kink > 2e-6          # no (Leaky)ReLU input of the fixture sits in the fp32 rounding band
