"""GPU suite, collected LAST on purpose (file name): HIP-against-HIP consistency of the training paths — graph
replay vs eager launches, the native step vs the autograd step, the opt-in variants, bitwise reproducibility.

These tests compare two of OUR OWN paths over several optimizer steps.  They are not parity tests: every
oracle / fixture test lives in the files that collect before this one, so a failure here can never keep them
from running (round 2: `pytest -x` stopped at a flaky test of this family and 13 oracle tests never ran).

What made the family flaky, measured (profiles/r03_grad_error_5steps_l3.txt, tests/diag_grad_error.py): not a
gradient-accuracy gap — both paths sit at the fp32 reference's own error against float64 at every step — but
(a) `gnn.bias`: its true gradient is 0 (a bias in front of a train-mode BatchNorm), what arrives is rounding
noise, and Adam turns noise into full +-lr steps; the noise came from float ATOMICS (d_bias), so the bias moved
differently run to run and perturbed every activation by an ulp; (b) the derivative of (Leaky)ReLU jumps at 0:
with ~5e5 pre-activations per step one of them sits within that ulp of 0 in ~5 % of the steps, flips in one run
and not in the other, and moves an attention gradient by ~1 % of its largest entry — which five Adam steps turn
into the 1e-5 .. 4e-5 parameter drift the old test tripped over.  Since ABI 20 the backward has no floating-point
atomics (gdn_colsum_ticket), a step is bitwise reproducible, and the tests below either demand EXACT equality
(same kernels, same inputs) or re-synchronise the two implementations every step so that no difference can be
amplified through a kink."""
import numpy as np
import pytest
import torch

from test_gpu_train_parity import FixedMaskDropout, _mix32_mask

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("split", [False, True])
def test_graphed_train_step_equals_eager_steps(split, gpu_device):
    """harness.GraphedTrainStep: replaying the captured step trains EXACTLY like launching it eagerly — the same
    kernels on the same inputs, and nothing in a step depends on the order workgroups finish in (to fp32: the
    head's column statistics are fp64 atomics whose order can move the fp64 sum by 1e-16, i.e. an fp32 result
    once in ~1e9 values) — so losses, parameters and BatchNorm buffers are compared bit for bit; capturing does
    not move the parameters.  split=True is the multi-rank form (forward+backward graph, all-reduce, optimizer
    graph)."""
    from gdn_amd.harness import GraphedTrainStep
    from test_gpu_forward_parity import random_params
    b = 64
    g = torch.Generator().manual_seed(5)
    xs = torch.rand((4, b, 27, 10), generator=g).to(gpu_device)
    ys = torch.rand((4, b, 27), generator=g).to(gpu_device)
    finals = []
    for use_graph in (False, True):
        model = random_params(27, 10, 8, 64, seed=3).to(gpu_device)
        model.dp.p = 0.0
        before = [p.detach().clone() for p in model.parameters()]
        step = GraphedTrainStep(model, b, use_graph=use_graph, split=split)
        if use_graph:
            step._capture()
            for p, q in zip(model.parameters(), before):
                assert torch.equal(p, q)
        losses = []
        for i in range(4):
            step.x.copy_(xs[i]); step.y.copy_(ys[i])
            losses.append(float(step.step()))
        finals.append((losses, [p.detach().clone() for p in model.parameters()],
                       [bf.detach().clone() for bf in model.buffers()]))
    assert finals[0][0] == finals[1][0]
    names = [n for n, _ in model.named_parameters()]
    for name, pa, pb in zip(names, finals[0][1], finals[1][1]):
        assert torch.equal(pa, pb), (name, float((pa - pb).abs().max()))
    for ba, bb in zip(finals[0][2], finals[1][2]):
        assert torch.equal(ba, bb)
    assert min(finals[1][0]) < finals[1][0][0]          # it trains


def test_harness_train_graph_mode_matches_eager(gpu_device, tmp_path):
    """train(use_graph=True): full minibatches replay the captured step, the ragged last batch runs
    eagerly with the same optimizer; same losses, same checkpoint as the eager loop."""
    from gdn_amd import harness
    from test_gpu_forward_parity import random_params
    g = torch.Generator().manual_seed(11)
    xs, ys = torch.rand((150, 27, 10), generator=g), torch.rand((150, 27), generator=g)
    loader = [(xs[s:s + 64], ys[s:s + 64], torch.zeros(len(xs[s:s + 64])), None) for s in range(0, 150, 64)]
    assert [b[0].shape[0] for b in loader] == [64, 64, 22]
    val = [(xs[:32], ys[:32], torch.zeros(32), None)]
    out = {}
    for mode in (False, True):
        model = random_params(27, 10, 8, 64, seed=4).to(gpu_device)
        model.dp.p = 0.0
        path = str(tmp_path / f"best_{mode}.pt")
        losses = harness.train(model, path, {"epoch": 2}, loader, val, use_graph=mode)
        out[mode] = (losses, torch.load(path, weights_only=True))
    assert len(out[True][0]) == 6
    np.testing.assert_allclose(out[False][0], out[True][0], atol=2e-5, rtol=0)
    for key, val_e in out[False][1].items():
        # zero-gradient bias (see the 2-step test) random-walks by +-lr per step; the BatchNorm behind it
        # tracks the mean of z, which contains that bias
        tol = 2e-2 if key.endswith("gnn.bias") or key.endswith("0.bn.running_mean") else 1e-4
        np.testing.assert_allclose(val_e.cpu().numpy(), out[True][1][key].cpu().numpy(), atol=tol, err_msg=key)




def test_back_to_back_graph_replays_without_host_sync(gpu_device):
    """Replays issued back to back (no host synchronisation, as in a real training loop) must train like
    per-step launches.  Shape = the SWaT one at 4096 windows: torch's multi-block mean reduction (the
    mse_loss the step used before the fused loss kernel) went wrong exactly here under replay."""
    from gdn_amd.harness import GraphedTrainStep
    from test_gpu_forward_parity import random_params
    b, steps = 4096, 10
    g = torch.Generator().manual_seed(0)
    x = torch.rand((b, 127, 15), generator=g).to(gpu_device)
    y = torch.rand((b, 127), generator=g).to(gpu_device)

    model = random_params(127, 15, 30, 64, seed=0).to(gpu_device).train()
    model.dp.p = 0.0
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    for _ in range(steps):                                  # plain eager loop, torch loss, synced every step
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(model(x, None), y)
        loss.backward()
        opt.step()
        ref_loss = loss.item()
    ref_params = [p.detach().clone() for p in model.parameters()]

    model = random_params(127, 15, 30, 64, seed=0).to(gpu_device).train()
    model.dp.p = 0.0
    step = GraphedTrainStep(model, b)
    step.x.copy_(x)
    step.y.copy_(y)
    for _ in range(steps):
        step.step()                                         # no .item(), no synchronize
    torch.cuda.synchronize()
    assert abs(float(step.loss) - ref_loss) < 2e-5
    for (name, p), q in zip(model.named_parameters(), ref_params):
        tol = 2e-2 if name.endswith("gnn.bias") else 2e-4
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.cpu().numpy(), atol=tol, err_msg=name)




def test_graph_mode_with_even_batches_keeps_validation_fresh(gpu_device, tmp_path):
    """Every batch is full-size, so a graph-mode epoch runs NO Python forward in train mode: the
    validation pass after each epoch must still see the parameters the replays wrote (the eval
    constants cache is keyed on versions a replay never bumps — GraphedTrainStep.step invalidates it).
    Same per-step losses, validation-selected checkpoint and early-stop bookkeeping as the eager loop."""
    from gdn_amd import harness
    from test_gpu_forward_parity import random_params
    g = torch.Generator().manual_seed(12)
    xs, ys = torch.rand((128, 27, 10), generator=g), torch.rand((128, 27), generator=g)
    loader = [(xs[s:s + 64], ys[s:s + 64], torch.zeros(64), None) for s in range(0, 128, 64)]
    val = [(xs[:32] * 0.5, ys[:32], torch.zeros(32), None)]
    seen = {}
    real_test = harness.test

    def spy(model, dataloader, device=None, **kw):
        loss, res = real_test(model, dataloader, device, **kw)
        seen.setdefault(spy.mode, []).append(loss)
        return loss, res
    harness.test = spy
    try:
        out = {}
        for mode in (False, True):
            spy.mode = mode
            model = random_params(27, 10, 8, 64, seed=5).to(gpu_device)
            model.dp.p = 0.0
            path = str(tmp_path / f"best_{mode}.pt")
            losses = harness.train(model, path, {"epoch": 3}, loader, val, use_graph=mode)
            out[mode] = (losses, torch.load(path, weights_only=True))
    finally:
        harness.test = real_test
    assert len(seen[False]) == len(seen[True]) == 3
    assert len(set(seen[True])) == 3, "validation loss did not move: stale eval constants"
    # (two correct runs drift apart by ~1e-5 per epoch: the zero-gradient gnn.bias random-walks by +-lr per Adam
    # step on rounding noise, see the 2-step test; stale constants would be off by ~1e-2)
    np.testing.assert_allclose(seen[True], seen[False], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out[True][0], out[False][0], atol=1e-4, rtol=0)
    for key, val_e in out[False][1].items():
        tol = 2e-2 if key.endswith("gnn.bias") or key.endswith("0.bn.running_mean") else 1e-4
        np.testing.assert_allclose(val_e.cpu().numpy(), out[True][1][key].cpu().numpy(), atol=tol, err_msg=key)




@pytest.mark.parametrize("p_drop,layers,inter", [(0.0, 1, 128), (0.2, 1, 128), (0.2, 2, 128), (0.0, 3, 128),
                                                 (0.2, 3, 50)])
def test_native_train_step_equals_the_autograd_step(p_drop, layers, inter, gpu_device):
    """NativeTrainStep (no autograd, gradients straight into the flat bucket, gdn_adam_step, dropout drawn in the
    kernels) against the autograd + torch.optim.Adam path fed THE SAME dropout masks (recomputed here from the
    documented hash), over 5 steps.  Every step BOTH paths start from the same parameters (the autograd model
    is reloaded from the native one), so the comparison is per step and nothing is amplified:
      * loss of the step: equal to 1e-6;
      * gradients native vs autograd: 1e-5 of each tensor's largest entry (different kernels only for the folded
        terms' chain rule and the embedding-gradient accumulation);
      * gradients of BOTH paths vs float64 at those parameters, relative bounds (tests/_grad_check.py), on the
        steps where no (Leaky)ReLU input that matters sits inside the fp32 rounding band (at least 1 of the 5);
      * the parameters after the optimizer: gdn_adam_step vs torch.optim.Adam, 2e-6 relative."""
    from gdn_amd import harness
    from _grad_check import KINK_BAND, NOISE_FLOOR, assert_grads_close, oracle_step
    from test_gpu_forward_parity import random_params
    b, n, w, k, d, steps, seed = 48, 27, 10, 8, 64, 5, 1234567890123
    g = torch.Generator().manual_seed(3)
    xs = torch.rand((steps, b, n, w), generator=g).to(gpu_device)
    ys = torch.rand((steps, b, n), generator=g).to(gpu_device)

    model = random_params(n, w, k, d, seed=9, out_layer_num=layers, inter=inter).to(gpu_device)
    model.dp.p = p_drop
    assert harness.NativeTrainStep.applicable(model)
    nat = harness.NativeTrainStep(model, b, use_graph=True, seed=seed)
    assert all(p.data_ptr() >= nat.flat_p.data_ptr() and p.data_ptr() < nat.flat_p.data_ptr() + 4 * nat.count
               for p in model.parameters())          # the parameters ARE views of the flat buffer

    ref = random_params(n, w, k, d, seed=9, out_layer_num=layers, inter=inter).to(gpu_device).train()
    masks = [(_mix32_mask(seed, t, b * n * d, p_drop, gpu_device).float() / (1.0 - p_drop)).view(b, n, d)
             for t in range(steps)]
    ref.dp = FixedMaskDropout(masks)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    names = [name for name, _ in ref.named_parameters()]
    compared = 0
    for t in range(steps):
        with torch.no_grad():                        # same starting point: the native path's state
            start = {key: v.detach().clone() for key, v in model.state_dict().items()}
            ref.load_state_dict(start)
        nat.x.copy_(xs[t]); nat.y.copy_(ys[t])
        loss_n = float(nat.step())
        g_n = {name: nat.flat_g[off:off + cnt].detach().clone().view(prm.shape)
               for name, prm, (off, cnt) in zip(names, ref.parameters(), nat.slices)}
        opt.zero_grad()
        loss_r = torch.nn.functional.mse_loss(ref(xs[t], None), ys[t])
        loss_r.backward()
        g_r = {name: prm.grad.detach().clone() for name, prm in ref.named_parameters()}
        opt.step()
        assert abs(loss_n - float(loss_r.detach())) < 1e-6, (t, loss_n, float(loss_r.detach()))
        assert torch.equal(nat.ws["topk"], ref.learned_graph)
        ref_loss, want, kink = oracle_step(start, xs[t], ys[t], ref.learned_graph, layers, masks[t])
        assert abs(loss_n - ref_loss) < 2e-6
        noise_only = {name for name in names if float(want[name].abs().max()) < NOISE_FLOOR}
        for name in names:
            top = float(want[name].abs().max())
            if name in noise_only:
                continue
            assert float((g_n[name] - g_r[name]).abs().max()) <= 1e-5 * top, (t, name)
        if kink > KINK_BAND:
            assert_grads_close(g_n, want, what=f"native step {t}")
            assert_grads_close(g_r, want, what=f"autograd step {t}")
            compared += 1
        for name, pa, pb in zip(names, model.parameters(), ref.parameters()):
            if name in noise_only:                   # +-lr on rounding noise, either sign
                assert float((pa - pb).detach().abs().max()) <= 2.1e-3, name
                continue
            # (from step 1 on torch's moments come from ITS gradient history: within the gradient agreement above)
            np.testing.assert_allclose(pa.detach().cpu().numpy(), pb.detach().cpu().numpy(), rtol=2e-6, atol=2e-7,
                                       err_msg=f"step {t} {name}")
    assert int(nat.state[1]) == steps
    assert compared >= 1, "no step was comparable with float64 (kinks): pick another seed"
    if p_drop > 0:
        kept = torch.stack(masks).ne(0).float().mean().item()
        assert abs(kept - (1.0 - p_drop)) < 2e-3         # the draw has the right rate


def test_native_step_variants_agree(gpu_device, monkeypatch):
    """The opt-in forms of the native step (loss folded into the head's last forward pass, GDN_FUSE_MSE=1; the
    row-gather backward, GDN_BWD_PATH is read once per process so only the loss variant is switched here) train
    exactly like the default: same losses, same parameters after 4 steps (same dropout stream)."""
    from gdn_amd import harness
    from test_gpu_forward_parity import random_params
    b, n, w, k, d = 32, 27, 10, 8, 64
    g = torch.Generator().manual_seed(5)
    xs = torch.rand((4, b, n, w), generator=g).to(gpu_device)
    ys = torch.rand((4, b, n), generator=g).to(gpu_device)
    results = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("GDN_FUSE_MSE", fuse)
        model = random_params(n, w, k, d, seed=3).to(gpu_device)
        step = harness.NativeTrainStep(model, b, use_graph=True, seed=99)
        assert step._fuse_mse == (fuse == "1")
        losses = []
        for t in range(4):
            step.x.copy_(xs[t]); step.y.copy_(ys[t])
            losses.append(float(step.step()))
        results.append((losses, step.flat_p.clone()))
    np.testing.assert_allclose(results[0][0], results[1][0], rtol=0, atol=1e-7)
    # (the loss is the only thing computed differently, and d_out = 2 (out - y) / count is the same expression in
    # both kernels: the parameters follow bit for bit, gnn.bias included now that d_bias has no atomics)
    assert torch.equal(results[0][1], results[1][1])




def test_matrix_core_backward_is_bitwise_reproducible(gpu_device):
    """The matrix-core backward has no atomics and no order-dependent reductions (d_bias: per-workgroup rows added
    in row order by the last workgroup to finish, gdn_colsum_ticket): 300 launches on the same inputs (512 windows:
    every workgroup takes two) give the same bits in all four outputs — a stale read across one of its ten
    barriers per window, or a ticket drawn before a row has landed, would show up here."""
    from gdn_amd import ops
    n, k, b, d = 127, 30, 512, 64
    g = torch.Generator().manual_seed(11)
    graph = ops.topk_graph(torch.randn((n, d), generator=g).to(gpu_device), k)
    xlin = torch.randn((b * n, d), generator=g).to(gpu_device)
    s_i, s_j = torch.randn((b * n,), generator=g).to(gpu_device), torch.randn((b * n,), generator=g).to(gpu_device)
    bias = torch.zeros((d,), device=gpu_device)
    d_z = (torch.randn((b * n, d), generator=g) * 1e-6).to(gpu_device)
    _z, alpha = ops.attn_aggregate_fwd(xlin, s_i, s_j, graph, bias, b, want_alpha=True)
    first = ops.attn_aggregate_bwd(d_z, xlin, alpha, s_i, s_j, graph, b)
    for _ in range(300):
        again = ops.attn_aggregate_bwd(d_z, xlin, alpha, s_i, s_j, graph, b)
        assert torch.equal(first[0], again[0]) and torch.equal(first[1], again[1]) and torch.equal(first[2], again[2])
        assert torch.equal(first[3], again[3])
    np.testing.assert_allclose(first[3].cpu().double().numpy(), d_z.double().sum(0).cpu().numpy(), rtol=1e-4, atol=1e-9)
