"""GPU suite: one training step (train.py:68-72 usage) — HIP forward + HIP backward of the graph
layer, torch for BN statistics / dropout / MLP — against the gradients the reference produced."""
import numpy as np
import pytest
import torch

from conftest import MODEL_CASES, load_golden, meta
from test_gpu_forward_parity import build_model

pytestmark = pytest.mark.gpu


class FixedMaskDropout(torch.nn.Module):
    def __init__(self, masks):
        super().__init__()
        self.masks, self.calls = list(masks), 0

    def forward(self, x):
        if not self.training:
            return x
        m = self.masks[self.calls]
        self.calls += 1
        return x * m


@pytest.mark.parametrize("case", MODEL_CASES)
def test_train_step_matches_reference_grads(case, gpu_device):
    data, p = load_golden(case)
    m = meta(data)
    model = build_model(p, m, gpu_device)
    model.injected_graph = torch.from_numpy(data["learned_graph"]).to(gpu_device)
    model.dp = FixedMaskDropout([torch.from_numpy(data["dropout_mask"]).to(gpu_device)])
    model.train()
    model.zero_grad()
    x = torch.from_numpy(data["x"]).to(gpu_device)
    y = torch.from_numpy(data["y"]).to(gpu_device)
    out = model(x, None)
    loss = torch.nn.functional.mse_loss(out, y, reduction="mean")       # train.py:20-23
    loss.backward()
    # train-mode BatchNorm normalises by the statistics of as few as B*N = 50..254 rows, which
    # amplifies fp32 association differences ~1/sigma: 5e-5 here (north_star bar: 1e-4)
    np.testing.assert_allclose(out.detach().cpu().numpy(), data["train_out"], atol=5e-5, rtol=0)
    np.testing.assert_allclose(loss.item(), float(data["train_loss"]), atol=5e-5, rtol=0)
    for name, prm in model.named_parameters():
        assert prm.grad is not None, name
        np.testing.assert_allclose(prm.grad.cpu().numpy(), data["g/" + name], atol=5e-5, rtol=1e-3,
                                   err_msg=name)
    for key, val in model.state_dict().items():
        if "running" in key or "num_batches" in key:
            np.testing.assert_allclose(val.cpu().numpy(), data["p_after_train_fwd/" + key], atol=2e-5,
                                       rtol=0, err_msg=key)


def test_two_adam_steps_match_reference_train_loop(gpu_device):
    """the reference's train() for 1 epoch x 2 batches (Adam lr 1e-3)."""
    data, p = load_golden("train_loop_2step")
    m = meta(data)
    model = build_model(p, m, gpu_device)
    masks = [torch.from_numpy(mk).to(gpu_device) for mk in data["masks"]]
    model.dp = FixedMaskDropout(masks)
    opt = torch.optim.Adam(model.parameters(), lr=0.001, weight_decay=0.0)   # train.py:31
    model.train()
    losses = []
    x = torch.from_numpy(data["x"]).to(gpu_device)
    y = torch.from_numpy(data["y"]).to(gpu_device)
    for s in range(0, x.shape[0], m["b"]):
        opt.zero_grad()
        out = model(x[s:s + m["b"]], None)
        loss = torch.nn.functional.mse_loss(out, y[s:s + m["b"]], reduction="mean")
        loss.backward()
        opt.step()
        losses.append(loss.item())
    np.testing.assert_allclose(losses, data["losses"], atol=2e-5, rtol=0)
    for key, val in model.state_dict().items():
        # Adam's first steps move every weight by ~lr regardless of gradient scale, so tiny
        # gradient differences are amplified: compare with an lr-sized tolerance
        atol = 2e-4
        if key == "gnn_layers.0.gnn.bias":
            # this bias feeds a train-mode BatchNorm, so its true gradient is exactly 0 and both
            # implementations see only rounding noise (~1e-9), which Adam turns into +-lr steps:
            # each step the two runs may move in opposite directions: up to ~2*lr apart per step
            atol = 4.5e-3
        np.testing.assert_allclose(val.cpu().numpy(), data["p_final/" + key], atol=atol, rtol=0, err_msg=key)
