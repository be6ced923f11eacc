"""Worker for tests/test_cpu_distributed.py: run under torch.distributed.run with gloo, 2 ranks.
Exercises the N>1 host logic on CPU: shard split, the all-to-all-by-sensor scoring exchange and the
flat-bucket gradient all-reduce.  The per-rank arithmetic is injected from the oracle (tests may do
that; the product default is the HIP backend)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gdn_amd import harness  # noqa: E402
from oracle import score_oracle  # noqa: E402


class OracleScoreBackend:
    """CPU stand-in for HipScoreBackend: same three entry points, arithmetic from the oracle / numpy."""

    @staticmethod
    def keys(pred_tn, gt_tn, pitch, out=None):
        t, n = pred_tn.shape
        k = torch.full((n, pitch), float("nan"), dtype=torch.float64)          # filler slots
        k[:, :t] = (pred_tn.double() - gt_tn.double()).abs().t()
        if out is not None:
            out.copy_(k)
            return out
        return k

    @staticmethod
    def select(keys_flat, blocks, n, pitch, total):
        k = keys_flat.reshape(blocks, n, pitch).numpy()
        rows = []
        for s in range(n):
            v = k[:, s, :].reshape(-1)
            v = v[~np.isnan(v)]
            assert v.size == total, (v.size, total)
            from scipy.stats import iqr
            rows.append((np.median(v), iqr(v)))                                # util/data.py:79-80
        return torch.tensor(rows, dtype=torch.float64).reshape(-1, 2)

    @staticmethod
    def smooth_max(pred_tn, gt_tn, med_iqr, first_tick, halo_pred, halo_gt):
        p, g = pred_tn.double().numpy(), gt_tn.double().numpy()
        if halo_pred is not None:
            p = np.concatenate([halo_pred.double().numpy(), p])
            g = np.concatenate([halo_gt.double().numpy(), g])
        pad = p.shape[0] - pred_tn.shape[0]
        mi = med_iqr.numpy()
        a = (np.abs(p - g) - mi[:, 0]) / (np.abs(mi[:, 1]) + 1e-2)
        out = np.zeros((pred_tn.shape[0], p.shape[1]))
        for t in range(pred_tn.shape[0]):
            if first_tick + t >= 3:
                r = t + pad
                out[t] = (((a[r - 3] + a[r - 2]) + a[r - 1]) + a[r]) / 4.0
        return torch.from_numpy(out.max(axis=1))


def main():
    dist.init_process_group("gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    for total, n in ((101, 7), (64, 5), (9, 3), (5, 2)):
        g = torch.Generator().manual_seed(total)
        pred = torch.rand((total, n), generator=g)
        gt = pred + 0.1 * torch.randn((total, n), generator=g)
        s, e = harness.shard_range(total, rank, size)
        local = harness.distributed_anomaly(pred[s:e].contiguous(), gt[s:e].contiguous(), total,
                                            backend=OracleScoreBackend)
        want = score_oracle.anomaly_score(score_oracle.full_err_scores(pred.numpy(), gt.numpy()))
        np.testing.assert_allclose(local.numpy(), want[s:e], rtol=1e-12, atol=1e-13)

    # the chunk-overlapped evaluator: several chunks, a partly filled one, ranks with fewer chunks than others
    for total, n, chunk in ((9000, 5, 2048), (4099, 7, 2048), (40, 3, 4096), (5, 2, 4096), (2, 4, 2048)):
        g = torch.Generator().manual_seed(total + 1)
        pred = torch.rand((total, n), generator=g)
        gt = pred + 0.1 * torch.randn((total, n), generator=g)
        s, e = harness.shard_range(total, rank, size)
        box = {}

        def fwd(a, b, box=box, s=s):                       # "forward" = copy the precomputed predictions
            box["ev"].pred[a:b] = pred[s + a:s + b]

        ev = harness.ShardedEvaluator(None, None, gt[s:e].contiguous(), total, chunk=chunk,
                                      backend=OracleScoreBackend, forward=fwd)
        box["ev"] = ev
        want = score_oracle.anomaly_score(score_oracle.full_err_scores(pred.numpy(), gt.numpy()))
        for _ in range(2):                                 # buffers are reused across steps
            local = ev.step()
            np.testing.assert_allclose(local.numpy(), want[s:e], rtol=1e-12, atol=1e-13)

    # flat-bucket gradient averaging
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.BatchNorm1d(3), torch.nn.Linear(3, 1))
    harness.broadcast_parameters(model)
    x = torch.full((4, 5), float(rank + 1))
    model(x).sum().backward()
    mine = [p.grad.clone() for p in model.parameters()]
    gathered = [[torch.empty_like(m) for _ in range(size)] for m in mine]
    for buf, m in zip(gathered, mine):
        dist.all_gather(buf, m)
    harness.sync_gradients(model)
    for p, buf in zip(model.parameters(), gathered):
        torch.testing.assert_close(p.grad, sum(buf) / size)
    # ---- the data-parallel TRAINING step of harness.NativeTrainStep, host logic on CPU: every rank runs the
    # model (here: the oracle, float64) on ITS shard of the global batch with ITS OWN BatchNorm statistics
    # (standard DDP), writes the gradients into the flat bucket (harness.flat_layout), the bucket
    # is all-reduced as it stands (harness.all_reduce_flat) and Adam runs with grad_scale = 1/ranks.
    # Reference: the same per-shard gradients averaged in one process + torch.optim.Adam.
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_golden, meta
    from oracle import gdn_oracle
    data, p = load_golden("msl_demo_w5_k5")
    m = meta(data)
    names = [k for k in p if "running" not in k and "num_batches" not in k]
    f64 = torch.float64

    def shard_grads(r):
        """Gradients of rank r's shard (batch rows r::size), BatchNorm statistics of that shard only."""
        leaf = {k: (v.to(f64).clone().requires_grad_(k in names) if v.is_floating_point() else v) for k, v in p.items()}
        xb = torch.from_numpy(data["x"])[r::size].to(f64)
        yb = torch.from_numpy(data["y"])[r::size].to(f64)
        out = gdn_oracle.forward(leaf, xb, m["k"], training=True,
                                 dropout_mask=torch.from_numpy(data["dropout_mask"])[r::size].to(f64),
                                 graph=torch.from_numpy(data["learned_graph"]))["out"]
        torch.nn.functional.mse_loss(out, yb).backward()
        return [leaf[k].grad for k in names]

    params = [p[k].to(f64).clone() for k in names]
    slices, count = harness.flat_layout(params)
    assert all(off % 4 == 0 for off, _ in slices) and count % 4 == 0
    flat_p = torch.zeros((count,), dtype=f64)
    flat_g = torch.zeros((count,), dtype=f64)
    for prm, (off, cnt) in zip(params, slices):
        flat_p[off:off + cnt] = prm.reshape(-1)
    for gr, (off, cnt) in zip(shard_grads(rank), slices):
        flat_g[off:off + cnt] = gr.reshape(-1)
    scale = harness.all_reduce_flat(flat_g)
    assert scale == 1.0 / size
    # gdn_adam_step's update (include/gdn_hip.h), first step: m = (1-b1) g, v = (1-b2) g^2
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    gs = flat_g * scale
    m1, v1 = (1 - b1) * gs, (1 - b2) * gs * gs
    flat_p = flat_p - (lr / (1 - b1)) * m1 / (v1.sqrt() / (1 - b2) ** 0.5 + eps)
    # reference
    ref_params = [torch.nn.Parameter(p[k].to(f64).clone()) for k in names]
    per_rank = [shard_grads(r) for r in range(size)]
    opt = torch.optim.Adam(ref_params, lr=lr)
    for i, prm in enumerate(ref_params):
        prm.grad = sum(per_rank[r][i] for r in range(size)) / size
    opt.step()
    for prm, (off, cnt), name in zip(ref_params, slices, names):
        torch.testing.assert_close(flat_p[off:off + cnt].view(prm.shape), prm.detach(), rtol=1e-9, atol=1e-12, msg=name)
    # per-rank BatchNorm statistics really differ from the global-batch ones (the step is DDP, not SyncBN)
    if size > 1:
        assert not torch.allclose(per_rank[0][names.index("gnn_layers.0.gnn.lin.weight")],
                                  per_rank[1][names.index("gnn_layers.0.gnn.lin.weight")])
    # identical parameters on every rank after the broadcast
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(flat, ref)
    dist.barrier()
    if rank == 0:
        print("DIST_WORKER_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
