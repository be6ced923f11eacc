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
