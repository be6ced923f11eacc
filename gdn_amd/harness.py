"""Callers of the hot path, mirroring the reference's loops:

  test()            <- test.py:21-79   (eval loop: forward per batch, MSE, concatenated outputs)
  train()           <- train.py:27-112 (Adam lr 1e-3, MSE, per-epoch validation, best checkpoint,
                                        early stop after 15 epochs without improvement)
  SeriesEvaluator   the throughput form of test() + evaluate.get_full_err_scores for a series
                    that is already resident in HBM: forward launches write straight into one
                    [T, N] prediction buffer, scoring runs on device, the whole step can be
                    captured in a HIP graph and replayed.
  shard / DDP       one process per GPU: windows are independent, so a series is split into
                    contiguous shards (no collective in the forward); scoring needs per-sensor
                    order statistics over ALL ticks -> one all-to-all by sensor + one all-gather
                    of the [N,2] median/IQR table; training all-reduces one flat gradient bucket.
"""
from __future__ import annotations

import contextlib
import ctypes
import gc
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import ops


# --------------------------------------------------------------------------- eval loop

@contextlib.contextmanager
def capture(graph, pool=None):
    """`torch.cuda.graph(graph)` with the cyclic garbage collector out of the way.  torch 2.10 no longer collects
    on entry, and a collection that fires INSIDE the captured region can finalise an older `CUDAGraph` or free a
    cached block of a dead graph pool — HIP calls that are illegal while a stream is capturing; the error is
    raised from a destructor and ends the process (seen once in the GPU suite: `Fatal Python error: Aborted` under
    `Garbage-collecting`).  So: collect before, keep the collector disabled until the capture has ended."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        with torch.cuda.graph(graph, pool=pool):
            yield
    finally:
        if was_enabled:
            gc.enable()

def test(model, dataloader, device=None, as_tensors: bool = False):
    """Mirror of the reference test() (test.py:21-79).  Returns (avg_loss, [predictions, ground
    truth, labels]) with the three results as python lists, exactly like the reference; per-batch
    losses stay on the device and are read once at the end (the reference syncs every batch).
    `as_tensors=True` keeps the three results as device tensors [T, N] (gdn_amd.evaluate accepts
    them directly): the `.tolist()` of the reference costs seconds at T = 10^5 windows."""
    device = device or next(model.parameters()).device
    model.eval()
    preds, gts, labs, losses = [], [], [], []
    for x, y, labels, edge_index in dataloader:
        x, y, labels = [item.to(device).float() for item in (x, y, labels)]
        with torch.no_grad():
            predicted = model(x, edge_index)
            losses.append(F.mse_loss(predicted, y, reduction="mean"))
            preds.append(predicted)
            gts.append(y)
            labs.append(labels.unsqueeze(1).repeat(1, predicted.shape[1]))
    avg_loss = float(torch.stack(losses).double().sum().item() / len(losses)) if losses else 0.0
    if not preds:
        return avg_loss, [[], [], []]
    out = [torch.cat(preds), torch.cat(gts), torch.cat(labs)]
    return avg_loss, (out if as_tensors else [t.tolist() for t in out])


def train(model=None, save_path="", config=None, train_dataloader=None, val_dataloader=None, use_graph=False,
          **_ignored):
    """Mirror of the reference train() (train.py:27-112): same optimizer, loss, checkpoint and
    early-stop rule.  Returns the list of per-step losses.

    `use_graph=True` (or config["hip_graph"]) runs every full-size minibatch as one replay of a
    `GraphedTrainStep` (captured at the first batch's size) and the ragged last batch of an epoch
    eagerly with the same optimizer; losses stay on the device until the epoch ends (the
    reference syncs with `.item()` every step)."""
    config = config or {}
    use_graph = use_graph or bool(config.get("hip_graph", False))
    device = next(model.parameters()).device
    graphed = None
    if not use_graph:
        optimizer = torch.optim.Adam(model.parameters(), lr=0.001, weight_decay=config.get("decay", 0))
    losses, min_loss, stale = [], 1e8, 0
    # Range of the training inputs (include/gdn_hip.h "range guard"): decided ONCE — config["wide"] when the caller
    # knows (python -m gdn_amd.main looks at its resident series), else from the first batch — and pinned on the
    # model for the whole run: no per-step host check, and the captured step needs the answer before the capture
    range_before = getattr(model, "operand_range", "auto")
    wide = config.get("wide", None)
    for _epoch in range(config.get("epoch", 1)):
        model.train()
        epoch_losses = []
        for x, labels, _attack, edge_index in train_dataloader:
            x, labels = x.float().to(device), labels.float().to(device)
            if wide is None and range_before == "auto":
                wide = model.input_exceeds_limit(x, margin=16.0)      # (the weights move during training)
            if range_before == "auto":
                model.operand_range = "wide" if wide else "narrow"
            if use_graph and graphed is None:
                graphed = GraphedTrainStep(model, x.shape[0], lr=0.001, weight_decay=config.get("decay", 0),
                                           wide=model.operand_range == "wide")
                optimizer = graphed.optimizer
            if graphed is not None and x.shape[0] == graphed.x.shape[0]:
                graphed.x.copy_(x)
                graphed.y.copy_(labels)
                epoch_losses.append(graphed.step().clone())
                continue
            optimizer.zero_grad()
            out = model(x, edge_index)
            loss = F.mse_loss(out, labels, reduction="mean")
            loss.backward()
            sync_gradients(model)
            optimizer.step()
            epoch_losses.append(loss.detach())
        step_losses = torch.stack(epoch_losses).tolist() if epoch_losses else []
        losses.extend(step_losses)
        acc = float(sum(step_losses))
        if val_dataloader is not None:
            model.operand_range = range_before          # validation data: its own range (eval guard)
            val_loss, _ = test(model, val_dataloader, device)
            if val_loss < min_loss:
                if save_path:
                    torch.save(model.state_dict(), save_path)
                min_loss, stale = val_loss, 0
            else:
                stale += 1
            if stale >= 15:
                break
        elif acc < min_loss:
            if save_path:
                torch.save(model.state_dict(), save_path)
            min_loss = acc
    model.operand_range = range_before
    return losses


# --------------------------------------------------------------------------- sharding / DDP
def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(total: int, rank: int, world_size: int):
    """Contiguous, balanced split of `total` windows: the first (total % world) ranks get one extra."""
    base, extra = divmod(total, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def sensor_range(n: int, rank: int, world_size: int):
    return shard_range(n, rank, world_size)


def pack_gradients(model, flat: torch.Tensor | None = None) -> torch.Tensor:
    """All gradients in ONE flat fp32 bucket (9 729 values = 38 KiB at the SWaT shape), written into
    `flat` when given (a static buffer for HIP-graph capture)."""
    grads = [p.grad.reshape(-1) for p in model.parameters() if p.grad is not None]
    if flat is None:
        return torch.cat(grads)
    torch.cat(grads, out=flat)
    return flat


def unpack_gradients(model, flat: torch.Tensor, world_size: int) -> None:
    """Averaged bucket back into the parameters' gradients: one scale, one multi-tensor copy."""
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    if world_size > 1:
        flat.div_(world_size)
    views, off = [], 0
    for g in grads:
        views.append(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
    torch._foreach_copy_(grads, views)


def sync_gradients(model, group=None) -> None:
    """Data-parallel gradient averaging: pack, one all-reduce (RCCL over xGMI on GPUs; latency bound
    at this size), unpack.  No-op in a single process."""
    _rank, size = world()
    if size == 1 or not any(p.grad is not None for p in model.parameters()):
        return
    flat = pack_gradients(model)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    unpack_gradients(model, flat, size)


def broadcast_parameters(model, src: int = 0) -> None:
    _rank, size = world()
    if size == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src)


KEY_SLICE = 2048    # the select kernel's slice: exchanged key rows are padded to a multiple of it


class HipScoreBackend:
    """Per-rank compute of the distributed scoring; tests inject an oracle-backed stand-in to
    exercise the exchange logic on CPU (gloo)."""

    @staticmethod
    def keys(pred_tn, gt_tn, pitch, out=None):
        return ops.score_keys(pred_tn, gt_tn, pitch, out=out)

    @staticmethod
    def select(keys_flat, blocks, n, pitch, total, ws=None, out=None):
        return ops.score_select(keys_flat, blocks, n, pitch, total, ws=ws, out=out)

    @staticmethod
    def select_workspace(blocks, n, pitch, device):
        return ops.score_select_workspace(blocks, n, pitch, device)

    @staticmethod
    def smooth_max(pred_tn, gt_tn, med_iqr, first_tick, halo_pred, halo_gt, anomaly=None):
        return ops.score_smooth_max(pred_tn, gt_tn, med_iqr, want_scores=False, first_tick=first_tick,
                                    halo_pred=halo_pred, halo_gt=halo_gt, anomaly=anomaly)[1]


def distributed_anomaly(pred_local, gt_local, total_ticks: int, backend=HipScoreBackend, group=None,
                        rehearse: bool = False):
    """Anomaly score of a series whose ticks are sharded contiguously over the ranks.

    Step 1  every rank turns ITS ticks into radix keys |pred-gt| (float64), already transposed to
            [sensor, tick] and padded to a common pitch, so the rows of the sensors rank r owns are one
            contiguous chunk: `all_to_all_single` sends them with no packing, and rank r selects
            median / IQR of its sensors straight over the received [rank, sensor, pitch] blocks
            (padding is a filler the select ignores) — no unpacking either.
    Step 2  all-gather of the per-sensor [median, IQR] rows -> every rank has the [N,2] table.
    Step 3  each rank normalises / smooths / maxes its own ticks; the 3-tick halo before its first
            tick comes from its predecessors (all-gather of every rank's last <=3 rows).
    Returns anomaly[local ticks] (float64).  Single process: plain local scoring."""
    rank, size = world()
    t_local, n = pred_local.shape
    if size == 1 and not (rehearse and dist.is_initialized()):
        keys = backend.keys(pred_local, gt_local, t_local)
        med_iqr = backend.select(keys, 1, n, t_local, t_local)
        return backend.smooth_max(pred_local, gt_local, med_iqr, 0, None, None)
    # (rehearse: run the full exchange with a 1-rank process group — exercises the RCCL calls on a
    # one-GPU box)
    dev = pred_local.device
    bounds = [shard_range(total_ticks, r, size) for r in range(size)]
    sens = [sensor_range(n, r, size) for r in range(size)]
    s0, s1 = sens[rank]
    n_mine = s1 - s0
    # ---- step 1
    longest = max(e - s for s, e in bounds)
    pitch = max(KEY_SLICE, (longest + KEY_SLICE - 1) // KEY_SLICE * KEY_SLICE)
    if t_local > 0:
        send = backend.keys(pred_local, gt_local, pitch)                 # [n, pitch], rows grouped by owner
    else:
        send = torch.full((n, pitch), -1, dtype=torch.int64, device=dev).view(torch.float64)   # all filler
    in_sizes = [(b - a) * pitch for a, b in sens]
    out_sizes = [n_mine * pitch] * size
    recv = torch.empty((size * n_mine * pitch,), dtype=torch.float64, device=dev)
    dist.all_to_all_single(recv, send.reshape(-1), out_sizes, in_sizes, group=group)
    if n_mine:
        my_mi = backend.select(recv, size, n_mine, pitch, total_ticks).to(torch.float64)
    else:
        my_mi = torch.empty((0, 2), dtype=torch.float64, device=dev)
    # ---- steps 2+3 in ONE all-gather: every rank publishes [its median/IQR rows (padded to the
    # largest sensor share) | its last <=3 (pred, gt) rows, right-aligned, as float64 (lossless)].
    # The halo = the 3 ticks before my first one; a shard shorter than 3 ticks makes it span several
    # predecessors.
    cap = max(b - a for a, b in sens)
    pub = torch.zeros((cap * 2 + 6 * n,), dtype=torch.float64, device=dev)
    pub[: n_mine * 2] = my_mi.reshape(-1)
    take = min(3, t_local)
    if take:
        tail = pub[cap * 2:].view(2, 3, n)
        tail[0, 3 - take:] = pred_local[t_local - take:]
        tail[1, 3 - take:] = gt_local[t_local - take:]
    gathered = [torch.empty_like(pub) for _ in range(size)]
    dist.all_gather(gathered, pub, group=group)
    med_iqr = torch.cat([g[: (b - a) * 2].view(-1, 2) for g, (a, b) in zip(gathered, sens)]).contiguous()
    first_tick = bounds[rank][0]
    halo_p = halo_g = None
    if first_tick > 0:
        rows = [torch.zeros((2, 3, n), dtype=torch.float64, device=dev)]      # ticks "before 0": never read
        for r in range(rank):
            have = min(3, bounds[r][1] - bounds[r][0])
            if have:
                rows.append(gathered[r][cap * 2:].view(2, 3, n)[:, 3 - have:])
        prev = torch.cat(rows, dim=1)[:, -3:].to(pred_local.dtype)
        halo_p, halo_g = prev[0].contiguous(), prev[1].contiguous()
    if t_local == 0:
        return torch.empty((0,), dtype=torch.float64, device=dev)
    return backend.smooth_max(pred_local, gt_local, med_iqr, first_tick, halo_p, halo_g)


class ShardedEvaluator:
    """The multi-rank eval step with its one real exchange hidden behind the forward.

    Same arithmetic as `SeriesEvaluator.forward_only()` + `distributed_anomaly()`, reorganised for one
    process per GPU over xGMI:
      * the local shard runs in chunks of `chunk` ticks; as soon as a chunk's predictions exist its
        radix keys (already transposed to [sensor, tick], rows grouped by owning rank) leave in an
        ASYNC `all_to_all_single`, so the transfers (N*T*8 bytes per rank in total) overlap the
        forward of the following chunks; only the last chunk's transfer is exposed;
      * the receive side is one [chunk, rank, my sensors, pitch] buffer = `chunks*ranks` row blocks of
        the blocked radix select — no packing or unpacking on either side;
      * every buffer, and the index tables that pick the median/IQR rows and the 3-tick halo out of the
        ONE all-gather that follows, are built once: a step issues no allocation-heavy torch code.
    `forward(start, stop)` must fill `self.pred[start:stop]`; the default runs the fused HIP forward."""

    def __init__(self, model, x_local, y_local: torch.Tensor, total_ticks: int, chunk: int = 4096,
                 backend=HipScoreBackend, forward=None, group=None, use_graph: bool = False):
        self.rank, self.size = world()
        self.use_graph, self._graphs, self._warm, self._result = bool(use_graph), None, False, None
        self.group, self.backend, self.model = group, backend, model
        self.x, self.y, self.total = x_local, y_local, total_ticks
        self.t, self.n = y_local.shape
        dev = y_local.device
        size, n = self.size, self.n
        self.bounds = [shard_range(total_ticks, r, size) for r in range(size)]
        if self.bounds[self.rank][1] - self.bounds[self.rank][0] != self.t:
            raise ValueError("y_local does not have this rank's share of the ticks (harness.shard_range)")
        self.sens = [sensor_range(n, r, size) for r in range(size)]
        s0, s1 = self.sens[self.rank]
        self.n_mine = s1 - s0
        longest = max(e - s for s, e in self.bounds)
        # every rank must agree on chunk count and pitch: both follow from the LONGEST shard
        self.pitch = max(KEY_SLICE, (min(chunk, longest) + KEY_SLICE - 1) // KEY_SLICE * KEY_SLICE)
        self.nchunks = max(1, (longest + self.pitch - 1) // self.pitch)
        self.pred = torch.empty((self.t, n), dtype=torch.float32, device=dev)
        filler = torch.full((self.nchunks, n, self.pitch), -1, dtype=torch.int64, device=dev)
        self.send = filler.view(torch.float64)          # chunks this rank has no ticks for stay all filler
        self.recv = torch.empty((self.nchunks, size, self.n_mine, self.pitch), dtype=torch.float64, device=dev)
        self.in_sizes = [(b - a) * self.pitch for a, b in self.sens]
        self.out_sizes = [self.n_mine * self.pitch] * size
        # published row: [my median/IQR rows, padded to the largest share | last 3 (pred, y) rows, right aligned]
        self.cap = max(b - a for a, b in self.sens)
        self.row = self.cap * 2 + 6 * n
        self.pub = torch.zeros((self.row,), dtype=torch.float64, device=dev)
        self.take = min(3, self.t)
        if self.take:
            self.pub[self.cap * 2:].view(2, 3, n)[1, 3 - self.take:] = y_local[self.t - self.take:].double()
        self.gathered = torch.empty((size * self.row,), dtype=torch.float64, device=dev)
        mi_idx = [r * self.row + (s - a) * 2 + h for r, (a, b) in enumerate(self.sens) for s in range(a, b)
                  for h in (0, 1)]
        self.mi_idx = torch.tensor(mi_idx, dtype=torch.int64, device=dev)
        # the 3 ticks before my first one, newest last; a shard shorter than 3 ticks makes them span ranks
        self.first_tick = self.bounds[self.rank][0]
        src = []                                        # (rank, row in its right-aligned 3-row tail)
        for r in range(self.rank - 1, -1, -1):
            have = min(3, self.bounds[r][1] - self.bounds[r][0])
            src.extend((r, 2 - j) for j in range(have))
            if len(src) >= 3:
                break
        src = (src[:3] + [(0, 0)] * 3)[:3][::-1]        # ticks before tick 0 are never read: any valid index
        halo_idx = [[[r * self.row + self.cap * 2 + which * 3 * n + j * n + c for c in range(n)] for r, j in src]
                    for which in (0, 1)]
        self.halo_idx = torch.tensor(halo_idx, dtype=torch.int64, device=dev)
        self.forward = forward if forward is not None else self._hip_forward
        self._wide = None
        # per-step buffers of the scoring kernels, allocated once (backends without the hooks allocate per call)
        self._sel_ws = self._sel_out = self._anomaly = None
        if hasattr(backend, "select_workspace") and self.n_mine:
            self._sel_ws = backend.select_workspace(self.nchunks * size, self.n_mine, self.pitch, dev)
            self._sel_out = torch.empty((self.n_mine, 2), dtype=torch.float64, device=dev)
        if hasattr(backend, "select_workspace") and self.t:
            self._anomaly = torch.empty((self.t,), dtype=torch.float64, device=dev)
        self._med_iqr = torch.empty((n, 2), dtype=torch.float64, device=dev)
        self._halo64 = torch.empty((2, 3, n), dtype=torch.float64, device=dev)
        self._halo = torch.empty((2, 3, n), dtype=torch.float32, device=dev)

    def _hip_forward(self, start, stop):
        if self._wide is None:      # range of the resident shard, looked at once
            m = self.model
            self._wide = (m.operand_range == "wide" and self.x.dtype != torch.bfloat16) or (
                m.operand_range == "auto" and m.input_exceeds_limit(self.x))
        self.model.forward_into(self.x[start:stop], self.pred[start:stop], wide=self._wide)

    # A step = nchunks x [forward + keys of the chunk | async all-to-all of its key rows] -> [select of my sensors'
    # median / IQR + my last 3 prediction rows into the published row] -> all-gather -> [median/IQR table, halo,
    # smoothing + max].  The bracketed COMPUTE segments are plain launch sequences with static arguments: with
    # `use_graph` each is captured once in a HIP graph and a step is nchunks + 2 replays and nchunks + 1
    # collectives issued from Python — against ~10 launches per chunk before (a 0.35 ms GPU step does not survive
    # ~100 us of Python per rank).  The collectives themselves stay eager: RCCL inside a captured graph is not
    # something this one-GPU box can vouch for.
    def _seg_forward(self, c):
        a, b = min(self.t, c * self.pitch), min(self.t, (c + 1) * self.pitch)
        if b > a:
            self.forward(a, b)
            self.backend.keys(self.pred[a:b], self.y[a:b], self.pitch, out=self.send[c])

    def _seg_select(self):
        if self.n_mine:
            if self._sel_ws is not None:        # the select writes straight into the row this rank publishes
                self.backend.select(self.recv.reshape(-1), self.nchunks * self.size, self.n_mine, self.pitch, self.total,
                                    ws=self._sel_ws, out=self.pub[: self.n_mine * 2].view(self.n_mine, 2))
            else:
                mi = self.backend.select(self.recv.reshape(-1), self.nchunks * self.size, self.n_mine, self.pitch,
                                         self.total)
                self.pub[: self.n_mine * 2] = mi.reshape(-1)
        if self.take:
            self.pub[self.cap * 2:].view(2, 3, self.n)[0, 3 - self.take:] = self.pred[self.t - self.take:]

    def _seg_finish(self):
        torch.index_select(self.gathered, 0, self.mi_idx, out=self._med_iqr.view(-1))
        halo_p = halo_g = None
        if self.first_tick > 0:
            torch.index_select(self.gathered, 0, self.halo_idx.view(-1), out=self._halo64.view(-1))
            self._halo.copy_(self._halo64)
            halo_p, halo_g = self._halo[0], self._halo[1]
        if self.t == 0:
            return torch.empty((0,), dtype=torch.float64, device=self.pred.device)
        extra = {"anomaly": self._anomaly} if self._anomaly is not None else {}
        return self.backend.smooth_max(self.pred, self.y, self._med_iqr, self.first_tick, halo_p, halo_g, **extra)

    def _capture_segments(self):
        """One eager step has run (plans, constants, RCCL channels are warm): capture the compute segments.  Returns
        False (and stays eager) when anything about the capture fails."""
        try:
            torch.cuda.synchronize()
            graphs = []
            for fn, args in [(self._seg_forward, (c,)) for c in range(self.nchunks)] + [(self._seg_select, ()),
                                                                                         (self._seg_finish, ())]:
                g = torch.cuda.CUDAGraph()
                with capture(g):
                    out = fn(*args)
                graphs.append(g)
            self._graphs, self._result = graphs, out
            return True
        except Exception as exc:     # noqa: BLE001 — a capture that fails must not take the evaluator down with it
            import warnings
            warnings.warn(f"ShardedEvaluator: HIP-graph capture of the compute segments failed ({exc!r}); running eagerly")
            self._graphs = None
            torch.cuda.synchronize()
            return False

    def step(self):
        graphs = self._graphs
        if self.use_graph and graphs is None and self._warm:
            self.use_graph = self._capture_segments()
            graphs = self._graphs
        works = []
        for c in range(self.nchunks):
            if graphs is not None:
                graphs[c].replay()
            else:
                self._seg_forward(c)
            works.append(dist.all_to_all_single(self.recv[c].reshape(-1), self.send[c].reshape(-1),
                                                self.out_sizes, self.in_sizes, group=self.group, async_op=True))
        for w in works:
            w.wait()
        if graphs is not None:
            graphs[self.nchunks].replay()
        else:
            self._seg_select()
        dist.all_gather_into_tensor(self.gathered, self.pub, group=self.group)
        self._warm = True
        if graphs is not None:
            graphs[self.nchunks + 1].replay()
            return self._result
        return self._seg_finish()


# --------------------------------------------------------------------------- resident-series evaluator
class SeriesEvaluator:
    """Eval forward + anomaly score over a series of T windows resident in HBM.

    One `step()` = the reference's `test()` over the series in batches of `batch` windows
    (test.py:43-62) followed by `get_full_err_scores` + max over sensors (evaluate.py:6-68,
    :131-139), with nothing leaving the device.  With `use_graph=True` the launches of a step
    are captured once in a HIP graph and replayed."""

    def __init__(self, model, x_all: torch.Tensor | None, y_all: torch.Tensor, batch: int, use_graph: bool = True,
                 want_scores: bool = False, streams: int = 3, coalesce: int = 1,
                 series: torch.Tensor | None = None):
        """`batch` = the logical minibatch of the reference's loader; `coalesce` consecutive batches
        (contiguous in the resident series) go out as ONE launch — eval results do not depend on the
        minibatch size, and launches of a few thousand windows amortise the per-workgroup prologue."""
        # `series` [N, T_raw] (datasets/TimeDataset.py:42 layout) replaces x_all: window t is
        # series[:, t : t+W], built inside the kernel — no [T, N, W] tensor
        assert (x_all is None) != (series is None), "give either the window tensor or the raw series"
        assert y_all.is_cuda and (x_all if x_all is not None else series).is_cuda
        self.series = series
        self.model, self.x, self.y, self.batch = model.eval(), x_all, y_all, batch * max(1, coalesce)
        # range of the resident data, looked at ONCE (include/gdn_hip.h "range guard"): beyond the 16-bit operand
        # range of the matrix-core kernels the whole evaluator runs on the fp32 row-gather kernels
        src0 = series if series is not None else x_all
        self.wide = (model.operand_range == "wide" and src0.dtype != torch.bfloat16) or (
            model.operand_range == "auto" and model.out_layer_num == 1 and model.input_exceeds_limit(src0))
        self.logical_batch, self.coalesce = batch, max(1, coalesce)
        self.t, self.n = y_all.shape
        dev = y_all.device
        self.pred = torch.empty((self.t, self.n), dtype=torch.float32, device=dev)
        self.ws = ops.score_workspace(self.t, self.n, dev)
        self.med_iqr = torch.empty((self.n, 2), dtype=torch.float64, device=dev)
        self.anomaly = torch.empty((self.t,), dtype=torch.float64, device=dev)
        self.scores = torch.empty((self.n, self.t), dtype=torch.float64, device=dev) if want_scores else None
        self.graph = None
        self.fgraph = None
        self.use_graph = use_graph
        # GDN_FUSE_KEYS=1: the forward's epilogue writes the scoring keys itself (gdn_forward_fused_plan_keys) and
        # the gdn_score_keys launch disappears.  Measured SLOWER and therefore off by default: the 127 scattered
        # 8-byte stores per window (row pitch 256 KB) cost the forward ~40 us per 32768 windows, the transposing
        # keys kernel 14 us (step 0.379 vs 0.352 ms).
        self.fuse_keys = os.environ.get("GDN_FUSE_KEYS", "0") == "1"
        # independent batches are launched round-robin on side streams (fork/join around the
        # forward), so consecutive launches overlap each other's ramp-up and tail.  Round 1, 4096-window launches:
        # two streams 0.634 ms/step vs 0.697 with one and 0.676 with four.  Round 3, 512-window launches of two
        # windows per workgroup (gdn_forward_dense.hip, fused_op): 72.2 M windows/s on two streams, 74.4 M on
        # three, 69.9 M on four — three is the default
        n_launch = (self.t + self.batch - 1) // self.batch
        self.side = [torch.cuda.Stream(device=dev) for _ in range(min(streams, n_launch))] if streams > 1 else []

    def _launch_forward(self, with_keys: bool = False):
        m = self.model
        # constants and the plan are built (when stale) HERE, on the caller's stream, before the fork: built
        # lazily inside the first side-stream launch, the launches on the other side streams would read them
        # unordered (first eager step after a parameter update: garbage in some windows)
        if m.out_layer_num == 1 and not m.training and not self.wide:
            src = self.series if self.series is not None else self.x
            m._plan(m._constants(), src.dtype == torch.bfloat16)
        spans = [(s, min(self.t, s + self.batch)) for s in range(0, self.t, self.batch)]
        # scoring hand-off: the forward's epilogue writes the float64 radix keys |pred - y| of its windows into
        # their columns of the [n, t] key block at the head of the scoring workspace (no gdn_score_keys launch)
        def keys(s, e):
            return (self.y[s:e], self.ws.data_ptr() + 8 * s, self.t) if with_keys else None
        if self.series is not None:
            def launch(s, e):
                m.forward_series(self.series, s, e - s, out=self.pred[s:e], keys=keys(s, e), wide=self.wide)
        else:
            def launch(s, e):
                m.forward_into(self.x[s:e], self.pred[s:e], keys=keys(s, e), wide=self.wide)
        if len(self.side) < 2:
            for s, e in spans:
                launch(s, e)
            return
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        for st in self.side:
            st.wait_event(fork)
        for i, (s, e) in enumerate(spans):
            with torch.cuda.stream(self.side[i % len(self.side)]):
                launch(s, e)
        for st in self.side:
            join = torch.cuda.Event()
            join.record(st)
            main.wait_event(join)

    def _launch_score(self, have_keys: bool = False):
        from . import _lib
        st = torch.cuda.current_stream().cuda_stream
        if have_keys:       # keys [n, t] already sit at the head of the workspace (same layout gdn_score_quantiles uses)
            _lib.call("gdn_score_select", self.ws.data_ptr(), 1, self.n, self.t, self.t,
                      self.ws.data_ptr() + 8 * self.t * self.n, self.med_iqr.data_ptr(), st)
        else:
            _lib.call("gdn_score_quantiles", self.pred.data_ptr(), self.y.data_ptr(), self.t, self.n,
                      self.ws.data_ptr(), self.med_iqr.data_ptr(), st)
        _lib.call("gdn_score_smooth_max", self.pred.data_ptr(), self.y.data_ptr(), self.med_iqr.data_ptr(),
                  self.t, self.n, 0, None, None,
                  None if self.scores is None else self.scores.data_ptr(), self.anomaly.data_ptr(), st)

    def _launch_all(self):
        src = self.series if self.series is not None else self.x
        fuse = self.fuse_keys and not self.wide and self.model.fused_keys_supported(src.dtype == torch.bfloat16)
        self._launch_forward(with_keys=fuse)
        self._launch_score(have_keys=fuse)

    def _fresh(self):
        """Captured graphs bake in the pointers of the model's folded constants: drop them when a
        parameter changed since the capture."""
        key = self.model._constants().key
        if key != getattr(self, "_graph_key", None):
            self.graph = self.fgraph = None
            if getattr(self, "_graph_key", None) is not None:      # the x limit follows the parameters: look again
                src0 = self.series if self.series is not None else self.x
                m = self.model
                self.wide = (m.operand_range == "wide" and src0.dtype != torch.bfloat16) or (
                    m.operand_range == "auto" and m.out_layer_num == 1 and m.input_exceeds_limit(src0))
            self._graph_key = key

    def _capture(self, fn):
        self.model._constants()                      # build graph/constants outside the capture
        fn()                                         # warm-up (occupancy queries, attributes)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with capture(g):
            fn()
        return g

    def forward_only(self):
        """Forward launches only (multi-GPU: scoring then goes through distributed_anomaly)."""
        if not self.use_graph:
            self._launch_forward()
            return self.pred
        self._fresh()
        if self.fgraph is None:
            self.fgraph = self._capture(self._launch_forward)
        self.fgraph.replay()
        return self.pred

    def step(self):
        if not self.use_graph:
            self._launch_all()
            return self.anomaly
        self._fresh()
        if self.graph is None:
            self.graph = self._capture(self._launch_all)
        self.graph.replay()
        return self.anomaly


# --------------------------------------------------------------------------- graphed train step
class AutogradTrainStep:
    """One optimisation step of the reference's train() (train.py:52-66: zero_grad, forward, MSE,
    backward, Adam) captured once in a HIP graph and replayed per minibatch — the form that goes through
    torch autograd and torch.optim.Adam(fused=True): used when `NativeTrainStep` does not apply
    (a custom `model.dp` module, an injected graph, an OutLayer or a shape the training kernels do not take).

    At the reference's batch sizes a step is ~30 launches of a few microseconds each, so issuing
    them from Python costs more than running them; a replayed graph removes that.  The graph
    covers the per-step rebuild of the sensor graph and the folded attention terms (they depend
    on the parameters Adam has just changed), the HIP forward/backward of the graph layer and of
    the train-mode head, the fused MSE loss + gradient kernel, torch's dropout draw and a fused,
    capturable Adam.  With more than one rank the step is
    two graphs around ONE eager op, the all-reduce of the flat gradient bucket (packing is the
    tail of the first graph, averaging + unpacking the head of the second).

    `x` / `y` are the static input buffers: copy each minibatch into them, call `step()`, read
    `loss` (a device scalar) whenever convenient."""

    def __init__(self, model, batch: int, lr: float = 1e-3, weight_decay: float = 0.0, use_graph: bool = True,
                 split: bool | None = None, wide: bool = False):
        self.wide = bool(wide)
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise RuntimeError("GraphedTrainStep needs the model on a HIP device")
        dev = p0.device
        n, w = model.embedding.weight.shape[0], model.gnn_layers[0].gnn.lin.weight.shape[1]
        self.model = model.train()
        self.x = torch.zeros((batch, n, w), dtype=torch.float32, device=dev)
        self.y = torch.zeros((batch, n), dtype=torch.float32, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self._d_out = torch.empty((batch, n), dtype=torch.float32, device=dev)
        self._mse_ws = ops.mse_workspace(dev)
        # fused: one multi-tensor launch for all 13 parameters instead of ~40 small ones
        self.optimizer = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay, capturable=True,
                                          fused=True)
        self.use_graph = use_graph
        self._graphs = None
        # two graphs around the (eager) gradient all-reduce; forced on by `split=True` for rehearsal
        self._split = world()[1] > 1 if split is None else bool(split)
        self._flat = None                                # static gradient bucket (split mode)
        import os
        self._torch_mse = bool(os.environ.get("GDN_TORCH_MSE"))
        self._dbg = None                                 # diagnostic snapshot buffers (tools/probe_mse_replay.py)

    # the two halves of a step; `loss` is written in place so it survives replays
    def _forward_backward(self):
        self.optimizer.zero_grad(set_to_none=True)      # backward then writes fresh gradients: no fill, no add
        before = self.model.operand_range
        self.model.operand_range = "wide" if self.wide else "narrow"      # no host check inside a captured step
        try:
            out = self.model(self.x, None)
        finally:
            self.model.operand_range = before
        if self._torch_mse:     # diagnostic (tools/probe_mse_replay.py): the round-1 form with torch's reduction
            loss = F.mse_loss(out, self.y, reduction="mean")
            loss.backward()
            self.loss.copy_(loss.detach())
            if self._dbg is not None:
                self._dbg["out"].copy_(out.detach())
                self._dbg["loss_raw"].copy_(loss.detach())
        else:
            # loss + its gradient in one launch (train.py:20-23, :72); autograd starts from d_out
            ops.mse_loss_grad(out.detach(), self.y, self._mse_ws, loss=self.loss, d_out=self._d_out)
            out.backward(self._d_out)
        if self._dbg is not None:
            for name, prm in self.model.named_parameters():
                self._dbg["g/" + name].copy_(prm.grad)
        if self._split:                                  # the bucket is part of the first graph
            if self._flat is None:
                self._flat = pack_gradients(self.model)
            else:
                pack_gradients(self.model, self._flat)

    def _update(self):
        if self._split:                                  # ... and its unpacking part of the second
            unpack_gradients(self.model, self._flat, world()[1])
        self.optimizer.step()

    def _all_reduce(self):
        if world()[1] > 1:
            dist.all_reduce(self._flat, op=dist.ReduceOp.SUM)

    def _snapshot(self):
        tensors = list(self.model.parameters()) + list(self.model.buffers())
        return tensors, [t.detach().clone() for t in tensors]

    def _capture(self):
        """Warm up on a side stream, capture, then put parameters, BN statistics and the Adam
        state back to what they were: capturing must not train the model."""
        tensors, saved = self._snapshot()
        side = torch.cuda.Stream(device=self.x.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self._forward_backward()
                self._all_reduce()
                self._update()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        graphs = []
        if self._split:
            for fn in (self._forward_backward, self._update):
                g = torch.cuda.CUDAGraph()
                with capture(g, pool=pool):
                    fn()
                graphs.append(g)
        else:
            g = torch.cuda.CUDAGraph()
            with capture(g, pool=pool):
                self._forward_backward()
                self._update()
            graphs.append(g)
        with torch.no_grad():
            for t, s in zip(tensors, saved):
                t.copy_(s)
            for state in self.optimizer.state.values():
                for v in state.values():
                    if torch.is_tensor(v):
                        v.zero_()
        self._graphs = graphs

    def step(self):
        if not self.use_graph:
            self._forward_backward()
            if self._split:
                self._all_reduce()
            self._update()
            return self.loss
        if self._graphs is None:
            self._capture()
        self._graphs[0].replay()
        if self._split:
            self._all_reduce()                           # the only eager op of a multi-rank step
            self._graphs[1].replay()
        # a replay writes parameters and BatchNorm statistics through raw pointers: no Python forward
        # runs and no version counter moves, so the model's cached eval constants (sensor graph,
        # attention terms, BatchNorm folds) must be dropped here or eval after training serves stale ones
        self.model.invalidate_constants()
        return self.loss


def flat_layout(params):
    """Slots of the flat parameter / gradient / optimizer-state buffers: [(offset, count)] per parameter, every
    slot on a 16-byte boundary (the kernels read rows as float4), and the padded total."""
    slices, total = [], 0
    for p in params:
        slices.append((total, p.numel()))
        total += (p.numel() + 3) & ~3
    return slices, total


def all_reduce_flat(flat_g, group=None):
    """The one collective of a data-parallel training step: the flat gradient buffer is summed over the ranks
    as it stands (the 1/ranks is applied by gdn_adam_step's grad_scale).  No-op in a single process."""
    if world()[1] > 1:
        dist.all_reduce(flat_g, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / max(1, world()[1])


class _FlatAdam:
    """What `train()` needs from an optimizer for the batches that do not go through the captured step (the
    ragged last batch of an epoch): zero_grad() / step() over the SAME flat Adam state as the native step."""

    def __init__(self, owner):
        self.owner = owner

    def zero_grad(self, set_to_none: bool = True):
        for p in self.owner.params:
            p.grad = None

    def step(self, grad_scale: float = 1.0):
        """`p.grad` -> the flat gradient buffer -> gdn_adam_step.  The gradients are taken AS THEY ARE: train()
        has already averaged them over the ranks (sync_gradients), so the 1/ranks the captured step folds into
        the optimizer kernel must not be applied a second time here."""
        o = self.owner
        with torch.no_grad():
            for p, (off, cnt) in zip(o.params, o.slices):
                if p.grad is not None:
                    o.flat_g[off:off + cnt].copy_(p.grad.reshape(-1))
                else:
                    o.flat_g[off:off + cnt].zero_()
        o._adam(grad_scale)
        o.model.invalidate_constants()


class NativeTrainStep:
    """the training step of train.py:63-79 with nothing but this library's kernels between the
    input batch and the updated parameters — no autograd graph, no torch optimizer, no mask tensor.

      * every parameter is a VIEW into one flat fp32 buffer; gradients, Adam's exp_avg / exp_avg_sq are flat
        buffers of the same layout;
      * forward: sensor graph + folded attention terms (they depend on the parameters of this step) ->
        projection -> attention/aggregate (keeps alpha) -> train-mode head with the dropout mask DRAWN in the
        kernels from (seed, step) -> fused MSE loss + its gradient;
      * backward: every kernel writes its parameter gradients straight into their slots of the flat gradient
        buffer (the two shares of the embedding gradient meet in one slot: gdn_terms_bwd_acc), so there is
        no zero_grad, no packing and no unpacking;
      * with several ranks the flat gradient buffer IS the bucket of the one all-reduce (sum; the 1/ranks is
        folded into the optimizer kernel);
      * gdn_adam_step: one launch over the flat buffers, torch.optim.Adam's update.

    The launches of a step are captured once in a HIP graph (two graphs around the all-reduce with >1 rank).
    `x` / `y` are the static input buffers; `loss` a device scalar.  BatchNorm uses per-rank batch
    statistics (standard DDP)."""

    BETAS, EPS = (0.9, 0.999), 1e-8

    @staticmethod
    def applicable(model) -> bool:
        import torch.nn as nn
        if not (type(model.dp) is nn.Dropout and model._hip_train_head_ok() and model.injected_graph is None
                and next(model.parameters()).is_cuda):
            return False
        n, d = model.embedding.weight.shape
        w = model.gnn_layers[0].gnn.lin.weight.shape[1]
        from . import _lib
        if not _lib.load().gdn_train_supported(n, w, d, model.topk):     # shape outside the training kernels
            return False
        # out_layer_num > 1: the OutLayer MLP must be one gdn_mlp_train_fwd takes (any row count > 1)
        return model.out_layer_num == 1 or ops.mlp_train_supported(model.out_layer, model.embedding.weight.shape[1], 2)

    def __init__(self, model, batch: int, lr: float = 1e-3, weight_decay: float = 0.0, use_graph: bool = True,
                 split: bool | None = None, seed: int | None = None, wide: bool = False):
        from . import _lib
        self._lib = _lib
        # inputs beyond the 16-bit operand range of the matrix-core kernels: the `_wide` (fp32 row-gather) entry
        # points throughout, decided by the caller from its data (harness.train: first batch / config["wide"])
        self.wide = bool(wide)
        self._sfx = "_wide" if self.wide else ""
        self.model = model.train()
        dev = next(model.parameters()).device
        self.lr, self.wd = float(lr), float(weight_decay)
        self.params = list(model.parameters())
        # flat parameter buffer; every slot starts on a 16-byte boundary (the kernels read rows as float4)
        self.slices, total = flat_layout(self.params)
        self.count = total
        self.flat_p = torch.zeros((total,), dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, (off, cnt) in zip(self.params, self.slices):
                self.flat_p[off:off + cnt].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + cnt].view(p.shape)
        model.invalidate_constants()
        model._key_tensors = None
        self.flat_g = torch.zeros_like(self.flat_p)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        if seed is None:
            # derived from torch's seed WITHOUT drawing from the global generator: the loaders' shuffle stream stays
            # the reference's (main.py:221-228 seeds, main.py:128-148 draws)
            seed = (int(torch.initial_seed()) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & (2 ** 62 - 1)
        self.state = torch.tensor([seed, 0], dtype=torch.int64, device=dev)     # {dropout seed, steps taken}
        self.optimizer = _FlatAdam(self)

        gnn, layer = model.gnn_layers[0].gnn, model.gnn_layers[0]
        n, d = model.embedding.weight.shape
        w, k = gnn.lin.weight.shape[1], model.topk
        self.n, self.d, self.w, self.k, self.batch = n, d, w, k, batch
        lib = _lib.load()
        pitch = ops.nbr_pitch(k)
        f32 = dict(dtype=torch.float32, device=dev)
        self.x = torch.zeros((batch, n, w), **f32)
        self.y = torch.zeros((batch, n), **f32)
        self.loss = torch.zeros((), **f32)
        bn_rows = batch * n
        self.ws = dict(
            topk=torch.empty((n, k), dtype=torch.int64, device=dev),
            nbr=torch.empty((n, pitch), dtype=torch.uint16, device=dev),
            deg=torch.empty((n,), dtype=torch.int32, device=dev),
            rent=torch.empty((n, (n + 15) & ~15), dtype=torch.int32, device=dev),
            rlen=torch.empty((n,), dtype=torch.int32, device=dev),
            terms=torch.empty((128 + 2 * n,), **f32),
            xlin=torch.empty((bn_rows, d), **f32), s_i=torch.empty((bn_rows,), **f32), s_j=torch.empty((bn_rows,), **f32),
            z=torch.empty((bn_rows, d), **f32), alpha=torch.empty((bn_rows, pitch), **f32),
            out=torch.empty((batch, n), **f32), d_out=torch.empty((batch, n), **f32),
            # zero-filled ONCE: the head kernels are told so (buffers_zeroed = 1) and leave them zeroed after every backward
            stats=torch.zeros((lib.gdn_head_train_stats_bytes(d) // 8,), dtype=torch.float64, device=dev),
            head_ws=torch.zeros((lib.gdn_head_train_workspace_bytes(n, d) // 8,), dtype=torch.float64, device=dev),
            d_z=torch.empty((bn_rows, d), **f32), d_xlin=torch.empty((bn_rows, d), **f32),
            d_si=torch.empty((bn_rows,), **f32), d_sj=torch.empty((bn_rows,), **f32),
            proj_ws=torch.empty((lib.gdn_project_bwd_workspace_bytes(n, w, d) // 4,), **f32),
            # [ticket (zero, left zero by every call) | d_bias rows | d_pi tables of shapes beyond LDS, e.g. 512 sensors]
            bwd_ws=torch.zeros((lib.gdn_attn_aggregate_bwd_workspace_bytes(batch, n, d, k) // 4,), **f32),
            d_a=torch.empty((128,), **f32), d_c=torch.empty((2 * n,), **f32),
            mse_ws=ops.mse_workspace(dev),
            head_mse_ws=torch.zeros((lib.gdn_head_mse_workspace_bytes() // 8,), dtype=torch.float64, device=dev),
        )
        self._mlp = None
        if model.out_layer_num > 1:
            hidden, _last = ops.mlp_train_layers(model.out_layer)
            h, layers = hidden[0][0].out_features, len(hidden) + 1
            self._mlp = (h, layers, [bn for _lin, bn in hidden])
            self.ws.update(
                act=torch.empty((bn_rows, d), **f32), d_act=torch.empty((bn_rows, d), **f32),
                mlp_saved=torch.empty((lib.gdn_mlp_train_saved_bytes(bn_rows, d, h, layers),), dtype=torch.uint8, device=dev),
                mlp_ws=torch.empty((lib.gdn_mlp_train_workspace_bytes(bn_rows, d, h, layers),), dtype=torch.uint8, device=dev))
        self._need_reverse = self.wide or bool(lib.gdn_attn_aggregate_bwd_uses_reverse(n, d, k))
        self._side = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        self._fork = os.environ.get("GDN_TRAIN_FORK", "0") == "1"
        # GDN_FUSE_MSE=1: loss + d_out from the head's last forward pass (gdn_head_train_fwd_rng_mse) instead of
        # the gdn_mse_loss_grad launch — measured SLOWER (0.198 vs 0.193 ms: 512 workgroups each pay the block
        # reductions and the ticket), so off by default
        self._fuse_mse = os.environ.get("GDN_FUSE_MSE", "0") == "1"
        self.use_graph = use_graph
        self._graphs = None
        self._split = world()[1] > 1 if split is None else bool(split)
        name_of = {id(p): name for name, p in model.named_parameters()}
        self._off = {name_of[id(p)]: off for p, (off, _c) in zip(self.params, self.slices)}
        self._layer, self._gnn = layer, gnn

    # pointers -------------------------------------------------------------------------------------------------
    def _pp(self, name):
        return self.flat_p.data_ptr() + 4 * self._off[name]

    def _gp(self, name):
        return self.flat_g.data_ptr() + 4 * self._off[name]

    def _bn_run(self, bn):
        if not bn.track_running_stats or bn.running_mean is None:
            return 0.0, None, None, None
        return float(bn.momentum), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr()

    # the two halves of a step ---------------------------------------------------------------------------------
    def _forward_backward(self):
        call, ws, m = self._lib.call, self.ws, self.model
        st = torch.cuda.current_stream().cuda_stream
        n, d, w, k, b = self.n, self.d, self.w, self.k, self.batch
        P, G = self._pp, self._gp
        g = "gnn_layers.0.gnn."
        bn1, bn2 = self._layer.bn, m.bn_outlayer_in
        p_drop = float(m.dp.p) if m.dp.training else 0.0
        rng = self.state.data_ptr()
        pt = {key: t.data_ptr() for key, t in ws.items()}
        # graph + folded attention terms of THIS step's parameters (models/GDN.py:145-165, graph_layer.py:94-104).
        # GDN_TRAIN_FORK=1: three independent chains forked onto side streams (parallel branches of the captured
        # graph): [top-k graph] | [folded terms -> projection] | [reverse lists, needed by the backward only] —
        # measured SLOWER than the serial chain inside a HIP graph (0.235 vs 0.218 ms), so off by default
        main = torch.cuda.current_stream()
        if self._fork:
            side, side2 = self._side
            side.wait_stream(main)
            s2 = side.cuda_stream
            call("gdn_node_terms", P(g + "lin.weight"), P(g + "att_i"), P(g + "att_j"), P(g + "att_em_i"), P(g + "att_em_j"),
                 P("embedding.weight"), n, d, w, pt["terms"], s2)
            call("gdn_project_fwd" + self._sfx, self.x.data_ptr(), P(g + "lin.weight"), pt["terms"], b, n, w, d, pt["xlin"],
                 pt["s_i"], pt["s_j"], s2)
            call("gdn_topk_graph", P("embedding.weight"), n, d, k, pt["topk"], pt["nbr"], pt["deg"], None, st)
            side2.wait_stream(main)
            call("gdn_graph_reverse", pt["nbr"], pt["deg"], n, k, pt["rent"], pt["rlen"], side2.cuda_stream)
            main.wait_stream(side)
        else:
            side2 = main
            # graph rows and folded terms in one launch (independent work, one launch less on the critical path)
            call("gdn_topk_graph_terms", P("embedding.weight"), n, d, k, pt["topk"], pt["nbr"], pt["deg"], P(g + "lin.weight"),
                 P(g + "att_i"), P(g + "att_j"), P(g + "att_em_i"), P(g + "att_em_j"), w, pt["terms"], st)
            if self._need_reverse:       # only the row-gather backward reads the reverse lists
                call("gdn_graph_reverse", pt["nbr"], pt["deg"], n, k, pt["rent"], pt["rlen"], st)
            call("gdn_project_fwd" + self._sfx, self.x.data_ptr(), P(g + "lin.weight"), pt["terms"], b, n, w, d, pt["xlin"],
                 pt["s_i"], pt["s_j"], st)
        call("gdn_attn_aggregate_fwd" + self._sfx, pt["xlin"], pt["s_i"], pt["s_j"], pt["nbr"], pt["deg"], P(g + "bias"), b, n, d, k,
             pt["z"], pt["alpha"], st)
        m1, rm1, rv1, nb1 = self._bn_run(bn1)
        m2, rm2, rv2, nb2 = self._bn_run(bn2)
        bnp = (P("gnn_layers.0.bn.weight"), P("gnn_layers.0.bn.bias"), P("bn_outlayer_in.weight"), P("bn_outlayer_in.bias"))
        bng = (G("gnn_layers.0.bn.weight"), G("gnn_layers.0.bn.bias"), G("bn_outlayer_in.weight"), G("bn_outlayer_in.bias"))
        eps = (float(bn1.eps), float(bn2.eps))
        run = (m1, m2, rm1, rv1, nb1, rm2, rv2, nb2)
        if self._mlp is None:
            lw, lb = "out_layer.mlp.0.weight", "out_layer.mlp.0.bias"
            if self._fuse_mse:   # the loss and its gradient come out of the head's last forward pass
                call("gdn_head_train_fwd_rng_mse", pt["z"], P("embedding.weight"), *bnp, P(lw), P(lb), rng, p_drop, b, n, d,
                     *eps, *run, pt["stats"], pt["out"], self.y.data_ptr(), pt["head_mse_ws"], self.loss.data_ptr(),
                     pt["d_out"], 1, st)
            else:
                call("gdn_head_train_fwd_rng", pt["z"], P("embedding.weight"), *bnp, P(lw), P(lb), rng, p_drop, b, n, d,
                     *eps, *run, pt["stats"], pt["out"], 1, st)
        else:
            # out_layer_num > 1: head passes up to the dropped-out activation, then the MLP on the matrix cores
            h, layers, bns = self._mlp
            arr = lambda ptrs: (ctypes.c_void_p * len(ptrs))(*ptrs)
            names = [f"out_layer.mlp.{3 * l + j}.{kind}" for l in range(layers - 1) for j, kind in
                     ((0, "weight"), (0, "bias"), (1, "weight"), (1, "bias"))]
            lw, lb = f"out_layer.mlp.{3 * (layers - 1)}.weight", f"out_layer.mlp.{3 * (layers - 1)}.bias"
            runs = [self._bn_run(bn) for bn in bns]
            call("gdn_head_train_fwd_act", pt["z"], P("embedding.weight"), *bnp, None, None, 1.0, rng, p_drop, b, n, d,
                 *eps, *run, pt["stats"], pt["act"], 1, st)
            call("gdn_mlp_train_fwd", pt["act"], arr([P(nm) for nm in names]),
                 arr([q for r in runs for q in (r[1], r[2])]), arr([r[3] for r in runs]),
                 (ctypes.c_float * len(bns))(*[float(bn.eps) for bn in bns]),
                 (ctypes.c_float * len(bns))(*[r[0] for r in runs]), P(lw), P(lb), b * n, d, h, layers,
                 pt["mlp_saved"], pt["mlp_ws"], pt["out"], st)
        if self._mlp is not None or not self._fuse_mse:
            call("gdn_mse_loss_grad", pt["out"], self.y.data_ptr(), b * n, pt["mse_ws"], self.loss.data_ptr(), pt["d_out"], st)
        # backward: gradients land in their slots of flat_g
        if self._mlp is None:
            # (buffers_zeroed | 2: the head's small finishing reduction runs in the combined tail launch below)
            call("gdn_head_train_bwd_rng", pt["d_out"], pt["z"], P("embedding.weight"), *bnp, P(lw), rng, p_drop,
                 pt["stats"], b, n, d, *eps, pt["head_ws"], pt["d_z"], G("embedding.weight"), *bng, G(lw), G(lb), 3, st)
        else:
            call("gdn_mlp_train_bwd", pt["d_out"], pt["act"], arr([P(nm) for nm in names]), P(lw), b * n, d, h, layers,
                 pt["mlp_saved"], pt["mlp_ws"], arr([G(nm) for nm in names]), G(lw), G(lb), pt["d_act"], st)
            call("gdn_head_train_bwd_act", pt["d_act"], pt["z"], P("embedding.weight"), *bnp, None, None, 1.0, rng,
                 p_drop, pt["stats"], b, n, d, *eps, pt["head_ws"], pt["d_z"], G("embedding.weight"), *bng, 1, st)
        main.wait_stream(side2)                      # reverse lists
        call("gdn_attn_aggregate_bwd" + self._sfx, pt["d_z"], pt["xlin"], pt["alpha"], pt["s_i"], pt["s_j"], pt["nbr"], pt["rent"],
             pt["rlen"], b, n, d, k, pt["d_xlin"], pt["d_si"], pt["d_sj"], G(g + "bias"), pt["bwd_ws"], st)
        if self._mlp is None:
            # partial rows only; their reduction and the head's finishing reduction share ONE launch
            rows = ctypes.c_int(0)
            call("gdn_project_bwd_partials", self.x.data_ptr(), pt["d_xlin"], pt["d_si"], pt["d_sj"], b, n, w, d,
                 pt["proj_ws"], ctypes.byref(rows), st)
            call("gdn_train_finish", pt["head_ws"], pt["stats"], 1, b, n, d, G("embedding.weight"), *bng, G(lw), G(lb),
                 pt["proj_ws"], rows.value, w, G(g + "lin.weight"), pt["d_a"], pt["d_c"], st)
        else:
            call("gdn_project_bwd", self.x.data_ptr(), pt["d_xlin"], pt["d_si"], pt["d_sj"], b, n, w, d, pt["proj_ws"],
                 G(g + "lin.weight"), pt["d_a"], pt["d_c"], st)
        call("gdn_terms_bwd_acc", P(g + "lin.weight"), P(g + "att_i"), P(g + "att_j"), P(g + "att_em_i"), P(g + "att_em_j"),
             P("embedding.weight"), pt["d_a"], pt["d_c"], n, d, w, G(g + "lin.weight"), G(g + "att_i"), G(g + "att_j"),
             G(g + "att_em_i"), G(g + "att_em_j"), G("embedding.weight"), 1, st)

    def _all_reduce(self):
        all_reduce_flat(self.flat_g)

    def _adam(self, grad_scale: float | None = None):
        """gdn_adam_step over the flat buffers; `grad_scale` None = 1/ranks (flat_g holds the all-reduced SUM)."""
        if grad_scale is None:
            grad_scale = 1.0 / max(1, world()[1])
        self._lib.call("gdn_adam_step", self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                       self.exp_avg_sq.data_ptr(), self.state.data_ptr() + 8, self.count, self.lr, self.BETAS[0],
                       self.BETAS[1], self.EPS, self.wd, float(grad_scale), 0, 0,
                       torch.cuda.current_stream().cuda_stream)

    def _capture(self):
        """Warm up (lazy attribute / occupancy queries) and capture; the warm-up steps are undone: capturing must
        not train the model."""
        tensors = [self.flat_p, self.exp_avg, self.exp_avg_sq, self.state] + list(self.model.buffers())
        saved = [t.detach().clone() for t in tensors]
        for _ in range(2):
            self._forward_backward()
            self._all_reduce()
            self._adam()
        torch.cuda.synchronize()
        with torch.no_grad():
            for t, s_ in zip(tensors, saved):
                t.copy_(s_)
            self.flat_g.zero_()
        torch.cuda.synchronize()
        graphs = []
        if self._split:
            for fn in (self._forward_backward, self._adam):
                g_ = torch.cuda.CUDAGraph()
                with capture(g_):
                    fn()
                graphs.append(g_)
        else:
            g_ = torch.cuda.CUDAGraph()
            with capture(g_):
                self._forward_backward()
                self._adam()
            graphs.append(g_)
        self._graphs = graphs

    def step(self):
        if not self.use_graph:
            self._forward_backward()
            if self._split:
                self._all_reduce()
            self._adam()
        else:
            if self._graphs is None:
                self._capture()
            self._graphs[0].replay()
            if self._split:
                self._all_reduce()
                self._graphs[1].replay()
        self.model.invalidate_constants()       # parameters moved under the model's cached eval constants
        return self.loss


def GraphedTrainStep(model, batch: int, lr: float = 1e-3, weight_decay: float = 0.0, use_graph: bool = True,
                     split: bool | None = None, native: bool | None = None, wide: bool | None = None):
    """The captured training step: `NativeTrainStep` when the model and its shape allow it (plain nn.Dropout, an
    OutLayer and a sensor count the training kernels take), else `AutogradTrainStep`.  `native=False` forces the
    autograd form.  `wide`: the inputs exceed the 16-bit operand range (None: model.operand_range == "wide")."""
    if native is None:
        native = NativeTrainStep.applicable(model)
    if wide is None:
        wide = getattr(model, "operand_range", "auto") == "wide"
    if native:
        return NativeTrainStep(model, batch, lr=lr, weight_decay=weight_decay, use_graph=use_graph, split=split,
                               wide=wide)
    return AutogradTrainStep(model, batch, lr=lr, weight_decay=weight_decay, use_graph=use_graph, split=split,
                             wide=wide)
