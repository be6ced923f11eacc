#!/usr/bin/env python3
"""Headline benchmark: sliding-windows/sec, eval forward + anomaly score, SWaT-shape
(BASELINE.json metric; workload = configs[2]: 127 sensors, top-k 30, W=15, D=64, batch 512).
The series is resident, so by default all 64 consecutive 512-window minibatches of a step go out as ONE launch
(`--coalesce 8 --streams 2` was round 1's form: 5 % slower — every workgroup pays its prologue per launch;
`--coalesce 1` launches per minibatch; `value_per_batch_launches` reports that variant too).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One STEP = the reference's eval flow over a series of T windows resident in HBM on every rank
(weak scaling: T per rank is fixed): test.py's loop in launches of `batch` windows (test.py:43-62)
followed by evaluate.get_full_err_scores + the max over sensors (evaluate.py:6-68,131-139).
N>1: windows are sharded contiguously, no collective in the forward; the scoring needs each
sensor's order statistics over ALL ticks: one all-to-all by sensor + an all-gather of the [N,2]
median/IQR table (gdn_amd/harness.distributed_anomaly).

Prints ONE JSON line on rank 0.  Extra objects: `roofline` (the gather-aggregate kernel K8 the
north star names, measured live with HIP events in a staged-pipeline leg of this same process),
`roofline_fused` (the fused kernel that dominates the timed region), `cpu_baseline` (the oracle,
op-faithful CPU port, on a bounded sample; N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SENSORS, WINDOW, TOPK, DIM = 127, 15, 30, 64
# --workload: the headline is BASELINE configs[2]; configs[4] (the WADI-shape stress case) is a second line
WORKLOADS = {
    "swat": dict(n=127, w=15, k=30, label="BASELINE configs[2]: SWaT-shape eval forward + anomaly score"),
    "config4": dict(n=512, w=30, k=64, label="BASELINE configs[4]: WADI-shape stress (512 sensors, top-k 64, W=30) "
                                              "eval forward + anomaly score"),
}


def build_model(device):
    from gdn_amd import GDN
    torch.manual_seed(0)
    model = GDN([torch.zeros((2, 1), dtype=torch.long)], N_SENSORS, dim=DIM, input_dim=WINDOW, topk=TOPK)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():   # perturb what the reference zero-/identity-initialises
        gnn = model.gnn_layers[0].gnn
        for t in (gnn.att_em_i, gnn.att_em_j):
            t.copy_(torch.rand(t.shape, generator=g) * 0.2 - 0.1)
        for bn in (model.gnn_layers[0].bn, model.bn_outlayer_in):
            bn.running_mean.copy_(torch.randn(bn.running_mean.shape, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(bn.running_var.shape, generator=g) + 0.5)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return model.to(device).eval(), params


def event_time_launches(launch, count, settle_ms=60.0):
    """Average device time per launch: `count` back-to-back launches on the launch stream between two
    HIP events (the queue never drains, so this is kernel time + the ~1.5 us dependent-launch gap; an
    event pair per launch adds ~5 us of its own and was dropped).  Three such runs, recorded in-stream
    with no host synchronisation in between; (mean, best) in us.

    The runs are preceded, in the same stream and without a gap, by ~`settle_ms` of the same launches:
    this GPU needs ~50 ms of sustained load after an idle period to reach its steady clocks
    (tools/probe_ramp.py: the same graph replay takes 0.84 ms right after a synchronize and 0.67 ms
    sixty replays later), and a kernel's roofline fraction is a statement about steady state."""
    probe0, probe1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    probe0.record()
    for i in range(4):
        launch(i)
    probe1.record()
    torch.cuda.synchronize()
    per_launch_ms = max(probe0.elapsed_time(probe1) / 4, 1e-3)
    for i in range(int(settle_ms / per_launch_ms) + 1):
        launch(i)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    marks[0].record()
    for r in range(3):
        for i in range(count):
            launch(i)
        marks[r + 1].record()
    torch.cuda.synchronize()
    runs = [1e3 * marks[r].elapsed_time(marks[r + 1]) / count for r in range(3)]
    return sum(runs) / len(runs), min(runs)


def pmc_traffic(kernel_key, batch):
    """HBM bytes per launch from the committed rocprofv3 PMC summaries (profiles/r03_pmc_traffic.json, else round
    2's: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied) for this launch size, or None."""
    if N_SENSORS != 127:
        kernel_key = f"{kernel_key}_n{N_SENSORS}_d{DIM}"
    for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)["per_launch"][kernel_key][str(batch)]
            return int(rec["fetch_bytes"] + rec["write_bytes"])
        except (OSError, KeyError, ValueError):
            continue
    return None


def k8_roofline(model, x, batch, launches, storage="fp32"):
    """Staged eval pipeline leg (project -> attention/aggregate -> head); returns the roofline object of
    the gather-aggregate kernel at this launch size.  storage = "fp32" | "bf16" (x, xlin, z in HBM)."""
    from gdn_amd import _lib, ops
    c = model._constants()
    gnn = model.gnn_layers[0].gnn
    lin = model.out_layer.mlp[0]
    emb = model.embedding.weight
    sfx = "_bf16" if storage == "bf16" else ""
    esz = 2 if storage == "bf16" else 4
    if storage == "bf16" and not _lib.load().gdn_fused_plan_bytes(N_SENSORS, WINDOW, 64, TOPK, 1):
        return None      # the staged bf16-storage kernels exist on the matrix-core path only (n <= 127)
    xs = x[:batch].bfloat16() if storage == "bf16" else x[:batch]
    xlin, s_i, s_j = ops.project_fwd(xs, gnn.lin.weight, c.terms)
    z = torch.empty_like(xlin)
    st = torch.cuda.current_stream().cuda_stream
    nbr = c.graph.nbr_ordered()      # what ops.attn_aggregate_fwd hands the kernel when alpha is not asked for

    def k8(_i):
        _lib.call("gdn_attn_aggregate_fwd" + sfx, xlin.data_ptr(), s_i.data_ptr(), s_j.data_ptr(),
                  nbr.data_ptr(), c.graph.deg.data_ptr(), gnn.bias.data_ptr(),
                  batch, N_SENSORS, DIM, TOPK, z.data_ptr(), None, st)

    for i in range(3):
        k8(i)
    mean_us, med_us = event_time_launches(k8, launches)
    out = torch.empty((batch, N_SENSORS), device=x.device)

    def proj(_i):
        _lib.call("gdn_project_fwd" + sfx, xs.data_ptr(), gnn.lin.weight.data_ptr(), c.terms.data_ptr(), batch,
                  N_SENSORS, WINDOW, DIM, xlin.data_ptr(), s_i.data_ptr(), s_j.data_ptr(), st)

    def head(_i):
        _lib.call("gdn_head_fwd" + sfx, z.data_ptr(), emb.data_ptr(), c.bn1.data_ptr(), c.bn2.data_ptr(),
                  lin.weight.data_ptr(), lin.bias.data_ptr(), batch, N_SENSORS, DIM, out.data_ptr(), None, st)
    proj_us, _ = event_time_launches(proj, launches)
    head_us, _ = event_time_launches(head, launches)
    # read xlin once + write z once + the neighbour lists once (alpha fused, not stored)
    alg_bytes = 2 * batch * N_SENSORS * DIM * esz + N_SENSORS * c.graph.pitch * 2
    achieved = alg_bytes / (mean_us * 1e-6) / 1e9
    how = "on the matrix cores" if N_SENSORS <= 127 and DIM == 64 else "fp32 row gather (n > 127: VALU family)"
    return {"kernel": f"gdn_attn_aggregate_fwd{sfx} (K8 gather-aggregate {how}, staged eval leg, "
                      f"{storage} storage of xlin and z)",
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("k8" + sfx, batch),
            "launch_us": round(mean_us, 2), "launch_us_best": round(med_us, 2), "batch": batch,
            "algorithmic_bytes_per_launch": alg_bytes,
            "staged_pipeline_us": {"project": round(proj_us, 2), "attn_aggregate": round(mean_us, 2),
                                   "head": round(head_us, 2)}}


def fused_roofline(model, x, pred, batch, launches, storage="fp32"):
    from gdn_amd import _lib
    c = model._constants()
    gnn = model.gnn_layers[0].gnn
    lin = model.out_layer.mlp[0]
    emb = model.embedding.weight
    st = torch.cuda.current_stream().cuda_stream
    fixed = (gnn.lin.weight.data_ptr(), c.terms.data_ptr(), c.graph.nbr.data_ptr(), c.graph.deg.data_ptr(),
             gnn.bias.data_ptr(), emb.data_ptr(), c.bn1.data_ptr(), c.bn2.data_ptr(), lin.weight.data_ptr(),
             lin.bias.data_ptr())
    esz = 2 if storage == "bf16" else 4
    bf16 = storage == "bf16"
    name = "gdn_forward_fused_plan (" + ("bf16" if bf16 else "fp32") + " storage)"
    plan = model._plan(c, bf16)              # per-launch constants precomputed once (include/gdn_hip.h "plans")
    xs = x.bfloat16() if bf16 else x
    xstride, pstride = N_SENSORS * WINDOW * esz, N_SENSORS * 4
    nslots = max(1, x.shape[0] // batch)
    if plan is None:                         # n > 127: the row-gather kernel (no plan on that path)
        name = ("gdn_forward_fused_bf16" if bf16 else "gdn_forward_fused") + " (fp32 row gather, " + storage + " storage)"

    def launch(i):          # raw C-ABI call: host cost per launch stays below the kernel's duration
        s = (i % nslots) * batch
        if plan is None:
            _lib.call("gdn_forward_fused_bf16" if bf16 else "gdn_forward_fused", xs.data_ptr() + s * xstride, *fixed,
                      batch, N_SENSORS, WINDOW, DIM, TOPK, pred.data_ptr() + s * pstride, st)
            return
        _lib.call("gdn_forward_fused_plan", xs.data_ptr() + s * xstride, plan.data_ptr(), batch, N_SENSORS, WINDOW, DIM,
                  TOPK, int(bf16), pred.data_ptr() + s * pstride, None, st)
    for i in range(3):
        launch(i)
    mean_us, med_us = event_time_launches(launch, launches)
    alg_bytes = batch * N_SENSORS * WINDOW * esz + batch * N_SENSORS * 4    # SURVEY §8d "fused forward"
    achieved = alg_bytes / (mean_us * 1e-6) / 1e9
    return {"kernel": f"{name} (dominant kernel of the timed region; serial launches)", "bound": "hbm",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("fused" + ("_bf16" if storage == "bf16" else ""), batch),
            "launch_us": round(mean_us, 2), "launch_us_best": round(med_us, 2), "batch": batch,
            "algorithmic_bytes_per_launch": alg_bytes, "windows_per_s": round(batch / (mean_us * 1e-6), 1),
            "note": ("by design only x-in + out touch HBM; the aggregation runs as a dense [n x n] x [n x d] product on "
                     "v_mfma_f32_32x32x16 (two 16-bit terms per factor), bound by LDS operand traffic + VALU "
                     "(profiles/r02_sq_counters*.json)") if plan is not None else
                    "by design only x-in + out touch HBM; one workgroup per window, xlin tile in LDS, fp32 row gather on "
                    "the VALU (profiles/r03_sq_counters_config4*.json)"}


def streaming_copy_rate(device, mib=1024, reps=20):
    """Context for the roofline fractions (quoted against the 8 TB/s spec): what a plain device copy — read N bytes,
    write N bytes, larger than the 256 MB infinity cache — delivers on THIS GPU, measured the same way (HIP events)."""
    n = mib * 1024 * 1024 // 4
    x = torch.rand((n,), device=device)
    y = torch.empty_like(x)
    for _ in range(5):
        y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return {"achieved": round(2 * n * 4 / us / 1e3, 1), "unit": "GB/s", "bytes_each_way": n * 4,
            "note": "torch tensor copy, read + write; K8's roofline.achieved at the same traffic volume is the comparable figure"}


def train_step_line(device, n, w, batch, steps=50, dist=None, split=False):
    """Extra (not the headline): one optimisation step of the reference's train() (train.py:52-66) at the
    same shape — harness.NativeTrainStep: HIP forward/backward, in-kernel dropout draw, gdn_adam_step over flat
    buffers — replayed from one HIP graph.  With several ranks (BASELINE configs[3]: global batch = ranks x
    `batch`, DDP) the step is two graphs around ONE collective, the RCCL all-reduce of the flat gradient bucket
    (sum; 1/ranks folded into the optimizer kernel); every rank runs it, the time is the max over ranks."""
    from gdn_amd.harness import GraphedTrainStep, world
    model = build_model(device)[0].train()
    step = GraphedTrainStep(model, batch, split=True if split else None)
    g = torch.Generator(device=device).manual_seed(11 + world()[0])
    step.x.copy_(torch.rand(step.x.shape, device=device, generator=g))
    step.y.copy_(torch.rand(step.y.shape, device=device, generator=g))
    for _ in range(3):
        step.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ranks = world()[1]
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms = dt / steps * 1e3
    native = type(step).__name__ == "NativeTrainStep"
    return {"batch": batch, "global_batch": batch * ranks, "ranks": ranks, "ms_per_step": round(ms, 4),
            "windows_per_s": round(batch * ranks / ms * 1e3, 1),
            "step_impl": type(step).__name__, "optimizer": "gdn_adam_step (torch.optim.Adam update, flat buffers)"
            if native else "torch.optim.Adam(fused)",
            "hip_graph": True, "sensors": n, "window": w,
            "collective": ("one all_reduce (sum) of the flat fp32 gradient buffer, %d values, between two graph replays"
                           % step.count) if (native and getattr(step, "_split", False)) else None,
            "workload": "BASELINE configs[3] (global batch = ranks x 512, per-rank BatchNorm statistics)"
            if ranks > 1 else "single-GPU training step at the bench shape"}


def cpu_baseline(params, budget_s=12.0):
    """The oracle (op-faithful CPU port of the reference forward + numpy scoring) on a bounded
    sample of the same workload."""
    import numpy as np
    from oracle import gdn_oracle, score_oracle
    g = torch.Generator().manual_seed(0)
    # (the reference materialises [E, 1, 2d] edge tensors, E = b * n * (k+1): 512 windows of the 512-sensor shape
    # would be 17 M edges x 1 KB — a bounded sample of 32 windows there)
    b = 512 if N_SENSORS <= 127 else 32
    x = torch.rand((b, N_SENSORS, WINDOW), generator=g)
    with torch.no_grad():
        gdn_oracle.forward(params, x, TOPK)                       # warm-up
        t0 = time.perf_counter()
        reps = 0
        while True:
            out = gdn_oracle.forward(params, x, TOPK)["out"]
            reps += 1
            if time.perf_counter() - t0 > budget_s or reps >= 20:
                break
        fwd_s_per_window = (time.perf_counter() - t0) / (reps * b)
    t_score = 1024
    pred = torch.rand((t_score, N_SENSORS), generator=g).numpy()
    gt = torch.rand((t_score, N_SENSORS), generator=g).numpy()
    t0 = time.perf_counter()
    score_oracle.anomaly_score(score_oracle.full_err_scores(pred, gt))
    score_s_per_window = (time.perf_counter() - t0) / t_score
    del out, np
    return {"value": round(1.0 / (fwd_s_per_window + score_s_per_window), 1), "unit": "windows/s",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps}x oracle eval forward of {b} windows (fp32, torch CPU, edge-materialising op "
                      f"sequence of the reference) + oracle scoring of {t_score} ticks x {N_SENSORS} sensors; "
                      f"forward {1.0 / fwd_s_per_window:.0f} win/s, scoring {1.0 / score_s_per_window:.0f} win/s",
            "host_logical_cpus": os.cpu_count()}


def main():
    # stdout carries exactly one JSON line: RCCL prints a version banner to stdout when the first
    # communicator is created, so everything else goes to stderr until the result is ready
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        result = run()
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if result is not None:
        print(json.dumps(result), flush=True)


def run():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=512, help="logical minibatch (BASELINE configs[2])")
    ap.add_argument("--coalesce", type=int, default=64,
                    help="consecutive minibatches of the resident series sent as one launch (1 = per-batch launches)")
    ap.add_argument("--ticks", type=int, default=32768, help="windows per rank per step (SURVEY §8d)")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--streams", type=int, default=3,
                    help="side streams the forward launches of a step rotate over (matters with --coalesce 1: one launch "
                         "per minibatch; a step that is ONE launch uses no side stream)")
    ap.add_argument("--sharded-graph", action="store_true",
                    help="N>1: replay the ShardedEvaluator's compute segments from HIP graphs (measured slower than eager "
                         "launches in the 1-rank rehearsal: 0.452 vs 0.426 ms)")
    ap.add_argument("--exchange-chunk", type=int, default=32768,
                    help="N>1: ticks per async all-to-all of the scoring keys (overlaps the following forward chunks)")
    ap.add_argument("--sweep-max", type=int, default=262144,
                    help="largest launch of the K8 roofline sweep (0 = stop at --ticks)")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="1 rank, but run the N>1 step (RCCL process group, all-to-all scoring exchange)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="swat",
                    help="swat = BASELINE configs[2] (the headline); config4 = the 512-sensor stress shape")
    ap.add_argument("--dim", type=int, default=64, help="embedding / hidden width d (config4 is quoted at 64 and 128)")
    args = ap.parse_args()
    global N_SENSORS, WINDOW, TOPK, DIM
    wl = WORKLOADS[args.workload]
    N_SENSORS, WINDOW, TOPK, DIM = wl["n"], wl["w"], wl["k"], args.dim

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or args.rehearse_dist:
        import torch.distributed as dist
        if args.rehearse_dist and "RANK" not in os.environ:
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561")
        dist.init_process_group("nccl", device_id=device)

    from gdn_amd import harness
    model, params = build_model(device)
    t, batch = args.ticks, args.batch
    launch_batch = batch * max(1, args.coalesce)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.rand((t, N_SENSORS, WINDOW), generator=g).to(device)      # resident before the timed region
    y = torch.rand((t, N_SENSORS), generator=g).to(device)

    ev = harness.SeriesEvaluator(model, x, y, batch=batch, use_graph=not args.no_graph, streams=args.streams,
                                 coalesce=args.coalesce)
    if world == 1 and not args.rehearse_dist:
        step = ev.step
        step_impl = "SeriesEvaluator: forward launches + device scoring, one HIP graph replay per step"
    else:
        # N>1: windows shard by rank with no collective in the forward; the scoring exchange (all-to-all of
        # the radix keys by sensor) is issued per chunk of ticks and overlaps the forward of the following
        # chunks.  No fallback: a rank that cannot build or run this step fails the whole job (a silently
        # different timed path would be reported under this path's label).
        total = t * world
        sev = harness.ShardedEvaluator(model, x, y, total, chunk=args.exchange_chunk, use_graph=args.sharded_graph)
        sev.step()                      # eager: plans, constants, RCCL channels
        sev.step()                      # captures the compute segments (or decides to stay eager)
        torch.cuda.synchronize()
        step = sev.step
        step_impl = ("ShardedEvaluator: compute segments replayed from HIP graphs" if sev._graphs is not None else
                     "ShardedEvaluator: eager launches") + ", async all_to_all_single per chunk + one all_gather (eager RCCL)"

    # Kernel-level legs first, on every rank (their results are only reported by rank 0): K8 and fused
    # roofline with HIP events, each after its own ~60 ms of sustained launches.  Besides producing the
    # roofline numbers this leaves the GPU at steady clocks when the W warm-up steps begin — it needs ~50 ms
    # of load after an idle period to get there (tools/probe_ramp.py), and W steps alone are a few ms.
    launches_for = lambda b: max(12, min(64, 65536 // b))
    sizes = sorted({batch, min(4096, t), launch_batch, t})      # 4096: the launch size rounds 1-2 quoted
    sweep = [k8_roofline(model, x, b, launches=launches_for(b)) for b in sizes]
    if world == 1 and args.sweep_max > t and args.workload == "swat":   # the largest per-GPU launch
        xbig = torch.rand((args.sweep_max, N_SENSORS, WINDOW), device=device)     # 2 GB; xlin + z: 17 GB
        sweep.append(k8_roofline(model, xbig, args.sweep_max, launches=12))
        del xbig
        torch.cuda.empty_cache()
    fused_leg = fused_roofline(model, x, ev.pred, launch_batch, launches=launches_for(launch_batch))
    fused_sweep = [fused_leg if b == launch_batch else fused_roofline(model, x, ev.pred, b, launches=launches_for(b))
                   for b in sizes]
    # BASELINE configs[2] says bf16: the same kernels with x / xlin / z stored in bf16 (fp32 arithmetic)
    sweep_bf16 = [r for r in (k8_roofline(model, x, b, launches=launches_for(b), storage="bf16") for b in sizes)
                  if r is not None]
    fused_leg_bf16 = fused_roofline(model, x, ev.pred, launch_batch, launches=launches_for(launch_batch), storage="bf16")

    def timed(fn, steps):
        """`steps` steps bracketed by barrier + synchronize on both sides; max over ranks."""
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    def preroll(fn, ms=120.0):
        """>= `ms` of back-to-back steps right before a timed region (not part of --warmup): this GPU needs
        ~50 ms of sustained load after an idle period to reach its steady clocks (tools/probe_ramp.py), and a
        20-step region is 14 ms.  The step count is agreed across ranks (the N>1 step contains collectives)."""
        per_step = timed(fn, 8) / 8
        for _ in range(int(ms * 1e-3 / max(per_step, 1e-6)) + 1):
            fn()

    for _ in range(args.warmup):
        step()
    # the timed region: EXACTLY --steps steps, repeated (each repeat after its own pre-roll); the median
    # repeat is the reported one, all of them are listed
    repeats = []
    for _ in range(max(1, args.repeats)):
        preroll(step)
        repeats.append(timed(step, args.steps))
    elapsed = sorted(repeats)[len(repeats) // 2]

    result = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * t / (elapsed / args.steps)
        result = {
            "metric": "sliding-windows/sec forward+anomaly-score, SWaT-shape (127 sensors, W=15)" if args.workload == "swat"
            else "sliding-windows/sec forward+anomaly-score, WADI-shape stress (512 sensors, W=30)",
            "value": round(value, 1), "unit": "windows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["label"],
                       "sensors": N_SENSORS, "window": WINDOW, "topk": TOPK, "dim": DIM, "out_layer_num": 1,
                       "batch": batch, "batches_per_launch": max(1, args.coalesce),
                       "windows_per_launch": launch_batch, "windows_per_rank_per_step": t, "storage": "fp32",
                       "multi_gpu_measured": world > 1,
                       "hip_graph": not args.no_graph, "forward_streams": args.streams, "step_impl": step_impl,
                       "parallelism": f"windows sharded over {world} rank(s), no forward collective; scoring keys "
                                      f"all-to-all by sensor in chunks of {args.exchange_chunk} ticks (async, overlapping "
                                      "the forward) + one all-gather"
                       if world > 1 else "single GPU"},
        }
        result["timed_regions_ms_per_step"] = [round(1e3 * r / args.steps, 4) for r in repeats]
    if rank == 0:
        # the launch size the timed region uses
        result["roofline"] = next(r for r in sweep if r["batch"] == launch_batch)
        result["roofline_sweep"] = [{"batch": r["batch"], "achieved": r["achieved"], "frac": r["frac"],
                                     "launch_us": r["launch_us"]} for r in sweep]
        result["roofline_fused"] = fused_leg
        result["roofline_fused_sweep"] = [{"batch": r["batch"], "launch_us": r["launch_us"], "windows_per_s": r["windows_per_s"]}
                                          for r in fused_sweep]
        if sweep_bf16:
            result["roofline_bf16"] = next(r for r in sweep_bf16 if r["batch"] == launch_batch)
            result["roofline_bf16_sweep"] = [{"batch": r["batch"], "achieved": r["achieved"], "frac": r["frac"],
                                              "launch_us": r["launch_us"]} for r in sweep_bf16]
        result["roofline_fused_bf16"] = fused_leg_bf16
        if args.coalesce > 1 and world == 1 and not args.rehearse_dist:      # transparency: the same step with one launch per logical minibatch
            ev1 = harness.SeriesEvaluator(model, x, y, batch=batch, use_graph=not args.no_graph, streams=args.streams)
            for _ in range(args.warmup):
                ev1.step()
            preroll(ev1.step)
            result["value_per_batch_launches"] = round(t * args.steps / timed(ev1.step, args.steps), 1)
    if rank == 0 and world == 1 and not args.rehearse_dist:
        # the same step on bf16-stored windows (BASELINE configs[2] wording; `value` stays the fp32 line)
        evb = harness.SeriesEvaluator(model, x.bfloat16(), y, batch=batch, use_graph=not args.no_graph,
                                      streams=args.streams, coalesce=args.coalesce)
        for _ in range(args.warmup):
            evb.step()
        preroll(evb.step)
        result["value_bf16_storage"] = round(t * args.steps / timed(evb.step, args.steps), 1)
        del evb
    if rank == 0 and world == 1 and not args.rehearse_dist:
        # the same step with the windows built in-kernel from the raw [N, T+W] series
        raw = torch.rand((N_SENSORS, t + WINDOW), generator=torch.Generator().manual_seed(7)).to(device)
        ev2 = harness.SeriesEvaluator(model, None, raw[:, WINDOW:].t().contiguous(), batch=batch,
                                      use_graph=not args.no_graph, streams=args.streams, coalesce=args.coalesce,
                                      series=raw)
        from gdn_amd._lib import GdnHipError
        try:
            for _ in range(args.warmup):
                ev2.step()
            preroll(ev2.step)
            result["value_windows_from_raw_series"] = round(t * args.steps / timed(ev2.step, args.steps), 1)
        except GdnHipError as exc:      # n > 127 with n * w beyond the row-gather kernel's in-kernel window addressing
            result["value_windows_from_raw_series"] = None
            result["value_windows_from_raw_series_note"] = f"not available at this shape: {exc}"
        del ev2, raw
    # training step (BASELINE configs[3] with several ranks): every rank takes part in the gradient all-reduce
    if args.workload == "swat" or world == 1:
        from gdn_amd import _lib
        if _lib.load().gdn_train_supported(N_SENSORS, WINDOW, DIM, TOPK):
            line = train_step_line(device, n=x.shape[1], w=x.shape[2], batch=args.batch, dist=dist,
                                   split=world > 1 or args.rehearse_dist)
        else:       # e.g. > ~600 sensors: not even a 64-column slice of the backward's [n+1, d] fp32 tile fits LDS
            line = {"ms_per_step": None, "note": "gdn_train_supported() == 0 at this shape (the backward's tile slice "
                                                  "exceeds the 160 KB of LDS)"}
        if rank == 0:
            result["train_step"] = line
    if rank == 0 and world == 1:
        result["streaming_copy"] = streaming_copy_rate(device)
        if not args.skip_cpu:
            result["cpu_baseline"] = cpu_baseline(params)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return result if rank == 0 else None


if __name__ == "__main__":
    main()
