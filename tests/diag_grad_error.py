"""TEST INFRASTRUCTURE (not collected by pytest): where does the training step's gradient error come from?

    python tests/diag_grad_error.py [--layers 3] [--steps 1] [--b 48 --n 27 --w 10 --k 8]

Restates one training step (train.py:63-79 over models/GDN.py:122-187, models/graph_layer.py:53-117) in float64
IN THE DECOMPOSITION THE HIP KERNELS USE (folded attention terms a = lin^T att, c = emb . att_em; per-node
scalars s_i / s_j; dense per-target lists), keeps the gradient of every stage boundary, and then

  1. compares every parameter gradient of the HIP autograd path with float64, element by element, RELATIVE to
     the element (|dg| / max(|g|, 1e-3 max|g|)) — an absolute bound of 2e-6 cannot see a 1 % error on an
     attention gradient of 1e-5;
  2. replays every backward kernel ALONE on float64-exact inputs (rounded to fp32 once), so that each kernel's
     own error is separated from what it inherits:  head/MLP backward -> d_z,  gdn_attn_aggregate_bwd ->
     d_xlin / d_si / d_sj,  gdn_project_bwd -> d_lin (direct) / d_a / d_c,  gdn_terms_bwd -> parameters;
  3. runs the HIP step twice and prints the run-to-run difference (atomics).

GDN_BWD_PATH=valu in the environment selects the row-gather backward (read once per process by the library).
The float64 decomposition is checked against oracle.gdn_oracle (the op-faithful restatement) first.
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import gdn_oracle  # noqa: E402

f64 = torch.float64


from _grad_check import lists_of, staged_f64, rel_err, row   # noqa: E402,F401


def multi_step(a):
    """test_native_train_step_equals_the_autograd_step, with the float64 gradient AT EACH PATH'S OWN PARAMETERS
    beside every step: which path (if any) computes a wrong gradient, and when do the parameters part?"""
    from gdn_amd import harness
    from test_gpu_forward_parity import random_params
    dev = torch.device("cuda:0")
    b, n, w, k, d, steps = a.b, a.n, a.w, a.k, a.d, a.steps
    g = torch.Generator().manual_seed(3)
    xs = torch.rand((steps, b, n, w), generator=g)
    ys = torch.rand((steps, b, n), generator=g)
    mk = lambda: random_params(n, w, k, d, seed=a.seed, out_layer_num=a.layers, inter=a.inter).to(dev)
    model = mk()
    model.dp.p = 0.0
    nat = harness.NativeTrainStep(model, b, use_graph=not a.eager, seed=1234567890123)
    ref = mk().train()
    ref.dp.p = 0.0
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    names = [name for name, _ in ref.named_parameters()]
    watch = [nm for nm in names if "gnn.att" in nm or nm.endswith("lin.weight") or nm == "embedding.weight"
             or nm.endswith("gnn.bias")]

    def f64_grads(module, t):
        p = {key: (v.detach().cpu().to(f64).requires_grad_("running" not in key) if v.is_floating_point() else v.cpu())
             for key, v in module.state_dict().items()}
        own = gdn_oracle.learned_graph(p["embedding.weight"].detach().float(), k)
        graph = module.learned_graph.cpu()
        cos = gdn_oracle.cosine_matrix(p["embedding.weight"].detach()).sort(dim=-1, descending=True).values
        gap = float((cos[:, k - 1] - cos[:, k]).min())
        grads = staged_f64(p, xs[t].to(f64), ys[t].to(f64), graph, a.layers)[3]
        print("      smallest |pre-activation| (float64): " + ", ".join(f"{kk} {vv:.1e}" for kk, vv in staged_f64.kinks.items()))
        return grads, graph, bool(torch.equal(own, graph)), gap

    for t in range(steps):
        # float64 gradients at each path's parameters BEFORE the step (graphs: recomputed from those parameters)
        sd_n = {key: v.detach().clone() for key, v in model.state_dict().items()}
        nat.x.copy_(xs[t].to(dev)); nat.y.copy_(ys[t].to(dev))
        loss_n = float(nat.step())
        g_n = {nm: nat.flat_g[off:off + cnt].detach().clone().view(prm.shape)
               for nm, prm, (off, cnt) in zip(names, ref.parameters(), nat.slices)}
        opt.zero_grad()
        sd_r = {key: v.detach().clone() for key, v in ref.state_dict().items()}
        loss_r = F.mse_loss(ref(xs[t].to(dev), None), ys[t].to(dev))
        loss_r.backward()
        g_r = {nm: prm.grad.detach().clone() for nm, prm in ref.named_parameters()}
        opt.step()

        class _SD:      # state_dict holder for f64_grads
            def __init__(self, sd, graph): self.sd, self.learned_graph = sd, graph
            def state_dict(self): return self.sd
        g64_n, graph_n, own_n, gap_n = f64_grads(_SD(sd_n, nat.ws["topk"].clone()), t)
        g64_r, graph_r, own_r, gap_r = f64_grads(_SD(sd_r, ref.learned_graph.clone()), t)
        print(f"\n== step {t}: loss native {loss_n:.9f} autograd {float(loss_r.detach()):.9f}; the two HIP paths picked the "
              f"same top-k graph: {bool(torch.equal(graph_n, graph_r))}; equal to the fp32 CPU oracle's top-k: {own_n} / {own_r}; "
              f"smallest k-th/(k+1)-th cosine gap (float64) {gap_n:.2e}")
        for nm in watch:
            pd = float((sd_n[nm] - sd_r[nm]).abs().max())
            en = rel_err(g_n[nm], g64_n[nm])
            er = rel_err(g_r[nm], g64_r[nm])
            nr = rel_err(g_n[nm], g_r[nm].to(f64))
            print(f"    {nm:30s} |p_nat-p_ref| before {pd:8.2e} | native vs f64 {en[0]:8.2e} ({en[1]:8.2e} of max) | "
                  f"autograd vs f64 {er[0]:8.2e} ({er[1]:8.2e}) | native vs autograd {nr[0]:8.2e} ({nr[1]:8.2e})")
    print("\nparameters after the last step:")
    for nm, pa, pb in zip(names, model.parameters(), ref.parameters()):
        diff = (pa.detach() - pb.detach()).abs()
        print(f"    {nm:30s} max |p_nat - p_ref| {float(diff.max()):8.2e}   share above 2e-5: {float((diff > 2e-5).float().mean()):.3f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=48)
    ap.add_argument("--n", type=int, default=27)
    ap.add_argument("--w", type=int, default=10)
    ap.add_argument("--k", type=int, default=8)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--inter", type=int, default=128)
    ap.add_argument("--seed", type=int, default=9)
    ap.add_argument("--cpu-only", action="store_true", help="only check the float64 decomposition against the oracle")
    ap.add_argument("--init", default="random", help="'default' = the module's own initialisation + att_em ~ U(-0.1, 0.1) "
                    "(what __graft_entry__.smoke() trains): near-uniform attention, tiny att_i gradients")
    ap.add_argument("--eager", action="store_true", help="multi-step mode: NativeTrainStep without the HIP graph")
    ap.add_argument("--steps", type=int, default=0, help="> 0: the multi-step comparison of the two HIP training paths")
    a = ap.parse_args()
    if a.steps > 0:
        return multi_step(a)
    from test_gpu_forward_parity import random_params
    b, n, w, k, d = a.b, a.n, a.w, a.k, a.d
    g = torch.Generator().manual_seed(3)
    x = torch.rand((b, n, w), generator=g)
    y = torch.rand((b, n), generator=g)
    if a.init == "default":
        from gdn_amd import GDN
        torch.manual_seed(a.seed)
        model = GDN([torch.zeros((2, 1), dtype=torch.long)], n, dim=d, input_dim=w, topk=k, out_layer_num=a.layers,
                    out_layer_inter_dim=a.inter)
        with torch.no_grad():
            model.gnn_layers[0].gnn.att_em_i.uniform_(-0.1, 0.1)
            model.gnn_layers[0].gnn.att_em_j.uniform_(-0.1, 0.1)
    else:
        model = random_params(n, w, k, d, seed=a.seed, out_layer_num=a.layers, inter=a.inter)
    params = {key: v.detach().clone() for key, v in model.state_dict().items()}
    graph = gdn_oracle.learned_graph(params["embedding.weight"], k)

    def leaves(dtype):
        return {key: (v.to(dtype).requires_grad_("running" not in key) if v.is_floating_point() else v)
                for key, v in params.items()}

    # ---- float64: decomposition vs the op-faithful oracle
    p64 = leaves(f64)
    loss64, st, sg, g64 = staged_f64(p64, x.to(f64), y.to(f64), graph, a.layers)
    q64 = leaves(f64)
    r = gdn_oracle.forward(q64, x.to(f64), k, a.layers, training=True, dropout_mask=torch.ones((b, n, d), dtype=f64),
                           graph=graph)
    lo = F.mse_loss(r["out"], y.to(f64))
    lo.backward()
    worst = max(rel_err(g64[key], q64[key].grad)[0] for key in g64 if float(g64[key].abs().max()) > 1e-12)
    print(f"[f64] staged decomposition vs op-faithful oracle: loss {float(loss64):.12f} / {float(lo.detach()):.12f}, "
          f"worst relative gradient difference {worst:.2e}")
    # ---- the reference's own arithmetic (fp32, op-faithful oracle) against float64: the bar to match
    q32 = leaves(torch.float32)
    r = gdn_oracle.forward(q32, x, k, a.layers, training=True, dropout_mask=torch.ones((b, n, d)), graph=graph)
    F.mse_loss(r["out"], y).backward()
    print("[fp32 CPU oracle vs float64]  (the reference's own arithmetic)")
    for key in g64:
        row(key, q32[key].grad, g64[key])
    if a.cpu_only:
        return

    # ---- HIP, end to end through autograd; the backward ops are recorded
    from gdn_amd import ops
    dev = torch.device("cuda:0")
    rec = {}

    def recorder(name, fn):
        def wrapped(*args, **kw):
            out = fn(*args, **kw)
            rec[name] = (args, out)
            return out
        return wrapped

    orig = {nm: getattr(ops, nm) for nm in ("attn_aggregate_bwd", "project_bwd", "terms_bwd")}
    for nm, fn in orig.items():
        setattr(ops, nm, recorder(nm, fn))
    model = model.to(dev).train()
    model.dp.p = 0.0
    xg, yg = x.to(dev), y.to(dev)

    def hip_step():
        model.zero_grad()
        loss = F.mse_loss(model(xg, None), yg)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), {name: prm.grad.detach().clone() for name, prm in model.named_parameters()}

    loss_h, gh = hip_step()
    print(f"\n[HIP autograd path] loss {loss_h:.9f} (float64 {float(loss64):.9f}); BWD path = "
          f"{os.environ.get('GDN_BWD_PATH', 'matrix cores')}")
    for key in g64:
        row(key, gh[key], g64[key])
    (d_z_h, xlin_h, alpha_h, s_i_h, s_j_h, graph_h, _b), (d_xlin_h, d_si_h, d_sj_h, d_bias_h) = rec["attn_aggregate_bwd"]
    (_x, _dx, _dsi, _dsj, _d), (d_lin_h, d_a_h, d_c_h) = rec["project_bwd"]
    assert torch.equal(graph_h.topk.cpu(), graph), "HIP and oracle picked different top-k graphs"
    print("  stage boundaries as the HIP chain produced them (inherited + own error):")
    row("forward xlin", xlin_h.view(b, n, d), st["xlin"])
    row("forward s_i", s_i_h.view(b, n), st["s_i"])
    row("forward s_j", s_j_h.view(b, n), st["s_j"])
    row("d_z (head / MLP backward)", d_z_h.view(b, n, d), sg["z"])
    row("d_xlin", d_xlin_h.view(b, n, d), sg["xlin"])
    row("d_si", d_si_h.view(b, n), sg["s_i"])
    row("d_sj", d_sj_h.view(b, n), sg["s_j"])
    row("d_a", d_a_h[:, :w], sg["a_vec"])
    row("d_c", d_c_h, sg["c_vec"])

    # ---- every kernel alone on float64-exact inputs
    for nm, fn in orig.items():
        setattr(ops, nm, fn)
    c32 = lambda t: t.to(torch.float32).to(dev).contiguous()
    pitch = graph_h.pitch
    alpha64 = torch.zeros((b, n, pitch), dtype=f64)
    alpha64[:, :, :st["alpha"].shape[2]] = st["alpha"]
    print("  gdn_attn_aggregate_bwd alone (d_z, xlin, alpha, s_i, s_j = float64 values rounded to fp32):")
    o = ops.attn_aggregate_bwd(c32(sg["z"].reshape(b * n, d)), c32(st["xlin"].reshape(b * n, d)),
                               c32(alpha64.view(b * n, pitch)), c32(st["s_i"].reshape(-1)), c32(st["s_j"].reshape(-1)),
                               graph_h, b)
    row("d_xlin", o[0].view(b, n, d), sg["xlin"])
    row("d_si", o[1].view(b, n), sg["s_i"])
    row("d_sj", o[2].view(b, n), sg["s_j"])
    row("d_bias", o[3], g64["gnn_layers.0.gnn.bias"])
    print("  ... same with the HIP forward's own xlin / alpha / s_i / s_j (d_z still exact):")
    o = ops.attn_aggregate_bwd(c32(sg["z"].reshape(b * n, d)), xlin_h, alpha_h, s_i_h, s_j_h, graph_h, b)
    row("d_xlin", o[0].view(b, n, d), sg["xlin"])
    row("d_si", o[1].view(b, n), sg["s_i"])
    row("d_sj", o[2].view(b, n), sg["s_j"])
    print("  gdn_project_bwd alone (d_xlin, d_si, d_sj exact):")
    o = ops.project_bwd(xg, c32(sg["xlin"].reshape(b * n, d)), c32(sg["s_i"].reshape(-1)), c32(sg["s_j"].reshape(-1)), d)
    row("d_lin (direct)", o[0], sg["lin_direct"])
    row("d_a", o[1][:, :w], sg["a_vec"])
    row("d_c", o[2], sg["c_vec"])
    print("  gdn_terms_bwd alone (d_lin direct, d_a, d_c exact):")
    gnn = model.gnn_layers[0].gnn
    d_a32 = torch.zeros((2, 64), dtype=torch.float32, device=dev)
    d_a32[:, :w] = c32(sg["a_vec"])
    flat = torch.cat((c32(sg["lin_direct"]).reshape(-1), d_a32.reshape(-1), c32(sg["c_vec"]).reshape(-1)))
    o = ops.terms_bwd(gnn.lin.weight.detach(), gnn.att_i.detach(), gnn.att_j.detach(), gnn.att_em_i.detach(),
                      gnn.att_em_j.detach(), model.embedding.weight.detach(), flat[:d * w].view(d, w),
                      flat[d * w:d * w + 128].view(2, 64), flat[d * w + 128:].view(2, n))
    pre = "gnn_layers.0.gnn."
    row("lin.weight", o[0], g64[pre + "lin.weight"])
    row("att_i", o[1], g64[pre + "att_i"])
    row("att_j", o[2], g64[pre + "att_j"])
    row("att_em_i", o[3], g64[pre + "att_em_i"])
    row("att_em_j", o[4], g64[pre + "att_em_j"])

    # ---- run to run
    _l2, gh2 = hip_step()
    print("  run-to-run difference of the HIP step (same inputs):")
    for key in g64:
        row(key, gh2[key], gh[key].to(f64))


if __name__ == "__main__":
    main()
