"""TEST INFRASTRUCTURE: gradient checks against a float64 restatement of one training step.

`staged_f64` restates train.py:63-79 over models/GDN.py:122-187 / models/graph_layer.py:53-117 in float64 IN THE
DECOMPOSITION THE HIP KERNELS USE (folded attention terms a = lin^T att, c = emb . att_em; per-node scalars
s_i / s_j; per-target source lists), keeps the value and the gradient of every stage boundary, and reports the
smallest |pre-activation| in front of every ReLU / LeakyReLU.  tests/test_oracle_golden.py pins it to
oracle.gdn_oracle.forward (the op-faithful restatement the golden vectors pin) at 1e-10.

Why the pre-activations matter: the derivative of (Leaky)ReLU jumps at 0.  An fp32 implementation whose
pre-activation differs from the float64 one by rounding (~1e-7 .. 1e-6 behind a BatchNorm) lands on the other
side of the kink for elements that close to 0, and ONE flipped element moves an attention-parameter gradient by
up to ~1 % of its largest entry (`profiles/r03_grad_error_5steps_l3.txt`, step 2: both HIP paths agree with each
other to 1e-6 and differ from float64 by 1e-2).  The reference's own fp32 arithmetic does the same.  So a
gradient comparison is only meaningful when no pre-activation sits inside the rounding band: `KINK_BAND`.

Bounds (`assert_grads_close`), derived from `profiles/r03_grad_error_stages_l3_*.txt`: the HIP gradients sit at
max|d| / max|g| <= 3.6e-6 per tensor and |d| / max(|g|, 1e-3 max|g|) <= 7.6e-4 per element (the fp32 CPU oracle:
2.6e-6 and 4.1e-4); asserted: 2e-5 and 1e-2.
"""
import torch
import torch.nn.functional as F

from oracle import gdn_oracle

f64 = torch.float64
KINK_BAND = 1e-6        # a (Leaky)ReLU input closer to 0 than this can fall on either side in fp32 (the noise of a
                        # pre-activation behind a BatchNorm is ~3e-7; observed: flipped at 2.8e-7 and 8e-7, not at 1.8e-6)
TOL_MAX = 2e-5          # max |d| <= TOL_MAX * max|g| per tensor
TOL_ELEM = 1e-2         # |d| <= TOL_ELEM * max(|g|, 1e-3 max|g|) per element
NOISE_FLOOR = 1e-12     # a parameter whose float64 gradient is below this everywhere has a true gradient of 0


def lists_of(graph):
    """[n, k+1] source lists in the order of models/graph_layer.py:61-63 (non-self top-k entries, then self),
    -1 padded, and the validity mask."""
    n, k = graph.shape
    out = torch.full((n, k + 1), -1, dtype=torch.long)
    for i in range(n):
        src = [int(j) for j in graph[i] if int(j) != i] + [i]
        out[i, :len(src)] = torch.tensor(src)
    return out, out >= 0


def staged_f64(p, x, y, graph, layers, mask=None):
    """One training step's forward + backward in float64, stage boundaries kept.  `p`: float64 leaves keyed
    like state_dict.  Returns (loss, {stage name: tensor}, {stage name: gradient}, {param: gradient})."""
    b, n, w = x.shape
    pre = "gnn_layers.0.gnn."
    lin, emb = p[pre + "lin.weight"], p["embedding.weight"]
    d = lin.shape[0]
    a_vec = torch.stack((lin.T @ p[pre + "att_i"].view(d), lin.T @ p[pre + "att_j"].view(d)))        # [2, w]
    c_vec = torch.stack((emb @ p[pre + "att_em_i"].view(d), emb @ p[pre + "att_em_j"].view(d)))      # [2, n]
    lin_direct = lin + 0.0                       # the projection's own use of lin (its gradient = "d_lin direct")
    xlin = x @ lin_direct.T                                                                             # [b, n, d]
    s_i = x @ a_vec[0] + c_vec[0]
    s_j = x @ a_vec[1] + c_vec[1]
    lst, valid = lists_of(graph)
    safe = lst.clamp(min=0)
    sites = []                                   # (name, pre-activation, activation) of every (Leaky)ReLU

    def act_site(name, pre, fn):
        post = fn(pre)
        post.retain_grad()
        sites.append((name, pre, post))
        return post

    pre_logit = (s_i.unsqueeze(-1) + s_j[:, safe]).masked_fill(~valid, 1.0)      # (padding slots: masked out below)
    logit = act_site("leaky(logit)", pre_logit, lambda t: F.leaky_relu(t, gdn_oracle.NEG_SLOPE))      # [b, n, k+1]
    logit = logit.masked_fill(~valid, float("-inf"))
    e = (logit - logit.max(dim=-1, keepdim=True).values).exp()
    alpha = e / (e.sum(dim=-1, keepdim=True) + gdn_oracle.SOFTMAX_EPS)
    z = (alpha.unsqueeze(-1) * xlin[:, safe]).sum(dim=2) + p[pre + "bias"]                             # [b, n, d]
    stages = dict(a_vec=a_vec, c_vec=c_vec, lin_direct=lin_direct, xlin=xlin, s_i=s_i, s_j=s_j, z=z)
    for t in stages.values():
        t.retain_grad()
    new_stats = {}
    h = gdn_oracle.batch_norm(p, "gnn_layers.0.bn.", z.view(b * n, d), True, new_stats)
    h = act_site("relu(bn1)", h, F.relu).view(b, n, d)
    h = h * emb
    h = gdn_oracle.batch_norm(p, "bn_outlayer_in.", h.permute(0, 2, 1), True, new_stats)
    h = act_site("relu(bn2)", h, F.relu).permute(0, 2, 1)
    if mask is not None:
        h = h * mask
    act = h + 0.0
    act.retain_grad()
    stages["act"] = act
    hh = act
    for l in range(layers):                      # gdn_oracle.out_layer, with the pre-activations looked at
        key = f"out_layer.mlp.{3 * l}."
        hh = F.linear(hh, p[key + "weight"], p[key + "bias"])
        if l != layers - 1:
            hh = gdn_oracle.batch_norm(p, f"out_layer.mlp.{3 * l + 1}.", hh.permute(0, 2, 1), True, new_stats).permute(0, 2, 1)
            hh = act_site(f"relu(mlp{l})", hh, F.relu)
    out = hh.view(-1, n)
    loss = F.mse_loss(out, y)
    loss.backward()
    # smallest |pre-activation| per site, over the elements whose derivative MATTERS: where no gradient arrives at
    # the activation (a dropped-out element, a padding slot) either side of the kink gives the same result
    staged_f64.kinks = {}
    for name, pre, post in sites:
        live = post.grad != 0
        staged_f64.kinks[name] = float(pre.detach()[live].abs().min()) if bool(live.any()) else float("inf")
    stage_grads = {k_: v.grad.detach() for k_, v in stages.items()}
    stages = {k_: v.detach() for k_, v in stages.items()}
    stages["alpha"] = alpha.detach()
    stages["out"] = out.detach()
    grads = {k_: v.grad.detach() for k_, v in p.items() if torch.is_tensor(v) and v.requires_grad}
    return loss.detach(), stages, stage_grads, grads


def rel_err(got, want):
    """(max |d| / max(|g|, 1e-3 max|g|), max|d| / max|g|, max|g|) of a tensor against its float64 value."""
    got, want = got.detach().cpu().to(f64).reshape(-1), want.detach().cpu().to(f64).reshape(-1)
    top = float(want.abs().max())
    if top == 0.0:
        return float((got - want).abs().max()), 0.0, 0.0
    den = torch.clamp(want.abs(), min=1e-3 * top)
    diff = (got - want).abs()
    return float((diff / den).max()), float(diff.max()) / top, top


def row(name, got, want):
    e, a, top = rel_err(got, want)
    print(f"    {name:34s} max|g| {top:9.3e}   rel/element {e:9.2e}   max|d|/max|g| {a:9.2e}")
    return e




def f64_leaves(state_dict):
    """state_dict -> float64 CPU leaves (running statistics without gradient)."""
    return {key: (v.detach().cpu().to(f64).requires_grad_("running" not in key) if v.is_floating_point()
                  else v.detach().cpu()) for key, v in state_dict.items()}


def oracle_step(state_dict, x, y, graph, layers, mask=None):
    """float64 gradients of one training step at these parameters.  Returns (loss, {param: grad}, kink) with
    kink = the smallest |pre-activation| over every ReLU / LeakyReLU of the step."""
    p = f64_leaves(state_dict)
    loss, _st, _sg, grads = staged_f64(p, x.detach().cpu().to(f64), y.detach().cpu().to(f64), graph.cpu(), layers,
                                       None if mask is None else mask.detach().cpu().to(f64))
    return float(loss), grads, min(staged_f64.kinks.values())


def _group_of(name):
    """Tensors that come out of ONE reduction share a scale: d_att_i / d_att_j are the two rows of d_a times the same
    matrix, d_att_em_i / _j likewise, a BatchNorm's d_weight / d_bias are two sums of one pass.  When one of the pair
    is tiny through cancellation (att_i at initialisation: near-uniform attention, most d_s_i exactly 0) its error
    is still that of the shared sums, so the pair is judged against the larger of the two."""
    for a, b in (("att_em_i", "att_em"), ("att_em_j", "att_em"), ("att_i", "att"), ("att_j", "att")):
        if name.endswith(a):
            return name[: -len(a)] + b
    return name.rsplit(".", 1)[0] + ".<weight|bias>" if name.endswith((".weight", ".bias")) else name


def assert_grads_close(got, want, tol_max=TOL_MAX, tol_elem=TOL_ELEM, what=""):
    """Every tensor of `got` (name -> fp32 gradient) against its float64 value, relative to the largest entry of
    its group (`_group_of`; same-shape members only) and, element by element, relative to the element.
    Zero-gradient parameters (a bias in front of a train-mode BatchNorm) must come out as rounding noise.
    Returns the worst ratios seen (for reports)."""
    tops = {}
    for name, w in want.items():
        key = (_group_of(name), tuple(w.shape))
        tops[key] = max(tops.get(key, 0.0), float(w.abs().max()))
    worst = (0.0, 0.0)
    everything = max(1.0, max(tops.values()))            # noise is judged against the step's largest gradient
    for name, w in want.items():
        g = got[name].detach().cpu().to(f64).reshape(w.shape)
        if float(w.abs().max()) < NOISE_FLOOR:
            assert float(g.abs().max()) < 1e-6 * everything, (what, name, "expected rounding noise", float(g.abs().max()))
            continue
        top = tops[(_group_of(name), tuple(w.shape))]
        diff = (g - w).abs()
        r_max = float(diff.max()) / top
        r_el = float((diff / torch.clamp(w.abs(), min=1e-3 * top)).max())
        assert r_max <= tol_max, (what, name, f"max|d|/max|g| = {r_max:.2e} > {tol_max:.0e} (max|g| {top:.2e})")
        assert r_el <= tol_elem, (what, name, f"per-element relative error {r_el:.2e} > {tol_elem:.0e}")
        worst = (max(worst[0], r_max), max(worst[1], r_el))
    return worst
