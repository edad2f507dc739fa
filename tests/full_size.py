"""Full-size parity of a whole GAT/GCN train step: the HIP path (bot_amd, through the C ABI) against the oracle's C
restatement of DGL's CPU kernels (oracle/c_ops.py + oracle/ref_models.py) on the SAME weights, the SAME label mask and
dropout 0 — forward logits of every node and the gradient of every parameter (run.py:252-284).

TEST INFRASTRUCTURE: imported by tests/test_gpu_parity.py and by bench.py's `cpu_baseline` leg (which reports the same
comparison as the bench line's "parity" object); never by bot_amd.
"""
from __future__ import annotations

import contextlib
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

GAT_ARXIV = dict(n_layers=3, n_heads=3, n_hidden=250, norm="batch", non_interactive_attn=False, use_symmetric_norm=False,
                 linear=True, residual=False)   # BASELINE config 2 (run.py:1011-1013), drop rates set by the caller


def init_state(cfg, fin, n_classes, seed=0):
    """Parameters of the reference-shaped stack (random init of that architecture), as a CPU state_dict."""
    from bot_amd import nn as bnn
    torch.manual_seed(seed)
    model = bnn.GAT(dim_node=fin, dim_edge=0, dim_output=n_classes, activation=F.relu, dropout=0.0, input_drop=0.0,
                    attn_drop=0.0, edge_drop=0.0, **cfg)
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


class KinkGates:
    """The piecewise-linear activations of the stack evaluated with the 0/1 side of every kink GIVEN (taken from the HIP run)
    instead of derived from the oracle's own pre-activations: `relu` for the ReLU behind every hidden BatchNorm
    (models.py:726-730), `leaky` for the leaky-ReLU of the attention logits (models.py:526).

    Why: a pre-activation within fp32 rounding of zero can fall on either side in two fp32 implementations that sum in
    different orders.  The forward value barely moves (it is ~0 either way) but the derivative jumps: a hidden unit's whole
    contribution enters or leaves one row of the layer's weight gradients (~1/sqrt(N) of the row, far above 1e-4), and an
    attention logit of the reference's default scoring depends on the SOURCE node only (models.py:525), so one node's sign
    flips the derivative on all its out-edges at once.  Two CPU restatements of this step (oracle/ref_ops.py vs
    oracle/c_ops.py) differ by 0.75 % on such a row and agree to 2.6e-6 at equal gates; both are within 4e-6 of an fp64 run
    at equal gates.  Evaluating the oracle AT THE TESTED RUN'S GATES removes exactly that ambiguity and nothing else: the
    forward changes only where |pre-activation| is rounding noise (recorded in `stats`, bounded by the caller), and the
    gradients of the two implementations become comparable entry by entry."""

    def __init__(self, relu_gates, leaky_gates=None):
        self.rg, self.lg, self.ri, self.li, self.stats = relu_gates, leaky_gates, 0, 0, []

    def _note(self, kind, h, g):
        diff = (h.detach() > 0) != g
        self.stats.append({"kind": kind, "differ": int(diff.sum()), "of": diff.numel(),
                           "max_abs_preact_where_differ": float(h.detach().abs()[diff].max()) if bool(diff.any()) else 0.0})

    def relu(self, h):
        g = self.rg[self.ri % len(self.rg)].bool()
        self.ri += 1
        self._note("relu", h, g)
        return h * g.to(h.dtype)

    def leaky(self, e, slope):
        if self.lg is None:
            return F.leaky_relu(e, slope)
        g = self.lg[self.li % len(self.lg)].bool().view(e.shape)
        self.li += 1
        self._note("leaky", e, g)
        return torch.where(g, e, e * slope)


def oracle_step(src, dst, n, feat, labels, train_idx, mask, sd, cfg, n_classes, loss="loge", threads=None, steps=1, gates=None,
                dtype=torch.float32):
    """One train step (forward + loss + backward, training-mode BatchNorm, no dropout) on the oracle's C kernels.
    `gates`: (relu gates, leaky gates) as returned by hip_step, see KinkGates (None: the oracle's own).
    `dtype=torch.float64`: the whole step in double (liboracle_f64.so + torch fp64) — the exact side when two fp32 runs are
    ranked against each other (`logits_ok`, `rank_against_exact`).
    Returns (pred [N,C], {param name: grad}, [seconds per step], threads, gate stats)."""
    from oracle import c_ops
    from oracle import ref_models as RM
    if threads is None:
        threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    c_ops.set_num_threads(threads)
    g = c_ops.CGraph(src, dst, n)
    sdg = {k: (v.to(dtype).requires_grad_() if v.is_floating_point() and "running" not in k
               else (v.to(dtype) if v.is_floating_point() else v.clone())) for k, v in sd.items()}
    feat = feat.to(dtype)
    names = [k for k, v in sdg.items() if v.requires_grad]
    times, pred, grads, kg = [], None, None, None
    for _ in range(steps):
        t0 = time.perf_counter()
        kg = None if gates is None else KinkGates(*gates)
        x = RM.add_labels(feat, labels, train_idx[mask], n_classes)
        pred = RM.gat_forward(g, x, sdg, n_layers=cfg["n_layers"], n_heads=cfg["n_heads"], n_hidden=cfg["n_hidden"],
                              n_classes=n_classes, norm=cfg["norm"], non_interactive_attn=cfg["non_interactive_attn"],
                              use_symmetric_norm=cfg["use_symmetric_norm"], linear=cfg["linear"], residual=cfg["residual"],
                              activation=F.relu if kg is None else kg.relu, leaky=None if kg is None else kg.leaky, training=True)
        out = RM.compute_loss(pred[train_idx[~mask]], labels[train_idx[~mask]], loss)
        grads = torch.autograd.grad(out, [sdg[k] for k in names])
        times.append(time.perf_counter() - t0)
    return pred.detach(), dict(zip(names, grads)), times, int(c_ops.num_threads()), (kg.stats if kg is not None else None)


@contextlib.contextmanager
def tap_kinks():
    """While active, records from a bot_amd.nn stack (a) the output of every hidden layer's BatchNorm+ReLU(+dropout) epilogue
    — the fused layer node (bot_amd.nn.fused.gat_hidden_layer) and the modular epilogue (bot_amd.nn._epilogue) alike — and (b)
    the sign of every attention logit handed to the fused attention kernel (bot_amd._C.gat_attn_fwd), recomputed from the
    kernel's own operands in the kernel's order of additions, in edge-id order.  Wraps those three functions from the outside
    (test code; the product has no hook)."""
    import bot_amd.nn as bnn
    from bot_amd import _C
    from bot_amd.nn import fused
    from bot_amd.nn import edge_gat
    relu_taps, leaky_taps = [], []
    orig_e, orig_f, orig_a = bnn._epilogue, fused.gat_hidden_layer, _C.gat_attn_fwd

    def epi(h, norm, activation, dropout, training, **kw):
        y = orig_e(h, norm, activation, dropout, training, **kw)
        relu_taps.append((y.detach() > 0).to(torch.uint8).cpu())
        return y

    def hid(conv, bn, graph, h, dropout_p, training, **kw):
        # the tap reads the layer's fp32 output: keep it stored here (y_needed; `check_halves_only_hidden_states` asserts that the
        # halves-only form changes no bit of the results)
        y = orig_f(conv, bn, graph, h, dropout_p, training)
        if bn is not None:
            relu_taps.append((y.detach() > 0).to(torch.uint8).cpu())
        return y

    def attn(d, el, er, ee, eperm, keep, slope, H, aperm, zsign=None, **kw):
        # in pieces of 2^24 edges: torch-ROCm's row gather is wrong from 2^26 indices on (bot_amd.graph.take_rows)
        with torch.no_grad():
            dev = d.indptr.device
            gate = torch.empty((d.nnz, H), dtype=torch.uint8, device=dev)
            indptr = d.indptr.long()
            for a in range(0, d.nnz, 1 << 24):
                b = min(d.nnz, a + (1 << 24))
                pos = torch.arange(a, b, device=dev)
                z = torch.zeros((b - a, H), dtype=torch.float32, device=dev)
                if er is not None:
                    z = z + er.reshape(-1, H)[torch.searchsorted(indptr, pos, right=True) - 1]
                if el is not None:
                    z = z + el.reshape(-1, H)[d.indices[a:b].long()]
                if ee is not None:
                    z = z + (ee.reshape(-1, H)[a:b] if eperm is None else ee.reshape(-1, H)[eperm[a:b].long()])
                gate[d.eid[a:b].long()] = (z > 0).to(torch.uint8)
            leaky_taps.append(gate.cpu())
        return orig_a(d, el, er, ee, eperm, keep, slope, H, aperm, zsign, **kw)

    bnn._epilogue, fused.gat_hidden_layer, _C.gat_attn_fwd, edge_gat._epilogue = epi, hid, attn, epi
    try:
        yield relu_taps, leaky_taps
    finally:
        bnn._epilogue, fused.gat_hidden_layer, _C.gat_attn_fwd, edge_gat._epilogue = orig_e, orig_f, orig_a, orig_e


def hip_step(g, feat, labels, train_idx, mask, sd, cfg, n_classes, loss="loge", fuse=True):
    """The same step on the HIP path: bot_amd.nn.GAT + bot_amd.train.forward_backward on the device of `g`.
    Returns (pred, {param: grad}, (ReLU gates per hidden layer, leaky-ReLU gates per layer [E,H] in edge-id order))."""
    from bot_amd import nn as bnn
    from bot_amd import train as T
    dev = feat.device
    model = bnn.GAT(dim_node=feat.shape[1] + n_classes, dim_edge=0, dim_output=n_classes, activation=F.relu, dropout=0.0,
                    input_drop=0.0, attn_drop=0.0, edge_drop=0.0, **cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    model.fuse_layers = fuse
    with tap_kinks() as (relu_gates, leaky_gates):
        _, pred, _ = T.forward_backward(model, g, feat, labels, train_idx, train_idx[:0], train_idx[:0], use_labels=True,
                                        loss=loss, n_classes=n_classes, mask=mask.to(dev))
    assert len(relu_gates) == cfg["n_layers"] - 1 and len(leaky_gates) == cfg["n_layers"]
    pred, grads = pred.detach(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    # the tap keeps every hidden state stored in fp32 (it reads them), so the step above is not the production path
    # (hidden states as fp16 halves only, bot_amd.nn.fused `y_needed=False`).  The SAME step once more with no tap — what bench.py
    # times — must give the same logits and gradients, bit for bit, so that the parity verdict below covers the production path.
    model.zero_grad(set_to_none=True)
    _, pred2, _ = T.forward_backward(model, g, feat, labels, train_idx, train_idx[:0], train_idx[:0], use_labels=True,
                                     loss=loss, n_classes=n_classes, mask=mask.to(dev))
    assert torch.equal(pred2.detach(), pred), "the untapped (production) step gives different logits from the tapped one"
    for k, p in model.named_parameters():
        assert torch.equal(p.grad, grads[k]), f"the untapped (production) step gives a different gradient for {k}"
    return pred, grads, (relu_gates, leaky_gates)


def rank_against_exact(grads_hip, grads_ref, grads_exact, zero_grads=None):
    """Per parameter: (error of the HIP gradient, error of the fp32 oracle's gradient), both against the fp64 run and relative
    to the exact gradient's largest entry (the companion's for `zero_grads`, see compare)."""
    out = {}
    for k, gx in grads_exact.items():
        scale = max(float(grads_exact[(zero_grads or {}).get(k, k)].abs().max()), 1e-300)
        out[k] = (float((grads_hip[k].detach().cpu().double() - gx).abs().max()) / scale,
                  float((grads_ref[k].double() - gx).abs().max()) / scale)
    return out


CRITERIA = {
    "abs": "every logit within 1e-4 ABSOLUTE of the fp32 oracle's; every gradient entry within 1e-4 of its gradient's largest entry",
    "fp64-ranked": ("against the SAME step in fp64 (liboracle_f64.so): every logit / gradient entry within 1e-4 (absolute / of the "
                    "gradient's largest entry) of the exact one, or at most twice as far from it as the reference-order fp32 oracle is — "
                    "used where two fp32 runs cannot agree to 1e-4 (logits far above O(10), cancelling reductions)"),
}


def logits_ok(pred_hip, pred_ref, pred_exact=None, atol=1e-4):
    """(ok, criterion name, numbers).  Plain criterion: max |hip - fp32 oracle| <= atol.  With an fp64 run of the same step:
    max |hip - exact| <= max(atol, 2 max |fp32 oracle - exact|)."""
    hp = pred_hip.detach().cpu().double()
    d = float((hp - pred_ref.double()).abs().max())
    if pred_exact is None:
        return d <= atol, "abs", {"max_abs_logit_diff": d}
    eh = float((hp - pred_exact.double()).abs().max())
    eo = float((pred_ref.double() - pred_exact.double()).abs().max())
    return eh <= max(atol, 2 * eo), "fp64-ranked", {"max_abs_logit_diff": d, "logit_err_vs_fp64": eh, "oracle_logit_err_vs_fp64": eo}


def compare(pred_hip, grads_hip, pred_ref, grads_ref, gate_stats=None, tol=1e-4, zero_grads=None):
    """max |logit diff| over all nodes; the worst parameter-gradient error relative to that gradient's largest entry; the
    number of gradient entries beyond `tol` of it; and (with gates given to the oracle) how many ReLU / leaky-ReLU gates the two
    runs would have set differently and how large the pre-activation was there.
    `zero_grads` {param: companion param}: gradients that are identically zero in exact arithmetic (a bias in front of a
    training-mode BatchNorm: the batch mean is subtracted again) — both runs hold rounding noise there, so those are measured
    against the companion weight gradient's largest entry instead of their own."""
    d = float((pred_hip.detach().cpu().double() - pred_ref.double()).abs().max())
    worst, which, over, total, table = 0.0, None, 0, 0, {}
    for k, gr in grads_ref.items():
        gh = grads_hip[k].detach().cpu().double()
        scale = max(float(grads_ref[(zero_grads or {}).get(k, k)].double().abs().max()), 1e-30)
        e = (gh - gr.double()).abs() / scale
        table[k] = (scale, float(gh.abs().max()), float(e.max()), int((e > tol).sum()), e.numel())
        over += int((e > tol).sum())
        total += e.numel()
        if float(e.max()) > worst:
            worst, which = float(e.max()), k
    if os.environ.get("BOT_PARITY_TABLE"):
        for k, t in table.items():
            print("  %-28s |ref|max %.3e |hip|max %.3e  max err/|ref|max %.3e  over %d of %d" % ((k,) + t))
    r = {"max_abs_logit_diff": d, "max_rel_grad_err": worst, "worst_grad": which, "grad_entries_over_1e-4": over,
         "grad_entries": total, "n": int(pred_ref.shape[0]), "logit_scale": float(pred_ref.abs().max())}
    if gate_stats is not None:
        for kind in ("relu", "leaky"):
            st = [s for s in gate_stats if s["kind"] == kind]
            r[f"{kind}_gates_differing"] = sum(s["differ"] for s in st)
            r[f"{kind}_gates"] = sum(s["of"] for s in st)
        r["max_abs_preact_at_differing_gate"] = max(s["max_abs_preact_where_differ"] for s in gate_stats)
    return r


# ---------------------------------------------------------------------------------------------- GCN (BASELINE configs 1 / 3)
def gcn_oracle_step(src, dst, n, feat, labels, train_idx, sd, cfg, loss="logit", threads=None, gates=None):
    """One GCN train step (run.py:252-284 without --labels: the loss over `train_idx`) on the oracle's C kernels."""
    from oracle import c_ops
    from oracle import ref_models as RM
    if threads is None:
        threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    c_ops.set_num_threads(threads)
    g = c_ops.CGraph(src, dst, n)
    sdg = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    names = [k for k, v in sdg.items() if v.requires_grad]
    kg = None if gates is None else KinkGates(gates[0], None)
    t0 = time.perf_counter()
    pred = RM.gcn_forward(g, feat, sdg, n_layers=cfg["n_layers"], norm=cfg["norm"], norm_adj=cfg["norm_adj"],
                          use_linear=cfg["use_linear"], residual=cfg["residual"], activation=F.relu if kg is None else kg.relu,
                          training=True)
    out = RM.compute_loss(pred[train_idx], labels[train_idx], loss)
    grads = torch.autograd.grad(out, [sdg[k] for k in names])
    return pred.detach(), dict(zip(names, grads)), time.perf_counter() - t0, (kg.stats if kg is not None else None)


def gcn_hip_step(g, feat, labels, train_idx, sd, cfg, n_classes, loss="logit"):
    from bot_amd import nn as bnn
    from bot_amd import train as T
    dev = feat.device
    plain = []                   # without BatchNorm the stack applies `activation` itself (models.py:638-639): record it there

    def act(h):
        y = F.relu(h)
        plain.append((y.detach() > 0).to(torch.uint8).cpu())
        return y
    model = bnn.GCN(in_feats=feat.shape[1], n_classes=n_classes, activation=F.relu if cfg["norm"] == "batch" else act, dropout=0.0,
                    input_drop=0.0, **cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    with tap_kinks() as (relu_gates, leaky_gates):
        pred = model(g, feat)
        out = T.compute_loss(pred[train_idx], labels[train_idx], loss)
        out.backward()
    relu_gates = relu_gates + plain
    assert len(relu_gates) == cfg["n_layers"] - 1
    return pred.detach(), {k: p.grad.detach() for k, p in model.named_parameters()}, (relu_gates, None)


# ---------------------------------------------------------------------------------------------- edge-feature GAT (configs 4 / 5)
def _oracle_threads(default=32):
    """Host threads for the oracle's C kernels and torch CPU ops; BOT_ORACLE_THREADS overrides.  32 is the fastest on the
    256-thread hosts of the GPU boxes (S-products oracle step: 104-130 s at 32 threads, 147 s at 64, 216 s at 128)."""
    return min(os.cpu_count() or 1, int(os.environ.get("BOT_ORACLE_THREADS", default)))


def edge_gat_oracle_step(src, dst, n, nfeat, efeat, labels, train_idx, sd, *, n_layers, n_heads, n_hidden, node_loss, use_node_encoder,
                         residual, threads=None, gates=None, f64_weight_grads=False, dtype=torch.float32):
    """One train step of the ogbn-proteins / ogbn-products stack (full-graph branch) on the oracle's C kernels.
    `f64_weight_grads`: the Linear weight gradients accumulated in fp64 (oracle.ref_models.linear_f64grad).
    `dtype=torch.float64`: the whole step in double (liboracle_f64.so + torch fp64) — the exact side when two fp32 runs are
    ranked against each other."""
    from oracle import c_ops
    from oracle import ref_models as RM
    if threads is None:
        threads = _oracle_threads(32)
    torch.set_num_threads(threads)
    c_ops.set_num_threads(threads)
    g = c_ops.CGraph(src, dst, n)
    sdg = {k: (v.to(dtype).requires_grad_() if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    if dtype != torch.float32:
        sdg = {k: (v.to(dtype) if v.is_floating_point() and not v.requires_grad else v) for k, v in sdg.items()}
        nfeat = None if nfeat is None else nfeat.to(dtype)
        efeat = None if efeat is None else efeat.to(dtype)
    names = [k for k, v in sdg.items() if v.requires_grad and (use_node_encoder or not k.startswith("node_encoder"))]
    kg = None if gates is None else KinkGates(*gates)
    t0 = time.perf_counter()
    pred = RM.proteins_gat_forward(g, nfeat, efeat, sdg, n_layers=n_layers, n_heads=n_heads, n_hidden=n_hidden, training=True,
                                   use_node_encoder=use_node_encoder, residual=residual,
                                   activation=F.relu if kg is None else kg.relu, leaky=None if kg is None else kg.leaky,
                                   linear=RM.linear_f64grad if f64_weight_grads else F.linear)
    out = node_loss(pred[train_idx], labels[train_idx]).mean()
    grads = torch.autograd.grad(out, [sdg[k] for k in names], allow_unused=True)
    return pred.detach(), {k: g_ for k, g_ in zip(names, grads) if g_ is not None}, time.perf_counter() - t0, (kg.stats if kg else None)


def edge_gat_hip_step(model, g, labels, train_idx, node_loss):
    """The same step on the HIP path (model already on the device, inputs in g.ndata / g.edata)."""
    model.train()
    model.zero_grad(set_to_none=True)
    with tap_kinks() as (relu_gates, leaky_gates):
        pred = model(g)
        node_loss(pred[train_idx], labels[train_idx]).mean().backward()
    return pred.detach(), {k: p.grad.detach() for k, p in model.named_parameters() if p.grad is not None}, (relu_gates, leaky_gates)


# ---------------------------------------------------------------------------------------------- one entry point per BASELINE config
PARITY_SCALE = {"cora": 1.0, "arxiv": 1.0, "reddit": 1.0, "proteins": 0.125, "products": 0.125}   # bench.py's bounded CPU sample


def workload_parity(name, dev, scale=1.0, exact="auto", timed=False, warm=False, gpu_steps=0):
    """One train step (drop rates 0) of BASELINE config `name` (bot_amd.workloads) on the HIP path against the oracle's C
    kernels on the host cores, graph of the workload's generator at `scale` (1.0 = the size the bench line is quoted on; the
    density — mean degree — does not depend on it).  The oracle runs at the HIP run's ReLU / leaky-ReLU gates (KinkGates).
    `exact`: also run the step in fp64 and rank the two fp32 runs against it ("auto": config 4, whose logits reach 125).
    `timed`: the fp32 oracle step's seconds are reported as a CPU baseline — run it alone, not beside the fp64 leg.
    `warm`: with `timed`, the oracle step runs TWICE and the second one is the timing (first-touch page faults of its GB-sized
    temporaries are in the first).  `gpu_steps` > 0: that many train steps of the SAME graph on the HIP path are timed too
    (after 2 warm-ups, drop rates 0) and come back as "gpu_seconds_per_step" — the GPU rate on the graph the CPU number is from.
    Returns (parity dict incl. "criterion" / "ok", {"seconds", "threads", "edges", "nodes"} of the fp32 oracle step).
    Config 2 (arxiv) has its own entry points (oracle_step / hip_step: label mask, fused / modular variants)."""
    from bot_amd import workloads
    t_start = time.perf_counter()
    wl = workloads.build(name, dev, drop=False, scale=scale)
    model, g, ds = wl.model, wl.graph, wl.dataset
    t_built = time.perf_counter()
    s, d = (t.cpu() for t in g.edges())
    n = g.number_of_nodes()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    if exact == "auto":
        exact = name == "proteins"
    xp = xg = None
    zero = None
    if name in ("cora", "reddit"):
        hid, layers = (16, 2) if name == "cora" else (256, 3)
        cfg = dict(n_layers=layers, n_hidden=hid, norm="none" if name == "cora" else "batch", norm_adj="symm", use_linear=False,
                   residual=False)
        pred, grads, gates = gcn_hip_step(g, ds.feat, ds.labels, ds.train_idx, sd, cfg, ds.n_classes)
        rp, rg, secs, gstats = gcn_oracle_step(s, d, n, ds.feat.cpu(), ds.labels.cpu(), ds.train_idx.cpu(), sd, cfg, gates=gates)
        threads = _oracle_threads(32)
    else:
        prot = name == "proteins"
        kw = dict(n_layers=6 if prot else 3, n_heads=6 if prot else 4, n_hidden=80 if prot else 120,
                  node_loss=workloads._bce if prot else workloads._loge, use_node_encoder=prot, residual=prot)
        pred, grads, gates = edge_gat_hip_step(model, g, ds.labels, ds.train_idx, kw["node_loss"])
        t_hip = time.perf_counter()
        args = (s, d, n, ds.feat.cpu(), None if ds.efeat is None else ds.efeat.cpu(), ds.labels.cpu(), ds.train_idx.cpu(), sd)
        # products: Linear weight gradients of the ORACLE accumulated in fp64 (its sgemm over 2.45 M rows is 2.1e-4 off, see
        # oracle.ref_models.linear_f64grad); dst_fc biases sit in front of a training-mode BatchNorm (exact gradient 0)
        if exact and not timed:
            # the fp32 and the fp64 oracle steps side by side on two host threads (each brings its own OpenMP team: 2 x 32 of the
            # GPU box's 256 hardware threads; ctypes and torch release the GIL inside the kernels) — config 4's two legs are 115 s +
            # 200 s one after the other, most of the GPU suite's wall time
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(2) as pool:
                f32 = pool.submit(edge_gat_oracle_step, *args, gates=gates, f64_weight_grads=not prot, **kw)
                f64 = pool.submit(edge_gat_oracle_step, *args, gates=gates, dtype=torch.float64,
                                  threads=_oracle_threads(int(os.environ.get("BOT_ORACLE_THREADS_F64", 32))), **kw)
                rp, rg, secs, gstats = f32.result()
                xp, xg, secs64, _ = f64.result()
            print(f"workload_parity({name}): build {t_built - t_start:.1f} s, HIP step + taps {t_hip - t_built:.1f} s, fp32 oracle {secs:.1f} s || "
                  f"fp64 oracle {secs64:.1f} s (side by side), total so far {time.perf_counter() - t_start:.1f} s")
        else:
            rp, rg, secs, gstats = edge_gat_oracle_step(*args, gates=gates, f64_weight_grads=not prot, **kw)
            if exact:
                xp, xg, _, _ = edge_gat_oracle_step(*args, gates=gates, dtype=torch.float64, **kw)
        zero = {f"convs.{i}.dst_fc.bias": f"convs.{i}.dst_fc.weight" for i in range(kw["n_layers"])}
        threads = _oracle_threads(32)
    r = compare(pred, grads, rp, rg, gstats, zero_grads=zero)
    ok, crit, nums = logits_ok(pred, rp, xp)
    r.update(nums)
    if xg is not None:
        rank = rank_against_exact(grads, rg, xg, zero)
        worst = max(rank, key=lambda k: rank[k][0] / max(1e-4, 2 * rank[k][1]))
        r.update(worst_ranked=worst, hip_err_vs_fp64=rank[worst][0], oracle_err_vs_fp64=rank[worst][1],
                 max_hip_err_vs_fp64=max(v[0] for v in rank.values()), max_oracle_err_vs_fp64=max(v[1] for v in rank.values()))
        ok = ok and all(eh <= max(1e-4, 2 * eo) for eh, eo in rank.values())
        r["rank"] = rank
    else:
        ok = ok and r["max_rel_grad_err"] <= 1e-4
    # gates the two runs set differently may only be rounding noise: |pre-activation| there <= 1e-4, relative to the logit scale where
    # the fp64-ranked criterion applies (config 4: attention logits and activations of O(100))
    gate_tol = 1e-4 * (max(1.0, r["logit_scale"] / 10) if crit == "fp64-ranked" else 1.0)
    r["gate_tolerance"] = gate_tol
    ok = ok and r.get("max_abs_preact_at_differing_gate", 0.0) <= gate_tol
    r.update(criterion=crit, criterion_text=CRITERIA[crit], ok=bool(ok), edges=int(s.numel()), scale=scale)
    cpu = {"seconds": secs, "threads": threads, "edges": int(s.numel()), "nodes": n, "warm": False}
    if timed and warm:
        if name in ("cora", "reddit"):
            cpu["seconds"] = gcn_oracle_step(s, d, n, ds.feat.cpu(), ds.labels.cpu(), ds.train_idx.cpu(), sd, cfg, gates=gates)[2]
        else:
            cpu["seconds"] = edge_gat_oracle_step(*args, gates=gates, f64_weight_grads=not prot, **kw)[2]
        cpu["warm"], cpu["first_step_seconds"] = True, secs
    if gpu_steps > 0:
        for _ in range(2):
            wl.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(gpu_steps):
            wl.step()
        torch.cuda.synchronize()
        cpu["gpu_seconds_per_step"] = (time.perf_counter() - t0) / gpu_steps
    return r, cpu
