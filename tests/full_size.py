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


class GateAct:
    """ReLU with the 0/1 gates GIVEN (one uint8/bool [N, F] tensor per hidden layer, taken from the HIP run) instead of derived
    from the oracle's own pre-activations.  The stack has a ReLU behind every hidden BatchNorm (models.py:726-730); a
    pre-activation within fp32 rounding of zero can fall on either side in two fp32 implementations that sum in different
    orders, and the node's whole contribution then enters or leaves one row of the layer's weight gradients (~1/sqrt(N) of the
    row, far above 1e-4: two CPU restatements of this same step, oracle/ref_ops.py vs oracle/c_ops.py, differ by 0.75 % on such
    a row and by 1e-6 elsewhere).  Evaluating the oracle AT THE HIP RUN'S GATES removes exactly that ambiguity and nothing
    else: the forward changes only where |pre-activation| is rounding noise (recorded in `stats` and bounded by the caller),
    and the gradients of the two implementations become comparable entry by entry."""

    def __init__(self, gates):
        self.gates, self.i, self.stats = gates, 0, []

    def __call__(self, h):
        g = self.gates[self.i % len(self.gates)]
        self.i += 1
        own = h.detach() > 0
        diff = own != g.bool()
        self.stats.append({"differ": int(diff.sum()), "of": diff.numel(),
                           "max_abs_preact_where_differ": float(h.detach().abs()[diff].max()) if bool(diff.any()) else 0.0})
        return h * g.to(h.dtype)


def oracle_step(src, dst, n, feat, labels, train_idx, mask, sd, cfg, n_classes, loss="loge", threads=None, steps=1, gates=None):
    """One train step (forward + loss + backward, training-mode BatchNorm, no dropout) on the oracle's C kernels.
    `gates`: see GateAct (None: plain ReLU).  Returns (pred [N,C], {param name: grad}, [seconds per step], threads, gate stats)."""
    from oracle import c_ops
    from oracle import ref_models as RM
    if threads is None:
        threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    c_ops.set_num_threads(threads)
    g = c_ops.CGraph(src, dst, n)
    sdg = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    names = [k for k, v in sdg.items() if v.requires_grad]
    times, pred, grads, act = [], None, None, F.relu
    for _ in range(steps):
        t0 = time.perf_counter()
        act = F.relu if gates is None else GateAct(gates)
        x = RM.add_labels(feat, labels, train_idx[mask], n_classes)
        pred = RM.gat_forward(g, x, sdg, n_layers=cfg["n_layers"], n_heads=cfg["n_heads"], n_hidden=cfg["n_hidden"],
                              n_classes=n_classes, norm=cfg["norm"], non_interactive_attn=cfg["non_interactive_attn"],
                              use_symmetric_norm=cfg["use_symmetric_norm"], linear=cfg["linear"], residual=cfg["residual"],
                              activation=act, training=True)
        out = RM.compute_loss(pred[train_idx[~mask]], labels[train_idx[~mask]], loss)
        grads = torch.autograd.grad(out, [sdg[k] for k in names])
        times.append(time.perf_counter() - t0)
    return pred.detach(), dict(zip(names, grads)), times, int(c_ops.num_threads()), (act.stats if gates is not None else None)


@contextlib.contextmanager
def tap_hidden():
    """Records the output of every hidden layer's BatchNorm+ReLU(+dropout) epilogue of a bot_amd.nn stack while active — the
    fused layer node (bot_amd.nn.fused.gat_hidden_layer) and the modular epilogue (bot_amd.nn._epilogue) alike — by wrapping
    those two functions from the outside (test code; the product has no hook)."""
    import bot_amd.nn as bnn
    from bot_amd.nn import fused
    taps = []
    orig_e, orig_f = bnn._epilogue, fused.gat_hidden_layer

    def epi(h, norm, activation, dropout, training):
        y = orig_e(h, norm, activation, dropout, training)
        taps.append(y.detach())
        return y

    def hid(conv, bn, graph, h, dropout_p, training):
        y = orig_f(conv, bn, graph, h, dropout_p, training)
        if bn is not None:
            taps.append(y.detach())
        return y

    bnn._epilogue, fused.gat_hidden_layer = epi, hid
    try:
        yield taps
    finally:
        bnn._epilogue, fused.gat_hidden_layer = orig_e, orig_f


def hip_step(g, feat, labels, train_idx, mask, sd, cfg, n_classes, loss="loge", fuse=True):
    """The same step on the HIP path: bot_amd.nn.GAT + bot_amd.train.forward_backward on the device of `g`.
    Returns (pred, {param: grad}, [ReLU gates of the hidden layers as CPU uint8 tensors])."""
    from bot_amd import nn as bnn
    from bot_amd import train as T
    dev = feat.device
    model = bnn.GAT(dim_node=feat.shape[1] + n_classes, dim_edge=0, dim_output=n_classes, activation=F.relu, dropout=0.0,
                    input_drop=0.0, attn_drop=0.0, edge_drop=0.0, **cfg)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train()
    model.fuse_layers = fuse
    with tap_hidden() as taps:
        _, pred, _ = T.forward_backward(model, g, feat, labels, train_idx, train_idx[:0], train_idx[:0], use_labels=True,
                                        loss=loss, n_classes=n_classes, mask=mask.to(dev))
    assert len(taps) == cfg["n_layers"] - 1
    gates = [(t > 0).to(torch.uint8).cpu() for t in taps]
    return pred.detach(), {k: p.grad.detach() for k, p in model.named_parameters()}, gates


def compare(pred_hip, grads_hip, pred_ref, grads_ref, gate_stats=None, tol=1e-4):
    """max |logit diff| over all nodes; the worst parameter-gradient error relative to that gradient's largest entry; the
    number of gradient entries beyond `tol` of it; and (with gates given to the oracle) how many ReLU gates the two runs
    would have set differently and how large the pre-activation was there."""
    d = float((pred_hip.detach().cpu().double() - pred_ref.double()).abs().max())
    worst, which, over, total = 0.0, None, 0, 0
    for k, gr in grads_ref.items():
        gh = grads_hip[k].detach().cpu().double()
        scale = max(float(gr.double().abs().max()), 1e-30)
        e = (gh - gr.double()).abs() / scale
        over += int((e > tol).sum())
        total += e.numel()
        if float(e.max()) > worst:
            worst, which = float(e.max()), k
    r = {"max_abs_logit_diff": d, "max_rel_grad_err": worst, "worst_grad": which, "grad_entries_over_1e-4": over,
         "grad_entries": total, "n": int(pred_ref.shape[0]), "logit_scale": float(pred_ref.abs().max())}
    if gate_stats is not None:
        r["relu_gates_differing"] = sum(s["differ"] for s in gate_stats)
        r["relu_gates"] = sum(s["of"] for s in gate_stats)
        r["max_abs_preact_at_differing_gate"] = max(s["max_abs_preact_where_differ"] for s in gate_stats)
    return r
