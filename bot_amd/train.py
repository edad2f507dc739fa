"""The callers of the hot path: the train / evaluate step of the reference's full-batch harness
(src/no-sampling/run.py:229-322), restated around `bot_amd.nn` models.  Same tricks, same order of
operations: labels as input features (run.py:240-243, 256-263), label reuse (run.py:274-279),
logit / loge / savage losses (run.py:229-237), RMSprop warm-up (run.py:246-249)."""
from __future__ import annotations

import math
import os
import weakref

import torch
import torch.nn.functional as F

EPSILON = 1 - math.log(2)  # run.py:34
# One launch each for the label / prediction split, the input assembly (+ input dropout), the per-node loss with its gradient, and
# (bot_amd.optim.RMSprop) the optimizer update, instead of ~45 small tensor ops (include/bot_gnn.h "the train step's glue").
# BOT_FUSED_STEP=0 restores the tensor-op form of the same step (kept: label reuse takes it, and the tests compare the two).
FUSED_STEP = os.environ.get("BOT_FUSED_STEP", "1") != "0"
_SPLIT = {}   # (device, N, id(train_idx)) -> (weakref to train_idx, code int32 [N], wn float32 [N], train_idx._version)


def _split_buffers(train_idx, n):
    """`code` (-1) and `wn` (0) arrays of bot_label_split_f32: entries of non-training nodes are never written, so they are
    initialised once per train_idx TENSOR (identity, not address: a recycled address with other contents must not reuse them).
    One entry PER train_idx tensor, kept for as long as that tensor lives (a CapturedTrainStep bakes the buffers' addresses into its
    hipGraph and its step function holds train_idx: another split of the same N - a second model, a k-fold split, a test helper -
    must not evict them); the entry goes when the tensor is collected."""
    key = (train_idx.device, n, id(train_idx))
    ent = _SPLIT.get(key)
    if ent is None or ent[0]() is not train_idx or ent[3] != train_idx._version:
        ent = (weakref.ref(train_idx), torch.full((n,), -1, dtype=torch.int32, device=train_idx.device),
               torch.zeros(n, dtype=torch.float32, device=train_idx.device), train_idx._version)
        if key not in _SPLIT:
            weakref.finalize(train_idx, _SPLIT.pop, key, None)
        _SPLIT[key] = ent
    return ent[1], ent[2]


class _NodeLoss(torch.autograd.Function):
    """mean over the prediction nodes of the per-node loss (run.py:229-237 on pred[train_pred_idx]) as ONE kernel that also
    leaves the gradient: sum_n wn[n] y_n / count."""

    @staticmethod
    def forward(ctx, pred, labels, wn, count, kind):
        from . import _C
        y, dx = _C.node_loss(pred, labels, wn, count, kind, EPSILON, want_grad=pred.requires_grad)
        ctx.save_for_backward(dx)
        return _C.colsum(y.view(-1, 64)).sum() / count[0]      # fixed-order two-stage sum, then 64 values in one workgroup

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        return dx * g, None, None, None, None


def _fused_forward_backward(model, graph, feat, labels, train_idx, *, use_labels, mask_rate, loss, n_classes, mask, count_reduce=None):
    """`count_reduce(count)`: called on the device word that holds the number of prediction nodes before the loss reads it - the
    partitioned step (bot_amd.dist) all-reduces it there, which makes the local loss this rank's additive share of the GLOBAL mean and
    scales its gradient accordingly."""
    from . import _C
    from .nn import fused
    from .ops import new_dropout_seed
    n = feat.shape[0]
    code, wn = _split_buffers(train_idx, n)
    count = _C.label_split(train_idx, labels, mask, mask_rate, new_dropout_seed(1.0) if mask is None else 0, use_labels,
                           code if use_labels else None, wn)
    if count_reduce is not None:
        count_reduce(count)
    from . import nn as bnn
    # only stacks known to honour `input_already_dropped` hand their input dropout over; any other model keeps its own
    drop = getattr(model, "input_drop", None) if isinstance(model, (bnn.GAT, bnn.GCN)) else None
    if use_labels:
        # layer 0's operand in one pass: features, one-hot label block of the input-label nodes, input dropout (models.py:711) —
        # the stack is told not to drop it again
        p = float(drop.p) if (drop is not None and model.training) else 0.0
        feat = _C.build_input(feat, code, n_classes, p, new_dropout_seed(p))
        with fused.input_already_dropped(drop is not None):
            pred = model(graph, feat)
    else:
        pred = model(graph, feat)
    if pred.shape[1] <= 128:
        out = _NodeLoss.apply(pred, labels, wn, count, loss)
    else:       # wider than the loss kernel's 128 classes: the same weighted mean with tensor ops
        from .ops import sum_all
        y = per_node_loss(pred, labels.clamp(0, pred.shape[1] - 1), loss)
        out = sum_all(torch.where(wn > 0, y, torch.zeros_like(y))) / count[0]
    out.backward()
    # (`wn` is the per-train_idx buffer the next step overwrites: callers get their own copy, except inside a hipGraph capture, where the
    # replay rewrites the buffer in place and the alias is what a caller wants to read)
    return out, pred, wn if (wn.is_cuda and torch.cuda.is_current_stream_capturing()) else wn.clone()


def add_labels(feat, labels, idx, n_classes):
    """One-hot of the known training labels appended to the features — run.py:240-243."""
    onehot = torch.zeros([feat.shape[0], n_classes], device=feat.device, dtype=feat.dtype)
    onehot[idx, labels[idx, 0]] = 1
    return torch.cat([feat, onehot], dim=-1)


def per_node_loss(x, labels, loss="logit"):
    """The per-node term of run.py:229-236: cross entropy, optionally reshaped by loge or savage."""
    y = F.cross_entropy(x, labels[:, 0], reduction="none")
    if loss == "loge":
        y = torch.log(EPSILON + y) - math.log(EPSILON)
    elif loss == "savage":
        y = (1 - torch.exp(-y)) ** 2
    elif loss != "logit":
        raise ValueError(f"unknown loss {loss!r}")
    return y


def compute_loss(x, labels, loss="logit"):
    """run.py:229-237 — mean over the nodes of the (reshaped) cross entropy."""
    return torch.mean(per_node_loss(x, labels, loss))


def adjust_learning_rate(optimizer, lr, epoch):
    """Linear warm-up over the first 50 epochs, used with RMSprop — run.py:246-249."""
    if epoch <= 50:
        for group in optimizer.param_groups:
            group["lr"] = lr * epoch / 50


def compute_acc(pred, labels):
    return ((torch.argmax(pred, dim=1) == labels[:, 0]).float().sum() / len(pred)).item()


def forward_backward(model, graph, feat, labels, train_idx, val_idx, test_idx, *, use_labels=True, mask_rate=0.5,
                     n_label_iters=0, loss="logit", n_classes=None, mask=None):
    """Forward + loss + backward of `train()` — run.py:252-284 without the optimizer step.
    Returns (loss tensor, pred, w) with w the 0/1 loss weights (1 = a prediction node of this step, run.py:259-261 / :267): per
    training node in the tensor-op form, per NODE ([N], zero outside the training set) in the fused form.  `mask` overrides the
    random split of run.py:258.

    Written with FIXED shapes: the reference's `train_idx[mask]` / `train_idx[~mask]` (boolean indexing) makes the host wait
    for the device and gives tensors whose size changes from step to step; here the one-hot label block is written with the
    mask as its values and the loss is the weighted mean over ALL training nodes — the same sets, the same numbers (up to the
    summation order of a mean), no synchronisation, and a launch sequence a hipGraph can capture (CapturedTrainStep)."""
    if FUSED_STEP and n_label_iters == 0 and feat.dtype == torch.float32 and loss in ("logit", "loge", "savage"):
        return _fused_forward_backward(model, graph, feat, labels, train_idx, use_labels=use_labels, mask_rate=mask_rate, loss=loss,
                                       n_classes=n_classes, mask=mask)
    if mask is None:
        mask = torch.rand(train_idx.shape, device=train_idx.device) < mask_rate
    if use_labels:
        onehot = torch.zeros([feat.shape[0], n_classes], device=feat.device, dtype=feat.dtype)
        onehot[train_idx, labels[train_idx, 0]] = mask.to(feat.dtype)          # run.py:240-243 for idx = train_idx[mask]
        feat = torch.cat([feat, onehot], dim=-1)
        w = (~mask).to(feat.dtype)                                              # train_pred_idx = train_idx[~mask]
    else:
        w = mask.to(feat.dtype)                                                 # run.py:265-267
    pred = model(graph, feat)
    if n_label_iters > 0 and use_labels:
        # label reuse (run.py:274-279): nodes without an input label — masked-out training, validation and test nodes — get
        # the previous prediction's softmax in their label columns, then the model runs again
        m = mask.unsqueeze(1)
        for _ in range(n_label_iters):
            pred = pred.detach()
            prob = F.softmax(pred, dim=-1)
            feat[train_idx, -n_classes:] = torch.where(m, feat[train_idx, -n_classes:], prob[train_idx])
            for idx in (val_idx, test_idx):
                feat[idx, -n_classes:] = prob[idx]
            pred = model(graph, feat)
    # weighted mean over ALL nodes (weight 0 outside the prediction set) instead of `pred[train_pred_idx]`: the backward of an
    # index gather is an index_put with accumulation, which sorts its indices on the device every step
    # Nodes outside the prediction set contribute nothing, as in `pred[train_pred_idx]` (run.py:281): their labels may be
    # placeholders (-1, NaN cast to int) — clamped into the class range before the cross entropy so that it neither traps nor
    # indexes out of range — and their per-node terms are dropped with where(), not multiplied by 0 (0 * inf = NaN)
    wn = torch.zeros(pred.shape[0], device=pred.device, dtype=pred.dtype)
    wn[train_idx] = w
    y = per_node_loss(pred, labels.clamp(0, pred.shape[1] - 1), loss)
    from .ops import sum_all
    out = sum_all(torch.where(wn > 0, y, torch.zeros_like(y))) / sum_all(w)   # N-sized sums: the library's kernel (ops._SumAll)
    out.backward()
    return out, pred, w


def train_step(model, graph, feat, labels, train_idx, val_idx, test_idx, optimizer, **kw):
    """One `train()` call — run.py:252-287: zero_grad, forward, loss, backward, optimizer step."""
    model.train()
    optimizer.zero_grad()
    loss, pred, _ = forward_backward(model, graph, feat, labels, train_idx, val_idx, test_idx, **kw)
    optimizer.step()
    return loss, pred


@torch.no_grad()
def evaluate(model, graph, feat, labels, train_idx, val_idx, test_idx, *, use_labels=True, n_label_iters=0, loss="logit",
             n_classes=None):
    """run.py:290-322 — eval-mode forward with every training label as input, optional label reuse."""
    import contextlib
    from .nn import fused
    model.eval()
    n_static = feat.shape[1]
    if use_labels:
        feat = add_labels(feat, labels, train_idx, n_classes)
    # between the 1 + n_label_iters forward calls only the label columns change (run.py:304-308): the inference path computes
    # the feature columns' share of the first projection once
    with (fused.label_reuse(n_static) if (use_labels and n_label_iters > 0) else contextlib.nullcontext()):
        pred = model(graph, feat)
        if n_label_iters > 0:
            unlabel_idx = torch.cat([val_idx, test_idx])
            for _ in range(n_label_iters):
                feat[unlabel_idx, -n_classes:] = F.softmax(pred[unlabel_idx], dim=-1)
                pred = model(graph, feat)
    losses = tuple(compute_loss(pred[i], labels[i], loss) for i in (train_idx, val_idx, test_idx))
    accs = tuple(compute_acc(pred[i], labels[i]) for i in (train_idx, val_idx, test_idx))
    return accs + losses + (pred,)


# Seconds drain_rccl_watchdog() sleeps after the device synchronisation: ProcessGroupNCCL's watchdog wakes every 100 ms
# (kWatchdogThreadSleepMillis) and retires every completed Work in one pass: ten periods of margin (a capture happens once per run; the
# experiments of profiles/r06_capture_watchdog.txt ran with five).
CAPTURE_DRAIN_S = float(os.environ.get("BOT_CAPTURE_DRAIN_S", "1.0"))


def drain_rccl_watchdog(device=None) -> bool:
    """Empty ProcessGroupNCCL's watchdog list before a hipGraph capture that contains collectives.  Returns whether it had to.

    ROOT CAUSE of round 5's SIGABRT (native trace: profiles/r06_abort_trace.txt; mechanism pinned by tools/exp_capture_watchdog.py,
    profiles/r06_capture_watchdog.txt): on this ROCm (HIP 7.0.51831) hipEventQuery refuses an event whose last-recorded STREAM is
    currently inside a capture - hipErrorCapturedEvent, "operation not permitted on an event last recorded in a capturing stream" -
    even when the record itself was EAGER and completed long before the capture began (CUDA refuses only events recorded during the
    capture).  ProcessGroupNCCL keeps every eager collective's Work in a list that its watchdog THREAD polls every 100 ms with exactly
    that query (WorkNCCL::isCompleted -> finishedGPUExecutionInternal -> hipEventQuery of the end event, recorded on RCCL's stream).  The
    warm-up steps in front of a capture leave such Works behind; the first captured collective makes RCCL's stream join the capture; a
    watchdog pass that falls between that and the join-back (the halo exchange's overlap window, a few ms per captured step) gets the error,
    rethrows it, and std::terminate ends the process - about one capture in ten.  Works issued DURING a capture are never enqueued
    (ProcessGroupNCCL checks the current stream's capture status), so an empty list stays empty: synchronise (every eager Work is complete)
    and give the watchdog ten of its periods to retire them.  `capture_error_mode="thread_local"` (below) is still needed: it covers the
    watchdog's OTHER illegal-under-global-capture calls."""
    import time
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or not torch.cuda.is_available():
        return False
    if "nccl" not in str(dist.get_backend()).lower():
        return False
    torch.cuda.synchronize(device)
    time.sleep(CAPTURE_DRAIN_S)
    return True


class CapturedTrainStep:
    """One `train()` call (run.py:252-287: forward, loss, backward, optimizer step) captured ONCE into a hipGraph and replayed:
    ~300 launches per step cost one graph launch on the host, which is what bounds a rank once its GPU work drops to a few ms
    (8-way partitions).  Works because the step has fixed shapes and no host synchronisation (forward_backward), the C ABI
    never allocates or synchronises (include/bot_gnn.h), and the optimizer is constructed with `capturable=True`.

    Fresh randomness per replay: torch's own generators are graph-safe (mask split, input / attention dropout); the fused
    BatchNorm+ReLU+dropout kernels bake their Philox seed into the launch, so they additionally read a device word
    (`bot_amd._C.SEED_OFFSET`) that the captured step bumps first — forward and backward of one replay see the same value.
    Training-time edge drop (`edge_drop > 0`, the edge-feature GATs) draws its mask seed on the host and is not supported here.

    `step_fn()` -> (loss, pred) must run the WHOLE step (model.train(), zero_grad(set_to_none=True), ..., optimizer.step())."""

    def __init__(self, step_fn, device, warmup: int = 3):
        from . import _C
        self.device = torch.device(device)
        if _C.SEED_OFFSET is None or _C.SEED_OFFSET.device != self.device:
            _C.SEED_OFFSET = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._off = _C.SEED_OFFSET

        def body():
            self._off.add_(1)
            return step_fn()

        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):                       # warm-up off the default stream: lazy plans, autotuned GEMMs, caches
            for _ in range(warmup):
                body()
        torch.cuda.current_stream(self.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # With a process group alive, RCCL's watchdog THREAD polls its work events at any time; under the default "global" capture mode
        # that poll is an illegal call during capture and aborts the process ("operation not permitted when stream is capturing" raised
        # from ProcessGroupNCCL's watchdog).  "thread_local" checks only the capturing thread's calls - and the watchdog must have NOTHING
        # to poll while RCCL's stream is inside the capture (drain_rccl_watchdog: the round-5 abort).
        import torch.distributed as dist
        mode = "thread_local" if dist.is_available() and dist.is_initialized() else "global"
        self.drained = drain_rccl_watchdog(self.device)         # the warm-up's eager collectives: retired before RCCL's stream joins the capture
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.loss, self.pred = body()

    def __call__(self):
        from .nn import fused
        self.graph.replay()
        fused.bump_generation()     # parameters and BatchNorm statistics moved inside the graph: no version counter saw it
        return self.loss, self.pred


def captured_train_step(model, graph, feat, labels, train_idx, val_idx, test_idx, optimizer, warmup=3, **kw) -> CapturedTrainStep:
    """`train_step` as a replayable hipGraph; `optimizer` must have been built with capturable=True."""
    if not all(g.get("capturable", False) for g in optimizer.param_groups):
        raise ValueError("captured_train_step needs an optimizer constructed with capturable=True")

    def step():
        model.train()
        optimizer.zero_grad(set_to_none=True)
        loss, pred, _ = forward_backward(model, graph, feat, labels, train_idx, val_idx, test_idx, **kw)
        optimizer.step()
        return loss, pred
    return CapturedTrainStep(step, feat.device, warmup)
