"""The callers of the hot path: the train / evaluate step of the reference's full-batch harness
(src/no-sampling/run.py:229-322), restated around `bot_amd.nn` models.  Same tricks, same order of
operations: labels as input features (run.py:240-243, 256-263), label reuse (run.py:274-279),
logit / loge / savage losses (run.py:229-237), RMSprop warm-up (run.py:246-249)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

EPSILON = 1 - math.log(2)  # run.py:34


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
    Returns (loss tensor, pred, train_pred_idx).  `mask` overrides the random split of run.py:258."""
    if mask is None:
        mask = torch.rand(train_idx.shape, device=train_idx.device) < mask_rate
    if use_labels:
        train_labels_idx, train_pred_idx = train_idx[mask], train_idx[~mask]
        feat = add_labels(feat, labels, train_labels_idx, n_classes)
    else:
        train_pred_idx = train_idx[mask]
    pred = model(graph, feat)
    if n_label_iters > 0:
        unlabel_idx = torch.cat([train_pred_idx, val_idx, test_idx])
        for _ in range(n_label_iters):
            pred = pred.detach()
            feat[unlabel_idx, -n_classes:] = F.softmax(pred[unlabel_idx], dim=-1)
            pred = model(graph, feat)
    out = compute_loss(pred[train_pred_idx], labels[train_pred_idx], loss)
    out.backward()
    return out, pred, train_pred_idx


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
