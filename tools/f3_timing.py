#!/usr/bin/env python3
"""what `evaluate()` (run.py:290-322) costs at BASELINE config 2 next to the train step — the inference-only layers
(one GEMM + one fused sweep per layer, nothing edge-sized written) against the generic eval-mode forward under no_grad (modular
layers: attention kernel writing a [E,H] + sign bytes, SpMM, BatchNorm kernel), with and without label-reuse iterations."""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd import nn as bnn, synth, train as T, tuning  # noqa: E402

tuning.enable()
dev = "cuda"
ds = synth.make_dataset("arxiv", device="cpu", seed=0)
g = ds.graph.to(dev)
g.create_formats_()
C = ds.n_classes
torch.manual_seed(0)
model = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, n_layers=3, n_heads=3, n_hidden=250,
                norm="batch", dropout=0.75, input_drop=0.25, attn_drop=0.1, linear=True).to(dev)
opt = torch.optim.RMSprop(model.parameters(), lr=0.002)
feat, labels = ds.feat.to(dev), ds.labels.to(dev)
tr, va, te = ds.train_idx.to(dev), ds.val_idx.to(dev), ds.test_idx.to(dev)


def timed(fn, it=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


step = timed(lambda: T.train_step(model, g, feat, labels, tr, va, te, opt, use_labels=True, mask_rate=0.5, loss="loge", n_classes=C))
res = {"train_step_ms": round(step, 3)}
for iters in (0, 1, 3):
    model.fuse_layers = True
    a = timed(lambda: T.evaluate(model, g, feat, labels, tr, va, te, use_labels=True, n_label_iters=iters, loss="loge", n_classes=C))
    model.fuse_layers = False
    b = timed(lambda: T.evaluate(model, g, feat, labels, tr, va, te, use_labels=True, n_label_iters=iters, loss="loge", n_classes=C))
    res[f"evaluate_n_label_iters={iters}"] = {"inference_layers_ms": round(a, 3), "generic_no_grad_ms": round(b, 3),
                                              "fraction_of_train_step": round(a / step, 3)}
model.fuse_layers = True
model.eval()
x = T.add_labels(feat, labels, tr, C)
with torch.no_grad():
    res["forward_only_inference_ms"] = round(timed(lambda: model(g, x)), 3)
    model.fuse_layers = False
    res["forward_only_generic_ms"] = round(timed(lambda: model(g, x)), 3)
print(json.dumps(res))
