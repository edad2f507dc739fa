#!/usr/bin/env python3
"""End-to-end training demo on the HIP path: the reference's recipe (labels as inputs, loge loss, RMSprop with warm-up,
BatchNorm, dropout; src/no-sampling/run.py:252-380) on a synthetic power-law graph with PLANTED labels
(label = argmax of a random linear map of the mean neighbour feature), so accuracy has something to learn.

    python tools/train_demo.py [--epochs 60] [--scale 0.2] [--model gat|gcn]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd import ops, synth, train, tuning  # noqa: E402
from bot_amd import nn as bnn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=60)
    ap.add_argument("--scale", type=float, default=0.2)
    ap.add_argument("--model", default="gat", choices=["gat", "gcn"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    tuning.enable()
    ds = synth.make_dataset("arxiv", device="cpu", scale=args.scale)
    g = ds.graph.to(dev)
    n, C = g.number_of_nodes(), ds.n_classes
    feat = ds.feat.to(dev)
    # planted task: class = argmax of a fixed random projection of the mean-aggregated features (needs message passing)
    torch.manual_seed(123)
    agg = ops.copy_u_sum(g, feat) / g.in_degrees().clamp(min=1).unsqueeze(1)
    labels = (agg @ torch.randn(feat.shape[1], C, device=dev)).argmax(1, keepdim=True)
    tr, va, te = ds.train_idx.to(dev), ds.val_idx.to(dev), ds.test_idx.to(dev)
    if args.model == "gat":
        model = bnn.GAT(dim_node=feat.shape[1] + C, dim_edge=0, dim_output=C, n_hidden=64, n_layers=3, n_heads=3, activation=F.relu,
                        norm="batch", dropout=0.5, input_drop=0.1, attn_drop=0.05, linear=True).to(dev)
    else:
        model = bnn.GCN(in_feats=feat.shape[1] + C, n_classes=C, n_hidden=128, n_layers=3, activation=F.relu, norm="batch",
                        dropout=0.5, use_linear=True).to(dev)
    lr = 0.002
    opt = torch.optim.RMSprop(model.parameters(), lr=lr)
    t0 = time.perf_counter()
    first = last = None
    for epoch in range(1, args.epochs + 1):
        train.adjust_learning_rate(opt, lr, epoch)
        loss, _ = train.train_step(model, g, feat, labels, tr, va, te, opt, use_labels=True, mask_rate=0.5, loss="loge", n_classes=C)
        if epoch == 1 or epoch % 10 == 0 or epoch == args.epochs:
            tra, vaa, tea, trl, val, tel, _ = train.evaluate(model, g, feat, labels, tr, va, te, use_labels=True, loss="loge",
                                                             n_classes=C)
            print(f"epoch {epoch:3d}  train loss {loss.item():.4f}  eval loss {trl.item():.4f}/{val.item():.4f}/{tel.item():.4f}  "
                  f"acc {tra:.3f}/{vaa:.3f}/{tea:.3f}", flush=True)
            if first is None:
                first = (val.item(), vaa)
            last = (val.item(), vaa)
    torch.cuda.synchronize()
    print(f"{args.model}: N={n} E={g.number_of_edges()}  {args.epochs} epochs in {time.perf_counter() - t0:.1f} s;  "
          f"val loss {first[0]:.3f} -> {last[0]:.3f}, val acc {first[1]:.3f} -> {last[1]:.3f} (chance {1 / C:.3f})")
    assert last[0] < first[0] and last[1] > 3.0 / C, "the model did not learn the planted task"


if __name__ == "__main__":
    main()
