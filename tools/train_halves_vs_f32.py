#!/usr/bin/env python3
"""Evidence that the fp16-halves GEMMs train like the stock fp32 GEMMs: the config-2 model (GAT 3 x 3 x 250, the reference's
recipe: labels as inputs, loge loss, RMSprop with warm-up, dropout 0.75 / 0.25 / 0.1) on the full-size S-arxiv graph with
planted labels, trained twice from the same seeds (same weights, same label masks, same Philox dropout streams) — once per GEMM
mode — loss by loss.   python tools/train_halves_vs_f32.py [epochs=60]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from bot_amd import gemm, ops, synth, train, tuning, workloads
from bot_amd import nn as bnn

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
tuning.enable()
ds = synth.make_dataset("arxiv", device="cpu", scale=1.0)
g = ds.graph.to(dev)
n, C = g.number_of_nodes(), ds.n_classes
feat = ds.feat.to(dev)
torch.manual_seed(123)
agg = ops.copy_u_sum(g, feat) / g.in_degrees().clamp(min=1).unsqueeze(1)
labels = (agg @ torch.randn(feat.shape[1], C, device=dev)).argmax(1, keepdim=True)
tr, va, te = ds.train_idx.to(dev), ds.val_idx.to(dev), ds.test_idx.to(dev)


def run(mode):
    gemm.MODE = mode
    torch.manual_seed(7)
    model = bnn.GAT(dim_node=feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, **workloads.ARXIV_GAT).to(dev)
    opt = torch.optim.RMSprop(model.parameters(), lr=0.002)
    losses, t0 = [], time.perf_counter()
    for epoch in range(1, epochs + 1):
        train.adjust_learning_rate(opt, 0.002, epoch)
        loss, _ = train.train_step(model, g, feat, labels, tr, va, te, opt, use_labels=True, mask_rate=0.5, loss="loge", n_classes=C)
        losses.append(loss.item())
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    ev = train.evaluate(model, g, feat, labels, tr, va, te, use_labels=True, loss="loge", n_classes=C)
    return losses, secs, ev


la, ta, ea = run("f32")
lb, tb, eb = run("halves")
lc, _, _ = run("f32")          # the same mode twice: bit-identical (deterministic kernels, same seeds)
print(f"S-arxiv N={n} E={g.number_of_edges()}, GAT 3x3x250 (config 2 recipe, planted labels), {epochs} epochs per run, same seeds")
print("epoch   loss (stock fp32)   loss (halves)    |diff| / loss")
for e in list(range(0, min(10, epochs))) + list(range(19, epochs, 10)):
    print(f"{e + 1:5d}   {la[e]:.6f}          {lb[e]:.6f}       {abs(la[e] - lb[e]) / abs(la[e]):.1e}")
print("largest |diff| / loss over all epochs: %.2e" % max(abs(a - b) / abs(a) for a, b in zip(la, lb)))
print("stock fp32 run repeated: losses %s" % ("bit-identical" if la == lc else "differ by %.1e" % max(abs(a - b) for a, b in zip(la, lc))))
for name, ev in (("stock fp32", ea), ("halves", eb)):
    tra, vaa, tea, trl, val, tel, _ = ev
    print(f"{name:11s} after {epochs} epochs: acc {tra:.4f}/{vaa:.4f}/{tea:.4f}  eval loss {trl.item():.4f}/{val.item():.4f}/{tel.item():.4f}")
