#!/usr/bin/env python3
"""Per-parameter breakdown of the full-size config-2 parity run (tests/full_size.py) on the GPU box."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd import synth
from tests import full_size as FS

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
ds = synth.make_dataset("arxiv", device="cpu", seed=0, scale=scale)
C = ds.n_classes
sd = FS.init_state(FS.GAT_ARXIV, ds.feat.shape[1] + C, C, seed=0)
mask = torch.rand(ds.train_idx.shape, generator=torch.Generator().manual_seed(7)) < 0.5
s, d = ds.graph.edges()
n = ds.graph.number_of_nodes()
g = ds.graph.to("cuda"); g.create_formats_()
for fuse in (True, False):
    pred, grads, gates = FS.hip_step(g, ds.feat.cuda(), ds.labels.cuda(), ds.train_idx.cuda(), mask, sd, FS.GAT_ARXIV, C, fuse=fuse)
    rp, rg, t, thr, gs = FS.oracle_step(s, d, n, ds.feat, ds.labels, ds.train_idx, mask, sd, FS.GAT_ARXIV, C, gates=gates)
    print("fused" if fuse else "modular", json.dumps(FS.compare(pred, grads, rp, rg, gs)))
    for k, gr in rg.items():
        gh = grads[k].cpu().double(); gr = gr.double()
        sc = gr.abs().max().item()
        e = (gh - gr).abs() / sc
        l2 = ((gh - gr).norm() / gr.norm()).item()
        print("   %-26s shape %-14s max|g| %.3e  maxerr/max %.2e  relL2 %.2e  frac>1e-4 %.5f" % (k, tuple(gr.shape), sc, e.max().item(), l2, (e > 1e-4).double().mean().item()))
    # a second CPU restatement at the same gates: how far apart are two CPU implementations of the same fp32 math?
    if fuse:
        from oracle import ref_models as RM
        gg = RM.CooGraph(s, d, n)
        sdg = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
        x = RM.add_labels(ds.feat, ds.labels, ds.train_idx[mask], C)
        kg = FS.KinkGates(*gates)
        p2 = RM.gat_forward(gg, x, sdg, n_layers=3, n_heads=3, n_hidden=250, n_classes=C, norm="batch", linear=True, training=True,
                            activation=kg.relu, leaky=kg.leaky)
        out = RM.compute_loss(p2[ds.train_idx[~mask]], ds.labels[ds.train_idx[~mask]], "loge")
        names = [k for k, v in sdg.items() if v.requires_grad]
        g2 = dict(zip(names, torch.autograd.grad(out, [sdg[k] for k in names])))
        print("torch-CPU restatement vs C restatement (same gates):", json.dumps(FS.compare(p2, g2, rp, rg)))
