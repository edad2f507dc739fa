"""Shipped GEMM selections for the config 3-5 shapes vs the library default: same model, same inputs, forward output and
parameter gradients of one training-mode step with TunableOp off and on (all dropout rates 0), next to the model's own
sensitivity to a 1e-6 relative change of its inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import bot_amd
from bot_amd import ops, synth, tuning
from bot_amd import nn as bnn
from bot_amd.nn import edge_gat
from torch.cuda import tunable
import scale_check as SC
DEV = "cuda"
for name in sys.argv[1:] or ["reddit", "proteins", "products"]:
    g, f, c, _ = SC.build(name)
    n, E = g.number_of_nodes(), g.number_of_edges()
    torch.manual_seed(0)
    if name == "reddit":
        model = bnn.GCN(in_feats=f, n_classes=c, n_hidden=256, n_layers=3, activation=F.relu, norm="batch", dropout=0.0).to(DEV)
        feat = torch.randn(n, f, device=DEV)
        fwd = lambda: model(g, feat)
    elif name == "products":
        model = edge_gat.ProductsGAT(node_feats=f, edge_feats=0, n_classes=c, n_layers=3, n_heads=4, n_hidden=120, edge_emb=0,
                                     activation=F.relu, dropout=0.0, input_drop=0.0, attn_drop=0.0, edge_drop=0.0).to(DEV)
        g.ndata["feat"] = torch.randn(n, f, device=DEV)
        fwd = lambda: model(g)
    else:
        model = edge_gat.ProteinsGAT(node_feats=f, edge_feats=8, n_classes=c, n_layers=6, n_heads=6, n_hidden=80, edge_emb=16,
                                     activation=F.relu, dropout=0.0, input_drop=0.0, attn_drop=0.0, edge_drop=0.0,
                                     allow_zero_in_degree=True).to(DEV)
        g.edata["feat"] = torch.rand(E, 8, device=DEV)
        g.ndata["feat"] = ops.copy_e_sum(g, g.edata["feat"])
        fwd = lambda: model(g)
    model.train()  # batch statistics in the norms (the training path); every dropout rate is 0, so the step is deterministic
    res = []
    for on in (False, False, "perturbed", True):
        if on == "perturbed":  # conditioning of the model itself: library-default kernels, node features scaled by (1 + 1e-6)
            key = "feat"
            if name == "reddit":
                feat.mul_(1.0 + 1e-6)
            else:
                g.ndata[key] = g.ndata[key] * (1.0 + 1e-6)
            op = fwd().detach().double()
            print(f"S-{name}: default kernels, inputs scaled by (1 + 1e-6): forward max|diff|/max|out| {float((op - res[0][0]).abs().max() / res[0][0].abs().max()):.2e}")
            if name == "reddit":
                feat.div_(1.0 + 1e-6)
            else:
                g.ndata[key] = g.ndata[key] / (1.0 + 1e-6)
            continue
        if on:
            print("tuning file loaded:", tuning.enable())
        else:
            tunable.enable(False)
        model.zero_grad(set_to_none=True)
        out = fwd()
        out.square().mean().backward()
        res.append((out.detach().double(), [p.grad.detach().double() for p in model.parameters() if p.grad is not None]))
    (oa, ga), (o0, g0), (o1, g1) = res
    print('run-to-run (both untuned): forward', float((oa - o0).abs().max()), 'grads', max(float((a - b).abs().max()) for a, b in zip(ga, g0)))
    eo = float((o0 - o1).abs().max() / o0.abs().max())
    gmax = max(float(a.abs().max()) for a in g0)   # one scale for all parameters: biases in front of a norm have zero gradient
    eg = max(float((a - b).abs().max()) for a, b in zip(g0, g1)) / gmax
    print(f"S-{name}: forward max|diff|/max|out| {eo:.2e}; parameter gradients max|diff| / largest |grad| {eg:.2e}")
    del model, g
    torch.cuda.empty_cache()
