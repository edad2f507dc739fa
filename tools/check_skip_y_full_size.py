"""One train step of BASELINE config 2 at full size (S-arxiv, drop rates 0 so that both runs draw nothing): hidden states stored as halves
only (BOT_SKIP_Y default) vs stored in fp32, and by-product maxima vs separate passes — logits and every gradient must be BITWISE equal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from bot_amd import nn as bnn, synth, train as T, tuning
from bot_amd.nn import fused
tuning.enable()
dev = "cuda"
ds = synth.make_dataset("arxiv", device="cpu", seed=0)
g = ds.graph.to(dev); g.create_formats_()
C = ds.n_classes
feat, labels = ds.feat.to(dev), ds.labels.to(dev)
tr, va, te = ds.train_idx.to(dev), ds.val_idx.to(dev), ds.test_idx.to(dev)
mask = torch.rand(tr.shape, generator=torch.Generator().manual_seed(1)) < 0.5
mask = mask.to(dev)
res = {}
for name, skip, by in (("fast", True, True), ("stored", False, True), ("passes", True, False)):
    fused.SKIP_Y, fused.ABSMAX_BYPRODUCT = skip, by
    torch.manual_seed(0)
    model = bnn.GAT(dim_node=feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, n_layers=3, n_heads=3, n_hidden=250,
                    norm="batch", dropout=0.0, input_drop=0.0, attn_drop=0.0, linear=True).to(dev).train()
    h0 = fused.HANDLES
    loss, pred, _ = T.forward_backward(model, g, feat, labels, tr, va, te, use_labels=True, loss="loge", n_classes=C, mask=mask)
    res[name] = (pred.detach().clone(), {k: p.grad.clone() for k, p in model.named_parameters()}, fused.HANDLES - h0, float(loss))
print("handles per step:", {k: v[2] for k, v in res.items()}, "loss", res["fast"][3])
for other in ("stored", "passes"):
    same = torch.equal(res["fast"][0], res[other][0]) and all(torch.equal(res["fast"][1][k], res[other][1][k]) for k in res["fast"][1])
    print(f"fast vs {other}: bitwise equal = {same}")
    assert same
