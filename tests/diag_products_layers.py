"""Diagnostic (not collected by pytest): the S-products GAT forward at full size, layer by layer, HIP path vs the oracle's C
kernels — the conv output and the post-BatchNorm+ReLU output of every layer.     python tests/diag_products_layers.py [scale]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from bot_amd import workloads
from oracle import c_ops
from oracle import ref_models as RM

dev = torch.device("cuda", 0)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
wl = workloads.build("products", dev, drop=False, scale=scale)
model, g, ds = wl.model.train(), wl.graph, wl.dataset
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
hip = {}
for i, conv in enumerate(model.convs):
    conv.register_forward_hook(lambda m, a, out, i=i: hip.__setitem__(("conv", i), out.detach().flatten(1).cpu()))
from bot_amd.nn import edge_gat
orig = edge_gat._epilogue
cnt = [0]
def epi(h, norm, activation, dropout, training, **kw):
    y = orig(h, norm, activation, dropout, training, **kw)
    hip[("act", cnt[0])] = y.detach().cpu(); cnt[0] += 1
    return y
edge_gat._epilogue = epi
with torch.no_grad():
    pred = model(g).cpu()
threads = min(os.cpu_count() or 1, 32)
torch.set_num_threads(threads); c_ops.set_num_threads(threads)
s, d = (t.cpu() for t in g.edges())
cg = c_ops.CGraph(s, d, g.number_of_nodes())
ref = {}
oc = RM.proteins_gatconv_forward
ci = [0]
def conv_tap(*a, **k):
    out = oc(*a, **k)
    ref[("conv", ci[0])] = out.detach().flatten(1); ci[0] += 1
    return out
RM.proteins_gatconv_forward = conv_tap
ai = [0]
def act_tap(h):
    y = F.relu(h)
    ref[("act", ai[0])] = y.detach(); ai[0] += 1
    return y
with torch.no_grad():
    rp = RM.proteins_gat_forward(cg, ds.feat.cpu(), None, sd, n_layers=3, n_heads=4, n_hidden=120, training=True, use_node_encoder=False,
                                 residual=False, activation=act_tap)
for k in sorted(ref, key=lambda t: (t[1], t[0] == "act")):
    dlt = (hip[k] - ref[k]).abs()
    rows = dlt.amax(1) > 1e-3
    idx = rows.nonzero().flatten()
    print(k, "max diff %.3e" % dlt.max().item(), "rows over 1e-3: %d of %d" % (int(rows.sum()), rows.numel()),
          ("first %d last %d" % (int(idx[0]), int(idx[-1]))) if idx.numel() else "", "finite", bool(torch.isfinite(hip[k]).all()))
print("logits max diff %.3e" % (pred - rp).abs().max().item())
