"""Diagnostic (not collected by pytest): fp32 reduction noise of the dst_fc weight gradient of layer 0 at S-products size
(a [480, N] x [N, 100] product over N = 2.45 M rows) — the HIP run's (hipBLASLt) and the oracle's (torch CPU sgemm), each
against an fp64 product of ITS OWN operands.   python tests/diag_products_dw.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from bot_amd import workloads
from oracle import c_ops
from oracle import ref_models as RM

dev = torch.device("cuda", 0)
wl = workloads.build("products", dev, drop=False)
model, g, ds = wl.model.train(), wl.graph, wl.dataset
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
keep = {}
def fwd_hook(m, a, out):
    out.register_hook(lambda gr: keep.__setitem__("g", gr.detach().flatten(1)))
model.convs[0].register_forward_hook(fwd_hook)
pred = model(g)
workloads._loge(pred[ds.train_idx], ds.labels[ds.train_idx]).mean().backward()
dw = model.convs[0].dst_fc.weight.grad
dw64 = (keep["g"].double().t() @ ds.feat.double())
scale = dw64.abs().max()
print("HIP    dst_fc.weight grad vs fp64 of its operands: max err / max entry = %.3e" % ((dw.double() - dw64).abs().max() / scale).item())
threads = min(os.cpu_count() or 1, 32)
torch.set_num_threads(threads); c_ops.set_num_threads(threads)
s, d = (t.cpu() for t in g.edges())
cg = c_ops.CGraph(s, d, g.number_of_nodes())
sdg = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
oc, first = RM.proteins_gatconv_forward, [True]
def tap(*a, **k):
    out = oc(*a, **k)
    if first[0]:
        first[0] = False
        out.register_hook(lambda gr: keep.__setitem__("go", gr.detach().flatten(1)))
    return out
RM.proteins_gatconv_forward = tap
x = ds.feat.cpu()
rp = RM.proteins_gat_forward(cg, x, None, sdg, n_layers=3, n_heads=4, n_hidden=120, training=True, use_node_encoder=False, residual=False)
tr = ds.train_idx.cpu()
workloads._loge(rp[tr], ds.labels.cpu()[tr]).mean().backward()
odw = sdg["convs.0.dst_fc.weight"].grad
odw64 = keep["go"].double().t() @ x.double()
print("oracle dst_fc.weight grad vs fp64 of its operands: max err / max entry = %.3e" % ((odw.double() - odw64).abs().max() / odw64.abs().max()).item())
print("HIP vs oracle fp32: %.3e   HIP fp64 vs oracle fp64: %.3e" % (((dw.cpu().double() - odw.double()).abs().max() / odw64.abs().max()).item(),
                                                                   ((dw64.cpu() - odw64).abs().max() / odw64.abs().max()).item()))
