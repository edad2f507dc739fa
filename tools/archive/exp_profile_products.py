"""Experiment: per-op device time (torch.profiler) of one S-products / S-proteins train step: what torch itself still runs."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import scale_check as SC
from bot_amd.nn import edge_gat
from bot_amd import tuning
tuning.enable()
name = sys.argv[1] if len(sys.argv) > 1 else "products"
g, f, c, _ = SC.build(name)
n, E = g.number_of_nodes(), g.number_of_edges()
if name == "reddit":
    from bot_amd import nn as bnn
    gcn = bnn.GCN(in_feats=f, n_classes=c, n_hidden=256, n_layers=3, activation=F.relu, norm="batch", dropout=0.5).to("cuda").train()
    feat = torch.randn(n, f, device="cuda")
    model = lambda graph: gcn(graph, feat)
    model.zero_grad = lambda set_to_none=True: gcn.zero_grad(set_to_none=set_to_none)
elif name == "products":
    model = edge_gat.ProductsGAT(node_feats=f, edge_feats=0, n_classes=c, n_layers=3, n_heads=4, n_hidden=120, edge_emb=0,
                                 activation=F.relu, dropout=0.5, input_drop=0.1, attn_drop=0.0, edge_drop=0.1).to("cuda").train()
    g.ndata["feat"] = torch.randn(n, f, device="cuda")
else:
    from bot_amd import ops
    model = edge_gat.ProteinsGAT(node_feats=f, edge_feats=8, n_classes=c, n_layers=6, n_heads=6, n_hidden=80, edge_emb=16,
                                 activation=F.relu, dropout=0.25, input_drop=0.1, attn_drop=0.0, edge_drop=0.1,
                                 allow_zero_in_degree=True).to("cuda").train()
    g.edata["feat"] = torch.rand(E, 8, device="cuda")
    g.ndata["feat"] = ops.copy_e_sum(g, g.edata["feat"])
def step():
    model.zero_grad(set_to_none=True)
    out = model(g); out.square().mean().backward()
step(); step(); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:30]:
    print(f"{e.self_device_time_total/1e3:8.2f}ms x{e.count:<4d} {e.key:30s} {str(e.input_shapes)[:90]}")
