import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import scale_check as SC
from bot_amd.nn import edge_gat
from bot_amd import tuning
tuning.enable()
g, f, c, _ = SC.build("products")
n = g.number_of_nodes()
model = edge_gat.ProductsGAT(node_feats=f, edge_feats=0, n_classes=c, n_layers=3, n_heads=4, n_hidden=120, edge_emb=0,
                             activation=F.relu, dropout=0.5, input_drop=0.1, attn_drop=0.0, edge_drop=0.1).to("cuda").train()
g.ndata["feat"] = torch.randn(n, f, device="cuda")
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
