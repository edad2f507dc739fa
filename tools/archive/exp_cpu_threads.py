"""Pick the host thread count for the CPU baseline: time the oracle's GAT step at 1/4 scale for several counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import c_ops, ref_models as RM
from bot_amd import synth, nn as bnn
from bot_amd import workloads
ds = synth.make_dataset("arxiv", device="cpu", scale=0.25)
s, d = ds.graph.edges(); n = ds.graph.number_of_nodes(); C = ds.n_classes
g = c_ops.CGraph(s, d, n)
torch.manual_seed(0)
model = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, **workloads.ARXIV_GAT)
sd = {k: (v.clone().requires_grad_() if v.is_floating_point() and "running" not in k else v.clone()) for k, v in model.state_dict().items()}
params = [v for v in sd.values() if v.requires_grad]
x = RM.add_labels(ds.feat, ds.labels, ds.train_idx[:1000], C)
print("cpu_count", os.cpu_count())
for nt in (8, 16, 32, 64, 128, 256):
    if nt > os.cpu_count(): break
    torch.set_num_threads(nt); c_ops.set_num_threads(nt)
    ts = []
    for it in range(3):
        t0 = time.perf_counter()
        pred = RM.gat_forward(g, x, sd, n_layers=3, n_heads=3, n_hidden=250, n_classes=C, norm="batch", linear=True, training=True)
        loss = RM.compute_loss(pred[ds.train_idx], ds.labels[ds.train_idx], "loge")
        torch.autograd.grad(loss, params)
        ts.append(time.perf_counter() - t0)
    print(nt, "threads: best %.3f s" % min(ts), flush=True)
