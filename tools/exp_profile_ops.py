"""Experiment: per-op device time of one config-2 train step (torch.profiler, grouped by op and input shape): what is left
outside the GEMMs and the HIP kernels."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from bot_amd import workloads
from bot_amd import synth, train, tuning
from bot_amd import nn as bnn
tuning.enable()
dev = torch.device("cuda", 0)
ds = synth.make_dataset("arxiv", device="cpu", seed=0)
n, C = ds.graph.number_of_nodes(), ds.n_classes
torch.manual_seed(0)
model = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, **workloads.ARXIV_GAT).to(dev)
opt = torch.optim.RMSprop(model.parameters(), lr=0.002)
g = ds.graph.to(dev); g.create_formats_()
feat, labels = ds.feat.to(dev), ds.labels.to(dev)
tr, va, te = ds.train_idx.to(dev), ds.val_idx.to(dev), ds.test_idx.to(dev)
step = lambda: train.train_step(model, g, feat, labels, tr, va, te, opt, use_labels=True, mask_rate=0.5, loss="loge", n_classes=C)
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
print("ops with device time: %d kinds, %d calls, %.1f us" % (len(rows), sum(e.count for e in rows), sum(e.self_device_time_total for e in rows)))
for e in rows[:60]:
    if e.key in ("aten::mm", "aten::addmm", "aten::addmm_"): continue
    print(f"{e.self_device_time_total:8.1f}us x{e.count:<3d} {e.key:34s} {str(e.input_shapes)[:80]}")
