import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
import bench
from bot_amd import synth, train, tuning
from bot_amd import nn as bnn
tuning.enable()
dev = torch.device("cuda", 0)
ds = synth.make_dataset("arxiv", device="cpu", seed=0)
n, C = ds.graph.number_of_nodes(), ds.n_classes
torch.manual_seed(0)
model = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, **bench.CFG).to(dev)
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
evs = [e for e in prof.events() if e.device_time_total > 40 and e.name.startswith("aten::") and not e.name.startswith("aten::mm") and not e.name.startswith("aten::addmm")]
evs.sort(key=lambda e: -e.device_time_total)
for e in evs[:25]:
    st = [s for s in (e.stack or []) if "repo" in s][:3]
    print(f"{e.device_time_total:8.1f}us {e.name:28s} {str(e.input_shapes)[:70]:70s} {' | '.join(s.split('/')[-1] for s in st)}")
