"""How long does the host take to ENQUEUE one train step (no device sync)?  Bounds the step time when per-rank GPU work shrinks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from bot_amd import workloads
from bot_amd import synth, train, tuning, nn as bnn
tuning.enable()
dev = torch.device("cuda:0")
for scale in (1.0, 0.125):
    ds = synth.make_dataset("arxiv", device="cpu", scale=scale)
    C = ds.n_classes
    torch.manual_seed(0)
    model = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, **workloads.ARXIV_GAT).to(dev)
    opt = torch.optim.RMSprop(model.parameters(), lr=0.002)
    g = ds.graph.to(dev); g.create_formats_()
    feat, labels = ds.feat.to(dev), ds.labels.to(dev)
    tr, va, te = ds.train_idx.to(dev), ds.val_idx.to(dev), ds.test_idx.to(dev)
    step = lambda: train.train_step(model, g, feat, labels, tr, va, te, opt, use_labels=True, mask_rate=0.5, loss="loge", n_classes=C)
    for _ in range(5): step()
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        step(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    print(f"scale {scale}: N={g.number_of_nodes()} host enqueue {1e3*sorted(enq)[10]:.2f} ms, step {1e3*sorted(tot)[10]:.2f} ms")
