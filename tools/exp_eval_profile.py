"""Kernel trace target: the inference-only forward of evaluate() at BASELINE config 2 (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from bot_amd import nn as bnn, synth, train as T, tuning
tuning.enable()
dev = "cuda"
ds = synth.make_dataset("arxiv", device="cpu", seed=0)
g = ds.graph.to(dev); g.create_formats_()
C = ds.n_classes
torch.manual_seed(0)
model = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, n_layers=3, n_heads=3, n_hidden=250,
                norm="batch", dropout=0.75, input_drop=0.25, attn_drop=0.1, linear=True).to(dev).eval()
feat, labels, tr = ds.feat.to(dev), ds.labels.to(dev), ds.train_idx.to(dev)
x = T.add_labels(feat, labels, tr, C)
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
        model(g, x)
torch.cuda.synchronize()
