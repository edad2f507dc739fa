#!/usr/bin/env python3
"""Regenerate bot_amd/tuning/tunableop_gfx950.csv: time the hipBLASLt/rocBLAS candidates (PyTorch TunableOp) for every fp32
GEMM shape of the config-2 train step — single GPU (N rows) and the per-rank row counts of the node-balanced 2/4/8-way
partitions — and write the winners.  Run on an MI355X:  python tools/tune_gemm.py gpurun_out/tunableop_gfx950.csv
"""
import os
import sys

import torch
from torch.cuda import tunable

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd import tuning  # noqa: E402

out = sys.argv[1]
tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename(out)
if os.path.exists(tuning.FILE):
    tunable.read_file(tuning.FILE)
tunable.set_max_tuning_iterations(30)
tunable.set_max_tuning_duration(15)
N = 169343
rows = [N]
for w in (2, 4, 8):
    rows += sorted({(N * (k + 1) + w - 1) // w - (N * k + w - 1) // w for k in range(w)}, reverse=True)
layers = [(168, 1536), (750, 1536), (750, 128)]  # (Fin, padded merged width) of the three GAT layers
dev = "cuda"
from bot_amd.nn import fused  # noqa: E402
KP = fused.WEIGHT_KP  # merged weights handed to the layer node as [K, P] (True) or [P, K]
for n in rows:
    for K, P in layers:
        h = torch.randn(n, K, device=dev)
        W = torch.randn(K, P, device=dev) if KP else torch.randn(P, K, device=dev)
        d = torch.randn(n, P, device=dev)
        if KP:
            torch.mm(h, W); torch.mm(h.t(), d); torch.mm(d, W.t())      # forward, dW, dh
        else:
            torch.mm(h, W.t()); torch.mm(d.t(), h); torch.mm(d, W)
        torch.cuda.synchronize()
        print("tuned", n, K, P, flush=True)
    # aggregate-before-project layer 0 (bot_amd/nn/fused.py _GATHiddenAggFirst): Fin = 168, H = 3, D = 250, P2 = 768
    Fin, H, D, P2 = 168, 3, 250, 768
    h = torch.randn(n, Fin, device=dev)
    Wr = torch.randn(Fin, P2, device=dev) if KP else torch.randn(P2, Fin, device=dev)
    W = torch.randn(H, D, Fin, device=dev)
    z = torch.randn(H, n, Fin, device=dev)
    dout2 = torch.randn(n, P2, device=dev)
    agg = torch.empty(H, n, D, device=dev)
    dz = torch.empty(H, n, Fin, device=dev)
    dW3 = torch.empty(H, D, Fin, device=dev)
    if KP:
        torch.mm(h, Wr); torch.mm(h.t(), dout2); torch.mm(dout2, Wr.t())
    else:
        torch.mm(h, Wr.t()); torch.mm(dout2.t(), h); torch.mm(dout2, Wr)
    for i in range(1):  # the three heads share one shape / leading dimensions
        torch.mm(z[i], W[i].t(), out=agg[i])
        res = torch.randn(n, P2, device=dev)
        res[:, i * D:(i + 1) * D].addmm_(z[i], W[i].t())               # the in-place form on the residual columns (ldc = P2)
        dxi = dout2[:, i * D:(i + 1) * D]
        torch.mm(dxi, W[i], out=dz[i])
        torch.mm(dxi.t(), z[i], out=dW3[i])
    torch.cuda.synchronize()
    print("tuned agg-first", n, flush=True)
getattr(tunable, "write_file", lambda f: None)(out)  # older TunableOp: the file is written at exit
print("results:", len(tunable.get_results()))
