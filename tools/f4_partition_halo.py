#!/usr/bin/env python3
"""halo rows / cut edges of the 1-D partition at W = 2 / 4 / 8 — contiguous ranges of the graph's own
(random) numbering vs ranges of the label-propagation community order (bot_amd.dist `partitioner="community"`) — on the
headline S-arxiv graph and on S-arxiv-comm (planted communities).  Pure integer work; runs on the CPU or the GPU."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = "cuda" if torch.cuda.is_available() else "cpu"
if dev == "cpu":  # the product has no CPU kernels; only the integer planning code runs here (tests' emulation for degrees)
    from tests import _oracle_backend
    _oracle_backend.install_direct()
from bot_amd import dist as bdist, synth  # noqa: E402
from bot_amd.graph import reorder_permutation  # noqa: E402

for name in ("arxiv", "arxiv-comm"):
    ds = synth.make_dataset(name, device=dev, seed=0)
    g = ds.graph
    s, d = g.edges()
    n = g.number_of_nodes()
    p2, labels = reorder_permutation(g, "community")
    inv = torch.empty_like(p2)
    inv[p2] = torch.arange(n, device=p2.device)
    structured = int(torch.bincount(labels).max()) * 4 <= n   # what partition_dataset(partitioner="community") checks
    for W in (2, 4, 8):
        a = bdist.halo_statistics(s, d, n, W)
        b = bdist.halo_statistics(inv[s], inv[d], n, W)
        print(json.dumps({"graph": name, "world": W, "communities_found": int(torch.unique(labels).numel()),
                          "largest_label_share": round(float(torch.bincount(labels).max()) / n, 3), "community_partitioner_applies": structured,
                          "contiguous": {"halo_rows_per_rank_max": max(a["halo_rows_per_rank"]), "halo_rows_total": sum(a["halo_rows_per_rank"]),
                                         "cut_edge_fraction": round(a["cut_edges"] / a["edges"], 4)},
                          "community": {"halo_rows_per_rank_max": max(b["halo_rows_per_rank"]), "halo_rows_total": sum(b["halo_rows_per_rank"]),
                                        "cut_edge_fraction": round(b["cut_edges"] / b["edges"], 4),
                                        "owned_rows_min_max": [min(b["owned_rows_per_rank"]), max(b["owned_rows_per_rank"])]}}))
