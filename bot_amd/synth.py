"""Seeded synthetic inputs shaped like the reference's datasets (no datasets or network here).

Directed power-law graph: endpoints drawn as floor(N * u^gamma) (heavy-tailed, Zipf-like degrees),
node ids relabelled at random (no free locality), then the reference's `preprocess` recipe
(run.py:133-148).  Features N(0,1), labels uniform, arxiv-like 54/18/28 % split.  SURVEY §8d."""
from __future__ import annotations

from dataclasses import dataclass

import torch

from .graph import Graph, preprocess

SHAPES = {
    # name: (nodes, raw directed edges, features, classes)
    "cora": (2708, 10556, 1433, 7),
    "arxiv": (169343, 1166243, 128, 40),
    "reddit": (232965, 57_307_946, 602, 41),       # 114.6M directed edges once symmetrised
    "proteins": (132534, 39_561_252, 8, 112),
    "products": (2449029, 61_859_140, 100, 47),
}
BASE_SEED = 20210325


def powerlaw_edges(n, e_raw, seed, gamma=2.0, device="cpu"):
    gen = torch.Generator().manual_seed(seed)
    src = (n * torch.rand(e_raw, generator=gen, dtype=torch.float64) ** gamma).long().clamp_(max=n - 1)
    dst = (n * torch.rand(e_raw, generator=gen, dtype=torch.float64) ** gamma).long().clamp_(max=n - 1)
    perm = torch.randperm(n, generator=gen)
    return perm[src].to(device), perm[dst].to(device)


@dataclass
class Dataset:
    graph: Graph
    feat: torch.Tensor
    labels: torch.Tensor      # int64 [N,1]
    train_idx: torch.Tensor
    val_idx: torch.Tensor
    test_idx: torch.Tensor
    n_classes: int
    raw_edges: int


def make_dataset(name="arxiv", device="cuda", seed=0, scale=1.0) -> Dataset:
    n, e_raw, f, c = SHAPES[name]
    n, e_raw = max(8, int(n * scale)), max(8, int(e_raw * scale))
    s, d = powerlaw_edges(n, e_raw, BASE_SEED + seed, device=device)
    g = preprocess(Graph(s, d, n))
    gen = torch.Generator().manual_seed(BASE_SEED + 1000 + seed)
    feat = torch.randn(n, f, generator=gen).to(device)
    labels = torch.randint(0, c, (n, 1), generator=gen).to(device)
    perm = torch.randperm(n, generator=gen).to(device)
    a, b = int(0.54 * n), int(0.72 * n)
    return Dataset(g, feat, labels, perm[:a], perm[a:b], perm[b:], c, e_raw)


def degree_stats(g: Graph):
    deg = g.in_degrees().float()
    return {"max": int(deg.max()), "median": float(deg.median()), "mean": float(deg.mean())}
