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


def community_edges(n, e_raw, seed, gamma=2.0, n_blocks=128, p_in=0.9, device="cpu"):
    """Power-law graph WITH planted communities (a second generator on which locality work is measurable).
    Before the random relabel, vertex i belongs to block i % n_blocks, so every block holds hubs and leaves alike; the source
    of an edge is drawn like `powerlaw_edges` (floor(N u^gamma)); with probability `p_in` the destination is drawn by the same
    law among the members of the source's block, otherwise over the whole graph.  Then ids are relabelled at random exactly as
    in `powerlaw_edges`: the structure is there, but nothing in the numbering shows it."""
    gen = torch.Generator().manual_seed(seed)
    src = (n * torch.rand(e_raw, generator=gen, dtype=torch.float64) ** gamma).long().clamp_(max=n - 1)
    glob = (n * torch.rand(e_raw, generator=gen, dtype=torch.float64) ** gamma).long().clamp_(max=n - 1)
    per = (n + n_blocks - 1) // n_blocks
    rank = (per * torch.rand(e_raw, generator=gen, dtype=torch.float64) ** gamma).long().clamp_(max=per - 1)
    local = (rank * n_blocks + src % n_blocks).clamp_(max=n - 1)
    inside = torch.rand(e_raw, generator=gen) < p_in
    dst = torch.where(inside, local, glob)
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


def make_dataset(name="arxiv", device="cuda", seed=0, scale=1.0, reorder=None) -> Dataset:
    """`name`: a key of SHAPES, or "<key>-comm" for the same shape with planted communities (`community_edges`; 128 blocks at
    S-arxiv size: ~1 300 vertices = ~4 MB of 3 x 250 fp32 rows each, p_in = 0.9).  `reorder`: passed to `preprocess`."""
    comm = name.endswith("-comm")
    n, e_raw, f, c = SHAPES[name[:-5] if comm else name]
    n, e_raw = max(8, int(n * scale)), max(8, int(e_raw * scale))
    if comm:
        s, d = community_edges(n, e_raw, BASE_SEED + seed, n_blocks=max(2, n // 1323), device=device)
    else:
        s, d = powerlaw_edges(n, e_raw, BASE_SEED + seed, device=device)
    g = preprocess(Graph(s, d, n), reorder=reorder)
    gen = torch.Generator().manual_seed(BASE_SEED + 1000 + seed)
    feat = torch.randn(n, f, generator=gen).to(device)
    labels = torch.randint(0, c, (n, 1), generator=gen).to(device)
    perm = torch.randperm(n, generator=gen).to(device)
    a, b = int(0.54 * n), int(0.72 * n)
    return Dataset(g, feat, labels, perm[:a], perm[a:b], perm[b:], c, e_raw)


def degree_stats(g: Graph):
    deg = g.in_degrees().float()
    return {"max": int(deg.max()), "median": float(deg.median()), "mean": float(deg.mean())}
