"""The five BASELINE.json configurations as runnable train steps on synthetic inputs of their shape, on one
GPU or 1-D vertex-partitioned over the ranks of a process group.  Used by bench.py (`--workload`) and the tools; the recipes
are the reference's (model shape, drop rates, loss, optimizer), the data is seeded noise — there are no datasets here.

| workload | BASELINE config | model | recipe (reference) |
|---|---|---|---|
| cora     | 1 | GCN 2 x 16                          | src/no-sampling/run.py --model=gcn, Adam, logit loss |
| arxiv    | 2 | GAT 3 layers x 3 heads x 250        | run.py:1011-1013: --labels --loss=loge --linear, RMSprop lr 0.002, dropout .75/.25/.1 |
| reddit   | 3 | GCN 3 x 256                         | run.py --model=gcn --norm=batch, Adam |
| proteins | 4 | edge-feature GAT 6 x 6 heads x 80   | src/ogbn-proteins/gat.py:203-207,320-327: BCE-with-logits over 112 tasks, AdamW lr 0.01, edge drop 0.1 |
| products | 5 | GAT 3 x 4 heads x 120               | src/ogbn-products/gat.py:107-118,238,379-386: loge loss, AdamW lr 0.01, edge drop 0.1 |
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import ops, synth
from . import nn as bnn
from . import train as T
from .graph import Graph, preprocess, take_rows, to_bidirected
from .nn import edge_gat

NAMES = ("cora", "arxiv", "reddit", "proteins", "products")

ARXIV_GAT = dict(n_layers=3, n_heads=3, n_hidden=250, norm="batch", dropout=0.75, input_drop=0.25, attn_drop=0.1, edge_drop=0.0,
                 non_interactive_attn=False, use_symmetric_norm=False, linear=True, residual=False)


@dataclass
class Workload:
    name: str
    describe: str                 # goes into the bench line's config.workload
    n_nodes: int
    n_edges: int                  # edges of the graph the sparse kernels sweep (after the reference's preprocessing)
    raw_edges: int
    step: object                  # callable: one full train step (zero_grad, forward, loss, backward, optimizer step)
    model: object
    dominant: tuple               # (kernel family, shape key) of the SpMM the roofline object reports, see bot_amd._C._timed
    dominant_shape: tuple         # (H, D, weighted)
    n_local: int                  # destination rows / edges this rank sweeps (== n_nodes / n_edges on one GPU)
    e_local: int
    dataset: object = None
    graph: object = None


def _loge(x, labels):
    y = F.cross_entropy(x, labels[:, 0], reduction="none")
    return torch.log(T.EPSILON + y) - math.log(T.EPSILON)


def _bce(x, labels):
    return F.binary_cross_entropy_with_logits(x, labels.to(x.dtype), reduction="none").mean(1)


def _edge_dataset(name, device, seed, scale):
    """Inputs of the two edge-GAT scripts.  proteins: the raw graph made bidirected, 8 edge features U(0,1), node features =
    sum of incident edge features (ogbn-proteins/gat.py:58), 112 binary tasks; products: `preprocess`, 100 node features."""
    n, e_raw, f, c = synth.SHAPES[name]
    n, e_raw = max(8, int(n * scale)), max(8, int(e_raw * scale))
    s, d = synth.powerlaw_edges(n, e_raw, synth.BASE_SEED + seed, device=device)
    g = Graph(s, d, n)
    g = to_bidirected(g) if name == "proteins" else preprocess(g)
    g.create_formats_()
    gen = torch.Generator().manual_seed(synth.BASE_SEED + 1000 + seed)
    E = g.number_of_edges()
    if name == "proteins":
        efeat = torch.rand(E, 8, generator=gen).to(device)
        labels = (torch.rand(n, c, generator=gen) < 0.5).to(torch.int64).to(device)
        feat = None
    else:
        efeat = None
        feat = torch.randn(n, f, generator=gen).to(device)
        labels = torch.randint(0, c, (n, 1), generator=gen).to(device)
    perm = torch.randperm(n, generator=gen).to(device)
    a, b = int(0.54 * n), int(0.72 * n)
    ds = synth.Dataset(g, feat, labels, perm[:a], perm[a:b], perm[b:], c, e_raw)
    ds.efeat = efeat
    return ds


MODEL_DIMS = {       # name: (layers, heads, head width, edge-feature GAT)
    "cora": (2, 1, 16, False), "arxiv": (3, 3, 250, False), "reddit": (3, 1, 256, False), "proteins": (6, 6, 80, True), "products": (3, 4, 120, True)}


def hbm_budget(name: str, world: int = 1, scale: float = 1.0) -> dict:
    """A COARSE per-rank estimate (bytes) of what `build` + one step allocate, printed by bench.py before anything is allocated so that a
    first multi-GPU run that cannot fit says so instead of dying in the allocator.  Every rank builds the whole seeded
    dataset on its own device and cuts its block out of it (`build`), so the build term does not shrink with the world size; the step
    term does.  Calibrated on the measured single-GPU peaks (tools/hbm_peak.py, profiles/r05_hbm_peak.txt: 0.18 / 7.1 / 23.8 / 39.4 /
    75.2 GiB for configs 1-5; the estimate lands within +25 % of each; bench.py treats it as a gate that ALL ranks take together and that
    `--ignore-hbm-budget` overrides):
      build  80 B per edge (COO pairs, CSC, CSR, permutations, plans) + node features (+ 32 B per edge of edge features, config 4)
      step   GAT family: 9 B per edge, head and layer (attention weights, their dropped copy / signs) + 7.5 fp32 [rows, H D] tensors per
             wide layer (projection output, gradient operand, pre-BatchNorm state, halves, temporaries); GCN: 150 B per edge (the
             L2-blocked sweep's streams) + the same node term; x 1.3 (the bench process also runs parity / baseline legs: its peaks were 31.6 / 47.2 / 80.3 GiB).  Rows = owned + halo, the halo bounded by all other ranks' rows."""
    n, e_raw, f, c = synth.SHAPES[name]
    n, e_raw = max(8, int(n * scale)), max(8, int(e_raw * scale))
    E = 2 * e_raw + n                                            # symmetrised + self-loops (an upper bound: duplicates merge)
    layers, H, D, edge = MODEL_DIMS[name]
    gcn = name in ("cora", "reddit")
    n_own = -(-n // world)
    n_ext = n if world > 1 else n_own                            # a power-law graph's halo is most of the other ranks' rows
    e_loc = -(-E // world)
    build = E * 80 + n * max(f, 8) * 4 + (E * 32 if name == "proteins" else 0)
    if world > 1:
        build += e_loc * 80 + n_ext * max(f, 8) * 4              # the rank's block beside the whole graph it was cut from
    edge_part = e_loc * 150 if gcn else layers * e_loc * H * 9
    # (the modular edge-GAT stacks keep fewer [rows, H D] temporaries alive than the fused GAT layers: 4.5 against 7.5 tensors per wide layer,
    # recalibrated on profiles/r05_hbm_peak.txt after ADVICE r5: S-products 75.2 GiB measured, 112.5 estimated before, 78 now)
    node_part = max(1, layers - 1) * n_ext * H * D * 4 * (4.5 if edge else 7.5)
    step = int(1.3 * (edge_part + node_part))
    return {"whole_graph_build": int(build), "step": step, "total": int(build) + step + (1 << 28)}      # + 0.25 GiB of fixed costs


def build(name: str, device, *, rank=0, world=1, partitioned=False, seed=0, scale=1.0, norm_adj="rw", partitioner="contiguous",
          group=None, drop=True, capture=False) -> Workload:
    """Dataset + model + optimizer + step of one configuration.  Every rank builds the same (seeded) whole dataset on its own
    device and cuts its block out of it there (bot_amd.dist.build_partition).  `drop=False` zeroes every drop rate (parity
    and CPU tests; the benchmark keeps the reference's rates).  `capture=True`: the step is captured into a hipGraph after 3 eager
    warm-up steps and replayed (bot_amd.train.CapturedTrainStep; the GCN / GAT stacks of configs 1-3, whose step draws no
    host-side seeds per step — the edge-drop mask of configs 4 / 5 does, those stay eager)."""
    k = 1.0 if drop else 0.0
    if name not in NAMES:
        raise ValueError(f"unknown workload {name!r}: {NAMES}")
    dev = torch.device(device)
    if partitioned:
        from . import dist as bdist
    edge = name in ("proteins", "products")
    ds = _edge_dataset(name, dev, seed, scale) if edge else synth.make_dataset(name, device=dev, seed=seed, scale=scale)
    g = ds.graph
    n, E, C = g.number_of_nodes(), g.number_of_edges(), ds.n_classes
    torch.manual_seed(seed)
    if name == "arxiv":
        cfg = dict(ARXIV_GAT, use_symmetric_norm=norm_adj == "symm")
        cfg.update(dropout=0.75 * k, input_drop=0.25 * k, attn_drop=0.1 * k)
        model = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, **cfg).to(dev)
        from . import optim as boptim
        opt = boptim.RMSprop(model.parameters(), lr=0.002, capturable=capture)      # torch.optim.RMSprop's update in one launch
        kw = dict(use_labels=True, mask_rate=0.5, loss="loge", n_classes=C)
        shape, desc = (3, 250, True), (f"GAT 3 layers x 3 heads x 250, --labels --loss=loge --linear --norm=batch"
                                       f"{' --norm-adj=symm' if norm_adj == 'symm' else ''}, dropout 0.75/0.25/0.1, RMSprop step included")
    elif name in ("cora", "reddit"):
        hid, layers = (16, 2) if name == "cora" else (256, 3)
        model = bnn.GCN(in_feats=ds.feat.shape[1], n_classes=C, n_hidden=hid, n_layers=layers, activation=F.relu,
                        norm="none" if name == "cora" else "batch", norm_adj="symm", dropout=0.5 * k).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=0.01, capturable=capture, fused=dev.type == "cuda")
        kw = dict(use_labels=False, mask_rate=0.5, loss="logit", n_classes=C)   # run.py:265-267: the mask split also without --labels
        shape, desc = (1, hid, False), f"GCN {layers} layers x {hid}, norm_adj=symm, dropout 0.5, logit loss, Adam step included"
    elif name == "proteins":
        model = edge_gat.ProteinsGAT(node_feats=8, edge_feats=8, n_classes=C, n_layers=6, n_heads=6, n_hidden=80, edge_emb=16,
                                     activation=F.relu, dropout=0.25 * k, input_drop=0.1 * k, attn_drop=0.0, edge_drop=0.1 * k,
                                     allow_zero_in_degree=True).to(dev)
        opt = torch.optim.AdamW(model.parameters(), lr=0.01, weight_decay=0, fused=dev.type == "cuda")
        shape, desc = (6, 80, True), ("edge-feature GAT 6 layers x 6 heads x 80, 8-d edge features, edge_emb 16, dropout 0.25/0.1, "
                                      "edge drop 0.1, BCE-with-logits over 112 tasks, AdamW step included")
    else:
        model = edge_gat.ProductsGAT(node_feats=ds.feat.shape[1], edge_feats=0, n_classes=C, n_layers=3, n_heads=4, n_hidden=120,
                                     edge_emb=0, activation=F.relu, dropout=0.5 * k, input_drop=0.1 * k, attn_drop=0.0, edge_drop=0.1 * k).to(dev)
        opt = torch.optim.AdamW(model.parameters(), lr=0.01, weight_decay=0, fused=dev.type == "cuda")
        shape, desc = (4, 120, True), ("GAT 3 layers x 4 heads x 120, dropout 0.5/0.1, edge drop 0.1, loge loss, AdamW step included")

    if name == "proteins":  # node features = sum of the incident edge features over the WHOLE graph (ogbn-proteins/gat.py:58)
        with torch.no_grad():
            ds.feat = ops.copy_e_sum(g, ds.efeat)
    n_local, e_local = n, E
    if not partitioned:
        if edge:
            g.ndata["feat"] = ds.feat
            if ds.efeat is not None:
                g.edata["feat"] = ds.efeat
            node_loss = _bce if name == "proteins" else _loge
            tr = ds.train_idx

            def step():
                model.train()
                opt.zero_grad()
                pred = model(g)
                loss = node_loss(pred[tr], ds.labels[tr]).mean()
                loss.backward()
                opt.step()
                return loss, pred
        else:
            def step():
                return T.train_step(model, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, opt, **kw)
    else:
        part = bdist.partition_dataset(ds, rank, world, dev, group, partitioner=partitioner)
        model = bdist.wrap_model(model, group)
        bdist.seed_rank_streams(seed, rank)
        n_local, e_local = part.n_owned, part.n_edges
        if edge:
            pg = part.graph
            pg.ndata["feat"] = part.feat
            if ds.efeat is not None:
                pg.edata["feat"] = take_rows(ds.efeat, part.edge_ids)
            node_loss = _bce if name == "proteins" else _loge

            def step():
                return bdist.step_generic(lambda: model(pg), part, node_loss, model=model, optimizer=opt, group=group)
        else:
            dkw = {k: v for k, v in kw.items()}

            def step():
                return bdist.train_step(model, part, opt, group=group, **dkw)
        ds.part = part
    captured = False
    if capture and not edge:
        eager, captured = step, True
        step = T.CapturedTrainStep(lambda: eager()[:2], dev)
    H, D, weighted = shape
    describe = (f"S-{name}: power-law graph N={n} E={E} (raw {ds.raw_edges}), F={0 if ds.feat is None else ds.feat.shape[1]}, C={C}; {desc}")
    wl = Workload(name, describe, n, E, ds.raw_edges, step, model, ("spmm", shape), shape, n_local, e_local, ds, g)
    wl.captured = captured
    wl.optimizer = opt
    return wl
