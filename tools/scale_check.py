#!/usr/bin/env python3
"""Full-size checks of the other BASELINE configs on one MI355X: S-reddit GCN (config 3), S-proteins edge-feature GAT
(config 4), S-products GAT (config 5).  For each: build the graph on the GPU, size-independent properties of the kernels at
that scale (degrees == copy_u_sum(ones) bit-exact, attention rows sum to one, adjoint identity), then one timed
forward+backward step of the reference-shaped model and the SpMM rate.

    python tools/scale_check.py [reddit proteins products]
"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot_amd  # noqa: E402
from bot_amd import _C, ops, synth  # noqa: E402
from bot_amd import nn as bnn  # noqa: E402
from bot_amd.nn import edge_gat  # noqa: E402

DEV = "cuda"


def build(name):
    n, e_raw, f, c = synth.SHAPES[name]
    t0 = time.perf_counter()
    s, d = synth.powerlaw_edges(n, e_raw, synth.BASE_SEED, device=DEV)
    g = bot_amd.Graph(s, d, n)
    if name == "proteins":  # the proteins script trains on the raw directed graph with edge features (no preprocess)
        g = bot_amd.to_bidirected(g)
    else:
        g = bot_amd.preprocess(g)
    g.create_formats_()
    torch.cuda.synchronize()
    return g, f, c, time.perf_counter() - t0


def properties(g, H, D):
    n, E = g.number_of_nodes(), g.number_of_edges()
    deg = ops.copy_u_sum(g, torch.ones(n, 1, device=DEV)).squeeze(1)
    assert torch.equal(deg.long(), g.in_degrees()), "copy_u_sum(ones) != in_degrees"
    el, er = torch.randn(n, H, 1, device=DEV), torch.randn(n, H, 1, device=DEV)
    a = ops.gat_attention(g, el, er, order="csc")
    ones = torch.ones(n, H, D, device=DEV)
    agg = ops.u_mul_e_sum(g, ones, a, order="csc")
    has = (g.in_degrees() > 0).view(-1, 1, 1)
    assert torch.allclose(agg[has.expand_as(agg)], ones[has.expand_as(agg)], atol=2e-5), "attention rows do not sum to 1"
    x = torch.randn(n, H, D, device=DEV, requires_grad=True)
    y = torch.randn(n, H, D, device=DEV)
    out = ops.u_mul_e_sum(g, x, a, order="csc")
    (out * y).sum().backward()
    lhs, rhs = (out.detach().double() * y.double()).sum(), (x.detach().double() * x.grad.double()).sum()
    # scale by the norms, not by |lhs|: the inner product of random tensors can be arbitrarily close to zero
    assert abs(lhs - rhs) <= 1e-5 * float(out.detach().double().norm() * y.double().norm()), "adjoint identity violated"
    return {"N": n, "E": E, "max_in_deg": int(g.in_degrees().max()), "long_rows": g.csc.n_long, "chunk": g.csc.chunk}


def timed(fn, it=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


def main():
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["reddit", "proteins", "products"]
    # --gemm-tuning=file (default): shipped TunableOp selections; =tune: also time missing shapes and write them to
    # gpurun_out/tunableop_scale.csv (maintenance); =off: library heuristics
    mode = ([a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--gemm-tuning=")] or ["file"])[0]
    if mode != "off":
        from bot_amd import tuning
        tuning.enable(tune_missing=mode == "tune")
        if mode == "tune":
            from torch.cuda import tunable
            tunable.set_max_tuning_iterations(5)
            tunable.set_max_tuning_duration(10)
            os.makedirs("gpurun_out", exist_ok=True)
            tunable.set_filename(os.path.join("gpurun_out", "tunableop_scale.csv"))
    for name in which:
        g, f, c, t_build = build(name)
        n, E = g.number_of_nodes(), g.number_of_edges()
        if name == "reddit":
            H, D = 1, 256
            model = bnn.GCN(in_feats=f, n_classes=c, n_hidden=256, n_layers=3, activation=F.relu, norm="batch", dropout=0.5).to(DEV)
            feat = torch.randn(n, f, device=DEV)
            fwd = lambda: model(g, feat)
        elif name == "products":
            H, D = 4, 120
            model = edge_gat.ProductsGAT(node_feats=f, edge_feats=0, n_classes=c, n_layers=3, n_heads=4, n_hidden=120, edge_emb=0,
                                         activation=F.relu, dropout=0.5, input_drop=0.1, attn_drop=0.0, edge_drop=0.1).to(DEV)
            g.ndata["feat"] = torch.randn(n, f, device=DEV)
            fwd = lambda: model(g)
        else:
            H, D = 6, 80
            model = edge_gat.ProteinsGAT(node_feats=f, edge_feats=8, n_classes=c, n_layers=6, n_heads=6, n_hidden=80, edge_emb=16,
                                         activation=F.relu, dropout=0.25, input_drop=0.1, attn_drop=0.0, edge_drop=0.1,
                                         allow_zero_in_degree=True).to(DEV)
            g.edata["feat"] = torch.rand(E, 8, device=DEV)
            g.ndata["feat"] = ops.copy_e_sum(g, g.edata["feat"])  # ogbn-proteins/gat.py:58
            fwd = lambda: model(g)
        props = properties(g, H, D)
        model.train()

        def step():
            model.zero_grad(set_to_none=True)
            out = fwd()
            out.square().mean().backward()
            return out

        out = step()
        assert torch.isfinite(out).all()
        ms = timed(step)
        x = torch.randn(n, H, D, device=DEV)
        a = torch.rand(E, H, device=DEV)
        w = None if name == "reddit" else a
        from bot_amd import blocked
        blocked.ENABLED = False
        ms_spmm_row = timed(lambda: _C.spmm(g.csc, x, w, None), 5)
        blocked.ENABLED = True
        ms_spmm = timed(lambda: _C.spmm(g.csc, x, w, None), 5)
        alg = 4 * (2 * n * H * D + E + n + 1 + (E * H if w is not None else 0))
        print(json.dumps({"workload": f"S-{name}", **props, "graph_build_s": round(t_build, 2), "fwd_bwd_ms": round(ms, 2),
                          "edges_per_s": round(E / ms * 1e3), "spmm_ms": round(ms_spmm, 3), "spmm_row_kernel_ms": round(ms_spmm_row, 3),
                          "spmm_blocked": blocked.plan_for(g.csc, n, H, D) is not None,
                          "spmm_alg_GBs": round(alg / ms_spmm / 1e6, 1), "spmm_frac_of_8TBs": round(alg / ms_spmm / 1e6 / 8000, 4),
                          "spmm_gather_model_GBs": round(4 * (E * H * D + n * H * D + E) / ms_spmm / 1e6, 1),
                          "hbm_GB_allocated": round(torch.cuda.max_memory_allocated() / 1e9, 1)}), flush=True)
        del model, g, x, a
        torch.cuda.empty_cache()
    if mode == "tune":
        from torch.cuda import tunable
        getattr(tunable, "write_file", lambda f=None: None)(os.path.join("gpurun_out", "tunableop_scale.csv"))


if __name__ == "__main__":
    main()
