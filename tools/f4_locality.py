#!/usr/bin/env python3
"""SURVEY §8 f4 measurements for the headline SpMM (u_mul_e_sum forward, H=3 D=250):

  python tools/f4_locality.py time     per graph {S-arxiv, S-arxiv-comm} x numbering {as generated, degree, community(+XCD order)}:
                                       kernel time, algorithmic GB/s and fraction of the 8 TB/s peak, gathered TB/s
  python tools/f4_locality.py ceiling  the gather-ceiling experiment on S-arxiv: the same kernel with every source id folded into
                                       a window of W rows (W * 3 000 B resident in L2 / Infinity Cache / neither)
  python tools/f4_locality.py pmc GRAPH NUMBERING    a few launches of the SpMM only, for the rocprofv3 --pmc passes
"""
import dataclasses
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bot_amd  # noqa: E402
from bot_amd import _C, synth  # noqa: E402

H, D = 3, 250
DEV = "cuda"


def build(name, numbering):
    ds = synth.make_dataset(name, device="cpu", seed=0, reorder=None if numbering == "none" else numbering)
    g = ds.graph.to(DEV)
    g.create_formats_()
    return g


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def operands(g):
    n, E = g.number_of_nodes(), g.number_of_edges()
    # F4_PITCH (floats): row pitch of x and out; 752 gives 16-byte aligned rows of 750 floats, the layout of the merged GEMM output,
    # on which the flat 16-byte-lane kernel runs (BOT_SPMM_FLAT=0: the 8-byte head-segment kernel on the same operands)
    pitch = int(os.environ.get("F4_PITCH", H * D))
    x = torch.randn(n, pitch, device=DEV)[:, :H * D].unflatten(1, (H, D))
    a = torch.rand(E, H, device=DEV)
    out = torch.empty(n, pitch, device=DEV)[:, :H * D].unflatten(1, (H, D))
    return x, a, out


def main():
    mode = sys.argv[1]
    if mode == "time":
        for name in ("arxiv", "arxiv-comm"):
            for numbering in ("none", "degree", "community"):
                t0 = time.perf_counter()
                g = build(name, numbering)
                prep = time.perf_counter() - t0
                n, E = g.number_of_nodes(), g.number_of_edges()
                x, a, out = operands(g)
                ms = timeit(lambda: _C.spmm(g.csc, x, a, None, out=out))
                y, _, dout = operands(g)          # same pitch as x: the fused backward takes the flat layout when all three slabs allow it
                ms_b = timeit(lambda: _C.spmm_dot(g.csr, x, a, g.csr2csc, y, out=dout))
                el = torch.randn(n, H, device=DEV)
                ms_i = timeit(lambda: _C.gat_infer(g.csc, x, el, out=out))
                alg = 4 * (2 * n * H * D + E + n + 1 + E * H)
                print(json.dumps({"graph": name, "numbering": numbering, "plan_order": g.plan_order, "N": n, "E": E,
                                  "kernel": "spmm fwd H=3 D=250", "ms": round(ms, 4), "alg_GBs": round(alg / ms / 1e6, 1),
                                  "frac_of_8TBs": round(alg / ms / 1e6 / 8000, 4), "gathered_TBs": round(E * H * D * 4 / ms / 1e9, 2),
                                  "spmm_dot_bwd_ms": round(ms_b, 4), "gat_infer_ms": round(ms_i, 4), "build_s": round(prep, 2)}), flush=True)
                del g, x, a, out
    elif mode == "ceiling":
        g = build("arxiv", "none")
        n, E = g.number_of_nodes(), g.number_of_edges()
        x, a, out = operands(g)
        ms = timeit(lambda: _C.spmm(g.csc, x, a, None, out=out))
        print("S-arxiv H=3 D=250, all %d sources (%.0f MB table): %.3f ms -> %.2f TB/s gathered" % (n, n * 3000 / 1e6, ms, E * 3000 / ms / 1e9))
        for win in (256, 1024, 4096, 16384, 65536):
            dd = dataclasses.replace(g.csc, indices=(g.csc.indices % win).contiguous(), blocked={})
            ms = timeit(lambda: _C.spmm(dd, x, a, None, out=out))
            print("sources folded into %6d rows (%6.1f MB): %.3f ms -> %.2f TB/s gathered, alg frac of 8 TB/s %.3f" % (
                win, win * 3000 / 1e6, ms, E * 3000 / ms / 1e9, 4 * (2 * n * H * D + E + n + 1 + E * H) / ms / 1e6 / 8000))
    elif mode == "step":   # the whole config-2 train step on the community graph, as generated vs renumbered by preprocess(reorder=...)
        import torch.nn.functional as F
        from bot_amd import nn as bnn, train as T, tuning, workloads
        tuning.enable()
        for name in ("arxiv", "arxiv-comm"):
            for numbering in ("none", "community"):
                ds = synth.make_dataset(name, device=DEV, seed=0, reorder=None if numbering == "none" else numbering)
                g, C = ds.graph, ds.n_classes
                torch.manual_seed(0)
                model = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, **workloads.ARXIV_GAT).to(DEV)
                opt = torch.optim.RMSprop(model.parameters(), lr=0.002)
                step = lambda: T.train_step(model, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, opt, use_labels=True,
                                            mask_rate=0.5, loss="loge", n_classes=C)
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    step()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 20 * 1e3
                print(json.dumps({"graph": name, "numbering": numbering, "plan_order": g.plan_order, "train_step_ms": round(ms, 3),
                                  "edges_per_s": round(g.number_of_edges() / ms * 1e3)}), flush=True)
    elif mode == "pmc":
        g = build(sys.argv[2], sys.argv[3])
        x, a, out = operands(g)
        for _ in range(5):
            _C.spmm(g.csc, x, a, None, out=out)
        torch.cuda.synchronize()
        print("kernel", _C._lib.bot_last_kernel().decode())


if __name__ == "__main__":
    main()
