#!/usr/bin/env python3
"""Kernel micro-benchmarks on the S-arxiv graph (HIP events on the launch stream); also the target of the
rocprofv3 --pmc passes whose summaries live under profiles/.

    python tools/microbench.py [--only spmm,sddmm,...] [--iters 10] [--workload arxiv]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd import _C, synth  # noqa: E402


def timeit(fn, iters, warmup=2):
    for _ in range(warmup):
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
    return ts[len(ts) // 2], sum(ts) / len(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--workload", default="arxiv")
    ap.add_argument("--chunk", type=int, default=0)
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    dev = torch.device("cuda:0")
    ds = synth.make_dataset(args.workload, device="cpu")
    g = ds.graph
    if args.chunk:
        g._chunk = args.chunk
    g = g.to(dev)
    g.create_formats_()
    n, E = g.number_of_nodes(), g.number_of_edges()
    print(json.dumps({"graph": args.workload, "N": n, "E": E, "chunk": g.csc.chunk, "long_rows": g.csc.n_long,
                      "items": g.csc.n_items, "deg": synth.degree_stats(g)}))
    res = []

    def report(name, ms, alg_bytes, gather_bytes):
        r = {"kernel": name, "device_kernel": _C._lib.bot_last_kernel().decode(), "ms": round(ms, 4), "alg_bytes": int(alg_bytes),
             "alg_bytes_per_edge": round(alg_bytes / E, 1), "alg_GBs": round(alg_bytes / ms / 1e6, 1),
             "gather_model_GBs": round(gather_bytes / ms / 1e6, 1), "frac_of_8TBs_alg": round(alg_bytes / ms / 1e6 / 8000, 4)}
        res.append(r)
        print(json.dumps(r), flush=True)

    for (H, D) in ((3, 250), (1, 128), (1, 256), (1, 40)):
        F = H * D
        x = torch.randn(n, H, D, device=dev)
        y = torch.randn(n, H, D, device=dev)
        a = torch.rand(E, H, device=dev)
        out = torch.empty(n, H, D, device=dev)
        alg_w = 4 * (2 * n * F + E + n + 1 + E * H)
        alg_0 = 4 * (2 * n * F + E + n + 1)
        gat = 4 * (E * F + n * F + E + n + 1)
        if not only or "spmm" in only:
            ms, _ = timeit(lambda: _C.spmm(g.csc, x, a, None, out=out), args.iters)
            report(f"spmm u_mul_e_sum fwd (CSC) H={H} D={D}", ms, alg_w, gat + 4 * E * H)
        if not only or "spmm_t" in only:
            ms, _ = timeit(lambda: _C.spmm(g.csr, x, a, g.csr2csc, out=out), args.iters)
            report(f"spmm u_mul_e_sum bwd (CSR, wperm) H={H} D={D}", ms, alg_w, gat + 4 * E * H)
        if not only or "copy" in only:
            ms, _ = timeit(lambda: _C.spmm(g.csc, x, None, None, out=out), args.iters)
            report(f"spmm copy_u_sum H={H} D={D}", ms, alg_0, gat)
        if (not only or "fused" in only) and D <= _C.spmm_dot_max_d(x):
            ms, _ = timeit(lambda: _C.spmm_dot(g.csr, x, a, g.csr2csc, y), args.iters)
            report(f"spmm_dot fused bwd (CSR) H={H} D={D}", ms, 4 * (3 * n * F + E + n + 1 + 2 * E * H), gat + 4 * (n * F + 2 * E * H))
        if not only or "sddmm" in only:
            ms, _ = timeit(lambda: _C.sddmm_dot(g.csc, x, y), args.iters)
            report(f"sddmm_dot H={H} D={D}", ms, 4 * (2 * n * F + E + E * H), 4 * (E * F + n * F + E + E * H))
    if not only or "bcast" in only:      # the aggregate-first input layer of config 2: 168-float source rows broadcast over 3 heads
        H, Fin = 3, 168
        x = torch.randn(n, Fin, device=dev)
        a = torch.rand(E, H, device=dev)
        dz = torch.randn(H, n, Fin, device=dev)
        ms, _ = timeit(lambda: _C.spmm_bcast(g.csc, x, a, None, head_outer=True), args.iters)
        report(f"spmm_bcast fwd H={H} Fin={Fin}", ms, 4 * (n * Fin + H * n * Fin + E + n + 1 + E * H), 4 * (E * Fin + H * n * Fin + E + E * H))
        ms, _ = timeit(lambda: _C.sddmm_dot_bcast(g.csc, x, dz), args.iters)
        report(f"sddmm_dot_bcast H={H} Fin={Fin}", ms, 4 * (n * Fin + H * n * Fin + E + n + 1 + E * H), 4 * (E * Fin + H * n * Fin + E + E * H))
    if not only or "halves" in only:     # ABI 17: the fused backward sweep with its first result written as a halves operand
        H, D = 3, 250
        x, y = torch.randn(n, H, D, device=dev), torch.randn(n, H, D, device=dev)
        a = torch.rand(E, H, device=dev)
        piece = 1536
        buf = torch.empty(n, 2 * piece, dtype=torch.float16, device=dev)
        scale = _C.halves_scale(x.reshape(n, -1) * 64)
        ms, _ = timeit(lambda: _C.spmm_dot_halves(g.csr, x, a, g.csr2csc, y, scale, buf, D, piece), args.iters)
        report(f"spmm_dot_halves fused bwd (CSR) H={H} D={D}", ms, 4 * (3 * n * H * D + E + n + 1 + 2 * E * H), 4 * (E * H * D + 2 * n * H * D + E + n + 1 + 2 * E * H))
    if not only or "attn" in only:
        for H in (3, 1):
            el, er = torch.randn(n, H, device=dev), torch.randn(n, H, device=dev)
            ms, _ = timeit(lambda: _C.gat_attn_fwd(g.csc, el, er, None, None, None, 0.2, H, None), args.iters)
            report(f"gat_attn_fwd H={H}", ms, 4 * (E * H + E + n + 1 + 2 * n * H), 4 * (2 * E * H + E + n + 1 + n * H))
            a = _C.gat_attn_fwd(g.csc, el, er, None, None, None, 0.2, H, None)
            da = torch.randn_like(a)
            ms, _ = timeit(lambda: _C.gat_attn_bwd(g.csc, el, er, None, None, 0.2, H, a, da, None, None, True), args.iters)
            report(f"gat_attn_bwd H={H}", ms, 4 * (3 * E * H + E + n + 1), 4 * (4 * E * H + E + n + 1))
            dz = torch.randn(E, H, device=dev)
            ms, _ = timeit(lambda: _C.segment_sum(g.csr, dz, g.csr2csc), args.iters)
            report(f"segment_sum (CSR, perm) W={H}", ms, 4 * (E * H + E + n * H + n + 1), 4 * (E * H + E + n * H + n + 1))


if __name__ == "__main__":
    main()
