"""Experiment: L2-blocked SpMM vs row-per-group kernel on S-reddit (F=256, unweighted)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bot_amd
from bot_amd import _C, synth, blocked
n, e_raw, f, c = synth.SHAPES["reddit"]
s, d = synth.powerlaw_edges(n, e_raw, synth.BASE_SEED, device="cuda")
g = bot_amd.preprocess(bot_amd.Graph(s, d, n)); g.create_formats_()
W = 256
for arg in sys.argv[1:]:
    k, v = arg.split("=")
    if k == "W": W = int(v)
    else: setattr(blocked, k, int(v))
x = torch.randn(n, 1, W, device="cuda")
def t(fn, it=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
blocked.ENABLED = False
print("row kernel  %.2f ms" % t(lambda: _C.spmm(g.csc, x, None, None)))
blocked.ENABLED = True
bp = blocked.plan_for(g.csc, n, 1, W)
print("W", W, "epi", bp.epi, "slots", bp.b_src.numel(), "tiles", bp.n_tiles, "nblk", bp.nblk, "T", bp.T, "round", bp.round_tiles, "blocked edges", bp.b_src.numel(), "hub rows", 0 if bp.heavy is None else bp.heavy.n_long, "hub edges", g.csc.nnz - bp.b_src.numel())
print("blocked+hub %.2f ms" % t(lambda: _C.spmm(g.csc, x, None, None)))
out = torch.empty(n, 1, W, device="cuda")
print("blocked only %.2f ms" % t(lambda: _C.spmm_blocked(bp, x, None, out)))
if bp.heavy is not None:
    print("hub only %.2f ms" % t(lambda: _C.spmm(bp.heavy, x, None, None, out=out)))
