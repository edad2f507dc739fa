"""Experiment: what does the row-per-group SpMM reach when EVERY gather hits the XCD L2 (sources folded into a small window)?"""
import os, sys, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bot_amd
from bot_amd import _C, synth, blocked
blocked.ENABLED = False
n, e_raw, f, c = synth.SHAPES["reddit"]
s, d = synth.powerlaw_edges(n, e_raw, synth.BASE_SEED, device="cuda")
g = bot_amd.preprocess(bot_amd.Graph(s, d, n)); g.create_formats_()
x = torch.randn(n, 1, 256, device="cuda")
def t(fn, it=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
E = g.csc.nnz
print("all sources (238 MB table): %.2f ms  -> %.1f TB/s gathered" % ((ms := t(lambda: _C.spmm(g.csc, x, None, None))), E * 1024 / ms / 1e9))
for win in (512, 2048, 8192, 32768):
    dd = dataclasses.replace(g.csc, indices=(g.csc.indices % win).contiguous(), blocked={})
    ms = t(lambda: _C.spmm(dd, x, None, None))
    print("sources folded into %6d rows (%5.1f MB): %.2f ms -> %.1f TB/s gathered" % (win, win * 1024 / 1e6, ms, E * 1024 / ms / 1e9))
