"""Experiment: fused backward spmm_dot at S-proteins (H=6, D=80): L2-blocked form vs the all-heads row kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bot_amd
from bot_amd import _C, synth, blocked
name = sys.argv[1] if len(sys.argv) > 1 else "proteins"
H, D = (6, 80) if name == "proteins" else (4, 120)
n, e_raw, f, c = synth.SHAPES[name]
s, d = synth.powerlaw_edges(n, e_raw, synth.BASE_SEED, device="cuda")
g = bot_amd.to_bidirected(bot_amd.Graph(s, d, n)) if name == "proteins" else bot_amd.preprocess(bot_amd.Graph(s, d, n))
g.create_formats_()
E = g.number_of_edges()
x = torch.randn(n, H, D, device="cuda"); y = torch.randn(n, H, D, device="cuda"); w = torch.rand(E, H, device="cuda")
def t(fn, it=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
blocked.ENABLED = False
ms = t(lambda: _C.spmm_dot(g.csr, x, w, g.csr2csc, y))
print("spmm_dot row kernel %.2f ms   (E=%d, gathered %.1f TB/s)" % (ms, E, E * H * D * 4 / ms / 1e9))
blocked.ENABLED = blocked.DOT_ENABLED = True
for mb in (1, 2, 4, 8, 16):
    blocked.DOT_L2_BLOCK_BYTES = mb << 20
    g.csr.blocked.clear()
    ms = t(lambda: _C.spmm_dot(g.csr, x, w, g.csr2csc, y))
    print("spmm_dot blocked, %2d MiB column blocks: %.2f ms" % (mb, ms))
blocked.DOT_L2_BLOCK_BYTES = 2 << 20
g.csr.blocked.clear()
bp = blocked.plan_for_dot(g.csr, n, H, D, g.csr2csc)
if bp is not None:
    print("plan: T=%d tiles=%d nblk=%d block_rows=%d heavy_rows=%s" % (bp.T, bp.n_tiles, bp.nblk, bp.block_rows, None if bp.heavy is None else bp.heavy.n_long))
    ms_h = t(lambda: _C.spmm_dot(bp.heavy, x, w, g.csr2csc, y)) if bp.heavy is not None else 0
    print("hub rows alone (row kernel): %.2f ms" % ms_h)
blocked.ENABLED = True
ms = t(lambda: _C.spmm(g.csc, x, w, None))
print("forward spmm blocked: %.2f ms" % ms)
