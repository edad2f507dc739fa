"""Experiment: head-major (group per (row, head)) vs row-major (group per row, all heads) gathers on S-products
(X = 4.7 GB >> Infinity Cache: every gathered row comes from HBM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bot_amd
from bot_amd import _C, synth
name = sys.argv[1] if len(sys.argv) > 1 else "products"
H, D = (4, 120) if name == "products" else (6, 80)
if len(sys.argv) > 3:
    H, D = int(sys.argv[2]), int(sys.argv[3])
n, e_raw, f, c = synth.SHAPES[name]
s, d = synth.powerlaw_edges(n, e_raw, synth.BASE_SEED, device="cuda")
g = bot_amd.preprocess(bot_amd.Graph(s, d, n)); g.create_formats_()
E = g.number_of_edges()
x = torch.randn(n, H, D, device="cuda")
w = torch.rand(E, H, device="cuda")
def t(fn, it=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
from bot_amd import blocked
blocked.ENABLED = os.environ.get("BOT_BLOCKED", "0") == "1"
print("E", E, "gather GB", E * H * D * 4 / 1e9)
print("spmm  [N,%d,%d] unweighted %.2f ms" % (H, D, t(lambda: _C.spmm(g.csc, x, None, None))))
print("spmm  [N,1,%d] unweighted %.2f ms" % (H * D, t(lambda: _C.spmm(g.csc, x.view(n, 1, H * D), None, None))))
print("spmm  [N,%d,%d] weighted   %.2f ms" % (H, D, t(lambda: _C.spmm(g.csc, x, w, None))))
y = torch.randn(n, H, D, device="cuda")
print("spmm_dot [N,%d,%d]        %.2f ms" % (H, D, t(lambda: _C.spmm_dot(g.csr, x, w, g.csr2csc, y))))
