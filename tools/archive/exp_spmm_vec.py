"""Experiment: the weighted all-heads SpMM on S-arxiv at D = 250 (8-byte lanes, 6 chunks) vs D = 252 (16-byte lanes, 3 chunks):
whole table (fabric-bound) and with every source folded into 1 024 rows (L2-resident)."""
import dataclasses, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bot_amd import _C, synth
ds = synth.make_dataset("arxiv", device="cpu", seed=0)
g = ds.graph.to("cuda"); g.create_formats_()
n, E = g.number_of_nodes(), g.number_of_edges()
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
for D in (250, 252, 256):
    x = torch.randn(n, 3, D, device="cuda"); a = torch.rand(E, 3, device="cuda"); out = torch.empty(n, 3, D, device="cuda"); y = torch.randn(n, 3, D, device="cuda")
    ms = t(lambda: _C.spmm(g.csc, x, a, None, out=out)); k = _C._lib.bot_last_kernel().decode()
    dd = dataclasses.replace(g.csc, indices=(g.csc.indices % 1024).contiguous(), blocked={})
    ms_l2 = t(lambda: _C.spmm(dd, x, a, None, out=out))
    ms_b = t(lambda: _C.spmm_dot(g.csr, x, a, g.csr2csc, y)); kb = _C._lib.bot_last_kernel().decode()
    print("D=%d  fwd %.3f ms (%s)  L2-resident %.3f ms   fused bwd %.3f ms (%s)" % (D, ms, k, ms_l2, ms_b, kb))
