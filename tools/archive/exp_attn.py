"""Experiment: attention forward/backward kernels alone (S-proteins H=6 with edge term, S-arxiv H=3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bot_amd
from bot_amd import _C, ops, synth
name = sys.argv[1] if len(sys.argv) > 1 else "proteins"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n, e_raw, f, c = synth.SHAPES[name]
s, d = synth.powerlaw_edges(n, e_raw, synth.BASE_SEED, device="cuda")
g = bot_amd.preprocess(bot_amd.Graph(s, d, n)); g.create_formats_()
E = g.number_of_edges()
el = torch.randn(n, H, 1, device="cuda", requires_grad=True)
er = torch.randn(n, H, 1, device="cuda", requires_grad=True)
ee = torch.randn(E, H, 1, device="cuda") if name == "proteins" else None
def t(fn, it=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
def fwd():
    with torch.no_grad():
        return ops.gat_attention(g, el, er, ee=ee, order="csc", ee_order="csc" if ee is not None else None) if ee is not None else ops.gat_attention(g, el, er, order="csc")
print(name, "E", E, "H", H, "fwd %.3f ms" % t(fwd))
a = ops.gat_attention(g, el, er, ee=ee, order="csc", ee_order="csc") if ee is not None else ops.gat_attention(g, el, er, order="csc")
ga = torch.randn_like(a)
def bwd():
    torch.autograd.grad(a, [el, er], ga, retain_graph=True)
print("bwd %.3f ms" % t(bwd))
if name == "proteins":
    with torch.no_grad():
        for label, kw in (("el+er+ee", dict(el=el, er=er, ee=ee)), ("er+ee (no gather)", dict(el=None, er=er, ee=ee)),
                          ("el+er (no edge term)", dict(el=el, er=er, ee=None)), ("er only", dict(el=None, er=er, ee=None))):
            f = lambda: ops.gat_attention(g, kw["el"], kw["er"], ee=kw["ee"], order="csc", ee_order="csc")
            print("fwd %-24s %.3f ms" % (label, t(f)))
