"""Experiment: primitives for the edge-drop draw at E = 77.7 M (randperm, sort, histograms, kthvalue) vs bot_random_keep_u8."""
import torch, time
E = 77753180
dev = "cuda"
def t(fn, it=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
bound = int(E * 0.1)
print("randperm %.2f ms" % t(lambda: torch.randperm(E, device=dev)))
r = torch.randint(0, 2**31 - 1, (E,), device=dev, dtype=torch.int32)
print("randint32 %.2f ms" % t(lambda: torch.randint(0, 2**31 - 1, (E,), device=dev, dtype=torch.int32)))
rf = torch.rand(E, device=dev)
print("rand f32 %.2f ms" % t(lambda: torch.rand(E, device=dev)))
try:
    print("kthvalue i32 %.2f ms" % t(lambda: torch.kthvalue(r, bound)))
except Exception as ex: print("kthvalue fail", ex)
try:
    print("kthvalue f32 %.2f ms" % t(lambda: torch.kthvalue(rf, bound)))
except Exception as ex: print("kthvalue fail", ex)
print("histc 65536 %.2f ms" % t(lambda: torch.histc(rf, bins=65536, min=0, max=1)))
print("histc 4096 %.2f ms" % t(lambda: torch.histc(rf, bins=4096, min=0, max=1)))
print("bincount(r>>19) %.2f ms" % t(lambda: torch.bincount((r >> 19).long(), minlength=4096)))
print("sort i32 %.2f ms" % t(lambda: torch.sort(r)))
print("cumsum u8->i32 %.2f ms" % t(lambda: torch.cumsum((r == 5), 0, dtype=torch.int32)))
print("compare %.2f ms" % t(lambda: (r > 12345).to(torch.uint8)))
perm = torch.randperm(E, device=dev)
def cur():
    p = torch.randperm(E, device=dev); eids = p[bound:]; keep = torch.zeros(E, dtype=torch.uint8, device=dev); keep[eids] = 1; return keep
print("current keep path %.2f ms" % t(cur))
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bot_amd import _C
print("bot random_keep %.2f ms" % t(lambda: _C.random_keep(E, E - bound, 99, dev)))
k = _C.random_keep(E, E - bound, 99, dev)
print("kept", int(k.sum()), "expected", E - bound)
