"""Accuracy of every tuned GEMM selection vs the library default, against an fp64 reference (same random operands)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bot_amd import tuning
dev = "cuda"
N = 169343
torch.manual_seed(0)
cases = []
for K, P in ((750, 1536), (168, 768), (750, 128)):
    h = torch.randn(N, K, device=dev); W = torch.randn(P, K, device=dev) * 0.05; d = torch.randn(N, P, device=dev) * 1e-3
    cases += [(f"fwd  [N,{K}]x[{K},{P}]", lambda h=h, W=W: torch.mm(h, W.t()), lambda h=h, W=W: torch.mm(h.double(), W.double().t())),
              (f"dW   [{P},N]x[N,{K}]", lambda h=h, d=d: torch.mm(d.t(), h), lambda h=h, d=d: torch.mm(d.double().t(), h.double())),
              (f"dX   [N,{P}]x[{P},{K}]", lambda W=W, d=d: torch.mm(d, W), lambda W=W, d=d: torch.mm(d.double(), W.double()))]
base = [(n, f().double(), r()) for n, f, r in cases]
print("tuning:", tuning.enable())
for (n, f, r), (_, o0, ref) in zip(cases, base):
    o1 = f().double()
    e0 = ((o0 - ref).abs().max() / ref.abs().max()).item()
    e1 = ((o1 - ref).abs().max() / ref.abs().max()).item()
    print(f"{n:28s} max err / max|ref|: default {e0:.2e}   tuned {e1:.2e}")
