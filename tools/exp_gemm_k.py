"""Experiment: the per-head GEMMs of the aggregate-first layer ([N,K] x [K,250], dz, dW) as a function of K (168 = 128 + 40)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.cuda import tunable
tunable.enable(True); tunable.tuning_enable(True)
tunable.set_max_tuning_iterations(20); tunable.set_max_tuning_duration(20)
tunable.set_filename("/tmp/exp_gemm_k.csv")
N, D = 169343, 250
dev = "cuda"
def t(fn, it=20):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
for K in (168, 176, 192):
    z = torch.randn(N, K, device=dev); W = torch.randn(D, K, device=dev); dx = torch.randn(N, 768, device=dev)[:, :D]
    out = torch.randn(N, 768, device=dev); dz = torch.empty(N, K, device=dev); dW = torch.empty(D, K, device=dev)
    f = t(lambda: out[:, :D].addmm_(z, W.t()))
    b1 = t(lambda: torch.mm(dx, W, out=dz))
    b2 = t(lambda: torch.mm(dx.t(), z, out=dW))
    print(f"K={K}: fwd {f:.3f} ms ({2*N*K*D/f/1e9:.0f} TF/s)  dz {b1:.3f} ms  dW {b2:.3f} ms", flush=True)
