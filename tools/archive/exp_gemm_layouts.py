"""Experiment: the three config-2 layer-1 GEMMs in both operand layouts, each with its best TunableOp kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.cuda import tunable
tunable.enable(True); tunable.tuning_enable(True)
tunable.set_max_tuning_iterations(20); tunable.set_max_tuning_duration(30)
tunable.set_filename("/tmp/exp_gemm_layouts.csv")
N, K, P = 169343, 750, 1536
dev = "cuda"
h = torch.randn(N, K, device=dev); W = torch.randn(P, K, device=dev); d = torch.randn(N, P, device=dev)
Wt = W.t().contiguous()          # [K, P]
def t(fn, it=10):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
fl = 2 * N * K * P / 1e9
for name, fn in (("fwd  h @ W.t()      (W [P,K])", lambda: torch.mm(h, W.t())),
                 ("fwd  h @ Wt         (Wt [K,P])", lambda: torch.mm(h, Wt)),
                 ("dX   d @ W          (W [P,K])", lambda: torch.mm(d, W)),
                 ("dX   d @ Wt.t()     (Wt [K,P])", lambda: torch.mm(d, Wt.t())),
                 ("dW   d.t() @ h      -> [P,K]", lambda: torch.mm(d.t(), h)),
                 ("dWt  h.t() @ d      -> [K,P]", lambda: torch.mm(h.t(), d))):
    ms = t(fn)
    print(f"{name:34s} {ms:7.3f} ms  {fl / ms:7.1f} TFLOP/s", flush=True)
