"""Experiment: fp32 projection GEMM shapes under sustained load (merged fc+res, NT vs NN)."""
import torch, time
import torch.nn.functional as F
dev = "cuda"
N = 169343
def t(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3
for K in (750, 168):
    h = torch.randn(N, K, device=dev)
    W1 = torch.randn(750, K, device=dev) * 0.05; W2 = torch.randn(750, K, device=dev) * 0.05
    Wc = torch.cat([W1, W2]); WcT = Wc.t().contiguous(); W1T = W1.t().contiguous()
    dy = torch.randn(N, 750, device=dev); dyc = torch.randn(N, 1500, device=dev)
    fl = 2 * N * K * 750 / 1e9
    a = t(lambda: F.linear(h, W1));  print(f"K={K} fwd linear NT  [N,{K}]x[750,{K}]^T : {a:.3f} ms  {fl/a:.1f} TF/s")
    a = t(lambda: h @ W1T);          print(f"K={K} fwd mm NN      [N,{K}]x[{K},750]  : {a:.3f} ms  {fl/a:.1f} TF/s")
    a = t(lambda: F.linear(h, Wc));  print(f"K={K} fwd merged NT  [N,{K}]x[1500,{K}]^T: {a:.3f} ms  {2*fl/a:.1f} TF/s")
    a = t(lambda: h @ WcT);          print(f"K={K} fwd merged NN                      : {a:.3f} ms  {2*fl/a:.1f} TF/s")
    a = t(lambda: dy @ W1);          print(f"K={K} bwd dX  [N,750]x[750,{K}]          : {a:.3f} ms  {fl/a:.1f} TF/s")
    a = t(lambda: dyc @ Wc);         print(f"K={K} bwd dX merged [N,1500]x[1500,{K}]  : {a:.3f} ms  {2*fl/a:.1f} TF/s")
    a = t(lambda: dy.t() @ h);       print(f"K={K} bwd dW  [750,N]x[N,{K}]            : {a:.3f} ms  {fl/a:.1f} TF/s")
    a = t(lambda: dyc.t() @ h);      print(f"K={K} bwd dW merged [1500,N]x[N,{K}]     : {a:.3f} ms  {2*fl/a:.1f} TF/s")
    a = t(lambda: h.t() @ dy);       print(f"K={K} bwd dW^T [{K},N]x[N,750]           : {a:.3f} ms  {fl/a:.1f} TF/s")
