"""One launch each of the three config-2 layer-0 shapes of bot_skinny_gemm_f32 (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bot_amd import _C
dev = torch.device("cuda", 0)
N, H, Fin, D, P2 = 169343, 3, 168, 250, 768
h = torch.randn(N, Fin, device=dev); Wr = torch.randn(Fin, P2, device=dev) * 0.1
z = torch.randn(H, N, Fin, device=dev); W = torch.randn(H, D, Fin, device=dev) * 0.1
out2 = torch.empty(N, P2, device=dev); dz = torch.empty(H, N, Fin, device=dev); dx = torch.randn(N, P2, device=dev)
for _ in range(2):
    _C.skinny_gemm(h, Wr, b_is_kn=True, out=out2)
    _C.skinny_gemm(z, W, b_is_kn=False, out=out2, accumulate=True, batch=H, strides=(N * Fin, D * Fin, D), m=N, n=D, k=Fin)
    _C.skinny_gemm(dx, W, b_is_kn=True, out=dz, batch=H, strides=(D, D * Fin, N * Fin), m=N, n=Fin, k=D)
torch.cuda.synchronize()
