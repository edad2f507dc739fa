"""Experiment: the per-head GEMMs of the aggregate-before-project layer (config-2 layer 0: 3 heads, Fin = 168, D = 250) as
strided-batch fp16 GEMMs over the three-fold reduction axis, against the fp32 GEMMs they would replace (time only).
    python tools/exp_layer0_halves.py"""
import time, torch
dev = torch.device("cuda", 0)
N, H, Fin, D = 169343, 3, 168, 250
pz, pd = 192, 256
def timed(f, k=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
z = torch.randn(H, N, Fin, device=dev); W = torch.randn(H, D, Fin, device=dev) * 0.1
dx = torch.randn(N, H * D, device=dev)
print("fp32 fwd   3 x [N,168]x[168,250]      : %.3f ms" % timed(lambda: [torch.mm(z[i], W[i].t()) for i in range(H)]))
print("fp32 dz    3 x [N,250]x[250,168]      : %.3f ms" % timed(lambda: [torch.mm(dx[:, i * D:(i + 1) * D], W[i]) for i in range(H)]))
print("fp32 dW3   3 x [250,N]x[N,168]        : %.3f ms" % timed(lambda: [torch.mm(dx[:, i * D:(i + 1) * D].t(), z[i]) for i in range(H)]))
zh = torch.randn(H, N, 3 * pz, device=dev).half(); wh = torch.randn(H, D, 3 * pz, device=dev).half()
print("halves fwd  bmm [3][N,576]x[576,250]  : %.3f ms" % timed(lambda: torch.bmm(zh, wh.transpose(1, 2), out_dtype=torch.float32)))
wh2 = torch.randn(H, 256, 3 * pz, device=dev).half()
print("halves fwd  bmm, D padded to 256      : %.3f ms" % timed(lambda: torch.bmm(zh, wh2.transpose(1, 2), out_dtype=torch.float32)))
dxh = torch.randn(H, N, 3 * pd, device=dev).half(); wt = torch.randn(H, Fin, 3 * pd, device=dev).half()
print("halves dz   bmm [3][N,768]x[768,168]  : %.3f ms" % timed(lambda: torch.bmm(dxh, wt.transpose(1, 2), out_dtype=torch.float32)))
wt2 = torch.randn(H, 192, 3 * pd, device=dev).half()
print("halves dz   bmm, Fin padded to 192    : %.3f ms" % timed(lambda: torch.bmm(dxh, wt2.transpose(1, 2), out_dtype=torch.float32)))
S = 20; R = N // S
x1 = torch.randn(H * S, R, D, device=dev).half(); d12 = torch.randn(H * S, R, 2 * pz, device=dev).half(); d1 = torch.randn(H * S, R, pz, device=dev).half()
print("halves dW3  2 bmm over 3x20 chunks    : %.3f ms" % timed(lambda: (torch.bmm(x1.transpose(1, 2), d12, out_dtype=torch.float32).view(H, S, D, -1).sum(1),
                                                                        torch.bmm(x1.transpose(1, 2), d1, out_dtype=torch.float32).view(H, S, D, -1).sum(1))))
# one block-diagonal-free alternative: the three heads as ONE GEMM over concatenated columns is 3x the flops; for reference
zc = torch.randn(N, H * 3 * pz, device=dev).half(); wc = torch.randn(H * 3 * pz, 768, device=dev).half()
print("halves fwd  one dense GEMM [N,1728]x[1728,768] (3x flops): %.3f ms" % timed(lambda: torch.mm(zc, wc, out_dtype=torch.float32)))
