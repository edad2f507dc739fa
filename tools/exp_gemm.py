"""Experiment: fp32-accurate projections from split-bf16/fp16 library GEMMs (hipBLASLt) vs the fp32 GEMM."""
import torch, time
dev = "cuda"
N, K, M = 169343, 750, 750
torch.manual_seed(0)
a = torch.randn(N, K, device=dev)
b = torch.randn(M, K, device=dev) * 0.05
ref64 = (a[:4096].double() @ b.double().t())

def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e3

def err(c):
    return ((c[:4096].double() - ref64).abs().max() / ref64.abs().max()).item(), ((c[:4096].double() - ref64).norm() / ref64.norm()).item()

print("fp32 mm: %.3f ms" % t(lambda: a @ b.t()), err(a @ b.t()))
for dt in (torch.bfloat16, torch.float16):
    ah = a.to(dt); al = (a - ah.float()).to(dt)
    bh = b.to(dt); bl = (b - bh.float()).to(dt)
    try:
        c = torch.mm(ah, bh.t(), out_dtype=torch.float32)
        print(dt, "out_dtype fp32 supported; single product: %.3f ms" % t(lambda: torch.mm(ah, bh.t(), out_dtype=torch.float32)), err(c))
    except Exception as e:
        print(dt, "out_dtype unsupported:", repr(e)[:200]); continue
    # x3 via K-concat: [ah, al] @ [bh; bh] + ah @ bl
    a2 = torch.cat([ah, al], 1); b2 = torch.cat([bh, bh], 1)
    def x3():
        c = torch.mm(a2, b2.t(), out_dtype=torch.float32)
        return c.addmm_(ah, bl.t()) if False else c + torch.mm(ah, bl.t(), out_dtype=torch.float32)
    print(dt, "x3 (2 GEMMs + add): %.3f ms" % t(x3), err(x3()))
    a3 = torch.cat([ah, al, ah], 1); b3 = torch.cat([bh, bh, bl], 1)
    print(dt, "x3 (one K-concat GEMM): %.3f ms" % t(lambda: torch.mm(a3, b3.t(), out_dtype=torch.float32)), err(torch.mm(a3, b3.t(), out_dtype=torch.float32)))
    print(dt, "split cost (to + sub + to): %.3f ms" % t(lambda: ((a - a.to(dt).float()).to(dt))))
    if dt == torch.bfloat16:
        am = al; r = a - ah.float() - am.float(); all_ = r.to(dt)
        bm = bl; rb = b - bh.float() - bm.float(); bll = rb.to(dt)
        a6 = torch.cat([ah, ah, am, ah, all_, am], 1); b6 = torch.cat([bh, bm, bh, bll, bh, bm], 1)
        print(dt, "x6 (one K-concat GEMM): %.3f ms" % t(lambda: torch.mm(a6, b6.t(), out_dtype=torch.float32)), err(torch.mm(a6, b6.t(), out_dtype=torch.float32)))
