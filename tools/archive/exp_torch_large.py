"""Probe: torch-ROCm row gathers x[idx] on index tensors of the S-products size (E = 1.26e8) — which shapes come back wrong
(compared with the same gather done in 8 M-index chunks), and whether index_select / gather are affected too."""
import torch
dev = torch.device("cuda", 0)
n = 2_449_029
gen = torch.Generator(device=dev).manual_seed(0)
def z_shape(E, W):
    return (E, W) if W else (E,)
def chunks(total, step=8_000_000):
    for a in range(0, total, step):
        yield slice(a, min(total, a + step))
for E in (30_000_000, 60_000_000, 67_108_864, 70_000_000, 126_144_421, 140_000_000):
    src = torch.randint(0, n, (E,), device=dev, generator=gen)
    for W in (0, 1, 2, 4, 8, 16):
        x = torch.randn((n, W) if W else (n,), device=dev, generator=gen)
        for name, fn in (("x[idx]", lambda: x[src]), ("index_select", lambda: torch.index_select(x, 0, src)),
                         ("int32 x[idx]", lambda: x[src.int()]), ("gather", lambda: torch.gather(x, 0, src if not W else src[:, None].expand(-1, W))),
                         ("take_along_dim/embedding", lambda: torch.nn.functional.embedding(src, x if W else x[:, None]).reshape(z_shape(E, W)))):
            try:
                z = fn()
                torch.cuda.synchronize()
            except Exception as ex:
                print(f"E={E} width={W or 'scalar'} {name}: raises {str(ex).splitlines()[0]}")
                continue
            bad, first = 0, None
            for c in chunks(E):
                w = (z[c] != x[src[c]])
                w = w.any(1) if W else w
                k = int(w.sum())
                if k and first is None:
                    first = c.start + int(w.nonzero()[0])
                bad += k
            if bad:
                print(f"E={E} width={W or 'scalar'} {name}: wrong rows {bad}, first at {first}")
    print("E", E, "done")
