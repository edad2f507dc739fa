#!/usr/bin/env python3
"""How much per-ROW accuracy do the fp16-halves GEMMs keep when the rows / columns of the left operand differ in magnitude?
One power-of-two scale per matrix (csrc/halves.hip) keeps 22 bits of every entry within 2^-17 of the largest; below that the
second half goes subnormal.  For row / column magnitudes log-uniform in 2^[-S, S]: error of every output row relative to that row's
own largest entry (worst row and the median row), halves vs the stock fp32 GEMM, both against fp64."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/tools", 1)[0])
from bot_amd import gemm  # noqa: E402

dev = "cuda"
gen = torch.Generator(device=dev).manual_seed(0)
n, K, P = 20000, 750, 1536
for which in ("rows", "cols", "rows+cols"):
    for S in (0, 4, 8, 12, 16, 20):
        x = torch.randn(n, K, device=dev, generator=gen)
        if "rows" in which:
            x = x * torch.exp2((torch.rand(n, 1, device=dev, generator=gen) * 2 - 1) * S)
        if "cols" in which:
            x = x * torch.exp2((torch.rand(1, K, device=dev, generator=gen) * 2 - 1) * S)
        w = torch.randn(P, K, device=dev, generator=gen) * 0.05
        ref = x.double() @ w.double().t()
        rowmax = ref.abs().amax(1).clamp_min(1e-300)
        out = {}
        for name, got in (("halves", gemm.mm_nt(gemm.split(x, 0), gemm.split(w, 1))), ("rowscaled", gemm.mm_nt_rowscaled(x, w) if hasattr(gemm, "mm_nt_rowscaled") else None),
                          ("stock", x @ w.t())):
            if got is None:
                continue
            e = (got.double() - ref).abs().amax(1) / rowmax
            out[name] = (float(e.max()), float(e.median()), float((got.double() - ref).abs().max() / ref.abs().max()))
        print(which, "S=%2d" % S, "  ".join(f"{k}: worst row {v[0]:.2e} median row {v[1]:.2e} normwise {v[2]:.2e}" for k, v in out.items()), flush=True)
