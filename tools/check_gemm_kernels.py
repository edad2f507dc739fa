"""Accuracy of EVERY shipped GEMM selection (bot_amd/tuning/tunableop_gfx950.csv) against an fp64 reference, next to the
library default on the same operands.  Each entry's BLAS call (column-major m, n, k, transposes, leading dimensions) is
replayed through the torch call that produces exactly that TunableOp key; an entry whose error is more than 4x the
default kernel's (and above 1e-5 of the result's scale) is reported as BAD.

    python tools/check_gemm_kernels.py [substring filter]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from torch.cuda import tunable
from bot_amd import tuning

dev = "cuda"
flt = sys.argv[1] if len(sys.argv) > 1 else ""
entries = []
for line in open(tuning.FILE):
    parts = line.strip().split(",")
    if len(parts) != 4 or parts[0] == "Validator" or flt not in parts[1]:
        continue
    op, key, sel, _ = parts
    t, m, n, k, _, lda, ldb, ldc = key.split("_")
    entries.append((op, key, sel, t[0], t[1], int(m), int(n), int(k), int(lda), int(ldb), int(ldc)))


def operands(ta, tb, m, n, k, lda, ldb):
    """X [n,k] and Y [k,m] (views with the entry's leading dimensions) such that torch.mm(X, Y) issues this BLAS call."""
    g = torch.Generator(device=dev).manual_seed(m * 31 + n * 7 + k)
    if tb == "n":
        X = torch.randn(n, ldb, device=dev, generator=g)[:, :k]
    else:
        X = torch.randn(k, ldb, device=dev, generator=g)[:, :n].t()
    if ta == "n":
        Y = torch.randn(k, lda, device=dev, generator=g)[:, :m]
    else:
        Y = torch.randn(m, lda, device=dev, generator=g)[:, :k].t()
    return X, Y * 0.05


bad = 0
for op, key, sel, ta, tb, m, n, k, lda, ldb, ldc in entries:
    X, Y = operands(ta, tb, m, n, k, lda, ldb)
    bias = torch.randn(m, device=dev) if op.startswith("GemmAndBias") else None
    if bias is not None and not (ta == "t" and tb == "n"):
        print("skip (unexpected bias layout)", key)
        continue
    call = (lambda: F.linear(X, Y.t(), bias)) if bias is not None else (lambda: torch.mm(X, Y))
    ref = X.double() @ Y.double() + (bias.double() if bias is not None else 0)
    scale = float(ref.abs().max())
    tunable.enable(False)
    e0 = float((call().double() - ref).abs().max()) / scale
    tuning.enable()
    before = len(tunable.get_results())
    e1 = float((call().double() - ref).abs().max()) / scale
    flag = "BAD" if (e1 > 4 * e0 and e1 > 1e-5) else "ok"
    bad += flag == "BAD"
    print(f"{flag:3s} {op[:-len('TunableOp_float_NN')]:12s} {key:44s} {sel:28s} default {e0:.2e}  selected {e1:.2e}")
print("entries checked:", len(entries), "bad:", bad)
sys.exit(1 if bad else 0)
