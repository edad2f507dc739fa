"""GEMM kernel selection for the dense projections (the only MFMA work on this path).

hipBLASLt / rocBLAS pick their fp32 kernels by heuristics that are well off for the tall-skinny shapes of this workload
([169 343, 750] x [750, 1536]: 101 TFLOP/s by default vs 150 TFLOP/s — 95 % of the fp32 MFMA peak — for the best kernel in
the same libraries).  PyTorch's TunableOp can time the candidates once and remember the winner per shape; the winners for
the shapes of BASELINE config 2 on gfx950 ship in `tunableop_gfx950.csv` (generated with
`PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 python bench.py`, ~2 min) and are loaded read-only here, so no
tuning time is spent inside a run.  Shapes not in the file, or a library/arch version mismatch (the file carries validators),
silently keep the library default.  Results are bit-identical math: same fp32 MFMA kernels family, different tile choice.
"""
from __future__ import annotations

import os

import torch

FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")


def enable(tune_missing: bool = False) -> bool:
    """Turn TunableOp on with the shipped selections.  `tune_missing=True` also times shapes that are not in the file
    (costs seconds per new shape on first use).  Returns False when TunableOp is unavailable."""
    try:
        from torch.cuda import tunable
        tunable.enable(True)
        tunable.tuning_enable(bool(tune_missing))
        if hasattr(tunable, "write_file_on_exit"):
            tunable.write_file_on_exit(False)
        # whatever TunableOp writes at exit goes to scratch, never into the package directory
        tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"bot_amd_tunableop_{os.getpid()}.csv"))
        return bool(tunable.read_file(FILE)) if os.path.exists(FILE) else False
    except Exception:  # noqa: BLE001 - optional optimisation only
        return False
