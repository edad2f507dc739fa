"""Records the fastest hipBLASLt solution for every fp16-halves GEMM shape of the given workloads (bot_amd.gemm) by timing ALL
the library's solutions for these types on the real operands (bot_gemm_halves_f32 with tune == 2: first call per shape), and
writes gpurun_out/halves_gemm.json — copy it to bot_amd/tuning/halves_gemm.json.  Indices are tied to the hipBLASLt build (tagged).
    python tools/tune_halves_gemm.py [arxiv reddit products proteins]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BOT_GEMM_ALGOS"] = "0"            # search from scratch
import torch
from bot_amd import _C, tuning, workloads, train as T

names = sys.argv[1:] or ["arxiv"]
dev = torch.device("cuda", 0)
tuning.enable()
_C.GEMM_TUNE, _C.GEMM_SEEN = 2, {}
for name in names:
    t0 = time.time()
    wl = workloads.build(name, dev)
    for _ in range(2):
        wl.step()
    if name == "arxiv":          # the inference layers' projections too (evaluate())
        ds = wl.dataset
        T.evaluate(wl.model, ds.graph, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, use_labels=True, loss="loge",
                   n_classes=ds.n_classes)
    torch.cuda.synchronize()
    del wl
    torch.cuda.empty_cache()
    print(f"{name}: {len(_C.GEMM_SEEN)} shapes so far, {time.time() - t0:.0f} s", flush=True)
rec = {"hipblaslt": _C._hipblaslt_tag(), "made_by": "tools/tune_halves_gemm.py " + " ".join(names),
       "shapes": {k: {"index": i, "ms": round(ms, 4)} for k, (i, ms) in sorted(_C.GEMM_SEEN.items()) if i >= 0 and ms > 0}}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "halves_gemm.json"), "w"), indent=1)
for k, v in rec["shapes"].items():
    print(k, v)
