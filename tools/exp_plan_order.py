"""Row-plan item order on the headline graph (random ids): the plan's longest-first order vs natural row order dealt over the XCDs
(graph.xcd_item_order) — forward SpMM, fused backward and the layer-0 broadcast sweep, H = 3, D = 250 / Fin = 168."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bot_amd
from bot_amd import _C, synth, graph as G
ds = synth.make_dataset("arxiv", device="cpu", seed=0)
H, D = 3, 250
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for order in ("degree", "xcd", "row"):
    g = G.preprocess(ds.raw_graph if hasattr(ds, "raw_graph") else ds.graph, plan_order="xcd" if order != "degree" else "degree") if False else ds.graph
    g = g.to("cuda"); g._csc = g._csr = g._csr2csc = None
    g.plan_order = "xcd" if order == "xcd" else "degree"
    g.create_formats_()
    if order == "row":          # plain ascending row order
        for d in (g.csc, g.csr):
            it = d.items
            whole = it[:, 3] < 0
            rest = it[whole]
            d.items = torch.cat([it[~whole], rest[torch.argsort(rest[:, 0], stable=True)]]).contiguous()
    n, E = g.number_of_nodes(), g.number_of_edges()
    x = torch.randn(n, 1536, device="cuda")[:, :750].unflatten(1, (H, D))
    w = torch.rand(E, H, device="cuda")
    y = torch.randn(n, H, D, device="cuda")
    x168 = torch.randn(n, 168, device="cuda")
    _C.SPMM_LAYOUT = "segments"
    r = {"order": order,
         "spmm_fwd": round(timeit(lambda: _C.spmm(g.csc, x, w, None)), 4),
         "spmm_dot": round(timeit(lambda: _C.spmm_dot(g.csr, x, w, g.csr2csc, y)), 4),
         "spmm_bcast": round(timeit(lambda: _C.spmm_bcast(g.csc, x168, w)), 4)}
    print(r)
