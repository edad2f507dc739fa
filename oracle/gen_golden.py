#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — generate tests/golden/*.npz by executing the reference's own modules.

Runs ONLY in the authoring container (needs /root/reference).  It puts the reference's source
directories on sys.path *without copying anything*, installs `oracle/dgl_standin.py` for the absent
`dgl`/`ogb` imports, instantiates the reference's `GraphConv`, `GATConv`, `GCN`, `GAT`
(src/no-sampling/models.py), the proteins `GATConv`/`GAT` (src/ogbn-proteins/models.py) and calls
`add_labels` / `compute_loss` / `train` (src/no-sampling/run.py) on small seeded graphs, then dumps
inputs, parameters, outputs and gradients as plain arrays.  The committed fixtures are data
(inputs + expected outputs); no reference source or bytecode is written anywhere.

    python -m oracle.gen_golden            # rewrites tests/golden/*.npz
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import types

sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only reference tree
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
OUT = os.path.join(REPO, "tests", "golden")

sys.path.insert(0, REPO)
from oracle import dgl_standin, ref_ops  # noqa: E402
from oracle.dgl_standin import StandinGraph  # noqa: E402


def _import_ref(subdir, name, alias):
    """Import /root/reference/src/<subdir>/<name>.py under a private alias (two dirs hold a models.py)."""
    path = os.path.join(REF, subdir)
    sys.path.insert(0, path)
    try:
        sys.modules.pop(name, None)
        m = importlib.import_module(name)
        sys.modules[alias] = m
        sys.modules.pop(name, None)
        return m
    finally:
        sys.path.remove(path)


def powerlaw_edges(n, e_raw, seed, gamma=2.0):
    g = torch.Generator().manual_seed(seed)
    src = (n * torch.rand(e_raw, generator=g, dtype=torch.float64) ** gamma).long().clamp_(max=n - 1)
    dst = (n * torch.rand(e_raw, generator=g, dtype=torch.float64) ** gamma).long().clamp_(max=n - 1)
    perm = torch.randperm(n, generator=g)
    return perm[src], perm[dst]


def t2n(t):
    return t.detach().cpu().numpy().copy()  # copy: later in-place updates (optimizer step, BN running stats) must not leak in


def sd2n(prefix, sd):
    return {f"{prefix}{k}": t2n(v) for k, v in sd.items()}


def grads2n(prefix, module):
    return {f"{prefix}{k}": t2n(p.grad) for k, p in module.named_parameters() if p.grad is not None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=OUT)
    ap.add_argument("--only", default="", help="\"products\": stop after the products fixtures")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)

    dgl_standin.install()
    M = _import_ref("no-sampling", "models", "ref_ns_models")
    # run.py does `from models import ...` at import time: expose the no-sampling models under that name
    sys.modules["models"] = M
    RUN = _import_ref("no-sampling", "run", "ref_ns_run")
    sys.modules.pop("models", None)
    P = _import_ref("ogbn-proteins", "models", "ref_proteins_models")
    torch.set_printoptions(precision=4)  # the reference sets 20 at import (models.py:15)

    # ------------------------------------------------------------------ graphs
    graphs = {}
    s, d = torch.tensor([0, 1, 2, 3, 2, 5]), torch.tensor([1, 2, 3, 4, 0, 3])  # models.py:187
    graphs["doc_noloop"] = (s, d, 6)
    graphs["doc_loop"] = (*ref_ops.add_self_loop(s, d, 6), 6)
    for name, n, e_raw, seed in (("g64", 64, 220, 11), ("g300", 300, 1500, 12)):
        rs, rd = powerlaw_edges(n, e_raw, seed)
        # reference preprocess(): run.py:133-148, executed through the stand-in graph
        g0 = StandinGraph(rs, rd, n)
        g0.ndata["feat"] = torch.zeros(n, 1)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            g1 = RUN.preprocess(g0)
        graphs[name] = (g1._src, g1._dst, n)
        graphs[name + "_raw"] = (rs, rd, n)

    out = {}
    for name, (s_, d_, n_) in graphs.items():
        out[f"{name}.src"], out[f"{name}.dst"], out[f"{name}.n"] = t2n(s_), t2n(d_), np.int64(n_)
        if not name.endswith("_raw"):
            out[f"{name}.in_deg"] = t2n(ref_ops.in_degrees(d_, n_))
            out[f"{name}.out_deg"] = t2n(ref_ops.out_degrees(s_, n_))
    np.savez_compressed(os.path.join(args.out, "graphs.npz"), **out)

    def G(name):
        s_, d_, n_ = graphs[name]
        return StandinGraph(s_, d_, n_)

    # ------------------------------------------------------------------ GraphConv (models.py:114-413)
    out = {}
    gen = torch.Generator().manual_seed(100)
    case = 0
    for gname in ("doc_loop", "g64", "g300"):
        for norm in ("both", "right", "none"):
            for fin, fout in ((10, 4), (4, 10)):
                for dtype in (torch.float32, torch.float64):
                    if dtype == torch.float64 and gname == "g300":
                        continue
                    g = G(gname)
                    torch.manual_seed(1000 + case)
                    conv = M.GraphConv(fin, fout, norm=norm, weight=True, bias=True).to(dtype)
                    with torch.no_grad():
                        conv.bias.normal_(generator=gen)
                    feat = torch.randn(g.number_of_nodes(), fin, generator=gen).to(dtype).requires_grad_()
                    gout = torch.randn(g.number_of_nodes(), fout, generator=gen).to(dtype)
                    rst = conv(g, feat)
                    (rst * gout).sum().backward()
                    k = f"c{case}."
                    out[k + "meta"] = np.array([gname, norm, str(fin), str(fout), str(dtype).split(".")[1]])
                    out.update(sd2n(k + "p.", conv.state_dict()))
                    out.update(grads2n(k + "g.", conv))
                    out[k + "feat"], out[k + "gout"], out[k + "rst"] = t2n(feat), t2n(gout), t2n(rst)
                    out[k + "dfeat"] = t2n(feat.grad)
                    case += 1
    out["n_cases"] = np.int64(case)
    # docstring known-answer rows: models.py:193-198 and :204-209 (data printed by the reference)
    out["doc.case1"] = np.array([[1.3326, -0.2797], [1.4673, -0.3080], [1.3326, -0.2797],
                                 [1.6871, -0.3541], [1.7711, -0.3717], [1.0375, -0.2178]])
    out["doc.case2"] = np.array([[-0.2473, -0.4631], [-0.3497, -0.6549], [-0.3497, -0.6549],
                                 [-0.4221, -0.7905], [-0.3497, -0.6549], [0.0, 0.0]])
    np.savez_compressed(os.path.join(args.out, "graphconv.npz"), **out)

    # ------------------------------------------------------------------ GATConv (models.py:416-566)
    # every section draws its inputs from its OWN generator: adding or reordering sections never changes another file
    gen = torch.Generator().manual_seed(200)
    out = {}
    case = 0
    real_randperm = torch.randperm
    for gname in ("g64", "g300"):
        for symm in (False, True):
            for attn_r in (False, True):
                for linear in (False, True):
                    for H, D, fin in ((3, 5, 12), (1, 7, 9)):
                        if gname == "g300" and H == 1 and not symm:
                            continue
                        for edge_drop in (0.0, 0.3):
                            if edge_drop > 0 and not (linear and attn_r):
                                continue
                            for dtype in (torch.float32, torch.float64):
                                if dtype == torch.float64 and (gname == "g300" or H == 1):
                                    continue
                                g = G(gname)
                                n = g.number_of_nodes()
                                torch.manual_seed(2000 + case)
                                conv = M.GATConv(fin, D, num_heads=H, edge_drop=edge_drop, linear=linear,
                                                 use_symmetric_norm=symm, non_interactive_attn=attn_r).to(dtype)
                                feat = torch.randn(n, fin, generator=gen).to(dtype).requires_grad_()
                                gout = torch.randn(n, H, D, generator=gen).to(dtype)
                                k = f"c{case}."
                                captured = {}

                                def spy(*a, **kw):
                                    p = real_randperm(*a, **kw)
                                    captured["perm"] = p.clone()
                                    return p

                                conv.train(edge_drop > 0)
                                torch.randperm = spy
                                try:
                                    rst = conv(g, feat)
                                finally:
                                    torch.randperm = real_randperm
                                (rst * gout).sum().backward()
                                if edge_drop > 0:
                                    bound = int(g.number_of_edges() * edge_drop)  # models.py:531
                                    out[k + "keep_eids"] = t2n(captured["perm"][bound:])
                                out[k + "meta"] = np.array([gname, str(int(symm)), str(int(attn_r)), str(int(linear)),
                                                            str(H), str(D), str(fin), str(edge_drop),
                                                            str(dtype).split(".")[1]])
                                out.update(sd2n(k + "p.", conv.state_dict()))
                                out.update(grads2n(k + "g.", conv))
                                out[k + "feat"], out[k + "gout"], out[k + "rst"] = t2n(feat), t2n(gout), t2n(rst)
                                out[k + "dfeat"] = t2n(feat.grad)
                                case += 1
    out["n_cases"] = np.int64(case)
    np.savez_compressed(os.path.join(args.out, "gatconv.npz"), **out)

    # ------------------------------------------------------------------ stacks GCN / GAT (models.py:569-736)
    gen = torch.Generator().manual_seed(300)
    out = {}
    case = 0
    stack_cfgs = [
        ("gcn", dict(n_layers=2, n_hidden=16, norm="none", norm_adj="symm", use_linear=False, residual=False)),
        ("gcn", dict(n_layers=3, n_hidden=12, norm="batch", norm_adj="rw", use_linear=True, residual=True)),
        ("gat", dict(n_layers=3, n_heads=3, n_hidden=6, norm="batch", non_interactive_attn=False,
                     use_symmetric_norm=False, linear=True, residual=False)),
        ("gat", dict(n_layers=3, n_heads=2, n_hidden=5, norm="none", non_interactive_attn=True,
                     use_symmetric_norm=True, linear=True, residual=True)),
        ("gat", dict(n_layers=2, n_heads=3, n_hidden=4, norm="batch", non_interactive_attn=True,
                     use_symmetric_norm=True, linear=False, residual=False)),
    ]
    for gname in ("g64", "g300"):
        for kind, cfg in stack_cfgs:
            for training in (False, True):
                g = G(gname)
                n, fin, C = g.number_of_nodes(), 11, 5
                torch.manual_seed(3000 + case)
                if kind == "gcn":
                    model = M.GCN(in_feats=fin, n_classes=C, activation=F.relu, **cfg)
                else:
                    model = M.GAT(dim_node=fin, dim_edge=0, dim_output=C, activation=F.relu, **cfg)
                # non-trivial BN running stats / affine so eval mode is a real test
                with torch.no_grad():
                    for m in model.modules():
                        if isinstance(m, torch.nn.BatchNorm1d):
                            m.running_mean.normal_(0, 0.3, generator=gen)
                            m.running_var.uniform_(0.5, 1.5, generator=gen)
                            m.weight.uniform_(0.5, 1.5, generator=gen)
                            m.bias.normal_(0, 0.3, generator=gen)
                    for name_, p in model.named_parameters():
                        if name_.endswith("bias") and p.dim() == 1 and "norms" not in name_:
                            p.normal_(0, 0.3, generator=gen)
                model.train(training)
                k = f"c{case}."
                out.update(sd2n(k + "p.", model.state_dict()))  # before forward: BN running stats as given
                feat = torch.randn(n, fin, generator=gen).requires_grad_()
                gout = torch.randn(n, C, generator=gen)
                logits = model(g, feat)
                (logits * gout).sum().backward()
                out[k + "meta"] = np.array([gname, kind, str(int(training)), repr(cfg)])
                out.update(grads2n(k + "g.", model))
                out[k + "feat"], out[k + "gout"], out[k + "logits"] = t2n(feat), t2n(gout), t2n(logits)
                out[k + "dfeat"] = t2n(feat.grad)
                out[k + "n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
                case += 1
    out["n_cases"] = np.int64(case)
    # parameter-count known answers recorded by the reference authors: run.py:723,828,1009
    RUN.n_node_feats, RUN.n_classes, RUN.n_edge_feats = 128, 40, 0
    ns = types.SimpleNamespace(labels=True, activation="relu", model="gat", n_hidden=250, n_layers=3, n_heads=3,
                               norm="batch", dropout=0.75, input_drop=0.25, attn_drop=0.1, edge_drop=0.0,
                               non_interactive_attn=False, norm_adj="rw", linear=True, residual=False)
    out["count.arxiv_gat_cfg2"] = np.int64(RUN.count_parameters(ns))  # reference prints 1441580 at run.py:1009
    ns2 = types.SimpleNamespace(**{**vars(ns), "model": "gcn", "labels": False, "n_hidden": 256, "linear": False,
                                   "norm_adj": "symm", "dropout": 0.5, "input_drop": 0.0})
    out["count.arxiv_gcn_h256"] = np.int64(RUN.count_parameters(ns2))  # reference prints 109608 at run.py:828
    np.savez_compressed(os.path.join(args.out, "stacks.npz"), **out)

    # ------------------------------------------------------------------ proteins GATConv / GAT
    gen = torch.Generator().manual_seed(400)
    out = {}
    case = 0
    for gname in ("g64", "g300"):
        for edge_feats, use_attn_dst, edge_drop in ((8, True, 0.0), (0, False, 0.0), (8, True, 0.25)):
            g = G(gname)
            n, E = g.number_of_nodes(), g.number_of_edges()
            H, D, fin = 3, 4, 10
            torch.manual_seed(4000 + case)
            conv = P.GATConv(fin, edge_feats, D, n_heads=H, edge_drop=edge_drop, use_attn_dst=use_attn_dst,
                             allow_zero_in_degree=False)
            with torch.no_grad():
                conv.dst_fc.bias.normal_(0, 0.3, generator=gen)
            feat = torch.randn(n, fin, generator=gen).requires_grad_()
            efeat = torch.rand(E, edge_feats, generator=gen).requires_grad_() if edge_feats else None
            gout = torch.randn(n, H, D, generator=gen)
            captured = {}

            def spy(*a, **kw):
                p = real_randperm(*a, **kw)
                captured["perm"] = p.clone()
                return p

            conv.train(edge_drop > 0)
            torch.randperm = spy
            try:
                rst = conv(g, feat, efeat)
            finally:
                torch.randperm = real_randperm
            (rst * gout).sum().backward()
            k = f"c{case}."
            if edge_drop > 0:
                out[k + "keep_eids"] = t2n(captured["perm"][int(E * edge_drop):])
            out[k + "meta"] = np.array([gname, str(edge_feats), str(int(use_attn_dst)), str(edge_drop), str(H), str(D)])
            out.update(sd2n(k + "p.", conv.state_dict()))
            out.update(grads2n(k + "g.", conv))
            out[k + "feat"], out[k + "gout"], out[k + "rst"], out[k + "dfeat"] = t2n(feat), t2n(gout), t2n(rst), t2n(feat.grad)
            if efeat is not None:
                out[k + "efeat"], out[k + "defeat"] = t2n(efeat), t2n(efeat.grad)
            case += 1
    out["n_conv_cases"] = np.int64(case)
    # full-graph GAT stack (src/ogbn-proteins/models.py:230-264), eval + train mode
    for training in (False, True):
        g = G("g64")
        n, E = g.number_of_nodes(), g.number_of_edges()
        torch.manual_seed(4100 + int(training))
        model = P.GAT(node_feats=9, edge_feats=8, n_classes=6, n_layers=2, n_heads=2, n_hidden=5, edge_emb=16,
                      activation=F.relu, dropout=0.0, input_drop=0.0, attn_drop=0.0, edge_drop=0.0)
        model.train(training)
        g.ndata["feat"] = torch.randn(n, 9, generator=gen)
        g.edata["feat"] = torch.rand(E, 8, generator=gen)
        k = f"s{int(training)}."
        out.update(sd2n(k + "p.", model.state_dict()))
        gout = torch.randn(n, 6, generator=gen)
        logits = model(g)
        (logits * gout).sum().backward()
        out.update(grads2n(k + "g.", model))
        out[k + "nfeat"], out[k + "efeat"], out[k + "gout"], out[k + "logits"] = (
            t2n(g.ndata["feat"]), t2n(g.edata["feat"]), t2n(gout), t2n(logits))
        out[k + "n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
    # reference-recorded count: ogbn-proteins/gat.py:377 (no labels) = 2475232
    big = P.GAT(node_feats=8, edge_feats=8, n_classes=112, n_layers=6, n_heads=6, n_hidden=80, edge_emb=16,
                activation=F.relu, dropout=0.25, input_drop=0.1, attn_drop=0.0, edge_drop=0.1)
    out["count.proteins_gat"] = np.int64(sum(p.numel() for p in big.parameters()))
    np.savez_compressed(os.path.join(args.out, "proteins.npz"), **out)

    # ------------------------------------------------------------------ products GAT stack (src/ogbn-products/models.py:170-265)
    if args.only in ("", "products"):
        PR = _import_ref("ogbn-products", "models", "ref_products_models")
        gen = torch.Generator().manual_seed(420)
        out = {}
        case = 0
        for residual, edge_emb, training in ((False, 0, True), (True, 0, True), (True, 16, True), (False, 0, False)):
            g = G("g64")
            n, E = g.number_of_nodes(), g.number_of_edges()
            torch.manual_seed(4200 + case)
            edge_feats = 8 if edge_emb else 0
            model = PR.GAT(node_feats=9, edge_feats=edge_feats, n_classes=6, n_layers=3, n_heads=2, n_hidden=5, edge_emb=edge_emb,
                           activation=F.relu, dropout=0.0, input_drop=0.0, attn_drop=0.0, edge_drop=0.0, residual=residual)
            model.train(training)
            g.ndata["feat"] = torch.randn(n, 9, generator=gen)
            if edge_emb:
                g.edata["feat"] = torch.rand(E, 8, generator=gen)
            k = f"s{case}."
            out.update(sd2n(k + "p.", model.state_dict()))
            gout = torch.randn(n, 6, generator=gen)
            logits = model(g)
            (logits * gout).sum().backward()
            out.update(grads2n(k + "g.", model))
            out[k + "nfeat"], out[k + "gout"], out[k + "logits"] = t2n(g.ndata["feat"]), t2n(gout), t2n(logits)
            if edge_emb:
                out[k + "efeat"] = t2n(g.edata["feat"])
            out[k + "meta"] = np.array([str(int(residual)), str(edge_emb), str(int(training))])
            out[k + "n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
            case += 1
        out["n_cases"] = np.int64(case)
        np.savez_compressed(os.path.join(args.out, "products.npz"), **out)
        if args.only == "products":
            return

    # ------------------------------------------------------------------ callers: run.py add_labels/compute_loss/train
    gen = torch.Generator().manual_seed(500)
    out = {}
    RUN.device = torch.device("cpu")
    # RMSprop warm-up (run.py:246-249, 351-352) at epochs 1 / 25 / 50 (last warm-up epoch) / 51 (first epoch past it);
    # label reuse (run.py:274-279, 304-308) once on a GCN and once on a GAT
    train_cases = (("g64", "gat", "rmsprop", "loge", 25, 0), ("g300", "gcn", "adam", "savage", 1, 1),
                   ("g64", "gat", "rmsprop", "logit", 51, 0), ("g300", "gat", "rmsprop", "loge", 50, 1),
                   ("g64", "gcn", "rmsprop", "logit", 1, 0))
    out["n_cases"] = np.int64(len(train_cases))
    for ci, (gname, kind, optim_name, loss_name, epoch, label_iters) in enumerate(train_cases):
        g = G(gname)
        n, fin, C = g.number_of_nodes(), 7, 4
        RUN.n_node_feats, RUN.n_classes, RUN.n_edge_feats = fin, C, 0
        a = types.SimpleNamespace(labels=True, activation="relu", model=kind, n_hidden=6, n_layers=2, n_heads=2,
                                  norm="batch", dropout=0.0, input_drop=0.0, attn_drop=0.0, edge_drop=0.0,
                                  non_interactive_attn=False, norm_adj="symm" if kind == "gcn" else "rw",
                                  linear=True, residual=False, mask_rate=0.5, n_label_iters=label_iters,
                                  loss=loss_name, optimizer=optim_name, lr=0.01, wd=0.0)
        torch.manual_seed(5000 + ci)
        model = RUN.build_model(a)
        feat = torch.randn(n, fin, generator=gen)
        labels = torch.randint(0, C, (n, 1), generator=gen)
        perm = torch.randperm(n, generator=gen)
        tr, va, te = perm[: n // 2], perm[n // 2: 3 * n // 4], perm[3 * n // 4:]
        g.ndata["feat"] = feat
        opt = (torch.optim.RMSprop if optim_name == "rmsprop" else torch.optim.Adam)(model.parameters(), lr=a.lr)
        if optim_name == "rmsprop":
            RUN.adjust_learning_rate(opt, a.lr, epoch)  # run.py:351-352
        k = f"t{ci}."
        out.update(sd2n(k + "p0.", model.state_dict()))
        torch.manual_seed(6000 + ci)
        mask = torch.rand(tr.shape) < a.mask_rate  # what train() will draw at run.py:258 under this seed
        torch.manual_seed(6000 + ci)
        acc, loss = RUN.train(a, model, g, labels, tr, va, te, opt, RUN.compute_acc)
        out[k + "meta"] = np.array([gname, kind, optim_name, loss_name, str(epoch), str(a.n_label_iters)])
        out[k + "feat"], out[k + "labels"], out[k + "mask"] = t2n(feat), t2n(labels), t2n(mask)
        out[k + "train_idx"], out[k + "val_idx"], out[k + "test_idx"] = t2n(tr), t2n(va), t2n(te)
        out[k + "loss"], out[k + "acc"] = np.float64(loss), np.float64(acc)
        out[k + "lr"] = np.float64(opt.param_groups[0]["lr"])
        out.update(grads2n(k + "g.", model))
        out.update(sd2n(k + "p1.", model.state_dict()))
        out[k + "aug"] = t2n(RUN.add_labels(feat, labels, tr[mask]))  # run.py:240-243
        # evaluate() on the post-step model — run.py:290-322 (eval mode, all training labels as inputs, label reuse)
        ev = RUN.evaluate(a, model, g, labels, tr, va, te, RUN.compute_acc, epoch)
        out[k + "eval_accs"] = np.array([float(v) for v in ev[:3]])
        out[k + "eval_losses"] = np.array([float(v) for v in ev[3:6]])
        out[k + "eval_pred"] = t2n(ev[6])
    RUN.n_classes = 5
    gen = torch.Generator().manual_seed(510)
    x = torch.randn(40, 5, generator=gen)
    y = torch.randint(0, 5, (40, 1), generator=gen)
    out["loss.x"], out["loss.y"] = t2n(x), t2n(y)
    for ln in ("logit", "loge", "savage"):  # run.py:229-237
        out[f"loss.{ln}"] = np.float64(RUN.compute_loss(types.SimpleNamespace(loss=ln), x, y).item())
    np.savez_compressed(os.path.join(args.out, "train.npz"), **out)

    sizes = {f: os.path.getsize(os.path.join(args.out, f)) for f in sorted(os.listdir(args.out))}
    print("wrote", sizes, "total", sum(sizes.values()))
    # leave no trace in the reference tree
    for root, dirs, _ in os.walk("/root/reference"):
        assert "__pycache__" not in dirs, root


if __name__ == "__main__":
    main()
