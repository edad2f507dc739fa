"""INTEGRATION.md §1 executed: the reference's OWN, unmodified `src/no-sampling/models.py` imported with its `dgl` names bound
to bot_amd's mirror of that surface — `bot_amd.function`, `bot_amd.ops.edge_softmax`, `bot_amd.utils.expand_as_pair`,
`bot_amd.DGLError` — and its `GraphConv` / `GATConv` / `GCN` / `GAT` run on `bot_amd.Graph`, checked against the golden vectors
(tests/golden/{graphconv,gatconv,stacks}.npz, which the same modules produced on the oracle's DGL stand-in).

Container-only: the reference tree does not exist on the GPU box and nothing of it is copied (it is imported where it lies, with
bytecode writing off).  The `dgl` namespace installed here holds NO arithmetic: every name is an alias of a bot_amd object or an
unused placeholder (`dgl.nn.pytorch`, `Identity` are imported by models.py:3,10 and never used).  Kernels: the emulated backend
(tests/_oracle_backend.py) — there is no GPU here; the same op sequence on the real kernels is `test_dgl_surface` in the GPU suite.
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import bot_amd
from tests import _oracle_backend, parity_cases as PC

REF_MODELS = "/root/reference/src/no-sampling/models.py"
pytestmark = pytest.mark.skipif(not os.path.isfile(REF_MODELS), reason="needs the reference tree (authoring container only)")


@pytest.fixture()
def ref_models(monkeypatch):
    """The reference's models.py as a module, its `import dgl…` lines (models.py:3-13) resolved to bot_amd."""
    _oracle_backend.install(monkeypatch)
    from bot_amd import function, ops, utils

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        monkeypatch.setitem(sys.modules, name, m)
        return m
    nn_utils = mod("dgl.nn.pytorch.utils", Identity=torch.nn.Identity)          # models.py:10, imported, never used
    nn_pytorch = mod("dgl.nn.pytorch", utils=nn_utils)                          # models.py:3 (`dglnn`), never used
    nn_ = mod("dgl.nn", pytorch=nn_pytorch)
    base = mod("dgl._ffi.base", DGLError=bot_amd.DGLError)                      # models.py:9
    ffi = mod("dgl._ffi", base=base)
    ops_ = mod("dgl.ops", edge_softmax=ops.edge_softmax)                        # models.py:11
    utils_ = mod("dgl.utils", expand_as_pair=utils.expand_as_pair)              # models.py:12
    monkeypatch.setitem(sys.modules, "dgl.function", function)                  # models.py:8
    mod("dgl", nn=nn_, _ffi=ffi, ops=ops_, utils=utils_, function=function)
    monkeypatch.setattr(sys, "dont_write_bytecode", True)                       # never drop __pycache__ into the reference tree
    opts = torch._tensor_str.PRINT_OPTS
    saved = (opts.precision, opts.threshold, opts.edgeitems, opts.linewidth, opts.sci_mode)
    spec = importlib.util.spec_from_file_location("_ref_no_sampling_models", REF_MODELS)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)                                                  # unmodified source, executed where it lies
    yield m
    opts.precision, opts.threshold, opts.edgeitems, opts.linewidth, opts.sci_mode = saved   # models.py:15 sets precision=20
    assert not os.path.isdir(os.path.join(os.path.dirname(REF_MODELS), "__pycache__"))


def test_reference_graphconv_on_bot_amd(golden, ref_models):
    n = 0
    for c in golden.cases("graphconv"):
        gname, norm, fin, fout, dt = (str(x) for x in c["meta"])
        if dt != "float32":
            continue
        g = PC.make_graph(golden, gname, "cpu")
        conv = PC.load_params(ref_models.GraphConv(int(fin), int(fout), norm=norm), c, "cpu")
        feat = PC.leaf(c.t("feat"))
        rst = conv(g, feat)
        PC.fwd_close(rst, c["rst"])
        (rst * c.t("gout")).sum().backward()
        PC.grad_close(feat.grad, c["dfeat"])
        PC.grad_close(conv.weight.grad, c["g.weight"])
        PC.grad_close(conv.bias.grad, c["g.bias"])
        n += 1
    assert n >= 4
    # the docstring known-answer rows of the reference itself (models.py:186-209) through its own module on bot_amd.Graph
    s, d, nn_ = golden.graph("doc_loop")
    conv = ref_models.GraphConv(10, 2, norm="both", weight=True, bias=True)
    res = conv(bot_amd.Graph(s, d, nn_), torch.ones(6, 10)).detach()
    ratio = res / torch.tensor([0.9082, 1.0000, 0.9082, 1.1498, 1.2071, 0.7071]).unsqueeze(1)
    assert torch.allclose(ratio, ratio[:1].expand_as(ratio), atol=2e-4)
    s, d, nn_ = golden.graph("doc_noloop")
    with pytest.raises(bot_amd.DGLError):                                        # models.py:334-346 raises the swapped-in DGLError
        conv(bot_amd.Graph(s, d, nn_), torch.ones(6, 10))


def test_reference_gatconv_on_bot_amd(golden, ref_models, monkeypatch):
    n = n_drop = 0
    real_randperm = torch.randperm
    for c in golden.cases("gatconv"):
        gname, symm, attn_r, linear, H, D, fin, edge_drop, dt = (str(x) for x in c["meta"])
        if dt != "float32":
            continue
        g = PC.make_graph(golden, gname, "cpu")
        conv = ref_models.GATConv(int(fin), int(D), num_heads=int(H), edge_drop=float(edge_drop), linear=bool(int(linear)),
                                  use_symmetric_norm=bool(int(symm)), non_interactive_attn=bool(int(attn_r)))
        conv = PC.load_params(conv, c, "cpu")
        if "keep_eids" in c:
            # the reference draws `perm = torch.randperm(E)` and keeps perm[int(E * p):] (models.py:528-532): hand it the
            # permutation the fixture recorded (dropped edges in front) so that the same subset is kept
            E = g.number_of_edges()
            keep = c.t("keep_eids").long()
            rest = torch.tensor(sorted(set(range(E)) - set(keep.tolist())), dtype=torch.int64)
            assert rest.numel() == int(E * float(edge_drop))
            monkeypatch.setattr(torch, "randperm", lambda *a, **kw: torch.cat([rest, keep]))
            conv.train()
            n_drop += 1
        feat = PC.leaf(c.t("feat"))
        rst = conv(g, feat)
        monkeypatch.setattr(torch, "randperm", real_randperm)
        PC.fwd_close(rst, c["rst"])
        (rst * c.t("gout")).sum().backward()
        PC.grad_close(feat.grad, c["dfeat"])
        for k, p in conv.named_parameters():
            PC.grad_close(p.grad, c[f"g.{k}"])
        assert "ft" not in g.ndata and "a" not in g.edata                        # graph.local_scope() (models.py:476) restored
        n += 1
    assert n >= 8 and n_drop >= 2


def test_reference_stacks_on_bot_amd(golden, ref_models):
    n = 0
    for c in golden.cases("stacks"):
        gname, kind, training, cfg = (str(x) for x in c["meta"])
        cfg = ast.literal_eval(cfg)
        g = PC.make_graph(golden, gname, "cpu")
        if kind == "gcn":
            model = ref_models.GCN(in_feats=11, n_classes=5, activation=F.relu, **cfg)
        else:
            model = ref_models.GAT(dim_node=11, dim_edge=0, dim_output=5, activation=F.relu, **cfg)
        model = PC.load_params(model, c, "cpu")
        assert sum(p.numel() for p in model.parameters()) == int(c["n_params"])
        model.train(bool(int(training)))
        feat = PC.leaf(c.t("feat"))
        logits = model(g, feat)
        PC.fwd_close(logits, c["logits"])
        (logits * c.t("gout")).sum().backward()
        PC.grad_close(feat.grad, c["dfeat"])
        for k, p in model.named_parameters():
            PC.grad_close(p.grad, c[f"g.{k}"])
        # the same state_dict drives bot_amd's own modules: identical keys, identical logits
        mine = PC.build_stack(kind, cfg)
        mine.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in c.sub("p.").items()}, strict=True)
        mine.train(bool(int(training)))
        with torch.no_grad() if not bool(int(training)) else torch.enable_grad():
            PC.fwd_close(mine(g, c.t("feat")), logits.detach().numpy())
        n += 1
    assert n == 20
