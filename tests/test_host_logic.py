"""CPU suite for the host logic above the C ABI: graph structures, transforms, autograd wrappers and
layer modules, run over the emulated backend (tests/_oracle_backend.py) against the golden vectors."""
import pytest
import torch

import bot_amd
from bot_amd import _C
from bot_amd import nn as bnn
from tests import _oracle_backend, parity_cases as PC


@pytest.fixture()
def cpu_backend(monkeypatch):
    _oracle_backend.install(monkeypatch)


def test_product_refuses_cpu_tensors(golden):
    """No CPU fallback: the real wrappers reject CPU tensors instead of computing."""
    s, d, n = golden.graph("g64")
    g = bot_amd.Graph(s, d, n)
    with pytest.raises(_C.BotKernelError):
        g.in_degrees()
    with pytest.raises(_C.BotKernelError):
        bot_amd.ops.copy_u_sum(g, torch.randn(n, 4))


def test_graph_structures(golden, cpu_backend):
    PC.check_graph_structures(golden, "cpu")


def test_preprocess_bit_exact(golden, cpu_backend):
    PC.check_preprocess(golden, "cpu")


def test_ops_against_oracle(golden, cpu_backend):
    PC.check_ops_against_oracle(golden, "cpu", gname="g64")


def test_graphconv_golden(golden, cpu_backend):
    PC.check_graphconv_golden(golden, "cpu")


def test_gatconv_golden(golden, cpu_backend):
    PC.check_gatconv_golden(golden, "cpu")


def test_dgl_surface(golden, cpu_backend):
    PC.check_dgl_surface_matches_fused(golden, "cpu")


@pytest.mark.parametrize("fuse", [False, True])
def test_stacks_golden(golden, cpu_backend, monkeypatch, fuse):
    from bot_amd.nn import fused
    monkeypatch.setattr(fused, "FORCE", True)
    PC.check_stacks_golden(golden, "cpu", fuse=fuse)


def test_state_dict_keys_and_param_counts():
    """Same state_dict keys / parameter counts as the reference records (run.py:1009, :828)."""
    import torch.nn.functional as F
    gat = bnn.GAT(dim_node=168, dim_edge=0, dim_output=40, n_hidden=250, n_layers=3, n_heads=3, activation=F.relu,
                  norm="batch", dropout=0.75, input_drop=0.25, attn_drop=0.1, linear=True)
    assert sum(p.numel() for p in gat.parameters()) == 1441580
    keys = set(gat.state_dict())
    assert {"convs.0.fc.weight", "convs.0.attn_l", "convs.0.res_fc.weight", "norms.1.running_var", "biases.0.bias"} <= keys
    assert "convs.0.attn_r" not in keys  # non_interactive_attn=False => no attn_r (models.py:444-447)
    gcn = bnn.GCN(in_feats=128, n_classes=40, n_hidden=256, n_layers=3, activation=F.relu, norm="batch", dropout=0.5)
    assert sum(p.numel() for p in gcn.parameters()) == 109608


def test_zero_in_degree_errors(golden, cpu_backend):
    s, d, n = golden.graph("doc_noloop")
    g = bot_amd.Graph(s, d, n)
    with pytest.raises(bot_amd.DGLError):
        bnn.GraphConv(3, 2)(g, torch.ones(n, 3))
    with pytest.raises(AssertionError):
        bnn.GATConv(3, 2)(g, torch.ones(n, 3))
    out = bnn.GraphConv(3, 2, allow_zero_in_degree=True)(g, torch.ones(n, 3))
    assert torch.all(out[5] == 0)  # node 5 has no in-edge: docstring example 2 (models.py:204-209)
    with pytest.raises(bot_amd.DGLError):
        bnn.GraphConv(3, 2, norm="left")


def test_empty_and_isolated(cpu_backend):
    g = bot_amd.Graph(torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), 5)
    assert g.in_degrees().tolist() == [0] * 5
    out = bot_amd.ops.copy_u_sum(g, torch.randn(5, 3))
    assert torch.all(out == 0)


def test_proteins_golden(golden, cpu_backend):
    PC.check_proteins_golden(golden, "cpu")


def test_copy_e_sum_preprocess(golden, cpu_backend):
    PC.check_copy_e_sum_preprocess(golden, "cpu")


def test_train_step_golden(golden, cpu_backend, monkeypatch):
    from bot_amd.nn import fused
    monkeypatch.setattr(fused, "FORCE", True)  # the GAT cases go through the fused hidden-layer node
    c0 = fused.CALLS
    PC.check_train_step_golden(golden, "cpu")
    assert fused.CALLS > c0


def test_agg_first_against_oracle(golden, cpu_backend, monkeypatch):
    from bot_amd.nn import fused
    monkeypatch.setattr(fused, "FORCE", True)
    PC.check_agg_first_against_oracle(golden, "cpu")
