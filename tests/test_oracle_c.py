"""The C restatement (oracle/csrc/oracle_ops.c) agrees with the torch restatement (oracle/ref_ops.py)
and, through GATConv, with the golden vectors from the reference modules."""
import numpy as np
import torch

from oracle import c_ops as C
from oracle import ref_models as RM
from oracle import ref_ops as R


def test_c_ops_match_torch_ops(golden):
    s, d, n = golden.graph("g300")
    g = C.CGraph(s, d, n)
    E = s.numel()
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, 3, 5, generator=gen)
    a = torch.rand(E, 3, 1, generator=gen)
    gout = torch.randn(n, 3, 5, generator=gen)
    for use_a in (False, True):
        xo, ao = x.clone().requires_grad_(), a.clone().requires_grad_()
        ref = R.u_mul_e_sum(s, d, n, xo, ao) if use_a else R.copy_u_sum(s, d, n, xo)
        (ref * gout).sum().backward()
        xc, ac = x.clone().requires_grad_(), a.clone().requires_grad_()
        out = C.u_mul_e_sum(g, xc, ac) if use_a else C.copy_u_sum(g, xc)
        (out * gout).sum().backward()
        assert torch.allclose(out, ref, atol=1e-5)
        assert torch.allclose(xc.grad, xo.grad, atol=1e-5)
        if use_a:
            assert torch.allclose(ac.grad, ao.grad, atol=1e-4)
    e = torch.randn(E, 3, 1, generator=gen) * 3
    ga = torch.randn(E, 3, 1, generator=gen)
    eo, ec = e.clone().requires_grad_(), e.clone().requires_grad_()
    ref, out = R.edge_softmax(d, n, eo), C.edge_softmax(g, ec)
    (ref * ga).sum().backward()
    (out * ga).sum().backward()
    assert torch.allclose(out, ref, atol=1e-6) and torch.allclose(ec.grad, eo.grad, atol=1e-5)
    el, er = torch.randn(n, 3, 1, generator=gen), torch.randn(n, 3, 1, generator=gen)
    lo, ro, lc, rc = (t.clone().requires_grad_() for t in (el, er, el, er))
    ref, out = R.u_add_v(s, d, lo, ro), C.u_add_v(g, lc, rc)
    (ref * ga).sum().backward()
    (out * ga).sum().backward()
    assert torch.equal(out, ref)
    assert torch.allclose(lc.grad, lo.grad, atol=1e-5) and torch.allclose(rc.grad, ro.grad, atol=1e-5)


def test_c_gatconv_matches_reference_golden(golden):
    n_checked = 0
    for c in golden.cases("gatconv"):
        gname, symm, attn_r, linear, H, D, fin, edge_drop, dt = (str(x) for x in c["meta"])
        if dt != "float32":
            continue
        s, d, n = golden.graph(gname)
        p = c.params()
        feat = c.t("feat").requires_grad_()
        rst = RM.gatconv_forward(C.CGraph(s, d, n), feat, p["fc.weight"], p["attn_l"], p.get("attn_r"), p.get("res_fc.weight"),
                                num_heads=int(H), out_feats=int(D), use_symmetric_norm=bool(int(symm)),
                                 keep_eids=c.t("keep_eids") if "keep_eids" in c else None)
        np.testing.assert_allclose(rst.detach().numpy(), c["rst"], rtol=1e-4, atol=1e-5)
        (rst * c.t("gout")).sum().backward()
        np.testing.assert_allclose(feat.grad.numpy(), c["dfeat"], rtol=1e-3, atol=1e-4)
        n_checked += 1
    assert n_checked > 10
