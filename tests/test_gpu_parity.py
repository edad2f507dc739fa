"""GPU parity suite: the HIP kernels, called through the C ABI, against the golden vectors and the
oracle.  Run with `-m gpu` on an MI355X."""
import os

import pytest
import torch

import bot_amd
from bot_amd import _C, ops
from oracle import ref_ops as R
from tests import parity_cases as PC

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_native_library_is_loaded():
    assert torch.cuda.is_available()
    assert _C._lib.bot_abi_version() == _C.ABI_VERSION == 19
    maps = open("/proc/self/maps").read()
    assert "libbot_gnn.so" in maps


def test_graph_structures(golden):
    PC.check_graph_structures(golden, DEV)


def test_preprocess_bit_exact(golden):
    PC.check_preprocess(golden, DEV)


@pytest.mark.parametrize("gname", ["g64", "g300"])
def test_ops_against_oracle(golden, gname):
    PC.check_ops_against_oracle(golden, DEV, gname=gname)


def test_graphconv_golden(golden):
    PC.check_graphconv_golden(golden, DEV)


def test_gatconv_golden(golden):
    PC.check_gatconv_golden(golden, DEV)


def test_dgl_surface(golden):
    PC.check_dgl_surface_matches_fused(golden, DEV)


@pytest.mark.parametrize("fuse", [False, True])
def test_stacks_golden(golden, fuse):
    PC.check_stacks_golden(golden, DEV, fuse=fuse)


def test_determinism(golden):
    """No float atomics: two launches on the same inputs are bitwise identical."""
    s, d, n = golden.graph("g300")
    g = bot_amd.Graph(s, d, n, chunk=8).to(DEV)
    x = torch.randn(n, 3, 250, device=DEV)
    a = torch.rand(s.numel(), 3, 1, device=DEV)
    o1, o2 = ops.u_mul_e_sum(g, x, a), ops.u_mul_e_sum(g, x, a)
    assert torch.equal(o1, o2)


def _powerlaw(n, e_raw, seed):
    gen = torch.Generator().manual_seed(seed)
    src = (n * torch.rand(e_raw, generator=gen, dtype=torch.float64) ** 2.0).long().clamp_(max=n - 1)
    dst = (n * torch.rand(e_raw, generator=gen, dtype=torch.float64) ** 2.0).long().clamp_(max=n - 1)
    perm = torch.randperm(n, generator=gen)
    return perm[src], perm[dst]


def test_midsize_against_oracle():
    """20k nodes / 300k edges, heavy tail (max degree in the thousands): long-row path at default chunk."""
    n = 20000
    rs, rd = _powerlaw(n, 150000, 5)
    s, d = R.preprocess_edges(rs, rd, n)
    g = bot_amd.Graph(s, d, n).to(DEV)
    assert g.csc.n_long > 0
    gen = torch.Generator().manual_seed(1)
    for H, D in ((3, 250), (1, 128), (1, 40)):
        x = torch.randn(n, H, D, generator=gen)
        el, er = torch.randn(n, H, 1, generator=gen), torch.randn(n, H, 1, generator=gen)
        gout = torch.randn(n, H, D, generator=gen)
        xo, lo, ro = PC.leaf(x), PC.leaf(el), PC.leaf(er)
        e = torch.nn.functional.leaky_relu(R.u_add_v(s, d, lo, ro), 0.2)
        ref = R.u_mul_e_sum(s, d, n, xo, R.edge_softmax(d, n, e))
        (ref * gout).sum().backward()
        xt, lt, rt = PC.leaf(x, DEV), PC.leaf(el, DEV), PC.leaf(er, DEV)
        a = ops.gat_attention(g, lt, rt, negative_slope=0.2, order="csc")
        out = ops.u_mul_e_sum(g, xt, a, order="csc")
        (out * gout.to(DEV)).sum().backward()
        PC.fwd_close(out, ref.detach().numpy(), 1e-4)
        PC.grad_close(xt.grad, xo.grad.numpy())
        PC.grad_close(lt.grad, lo.grad.numpy())
        PC.grad_close(rt.grad, ro.grad.numpy())


def test_dense_graph_attention_rows_against_oracle():
    """Round 6: the edge-sized row kernels (attention forward / backward, segment sum) on a DENSE graph at mid size - a 1 500-node graph
    of mean in-degree ~95 with a heavy tail (short rows of every length up to the long-row threshold, 16 lanes walking up to 8 edges deep,
    and rows beyond it: a path the sparse mid-size test does not reach and the full-size tests reach only inside whole steps), H = 1, 3, 6:
    logits / weights / aggregation and every gradient against the oracle, and the segment sum."""
    n = 1500
    rs, rd = _powerlaw(n, 80000, 11)
    s, d = R.preprocess_edges(rs, rd, n)
    g = bot_amd.Graph(s, d, n, chunk=128).to(DEV)
    assert g.number_of_edges() >= 64 * n and g.csc.n_long > 0 and g.csc.n_long < n // 2
    gen = torch.Generator().manual_seed(2)
    for H, D in ((1, 40), (3, 64), (6, 80)):
        x = torch.randn(n, H, D, generator=gen)
        el, er = torch.randn(n, H, 1, generator=gen), torch.randn(n, H, 1, generator=gen)
        gout = torch.randn(n, H, D, generator=gen)
        xo, lo, ro = PC.leaf(x), PC.leaf(el), PC.leaf(er)
        e = torch.nn.functional.leaky_relu(R.u_add_v(s, d, lo, ro), 0.2)
        ref = R.u_mul_e_sum(s, d, n, xo, R.edge_softmax(d, n, e))
        (ref * gout).sum().backward()
        xt, lt, rt = PC.leaf(x, DEV), PC.leaf(el, DEV), PC.leaf(er, DEV)
        a = ops.gat_attention(g, lt, rt, negative_slope=0.2, order="csc")
        out = ops.u_mul_e_sum(g, xt, a, order="csc")
        (out * gout.to(DEV)).sum().backward()
        PC.fwd_close(out, ref.detach().numpy(), 1e-4)
        PC.grad_close(xt.grad, xo.grad.numpy())
        PC.grad_close(lt.grad, lo.grad.numpy())
        PC.grad_close(rt.grad, ro.grad.numpy())
        # rows of the attention weights sum to one; the segment sum (copy_e_sum) against the oracle
        w = torch.rand(g.number_of_edges(), H, generator=gen)
        ref_s = R.copy_e_sum(d, n, w)
        PC.fwd_close(ops.copy_e_sum(g, w.to(DEV)), ref_s.numpy(), 2e-5 * float(ref_s.abs().max()))


def test_full_size_properties():
    """ogbn-arxiv-shaped graph (BASELINE config 2 size): size-independent properties.
    (1) linearity: spmm(x1 + 2*x2) == spmm(x1) + 2*spmm(x2);  (2) attention rows sum to 1, so aggregating
    a constant feature returns the constant;  (3) copy_u_sum of ones == in-degree (bit-exact integers);
    (4) adjoint identity <A x, y> == <x, A^T y> ties the forward (CSC) and backward (CSR) sweeps."""
    n, e_raw = 169343, 1166243
    rs, rd = _powerlaw(n, e_raw, 20210325)
    g = bot_amd.preprocess(bot_amd.Graph(rs, rd, n).to(DEV))
    E = g.number_of_edges()
    H, D = 3, 250
    x1, x2 = torch.randn(n, H, D, device=DEV), torch.randn(n, H, D, device=DEV)
    el, er = torch.randn(n, H, 1, device=DEV), torch.randn(n, H, 1, device=DEV)
    a = ops.gat_attention(g, el, er, order="csc")
    lhs = ops.u_mul_e_sum(g, x1 + 2 * x2, a, order="csc")
    rhs = ops.u_mul_e_sum(g, x1, a, order="csc") + 2 * ops.u_mul_e_sum(g, x2, a, order="csc")
    assert torch.allclose(lhs, rhs, rtol=1e-4, atol=1e-4)
    ones = torch.ones(n, H, D, device=DEV)
    agg = ops.u_mul_e_sum(g, ones, a, order="csc")
    assert torch.allclose(agg, ones, atol=1e-5)
    deg = ops.copy_u_sum(g, torch.ones(n, 1, device=DEV)).squeeze(1)
    assert torch.equal(deg.long(), g.in_degrees())
    y = torch.randn(n, H, D, device=DEV)
    xr = x1.clone().requires_grad_()
    out = ops.u_mul_e_sum(g, xr, a, order="csc")
    (out * y).sum().backward()
    lhs = (out.detach().double() * y.double()).sum()
    rhs = (x1.double() * xr.grad.double()).sum()
    # scaled by the norms, not by |lhs|: the inner product of random tensors can be arbitrarily close to zero
    assert abs(lhs - rhs) <= 1e-5 * float(out.detach().double().norm() * y.double().norm())
    assert E > 2_000_000


def test_row_gather_scatter():
    gen = torch.Generator().manual_seed(2)
    for F in (750, 40, 7):
        x = torch.randn(1000, F, generator=gen).to(DEV)
        rows = torch.sort(torch.randperm(1000, generator=gen)[:300]).values.to(torch.int32).to(DEV)
        out = _C.gather_rows(x, rows)
        assert torch.equal(out, x[rows.long()])
        vals = torch.randn(300, F, generator=gen).to(DEV)
        ref = x.clone()
        ref[rows.long()] += vals
        _C.scatter_add_rows(x, rows, vals)
        assert torch.equal(x, ref)


def test_block_graphs_reproduce_full_graph(golden):
    """The partitioned mode's local blocks (owned rows + halo sources) on the real kernels: two blocks computed
    side by side in one process, halo rows supplied by indexing, give the full-graph GAT aggregation."""
    from bot_amd import dist as bdist
    s, d, n = golden.graph("g300")
    g = bot_amd.Graph(s, d, n).to(DEV)
    gen = torch.Generator().manual_seed(4)
    H, D = 3, 250
    x = torch.randn(n, H, D, generator=gen).to(DEV)
    el, er = torch.randn(n, H, 1, generator=gen).to(DEV), torch.randn(n, H, 1, generator=gen).to(DEV)
    gout = torch.randn(n, H, D, generator=gen).to(DEV)
    xf, lf, rf = (t.clone().requires_grad_() for t in (x, el, er))
    full = ops.u_mul_e_sum(g, xf, ops.gat_attention(g, lf, rf, order="csc"), order="csc")
    (full * gout).sum().backward()
    dx = torch.zeros_like(x)
    dl, dr = torch.zeros_like(el), torch.zeros_like(er)
    for rank in range(2):
        p = bdist.build_partition(s, d, n, rank, 2, device=DEV)
        glob = torch.cat([torch.arange(p.lo, p.hi, device=DEV), p.halo_global.to(DEV)])
        xe, le = x[glob].clone().requires_grad_(), el[glob].clone().requires_grad_()
        re = er[p.lo:p.hi].clone().requires_grad_()
        out = ops.u_mul_e_sum(p.graph, xe, ops.gat_attention(p.graph, le, re, order="csc"), order="csc")
        assert torch.allclose(out, full[p.lo:p.hi], atol=1e-5)
        (out * gout[p.lo:p.hi]).sum().backward()
        dx.index_add_(0, glob, xe.grad)
        dl.index_add_(0, glob, le.grad)
        dr[p.lo:p.hi] += re.grad
        assert torch.equal(p.graph.in_degrees().cpu(), g.in_degrees().cpu()[p.lo:p.hi])
    assert torch.allclose(dx, xf.grad, atol=1e-4) and torch.allclose(dl, lf.grad, atol=1e-4) and torch.allclose(dr, rf.grad, atol=1e-4)


def test_f4_reorder_integer_invariants_on_device(golden):
    """SURVEY §8 f4 on the real device path: `preprocess(reorder="degree" | "community")` with the integer work done by device-side
    ops, forced and default XCD-aware plans — all integer invariants bit-exact against the oracle's preprocess (run.py:133-148)."""
    PC.check_f4_integer_invariants(golden, DEV)


def test_f4_kernels_on_reordered_graphs_against_oracle(golden):
    """The real kernels over XCD-ordered plans of renumbered graphs, in ORIGINAL id order, against oracle/ref_ops.py, the stacks
    golden (g300) and oracle/ref_models.py (20k-node planted-community graph, fused layer nodes)."""
    PC.check_f4_layers_in_original_order(golden, DEV)


def test_f4_community_partition_blocks_on_device(golden):
    """`partition_dataset(partitioner="community")` blocks of 2 and 3 ranks side by side on the real kernels vs the oracle."""
    PC.check_f4_community_partition_blocks(golden, DEV)


def test_halo_split_sweeps_on_device(golden):
    """The split structures and two-pass sweeps of the overlapped partitioned layer (bot_amd.nn.fused OVERLAP, Graph.halo_split) on the
    real kernels: blocks of 2- and 3-way partitions side by side in one process."""
    PC.check_halo_split_sweeps(golden, DEV)


def test_absmax_byproducts_on_device(golden):
    """max|value| delivered by the kernels that write the gradient buffer (fused backward sweep incl. long rows and the head-major
    fall-back, BatchNorm backward) is exact, and a fused stack's gradients are bitwise unchanged by it."""
    PC.check_absmax_byproducts(golden, DEV)


def test_halves_only_hidden_states_on_device(golden):
    """Hidden states that only a halves GEMM reads are stored as halves only: bitwise the results of the stack that stores them."""
    PC.check_halves_only_hidden_states(golden, DEV)


def test_merged_linear_blocks_on_device(golden):
    """The edge-feature GATConvs' merged projection on the halves path: column blocks out, gradients split in place into one operand."""
    PC.check_merged_linear_blocks(golden, DEV)


def test_halo_sums_on_device(golden):
    """bot_amd.halo's overlapped aggregations (what the modular layers call in partitioned mode) on the real kernels, the exchange
    replaced by indexing: forward, all gradients and the returned halo-row gradients against the one-exchange form."""
    PC.check_halo_sums(golden, DEV)


def test_proteins_golden(golden):
    PC.check_proteins_golden(golden, DEV)


def test_products_golden(golden):
    PC.check_products_golden(golden, DEV)


def test_copy_e_sum_preprocess(golden):
    PC.check_copy_e_sum_preprocess(golden, DEV)


def test_train_step_golden(golden):
    PC.check_train_step_golden(golden, DEV)


@pytest.mark.parametrize("F", [750, 256, 41])
def test_fused_bn_relu_dropout_matches_torch(F):
    """BatchNorm1d (batch statistics over the node axis) + ReLU, fused, against torch's own modules: output,
    input/affine gradients, running statistics; training and eval mode; also on a strided (row-padded) input."""
    n = 5000
    gen = torch.Generator().manual_seed(F)
    x0 = (torch.randn(n, F, generator=gen) * 1.7 + 0.4).to(DEV)
    gy = torch.randn(n, F, generator=gen).to(DEV)
    for training in (True, False):
        for strided in (False, True):
            ref_bn, bn = torch.nn.BatchNorm1d(F).to(DEV), torch.nn.BatchNorm1d(F).to(DEV)
            with torch.no_grad():
                for m in (ref_bn, bn):
                    m.weight.copy_(torch.linspace(0.5, 1.5, F))
                    m.bias.copy_(torch.linspace(-0.3, 0.3, F))
                    m.running_mean.fill_(0.1)
                    m.running_var.fill_(1.3)
            ref_bn.train(training), bn.train(training)
            xr = x0.clone().requires_grad_()
            if strided:
                buf = torch.zeros(n, F + 6, device=DEV)
                buf[:, :F] = x0
                xt = buf[:, :F].detach().requires_grad_()
            else:
                xt = x0.clone().requires_grad_()
            ref = torch.relu(ref_bn(xr))
            out = ops.bn_relu_dropout(xt, bn, relu=True, p=0.5, training=False)  # dropout off: comparable
            assert torch.allclose(out, ref, atol=2e-5, rtol=1e-5)
            (ref * gy).sum().backward()
            (out * gy).sum().backward()
            assert torch.allclose(xt.grad, xr.grad, atol=2e-5, rtol=1e-4)
            assert torch.allclose(bn.weight.grad, ref_bn.weight.grad, rtol=1e-4, atol=1e-3)
            assert torch.allclose(bn.bias.grad, ref_bn.bias.grad, rtol=1e-4, atol=1e-3)
            assert torch.allclose(bn.running_mean, ref_bn.running_mean, atol=1e-6)
            assert torch.allclose(bn.running_var, ref_bn.running_var, rtol=1e-5)
            assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked)


def test_fused_dropout_mask_consistency():
    """Dropout inside the fused epilogue: keep rate ~ 1-p, kept entries scaled by 1/(1-p), and the backward
    regenerates exactly the forward mask (no mask is stored)."""
    n, F, p = 20000, 750, 0.75
    x = torch.randn(n, F, device=DEV).requires_grad_()
    bn = torch.nn.BatchNorm1d(F).to(DEV).train()
    y = ops.bn_relu_dropout(x, bn, relu=True, p=p, training=True)
    with torch.no_grad():
        base = torch.relu(torch.nn.functional.batch_norm(x, None, None, bn.weight, bn.bias, True, 0.0, bn.eps))
    pos = base > 0
    kept = (y != 0) & pos
    rate = kept.sum().item() / pos.sum().item()
    assert abs(rate - (1 - p)) < 5e-3, rate
    assert torch.allclose(y[kept], base[kept] / (1 - p), rtol=1e-4, atol=1e-5)
    factor = torch.where(kept, torch.full_like(y, 1 / (1 - p)), torch.zeros_like(y))  # the mask the forward used
    gy = torch.randn_like(y)
    y.backward(gy)
    xr = x.detach().clone().requires_grad_()
    ref = torch.relu(torch.nn.functional.batch_norm(xr, None, None, bn.weight, bn.bias, True, 0.0, bn.eps)) * factor
    ref.backward(gy)
    assert torch.allclose(x.grad, xr.grad, atol=1e-4, rtol=1e-3)
    y2 = ops.bn_relu_dropout(x.detach(), bn, relu=True, p=p, training=True)
    assert not torch.equal(y2 != 0, y != 0)  # a fresh seed per call


def test_fused_dropout_mask_independent_of_alignment():
    """The Philox mask of element (r, c) must not depend on the vector width a launch picks (forward on 16-byte
    aligned x/y, backward with dy / dx that are only 8- or 4-byte aligned used to regenerate ANOTHER mask).  The forward runs
    at VEC=4; the backward kernels are driven through the raw wrappers with dy and dx shifted by 2 and by 1 floats."""
    from bot_amd import _C
    n, F, p, seed = 3000, 752, 0.6, 0x1234567890ABCDEF
    x = torch.randn(n, F, device=DEV)
    mean, m2 = _C.colstats(x)
    invstd = torch.rsqrt(m2 / n + 1e-5)
    w, b = torch.rand(F, device=DEV) + 0.5, torch.randn(F, device=DEV) * 0.1
    y = _C.bn_act_fwd(x, mean, invstd, w, b, True, p, seed)
    dy0 = torch.randn(n, F, device=DEV)
    sg0, sgx0 = _C.bn_act_bwd_reduce(dy0, x, mean, invstd, w, b, True, p, seed)
    dx0 = _C.bn_act_bwd_apply(dy0, x, mean, invstd, w, b, True, p, seed, sg0, sgx0, float(n))
    # the mask the forward used, recovered from its output
    kept = (y != 0)
    pre = torch.relu((x - mean) * invstd * w + b)
    assert torch.allclose(y[kept], (pre / (1 - p))[kept], rtol=1e-4, atol=1e-5)
    g_ref = torch.where(kept, dy0 / (1 - p), torch.zeros_like(dy0))
    assert torch.allclose(sg0, g_ref.sum(0), rtol=1e-4, atol=1e-2)
    for shift in (2, 1):  # 8-byte, then 4-byte aligned rows (row stride F + 4 keeps 16 bytes, the base pointer does not)
        buf = torch.zeros(n, F + 4, device=DEV)
        dy = buf[:, shift:shift + F]
        dy.copy_(dy0)
        sg, sgx = _C.bn_act_bwd_reduce(dy, x, mean, invstd, w, b, True, p, seed)
        assert torch.allclose(sg, sg0, rtol=1e-5, atol=1e-4) and torch.allclose(sgx, sgx0, rtol=1e-5, atol=1e-4)
        obuf = torch.zeros(n, F + 4, device=DEV)
        dx = _C.bn_act_bwd_apply(dy, x, mean, invstd, w, b, True, p, seed, sg0, sgx0, float(n), out=obuf[:, shift:shift + F])
        assert torch.equal(dx, dx0)
        xb = torch.zeros(n, F + 4, device=DEV)   # and a forward whose INPUT is misaligned writes the same mask
        xs = xb[:, shift:shift + F]
        xs.copy_(x)
        assert torch.equal(_C.bn_act_fwd(xs, mean, invstd, w, b, True, p, seed) != 0, kept)


def test_sddmm_dot_and_fused_backward_direct(golden):
    """The standalone SDDMM-dot kernel and the fused spmm_dot kernel against plain torch indexing."""
    s, d, n = golden.graph("g300")
    g = bot_amd.Graph(s, d, n, chunk=8).to(DEV)
    E = s.numel()
    gen = torch.Generator().manual_seed(6)
    # "rows" = the one-wave-per-item / all-heads layout of spmm_dot (the default where the shape fits), "heads" = the
    # head-major kernel: segments of 16 / 32 / 64 lanes per head, 1..4 chunks, heads of two chunks (D=250), odd head counts, vec 4 / 2 / 1
    cases = [((3, 250), "heads"), ((1, 40), "heads"), ((2, 7), "heads"), ((1, 1100), "heads")]
    cases += [(hd, "rows") for hd in ((4, 120), (6, 80), (2, 64), (3, 40), (8, 32), (5, 36), (2, 250), (3, 250), (3, 125), (3, 126), (2, 16))]
    for (H, D), layout in cases:
        os.environ["BOT_SPMM_DOT_LAYOUT"] = layout
        x = torch.randn(n, H, D, generator=gen).to(DEV)
        y = torch.randn(n, H, D, generator=gen).to(DEV)
        a = torch.rand(E, H, generator=gen).to(DEV)
        csc, csr = g.csc, g.csr
        rows = torch.repeat_interleave(torch.arange(n, device=DEV), (csc.indptr[1:] - csc.indptr[:-1]).long())
        ref = (x[csc.indices.long()] * y[rows]).sum(-1)
        assert torch.allclose(_C.sddmm_dot(csc, x, y), ref, atol=1e-3 * D ** 0.5, rtol=1e-4)
        if D <= _C.spmm_dot_max_d(x):
            out, dot = _C.spmm_dot(csr, x, a, g.csr2csc, y)
            if layout == "rows":  # and the strided-slab form the fused layer uses (row pitch > H*D)
                big = torch.randn(n, H * D + 8, generator=gen).to(DEV)
                xs = big[:, 4:4 + H * D].view(n, H, D)
                o2, d2 = _C.spmm_dot(csr, xs, a, g.csr2csc, y)
                o3, d3 = _C.spmm_dot(csr, xs.contiguous(), a, g.csr2csc, y)
                assert torch.equal(o2, o3) and torch.equal(d2, d3)
            rows_r = torch.repeat_interleave(torch.arange(n, device=DEV), (csr.indptr[1:] - csr.indptr[:-1]).long())
            w = a[g.csr2csc.long()]
            ref_out = torch.zeros(n, H, D, device=DEV).index_add_(0, rows_r, x[csr.indices.long()] * w.unsqueeze(-1))
            ref_dot = torch.empty(E, H, device=DEV)
            ref_dot[g.csr2csc.long()] = (x[csr.indices.long()] * y[rows_r]).sum(-1)
            assert torch.allclose(out, ref_out, atol=1e-4, rtol=1e-4)
            assert torch.allclose(dot, ref_dot, atol=1e-3 * D ** 0.5, rtol=1e-4)
    os.environ.pop("BOT_SPMM_DOT_LAYOUT", None)


@pytest.mark.parametrize("H", [1, 3, 6, 8])
def test_edge_mlp_kernels(H):
    """Fused 8->16->H edge MLP (ogbn-proteins) forward and its MFMA-reduced weight gradients vs torch autograd."""
    gen = torch.Generator().manual_seed(H)
    for E in (1, 5, 1000, 200003):
        ef = torch.rand(E, 8, generator=gen).to(DEV)
        W1 = (torch.randn(16, 8, generator=gen) * 0.5).to(DEV).requires_grad_()
        b1 = (torch.randn(16, generator=gen) * 0.3).to(DEV).requires_grad_()
        W2 = (torch.randn(H, 16, generator=gen) * 0.5).to(DEV).requires_grad_()
        dz = torch.randn(E, H, generator=gen).to(DEV)
        ref = torch.relu(ef @ W1.t() + b1) @ W2.t()
        out = _C.edge_mlp_fwd(ef, W1.detach(), b1.detach(), W2.detach())
        assert torch.allclose(out, ref, atol=1e-5, rtol=1e-5)
        ref.backward(dz)
        dW1, db1, dW2 = _C.edge_mlp_bwd(ef, W1.detach(), b1.detach(), W2.detach(), dz)
        scale = max(1.0, E ** 0.5)
        assert torch.allclose(dW1, W1.grad, atol=2e-5 * scale, rtol=1e-4)
        assert torch.allclose(db1, b1.grad, atol=2e-5 * scale, rtol=1e-4)
        assert torch.allclose(dW2, W2.grad, atol=2e-5 * scale, rtol=1e-4)


def test_edge_cases():
    """Empty graphs, isolated nodes, single node, degree == chunk and chunk + 1, many heads, D = 1, D beyond one launch tile."""
    # no edges at all
    g = bot_amd.Graph(torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), 7).to(DEV)
    assert g.in_degrees().tolist() == [0] * 7 and g.out_degrees().tolist() == [0] * 7
    x = torch.randn(7, 2, 5, device=DEV, requires_grad=True)
    out = ops.copy_u_sum(g, x)
    assert torch.all(out == 0)
    out.sum().backward()
    assert torch.all(x.grad == 0)
    a = ops.gat_attention(g, torch.randn(7, 2, 1, device=DEV), torch.randn(7, 2, 1, device=DEV), order="csc")
    assert a.shape == (0, 2, 1)
    assert torch.all(ops.u_mul_e_sum(g, x, a, order="csc") == 0)
    # single node with a self loop
    g1 = bot_amd.Graph(torch.tensor([0]), torch.tensor([0]), 1).to(DEV)
    x1 = torch.randn(1, 3, 4, device=DEV)
    a1 = ops.gat_attention(g1, torch.randn(1, 3, 1, device=DEV), None, order="csc")
    assert torch.allclose(a1, torch.ones_like(a1)) and torch.allclose(ops.u_mul_e_sum(g1, x1, a1, order="csc"), x1)
    # star + isolated nodes, degrees straddling the chunk size
    gen = torch.Generator().manual_seed(3)
    chunk = 16
    for hub_deg in (chunk - 1, chunk, chunk + 1, 5 * chunk + 3):
        n = hub_deg + 5
        src = torch.arange(1, hub_deg + 1)
        dst = torch.zeros(hub_deg, dtype=torch.int64)
        g = bot_amd.Graph(src, dst, n, chunk=chunk).to(DEV)
        for H, D in ((8, 3), (1, 1), (2, 1100)):
            x = torch.randn(n, H, D, generator=gen)
            el = torch.randn(n, H, 1, generator=gen)
            gout = torch.randn(n, H, D, generator=gen)
            xo, lo = PC.leaf(x), PC.leaf(el)
            e = torch.nn.functional.leaky_relu(R.copy_u(src, lo), 0.2)
            ref = R.u_mul_e_sum(src, dst, n, xo, R.edge_softmax(dst, n, e))
            (ref * gout).sum().backward()
            xt, lt = PC.leaf(x, DEV), PC.leaf(el, DEV)
            out = ops.u_mul_e_sum(g, xt, ops.gat_attention(g, lt, None, order="csc"), order="csc")
            (out * gout.to(DEV)).sum().backward()
            PC.fwd_close(out, ref.detach().numpy(), 1e-5)
            PC.grad_close(xt.grad, xo.grad.numpy())
            PC.grad_close(lt.grad, lo.grad.numpy())
            assert torch.all(out[1:] == 0)  # nodes without in-edges aggregate nothing
            if D <= 1024:  # the inference-only sweep on the same star (hub row chunked, everyone else without in-edges)
                inf = _C.gat_infer(g.csc, xt.detach(), lt.detach().squeeze(-1))
                PC.fwd_close(inf, ref.detach().numpy(), 1e-5)
            if H <= 4 and D <= 1024:  # first-layer weight gradient: <x[u], y[v,h,:]> per in-edge
                xs = torch.randn(n, D, generator=gen).to(DEV)
                ys = torch.randn(H, n, D, generator=gen).to(DEV)
                dots = _C.sddmm_dot_bcast(g.csc, xs, ys)
                refd = (xs[g.csc.indices.long()].unsqueeze(1) * ys[:, 0, :].unsqueeze(0)).sum(-1)   # every edge ends at node 0
                assert torch.allclose(dots, refd, rtol=1e-4, atol=1e-3 * D ** 0.5)
    # the new entry points on empty problems
    g = bot_amd.Graph(torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), 7).to(DEV)
    ad = torch.randn(7, 2, 5, device=DEV)
    o = _C.gat_infer(g.csc, torch.randn(7, 2, 5, device=DEV), torch.randn(7, 2, device=DEV), addend=ad, relu=True)
    assert torch.equal(o, torch.relu(ad))
    assert _C.sddmm_dot_bcast(g.csc, torch.randn(7, 5, device=DEV), torch.randn(2, 7, 5, device=DEV)).shape == (0, 2)


def test_midsize_edge_gat_stack_against_oracle():
    """The edge-feature GAT of configs 4 / 5 (ogbn-proteins/models.py, ogbn-products/models.py) as a whole stack on a 20 k-node
    power-law graph with hubs — fused edge MLP (fp32 MFMA weight-gradient kernel), merged input GEMM, residual in the SpMM
    epilogue, BatchNorm kernels — forward logits and every parameter gradient against the oracle's C kernels at the HIP run's
    ReLU / leaky-ReLU gates, ranked against the same step in fp64 (the criterion of the full-size config-4 test)."""
    import torch.nn.functional as F
    from bot_amd.nn import edge_gat
    from tests import full_size as FS
    n = 20000
    rs, rd = _powerlaw(n, 150000, 9)
    s, d = R.preprocess_edges(rs, rd, n)
    E = s.numel()
    gen = torch.Generator().manual_seed(2)
    nfeat, efeat = torch.randn(n, 8, generator=gen), torch.rand(E, 8, generator=gen)
    gout = torch.randn(n, 12, generator=gen)
    idx = torch.arange(n)
    loss = lambda x, y: (x * y).sum(1)
    for kind in ("proteins", "products"):
        torch.manual_seed(4)
        if kind == "proteins":
            model = edge_gat.ProteinsGAT(node_feats=8, edge_feats=8, n_classes=12, n_layers=3, n_heads=6, n_hidden=80, edge_emb=16,
                                         activation=F.relu, dropout=0.0, input_drop=0.0, attn_drop=0.0, edge_drop=0.0)
        else:
            model = edge_gat.ProductsGAT(node_feats=8, edge_feats=0, n_classes=12, n_layers=3, n_heads=4, n_hidden=120, edge_emb=0,
                                         activation=F.relu, dropout=0.0, input_drop=0.0, attn_drop=0.0, edge_drop=0.0)
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        g = bot_amd.Graph(s, d, n).to(DEV)
        g.ndata["feat"] = nfeat.to(DEV)
        if kind == "proteins":
            g.edata["feat"] = efeat.to(DEV)
        pred, grads, gates = FS.edge_gat_hip_step(model.to(DEV), g, gout.to(DEV), idx.to(DEV), loss)
        kw = dict(n_layers=3, n_heads=6 if kind == "proteins" else 4, n_hidden=80 if kind == "proteins" else 120, node_loss=loss,
                  use_node_encoder=kind == "proteins", residual=kind == "proteins", gates=gates)
        args = (s, d, n, nfeat, efeat if kind == "proteins" else None, gout, idx, sd)
        rp, rg, _, gstats = FS.edge_gat_oracle_step(*args, **kw)
        xp, xg, _, _ = FS.edge_gat_oracle_step(*args, dtype=torch.float64, **kw)
        assert set(rg) == set(xg) and set(rg) <= set(grads)
        assert max(st["max_abs_preact_where_differ"] for st in gstats) <= 1e-4
        PC.fwd_close(pred, rp.numpy(), 2e-4)
        assert float((pred.cpu().double() - xp).abs().max()) <= 2e-4
        zero = {f"convs.{i}.dst_fc.bias": f"convs.{i}.dst_fc.weight" for i in range(3)}
        for k, (eh, eo) in FS.rank_against_exact(grads, rg, xg, zero).items():
            assert eh <= max(PC.GRAD_RTOL, 2 * eo), (kind, k, eh, eo)


@pytest.mark.parametrize("l0_halves", [False, True])
def test_agg_first_against_oracle(golden, l0_halves):
    PC.check_agg_first_against_oracle(golden, DEV, l0_halves=l0_halves)


def test_dout_direct_against_oracle(golden):
    PC.check_dout_direct_against_oracle(golden, DEV)


@pytest.mark.parametrize("drop", [0.0, 0.5])
def test_bn_bwd_byproduct_against_oracle(golden, drop):
    PC.check_bn_bwd_byproduct_against_oracle(golden, DEV, drop=drop)


def test_abi18_bn_bwd_partials_from_the_nt_product():
    """bot_gemm_halves3_nt3_f32: the BatchNorm-backward reduce pass as a by-product of the product that writes dy.  Against the pass over the
    same dy (bot_bn_act_bwd_reduce_max_f32): the column maxima of |g| and |xhat| (through the bound) exactly, the sums to fp32 summation
    noise; dy itself bit for bit the product without the by-product; with and without dropout (the forward's Philox mask of (seed, r, c)) /
    ReLU / affine, rows of 750 floats at pitch 750 (8-byte quads) and 752, ragged m, row-major and fragment-major weights, two scales."""
    from bot_amd import gemm
    gen = torch.Generator(device=DEV).manual_seed(51)
    for (m, K, F, ldx, p, relu, affine) in ((1000, 96, 300, 300, 0.0, True, True), (5000, 1536, 750, 752, 0.75, True, True),
                                            (4099, 128, 750, 750, 0.5, True, False), (257, 64, 192, 192, 0.0, False, True),
                                            (20000, 1536, 750, 752, 0.75, True, True)):
        d = torch.randn(m, K, device=DEV, generator=gen) * 2
        w = torch.randn(F, K, device=DEV, generator=gen) * 0.1
        xbuf = torch.randn(m, ldx, device=DEV, generator=gen)
        x = xbuf[:, :F]
        mean, var = x.mean(0), x.var(0, unbiased=False)
        invstd = (var + 1e-5).rsqrt()
        bw = torch.randn(F, device=DEV, generator=gen) if affine else None
        bb = torch.randn(F, device=DEV, generator=gen) * 0.3 if affine else None
        seed = 123456789
        ws = gemm.split(w, 1)
        piece = ws.piece
        frag = _C.halves_split_frag(w, ws.scale, piece)
        sc = _C.halves_scale(d)
        db = _C.halves_split(d, sc, 2, piece)
        for b_frag in (False, True):
            for two in (False, True):
                kw = dict(a2_off=piece, b_frag=b_frag, n=F)
                if two:
                    if piece < 128:
                        continue
                    kw.update(scale_a2=torch.tensor([float(sc[0]) * 64, float(sc[1]) / 64], device=DEV), k_split=64)
                B = frag if b_frag else ws.buf
                plain = _C.gemm_halves3_nt(db, B, sc, ws.scale, piece, piece, piece, **kw)
                st = _C.BnBwdStats(x, mean, invstd, bw, bb, relu, p, seed)
                assert st.fits(m, F, piece)
                dy = _C.gemm_halves3_nt(db, B, sc, ws.scale, piece, piece, piece, bn=st, **kw)
                assert torch.equal(dy, plain), (m, K, F, b_frag, two)
                sg, sgx = st.sums()
                rg, rgx, rws = _C.bn_act_bwd_reduce(dy, x, mean, invstd, bw, bb, relu, p, seed, want_max=True)
                # (scale: a column's sum of |g| - the two forms add the same terms in different orders)
                for a, b in ((sg, rg), (sgx, rgx)):
                    tol = 2e-6 * float(dy.abs().sum(0).max()) * max(1.0, float(((x - mean) * invstd).abs().max()))
                    assert float((a - b).abs().max()) <= tol, (m, K, F, float((a - b).abs().max()), tol)
                s1, s2 = _C.absmax_slots(DEV), _C.absmax_slots(DEV)
                st.bound(sg, sgx, m, s1)
                _C.bn_bwd_bound(rws, m, sg, sgx, m, bw, invstd, s2)
                assert torch.equal(s1, s2) or float(_C.halves_scale_from_slots(s1)[0]) == float(_C.halves_scale_from_slots(s2)[0]), (m, K, F)
                s3 = _C.absmax_slots(DEV)               # both second stages in one launch (bot_bn_bwd_partials_finish_f32): the same sums, the same bound
                fg, fgx = st.finish(True, m, s3)
                assert torch.equal(fg, sg) and torch.equal(fgx, sgx) and torch.equal(s3, s1), (m, K, F)
                s4, s5 = _C.absmax_slots(DEV), _C.absmax_slots(DEV)
                st.finish(False, m, s4)
                st.bound(None, None, m, s5)
                assert torch.equal(s4, s5)
                if p > 0:        # the mask really is applied: the sums differ from the unmasked ones
                    st0 = _C.BnBwdStats(x, mean, invstd, bw, bb, relu, 0.0, seed)
                    _C.gemm_halves3_nt(db, B, sc, ws.scale, piece, piece, piece, bn=st0, **kw)
                    assert float((st0.sums()[0] - sg).abs().max()) > 1e-3 * float(sg.abs().max())
    # a product that cannot carry it is refused loudly (odd width), and the switch-off shape query
    st = _C.BnBwdStats(torch.randn(300, 33, device=DEV), torch.zeros(33, device=DEV), torch.ones(33, device=DEV), None, None, True, 0.0, 0)
    assert not st.fits(300, 33, 64)
    assert _C._lib.bot_gemm_halves3_nt_bn_rows(96) == 0 and _C._lib.bot_gemm_halves3_nt_bn_rows(128) == 256


@pytest.mark.isolated
def test_absmax_slot_sets_are_zero_and_disjoint():
    """_C.absmax_slots hands out sets of a block zeroed once per 512 requests: every set is zeros, no two sets overlap (also across the block
    boundary), a producer's maximum lands in its own set only, and inside a hipGraph capture each request is its own (captured) fill."""
    n = int(_C._lib.bot_absmax_slots())
    sets = [_C.absmax_slots(DEV) for _ in range(600)]
    ptrs = sorted(s.data_ptr() for s in sets)
    assert all(b - a >= 4 * n for a, b in zip(ptrs, ptrs[1:]))
    assert all(int(s.abs().sum()) == 0 and s.shape == (n,) and s.dtype == torch.int32 for s in sets)
    x = torch.randn(1000, 40, device=DEV)
    _C.absmax_into(x, sets[17])
    assert int((sets[17] != 0).sum()) > 0 and all(int(s.abs().sum()) == 0 for i, s in enumerate(sets) if i != 17)
    assert float(_C.halves_scale_from_slots(sets[17])[0]) == float(_C.halves_scale(x)[0])
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            s = _C.absmax_slots(DEV)
            _C.absmax_into(x, s)
            sc = _C.halves_scale_from_slots(s)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert float(sc[0]) == float(_C.halves_scale(x)[0])


def test_abi18_random_shapes():
    """Randomised differential test of the by-product epilogue: 20 random (m, K, F even, pitch of x, dropout, ReLU, affine, weight layout,
    second scale) - dy bit for bit the plain product, the column maxima (through the bound's slots) exactly and the sums to summation noise
    against the reduce pass over the same dy, nothing outside the partial buffers' [blocks, 2, F] touched."""
    import random
    from bot_amd import gemm
    rng = random.Random(18)
    gen = torch.Generator(device=DEV).manual_seed(71)
    for case in range(20):
        m = rng.choice([1, 17, 255, 256, 257, 1000, 5001, 12345])
        K = rng.choice([33, 64, 100, 250, 500, 750])
        F = rng.choice([2, 6, 40, 190, 192, 250, 256, 258, 750])
        ldx = F + rng.choice([0, 2, 4]) if F % 4 else F + rng.choice([0, 4])
        p = rng.choice([0.0, 0.0, 0.3, 0.75])
        relu, affine, b_frag = rng.random() < 0.7, rng.random() < 0.6, rng.random() < 0.5
        d = torch.randn(m, K, device=DEV, generator=gen) * 10 ** rng.uniform(-2, 2)
        w = torch.randn(F, K, device=DEV, generator=gen) * 10 ** rng.uniform(-2, 0)
        x = torch.randn(m, ldx, device=DEV, generator=gen)[:, :F] * 3 + 1
        mean = x.mean(0)
        invstd = (x.var(0, unbiased=False) + 1e-5).rsqrt() if m > 1 else torch.ones(F, device=DEV)
        bw = torch.randn(F, device=DEV, generator=gen) if affine else None
        bb = torch.randn(F, device=DEV, generator=gen) * 0.3 if affine else None
        ws = gemm.split(w, 1)
        piece = ws.piece
        B = _C.halves_split_frag(w, ws.scale, piece) if b_frag else ws.buf
        sc = _C.halves_scale(d)
        db = _C.halves_split(d, sc, 2, piece)
        kw = dict(a2_off=piece, b_frag=b_frag, n=F)
        if piece >= 128 and rng.random() < 0.4:
            kw.update(scale_a2=torch.tensor([float(sc[0]) * 16, float(sc[1]) / 16], device=DEV), k_split=rng.choice([32, 64, piece - 32]))
        plain = _C.gemm_halves3_nt(db, B, sc, ws.scale, piece, piece, piece, **kw)
        st = _C.BnBwdStats(x, mean, invstd, bw, bb, relu, p, 4242 + case)
        assert st.fits(m, F, piece), (case, m, K, F)
        guard = torch.full((st.nblk + 2, 2, F), 7.0, device=DEV)
        st.part, st.pmax = guard[1:-1], torch.full((st.nblk, 2, F), -1.0, device=DEV)
        dy = _C.gemm_halves3_nt(db, B, sc, ws.scale, piece, piece, piece, bn=st, **kw)
        assert torch.equal(dy, plain), (case, m, K, F)
        assert bool((guard[0] == 7.0).all()) and bool((guard[-1] == 7.0).all()) and bool((st.pmax >= 0).all())
        s1, s2 = _C.absmax_slots(DEV), _C.absmax_slots(DEV)
        sg, sgx = st.finish(True, m, s1)
        rg, rgx, rws = _C.bn_act_bwd_reduce(dy, x, mean, invstd, bw, bb, relu, p, 4242 + case, want_max=True)
        xh_max = float(((x - mean) * invstd).abs().max())
        tol = 3e-6 * float(dy.abs().sum(0).max()) * max(1.0, xh_max) + 1e-30
        assert float((sg - rg).abs().max()) <= tol and float((sgx - rgx).abs().max()) <= tol, (case, m, K, F, p, relu)
        _C.bn_bwd_bound(rws, m, sg, sgx, m, bw, invstd, s2)
        assert float(_C.halves_scale_from_slots(s1)[0]) == float(_C.halves_scale_from_slots(s2)[0]), (case, m, K, F)


def test_bcast_kernels_direct(golden):
    """spmm_bcast / spmm_dot_bcast (aggregate-before-project forms) against plain torch indexing."""
    s, d, n = golden.graph("g300")
    g = bot_amd.Graph(s, d, n, chunk=8).to(DEV)
    E = s.numel()
    gen = torch.Generator().manual_seed(8)
    for H, D in ((3, 168), (1, 40), (4, 9), (2, 250)):
        x = torch.randn(n, D, generator=gen).to(DEV)
        w = torch.rand(E, H, generator=gen).to(DEV)
        csc, csr = g.csc, g.csr
        rows = torch.repeat_interleave(torch.arange(n, device=DEV), (csc.indptr[1:] - csc.indptr[:-1]).long())
        ref = torch.zeros(n, H, D, device=DEV).index_add_(0, rows, w.unsqueeze(-1) * x[csc.indices.long()].unsqueeze(1))
        out = _C.spmm_bcast(csc, x, w, None, head_outer=True)
        assert torch.allclose(out.permute(1, 0, 2), ref, atol=1e-4, rtol=1e-4)
        assert torch.allclose(_C.spmm_bcast(csc, x, w, None, head_outer=False), ref, atol=1e-4, rtol=1e-4)
        dz = torch.randn(H, n, D, generator=gen).to(DEV)
        rows_r = torch.repeat_interleave(torch.arange(n, device=DEV), (csr.indptr[1:] - csr.indptr[:-1]).long())
        wr = w[g.csr2csc.long()]
        xs = dz[:, csr.indices.long(), :].permute(1, 0, 2)
        ref_out = torch.zeros(n, D, device=DEV).index_add_(0, rows_r, (wr.unsqueeze(-1) * xs).sum(1))
        ref_dot = torch.empty(E, H, device=DEV)
        ref_dot[g.csr2csc.long()] = (xs * x[rows_r].unsqueeze(1)).sum(-1)
        o, dot = _C.spmm_dot_bcast(csr, dz, w, g.csr2csc, x)
        assert torch.allclose(o, ref_out, atol=2e-4, rtol=1e-4)
        assert torch.allclose(dot, ref_dot, atol=1e-3 * D ** 0.5, rtol=1e-4)
        # the weight gradient alone, by the in-edge sweep that gathers the source row once for all heads (sddmm_dot_bcast):
        # same numbers as the dot output of spmm_dot_bcast; also from a row-strided x and through an output permutation
        dot2 = _C.sddmm_dot_bcast(csc, x, dz)
        assert torch.allclose(dot2, ref_dot, atol=1e-3 * D ** 0.5, rtol=1e-4)
        xb = torch.zeros(n, D + 4, device=DEV)
        xb[:, :D] = x
        assert torch.equal(_C.sddmm_dot_bcast(csc, xb[:, :D], dz), dot2)
        dot3 = _C.sddmm_dot_bcast(csc, x, dz, operm=csc.eid)
        assert torch.equal(dot3[csc.eid.long()], dot2)


def test_blocked_spmm_matches_row_kernel():
    """Dense graph (mean degree ~150, a few hubs): the L2-blocked SpMM + hub fallback equals the row-per-group kernel and
    plain torch, weighted and unweighted, through ops (forward of copy_u_sum / u_mul_e_sum and the GCN backward); the fused
    backward (row kernel) on the same graph."""
    from bot_amd import blocked
    n = 3000
    gen = torch.Generator().manual_seed(21)
    src = torch.randint(0, n, (450000,), generator=gen)
    dst = (n * torch.rand(450000, generator=gen, dtype=torch.float64) ** 1.6).long().clamp_(max=n - 1)  # skewed in-degrees
    src = torch.cat([src, torch.arange(n), torch.arange(n)])   # two hubs: every node points at nodes 0 and 1
    dst = torch.cat([dst, torch.zeros(n, dtype=torch.int64), torch.ones(n, dtype=torch.int64)])
    s, d = R.preprocess_edges(src, dst, n)
    g = bot_amd.Graph(s, d, n).to(DEV)
    assert g.csc.nnz / n > blocked.MIN_MEAN_DEGREE
    csc = g.csc
    rows = torch.repeat_interleave(torch.arange(n, device=DEV), (csc.indptr[1:] - csc.indptr[:-1]).long())
    # (1,128) / (2,60): 2 edges per gather instruction; (1,44) (2,24) (1,30) (1,16): 4 (lane groups, padded streams, T=256)
    for H, D in ((1, 256), (6, 80), (1, 41), (3, 250), (1, 128), (2, 60), (1, 44), (2, 24), (1, 30), (1, 16), (4, 2)):
        x = torch.randn(n, H, D, generator=gen).to(DEV)
        w = torch.rand(csc.nnz, H, generator=gen).to(DEV)
        for weights in (None, w):
            blocked.ENABLED = True
            out_b = _C.spmm(csc, x, weights, None)
            vec = 4 if D % 4 == 0 else 2 if D % 2 == 0 else 1
            bp = blocked.plan_for(csc, n, H, D)
            assert (bp is not None) == (blocked.MIN_ROW_FLOATS <= H * D <= 256 * vec)
            if bp is not None:
                lanes = -(-H * D // vec)
                assert bp.epi == (4 if lanes <= 16 else 2 if lanes <= 32 else 1)
            blocked.ENABLED = False
            out_r = _C.spmm(csc, x, weights, None)
            blocked.ENABLED = True
            xs = x[csc.indices.long()]
            if weights is not None:
                xs = xs * weights.unsqueeze(-1)
            ref = torch.zeros(n, H, D, device=DEV).index_add_(0, rows, xs)
            assert torch.allclose(out_b, ref, atol=2e-3, rtol=1e-4)
            assert torch.allclose(out_b, out_r, atol=2e-3, rtol=1e-4)
    bp = blocked.plan_for(csc, n, 1, 256)
    assert bp.heavy is not None  # the hub fallback is exercised
    # residual epilogue and row-strided operands (slices of one GEMM output), blocked rows and hub rows alike
    for H, D in ((6, 80), (1, 44), (2, 60)):
        big = torch.randn(n, 2 * H * D + 8, generator=gen).to(DEV)
        xs = big[:, :H * D].unflatten(1, (H, D))
        res = big[:, H * D:2 * H * D].unflatten(1, (H, D))
        w = torch.rand(csc.nnz, H, generator=gen).to(DEV)
        blocked.ENABLED = True
        out_b = _C.spmm(csc, xs, w, None, addend=res)
        blocked.ENABLED = False
        out_r = _C.spmm(csc, xs, w, None, addend=res)
        blocked.ENABLED = True
        ref = res + torch.zeros(n, H, D, device=DEV).index_add_(0, rows, xs[csc.indices.long()] * w.unsqueeze(-1))
        assert torch.allclose(out_b, ref, atol=2e-3, rtol=1e-4) and torch.allclose(out_b, out_r, atol=2e-3, rtol=1e-4)
    xg = torch.randn(n, 64, generator=gen).to(DEV).requires_grad_()
    y = ops.copy_u_sum(g, xg)
    gy = torch.randn(n, 64, generator=gen).to(DEV)
    (y * gy).sum().backward()
    ref_g = torch.zeros(n, 64, device=DEV).index_add_(0, s.to(DEV), gy[d.to(DEV)])
    assert torch.allclose(xg.grad, ref_g, atol=2e-3, rtol=1e-4)
    # the fused backward (spmm_dot: d ft and d a from one sweep of the transposed direction) on this dense graph — it has no
    # blocked form (bot_amd/blocked.py: built in round 2, never faster, removed in round 3): every head-segment layout of the row
    # kernel (16 / 32 / 64 lanes, 1-4 chunks), slabs of a wider buffer, hub rows, against plain torch; bitwise reproducible
    csr, c2c = g.csr, g.csr2csc
    rows_r = torch.repeat_interleave(torch.arange(n, device=DEV), (csr.indptr[1:] - csr.indptr[:-1]).long())
    for H, D in ((6, 80), (4, 120), (2, 64), (8, 16), (1, 128), (3, 40), (4, 250), (3, 7), (1, 20)):
        F = H * D
        big = torch.randn(n, 3 * F + 4, generator=gen).to(DEV)
        dx = big[:, :F].unflatten(1, (H, D))                      # upstream gradient (gathered)
        ft = big[:, F:2 * F].unflatten(1, (H, D))                 # the rows' own features
        w = torch.rand(csr.nnz, H, generator=gen).to(DEV)         # in CSC position order, reached through csr2csc
        outb = big[:, 2 * F:3 * F].unflatten(1, (H, D))
        o_b, dot_b = _C.spmm_dot(csr, dx, w, c2c, ft, out=outb)
        o_b, dot_b = o_b.clone(), dot_b.clone()
        wr = w[c2c.long()]
        xs = dx[csr.indices.long()]
        ref_o = torch.zeros(n, H, D, device=DEV).index_add_(0, rows_r, xs * wr.unsqueeze(-1))
        ref_dot = torch.empty(csr.nnz, H, device=DEV)
        ref_dot[c2c.long()] = (xs * ft[rows_r]).sum(-1)
        assert torch.allclose(o_b, ref_o, atol=2e-3, rtol=1e-4), (H, D)
        assert torch.allclose(dot_b, ref_dot, atol=1e-3 * D ** 0.5, rtol=1e-4), (H, D)
        o2, dot2 = _C.spmm_dot(csr, dx, w, c2c, ft)
        assert torch.equal(o2, o_b) and torch.equal(dot2, dot_b)


def test_spmm_flat_16_byte_lanes():
    """bot::spmm_flat_kernel — the weighted all-heads forward for head widths that are not a multiple of 4 floats (3 x 250), rows
    read as flat float4 lanes with the head resolved per element: against torch on a power-law graph with long rows (chunk 8 and
    default), slabs inside a wider 16-byte aligned buffer (the merged GEMM output), residual addend on an 8-byte aligned offset,
    row-contiguous and padded outputs; bitwise reproducible; and the dispatcher takes it exactly when the layout allows."""
    n = 3000
    rs, rd = _powerlaw(n, 40000, 9)
    s, d = R.preprocess_edges(rs, rd, n)
    gen = torch.Generator().manual_seed(5)
    _C.SPMM_LAYOUT = "flat"        # the hint the host gives for graphs whose plans are in XCD order (bot_amd/_C.py:spmm)
    for chunk in (8, None):
        g = bot_amd.Graph(s, d, n, chunk=chunk).to(DEV)
        csc = g.csc
        rows = torch.repeat_interleave(torch.arange(n, device=DEV), (csc.indptr[1:] - csc.indptr[:-1]).long())
        for H, D in ((3, 250), (2, 250), (4, 30), (3, 6), (2, 9), (4, 255)):
            F = H * D
            P = (2 * F + 3) // 4 * 4 + 8
            big = torch.randn(n, P, generator=gen).to(DEV)
            x = big[:, :F].unflatten(1, (H, D))                          # 16-byte aligned rows, pitch P
            res = big[:, F:2 * F].unflatten(1, (H, D))                   # starts at column F: 8-byte aligned only when F is even
            w = torch.rand(csc.nnz, H, generator=gen).to(DEV)
            ref = torch.zeros(n, H, D, device=DEV).index_add_(0, rows, x[csc.indices.long()] * w.unsqueeze(-1))
            for addend in (None, res):
                for padded_out in (False, True):
                    outbuf = torch.empty(n, (F + 3) // 4 * 4 if padded_out else F, device=DEV)
                    out = outbuf[:, :F].unflatten(1, (H, D))
                    got = _C.spmm(csc, x, w, None, out=out, addend=addend)
                    assert _C._lib.bot_last_kernel().decode().startswith("bot::spmm_flat_kernel") == (D % 4 != 0), (H, D, _C._lib.bot_last_kernel())
                    want = ref if addend is None else ref + res
                    assert torch.allclose(got, want, atol=2e-4, rtol=1e-5), (H, D, chunk, addend is not None, padded_out)
                    again = _C.spmm(csc, x, w, None, out=torch.empty_like(outbuf)[:, :F].unflatten(1, (H, D)), addend=addend)
                    assert torch.equal(again, got)
            xc = x.contiguous()                                             # pitch F: not 16-byte aligned rows unless F % 4 == 0
            got = _C.spmm(csc, xc, w, None)
            assert _C._lib.bot_last_kernel().decode().startswith("bot::spmm_flat_kernel") == (D % 4 != 0 and F % 4 == 0)
            assert torch.allclose(got, ref, atol=2e-4, rtol=1e-5)
            # both layouts: bitwise identical results (same per-element summation order)
            _C.SPMM_LAYOUT = "rows"
            rows_out = _C.spmm(csc, x, w, None, addend=res, out=torch.empty(n, F, device=DEV).unflatten(1, (H, D)))
            assert not _C._lib.bot_last_kernel().decode().startswith("bot::spmm_flat_kernel")
            _C.SPMM_LAYOUT = "flat"
            assert torch.equal(rows_out, _C.spmm(csc, x, w, None, addend=res, out=torch.empty(n, F, device=DEV).unflatten(1, (H, D))))
    _C.SPMM_LAYOUT = None
    # default: the hint follows the plan order of the direction
    h = bot_amd.reorder_graph(bot_amd.Graph(s, d, n), "degree", plan_order="xcd").to(DEV)
    big = torch.randn(n, 752, generator=gen).to(DEV)
    w = torch.rand(h.csc.nnz, 3, generator=gen).to(DEV)
    _C.spmm(h.csc, big[:, :750].unflatten(1, (3, 250)), w, None)
    assert _C._lib.bot_last_kernel().decode().startswith("bot::spmm_flat_kernel") and h.csc.plan_order == "xcd"
    _C.spmm(bot_amd.Graph(s, d, n).to(DEV).csc, big[:, :750].unflatten(1, (3, 250)), w, None)
    assert _C._lib.bot_last_kernel().decode().startswith("bot::spmm_rows_kernel")


def test_config1_cora_shape_gcn():
    """BASELINE config 1 at its exact shape (S-cora: 2 708 nodes / 10 556 raw edges / 1 433 features, 2-layer GCN hidden 16,
    7 classes): forward + backward of the HIP path against the CPU oracle."""
    import torch.nn.functional as F
    from bot_amd import nn as bnn, synth
    from oracle import ref_models as RM
    ds = synth.make_dataset("cora", device="cpu")
    s, d = ds.graph.edges()
    n = ds.graph.number_of_nodes()
    cfg = dict(n_layers=2, norm="none", norm_adj="symm", use_linear=False, residual=False)
    torch.manual_seed(5)
    model = bnn.GCN(in_feats=1433, n_classes=7, n_hidden=16, activation=F.relu, dropout=0.0, **cfg).train()
    p = {k: (v.clone().requires_grad_() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items()}
    gout = torch.randn(n, 7)
    ref = RM.gcn_forward(RM.CooGraph(s, d, n), ds.feat, p, **cfg)
    names = list(p)
    ref_grads = torch.autograd.grad((ref * gout).sum(), [p[k] for k in names])
    model = model.to(DEV)
    logits = model(ds.graph.to(DEV), ds.feat.to(DEV))
    (logits * gout.to(DEV)).sum().backward()
    PC.fwd_close(logits, ref.detach().numpy())
    got = dict(model.named_parameters())
    for k, rg in zip(names, ref_grads):
        PC.grad_close(got[k].grad, rg.numpy())


def test_random_shapes_against_torch():
    """Randomised sweep over (H, D, row padding, chunk) to cover every vector-width / lane-group / chunk dispatch branch of
    the gather kernels, including row-padded (strided) inputs."""
    import random
    rnd = random.Random(1234)
    gen = torch.Generator().manual_seed(1234)
    n = 257
    for trial in range(40):
        e = rnd.choice([0, 1, 50, 3000])
        src, dst = torch.randint(0, n, (e,), generator=gen), torch.randint(0, n, (e,), generator=gen)
        g = bot_amd.Graph(src, dst, n, chunk=rnd.choice([1, 4, 64])).to(DEV)
        H, D = rnd.choice([1, 2, 3, 5, 8]), rnd.choice([1, 2, 3, 4, 6, 7, 8, 16, 31, 33, 64, 100, 129, 250, 260, 513])
        pad = rnd.choice([0, 1, 2, 4])
        buf = torch.randn(n, H * D + pad, generator=gen).to(DEV)
        x = buf[:, :H * D].unflatten(1, (H, D))                       # row stride H*D + pad
        w = torch.rand(e, H, generator=gen).to(DEV)
        csc, csr = g.csc, g.csr
        rows = torch.repeat_interleave(torch.arange(n, device=DEV), (csc.indptr[1:] - csc.indptr[:-1]).long())
        ref = torch.zeros(n, H, D, device=DEV).index_add_(0, rows, x[csc.indices.long()] * w.unsqueeze(-1))
        tol = dict(atol=1e-4 * max(1.0, D ** 0.5), rtol=1e-4)
        assert torch.allclose(_C.spmm(csc, x, w, None), ref, **tol), (trial, H, D, pad, e)
        ref0 = torch.zeros(n, H, D, device=DEV).index_add_(0, rows, x[csc.indices.long()])
        assert torch.allclose(_C.spmm(csc, x, None, None), ref0, **tol), (trial, H, D, pad, e)
        y = torch.randn(n, H, D, generator=gen).to(DEV)
        refd = (x[csc.indices.long()] * y[rows]).sum(-1)
        assert torch.allclose(_C.sddmm_dot(csc, x, y), refd, **tol), (trial, H, D, pad, e)
        if D <= _C.spmm_dot_max_d(x):
            rows_r = torch.repeat_interleave(torch.arange(n, device=DEV), (csr.indptr[1:] - csr.indptr[:-1]).long())
            wr = w[g.csr2csc.long()]
            ref_o = torch.zeros(n, H, D, device=DEV).index_add_(0, rows_r, x[csr.indices.long()] * wr.unsqueeze(-1))
            ref_dot = torch.empty(e, H, device=DEV)
            ref_dot[g.csr2csc.long()] = (x[csr.indices.long()] * y[rows_r]).sum(-1)
            o, dot = _C.spmm_dot(csr, x, w, g.csr2csc, y)
            assert torch.allclose(o, ref_o, **tol) and torch.allclose(dot, ref_dot, **tol), (trial, H, D, pad, e)
        el, er = torch.randn(n, H, generator=gen).to(DEV), torch.randn(n, H, generator=gen).to(DEV)
        a = _C.gat_attn_fwd(csc, el, er, None, None, None, 0.2, H, None)
        if e:
            z = torch.nn.functional.leaky_relu(el[csc.indices.long()] + er[rows], 0.2)
            assert torch.allclose(a, R.edge_softmax(rows.cpu(), n, z.cpu()).to(DEV), atol=1e-6), (trial, H, e)


def test_random_keep_exact_uniform_and_reproducible():
    """Edge-drop mask (bot_random_keep_u8): exactly n_keep ones, a pure function of the seed, different for another seed,
    every element kept with probability n_keep / n (2 000 seeds), and the layers' edge drop leaves out exactly
    int(E * p) edges (models.py:528-532)."""
    for n, n_keep in ((1, 1), (1, 0), (2, 1), (7, 3), (1000, 900), (4097, 1), (100001, 100000), (3000017, 2700015)):
        a = _C.random_keep(n, n_keep, 1234, DEV)
        assert a.dtype == torch.uint8 and a.shape == (n,) and int(a.sum()) == n_keep and int(a.max()) <= 1
        assert torch.equal(a, _C.random_keep(n, n_keep, 1234, DEV))
        if 0 < n_keep < n and n > 7:
            assert not torch.equal(a, _C.random_keep(n, n_keep, 1235, DEV))
    n, n_keep, trials = 50, 35, 2000
    hits = torch.zeros(n, device=DEV)
    for seed in range(trials):
        hits += _C.random_keep(n, n_keep, seed * 7919 + 1, DEV)
    p = n_keep / n
    sigma = (trials * p * (1 - p)) ** 0.5
    assert float((hits - trials * p).abs().max()) < 4.5 * sigma   # 50 elements: 4.5 sigma never trips by chance
    pairs = torch.zeros((), device=DEV)                           # and pairs are (nearly) independent: P(both kept) = k(k-1)/(n(n-1))
    for seed in range(trials):
        k = _C.random_keep(n, n_keep, seed * 104729 + 3, DEV)
        pairs += k[0].float() * k[1].float()
    pp = n_keep * (n_keep - 1) / (n * (n - 1))
    assert abs(float(pairs) - trials * pp) < 4.5 * (trials * pp * (1 - pp)) ** 0.5
    import bot_amd.nn as bnn
    s, d = R.preprocess_edges(torch.randint(0, 300, (5000,)), torch.randint(0, 300, (5000,)), 300)
    g = bot_amd.Graph(s, d, 300).to(DEV)
    conv = bnn.GATConv(8, 4, num_heads=2, edge_drop=0.3).to(DEV).train()
    E = g.number_of_edges()
    keep = conv._kept_edges(g)
    assert int(keep.sum()) == E - int(E * 0.3)
    torch.manual_seed(0); k1 = conv._kept_edges(g)
    torch.manual_seed(0); k2 = conv._kept_edges(g)
    assert torch.equal(k1, k2)                                    # torch.manual_seed governs the mask


@pytest.mark.parametrize("name", ["reddit", "proteins", "products"])
def test_full_size_properties_configs345(name):
    """BASELINE configs 3-5 at their full synthetic sizes (114 M / 78 M / 126 M edges): size-independent properties of the
    HIP path — in-degrees == copy_u_sum(ones) bit-exact, attention rows sum to one, the SpMM / fused-backward pair satisfies
    the adjoint identity — on the code paths those sizes select (L2-blocked SpMM with lane groups and hub rows for S-reddit
    and S-proteins, the all-heads fused backward, long-row plans)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("scale_check", os.path.join(os.path.dirname(__file__), "..", "tools", "scale_check.py"))
    sc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sc)
    g, f, c, _ = sc.build(name)
    H, D = {"reddit": (1, 256), "proteins": (6, 80), "products": (4, 120)}[name]
    props = sc.properties(g, H, D)
    assert props["E"] == g.number_of_edges() and props["long_rows"] > 0
    from bot_amd import blocked
    assert (blocked.plan_for(g.csc, g.number_of_nodes(), H, D) is not None) == (name != "products")
    if name == "reddit":  # the 41-class output layer of the GCN: lane-group blocked kernel, odd width padded by the op
        ones = torch.ones(g.number_of_nodes(), 41, device=DEV)
        assert torch.equal(ops.copy_u_sum(g, ones)[:, 40].long(), g.in_degrees())
    del g
    torch.cuda.empty_cache()


def test_keep_mask_orders(golden):
    PC.check_keep_mask_orders(golden, DEV)


# ---------------------------------------------------------------------------------------------- full-size logits + gradients
_FULL = {}


def _full_size_inputs():
    """S-arxiv at BASELINE config 2's full size (169 343 nodes, 2.5 M edges after preprocess), config-2 weights, a fixed label mask."""
    if not _FULL:
        from bot_amd import synth
        from tests import full_size as FS
        ds = synth.make_dataset("arxiv", device="cpu", seed=0)
        C = ds.n_classes
        sd = FS.init_state(FS.GAT_ARXIV, ds.feat.shape[1] + C, C, seed=0)
        mask = torch.rand(ds.train_idx.shape, generator=torch.Generator().manual_seed(7)) < 0.5
        _FULL.update(ds=ds, sd=sd, mask=mask)
    return _FULL


@pytest.mark.parametrize("fuse,symm", [(True, False), (False, False), (True, True)])
def test_full_size_config2_logits_and_grads_against_c_oracle(fuse, symm):
    """at the size the headline number is quoted on, ONE whole train step (forward + loge loss + backward,
    dropout 0, training-mode BatchNorm) on the HIP path — fused layer nodes AND the modular DGL-surface path — against the
    oracle's C restatement of DGL's CPU kernels: every logit of the 169 343 nodes within 1e-4, every entry of every parameter
    gradient within 1e-4 of that gradient's largest entry (tolerances of tests/parity_cases.py).  The oracle is evaluated at the
    HIP run's ReLU gates (tests/full_size.py:KinkGates explains why); the gates the oracle would have chosen itself may differ
    only where the pre-activation is rounding noise, which is asserted too.  `symm`: the --norm-adj=symm variant of the same
    stack (models.py:500-505, 550-555; folded into the edge weights on the fused path)."""
    from tests import full_size as FS
    c = _full_size_inputs()
    ds = c["ds"]
    g = ds.graph.to(DEV)
    g.create_formats_()
    cfg = dict(FS.GAT_ARXIV, use_symmetric_norm=symm)
    pred, grads, gates = FS.hip_step(g, ds.feat.to(DEV), ds.labels.to(DEV), ds.train_idx.to(DEV), c["mask"], c["sd"],
                                     cfg, ds.n_classes, fuse=fuse)
    s, d = ds.graph.edges()
    rp, rg, times, threads, gstats = FS.oracle_step(s, d, ds.graph.number_of_nodes(), ds.feat, ds.labels, ds.train_idx, c["mask"],
                                                    c["sd"], cfg, ds.n_classes, gates=gates)
    r = FS.compare(pred, grads, rp, rg, gstats)
    assert r["n"] == 169343 and g.number_of_edges() > 2_000_000
    # Logits: 1e-4 ABSOLUTE against the fp32 oracle (the north star's tolerance) for the BASELINE configuration (|logit| <= 15).
    # The --norm-adj=symm variant multiplies the aggregation by in_deg^+1/2 (models.py:550-555): logits reach 57 and two fp32 runs
    # differ by 1.3e-4 there, so it is ranked against the SAME step in fp64, the criterion config 4 uses (FS.CRITERIA).
    xp = None
    if symm:
        xp, xg, t64, _, _ = FS.oracle_step(s, d, ds.graph.number_of_nodes(), ds.feat, ds.labels, ds.train_idx, c["mask"], c["sd"], cfg,
                                           ds.n_classes, gates=gates, dtype=torch.float64)
        rank = FS.rank_against_exact(grads, rg, xg)
        r.update(max_hip_grad_err_vs_fp64=max(v[0] for v in rank.values()), max_oracle_grad_err_vs_fp64=max(v[1] for v in rank.values()))
    ok, crit, nums = FS.logits_ok(pred, rp, xp, PC.FWD_ATOL)
    r.update(nums, logit_criterion=crit)
    print("full-size parity", "fused" if fuse else "modular", "symm" if symm else "", r, "oracle step %.2f s on %d threads" % (times[0], threads))
    assert ok and crit == ("fp64-ranked" if symm else "abs"), r
    assert r["max_rel_grad_err"] <= PC.GRAD_RTOL, r
    assert r["max_abs_preact_at_differing_gate"] <= 1e-4 and r["relu_gates_differing"] <= 1e-5 * r["relu_gates"] and r["leaky_gates_differing"] <= 1e-5 * r["leaky_gates"], r


# ---------------------------------------------------------------------------------------------- f3: inference-only GAT layer
def test_gat_infer_kernel_against_unfused_kernels_and_oracle(golden):
    """bot_gat_infer_f32 (logits + softmax + aggregation + residual + affine + ReLU in one sweep, nothing edge-sized written)
    against (a) the training kernels it replaces in eval mode — bot_gat_attn_fwd_f32 + bot_spmm_f32 + torch epilogue — on a
    power-law graph with long rows at the default chunk and on a small graph with chunk 8 (multi-slot rows everywhere), over
    head / width shapes that select every layout (all-heads with 1, 2, 4 heads per chunk and 2 chunks per head; head-major
    groups of 8..64 lanes and up to 4 chunks), aligned and misaligned strides; and (b) the oracle's restatement of the header
    contract (tests/_oracle_backend.gat_infer, plain torch ops on the CPU)."""
    from tests import _oracle_backend as OB
    n = 20000
    rs, rd = _powerlaw(n, 150000, 5)
    s, d = R.preprocess_edges(rs, rd, n)
    big = bot_amd.Graph(s, d, n).to(DEV)
    s2, d2, n2 = golden.graph("g300")
    small_cpu = bot_amd.Graph(s2, d2, n2, chunk=8)
    small = small_cpu.to(DEV)
    assert big.csc.n_long > 0 and small.csc.n_long > 10
    gen = torch.Generator().manual_seed(3)
    shapes = [(3, 250), (1, 40), (4, 120), (6, 80), (2, 64), (8, 16), (1, 128), (3, 7), (2, 250), (5, 33), (1, 1000), (4, 30), (1, 3)]
    kernels = set()
    for g, gc in ((big, None), (small, small_cpu)):
        N, E = g.number_of_nodes(), g.number_of_edges()
        for i, (H, D) in enumerate(shapes):
            F = H * D
            # even cases: row stride a multiple of 4 floats (16-byte lanes); odd cases: an odd stride forces the 4-byte instances
            pad = (-(F + H)) % 4 if i % 2 == 0 else (3 if (F + H + 3) % 2 else 4)
            buf = torch.randn(N, F + H + pad, generator=gen).to(DEV)
            x = buf[:, :F].unflatten(1, (H, D))                 # strided slab, as in the merged GEMM output
            el = buf[:, F:F + H]                                 # a column view, row stride F + H + pad
            er = torch.randn(N, H, generator=gen).to(DEV) if i % 3 != 1 else None
            ee = torch.randn(E, H, generator=gen).to(DEV) if i % 4 == 2 else None
            ew = (torch.rand(E, generator=gen) + 0.5).to(DEV) if i % 3 == 0 else None
            addend = torch.randn(N, H, D, generator=gen).to(DEV) if i % 2 == 0 else None
            scale = (torch.rand(F, generator=gen) + 0.5).to(DEV) if i % 5 != 4 else None
            shift = torch.randn(F, generator=gen).to(DEV) if i % 3 != 2 else None
            relu = i % 2 == 1
            out = _C.gat_infer(g.csc, x, el, er, ee, ew, 0.2, addend=addend, scale=scale, shift=shift, relu=relu)
            kernels.add(_C._lib.bot_last_kernel().decode())
            a = _C.gat_attn_fwd(g.csc, el.contiguous(), er, ee, None, None, 0.2, H, None)
            if ew is not None:
                a = a * ew.unsqueeze(1)
            ref = _C.spmm(g.csc, x, a, None, addend=addend).reshape(N, F)
            if scale is not None:
                ref = ref * scale
            if shift is not None:
                ref = ref + shift
            if relu:
                ref = torch.relu(ref)
            assert torch.allclose(out.reshape(N, F), ref, rtol=1e-4, atol=3e-5), (H, D, float((out.reshape(N, F) - ref).abs().max()))
            if gc is not None:                                   # the oracle, independent of every HIP kernel
                cpu = lambda t: None if t is None else t.cpu()
                oref = OB.gat_infer(gc.csc, x.cpu(), cpu(el), cpu(er), cpu(ee), cpu(ew), 0.2, addend=cpu(addend), scale=cpu(scale),
                                    shift=cpu(shift), relu=relu)
                PC.fwd_close(out, oref.numpy(), 1e-4)
            # el None: plain (ew-weighted) sum, the GraphConv aggregation
            out2 = _C.gat_infer(g.csc, x, None, None, None, ew, addend=addend, scale=scale, shift=shift, relu=relu)
            w1 = (ew if ew is not None else torch.ones(E, device=DEV)).unsqueeze(1).expand(E, H).contiguous()
            ref2 = _C.spmm(g.csc, x, w1, None, addend=addend).reshape(N, F)
            ref2 = ref2 * scale if scale is not None else ref2
            ref2 = ref2 + shift if shift is not None else ref2
            ref2 = torch.relu(ref2) if relu else ref2
            assert torch.allclose(out2.reshape(N, F), ref2, rtol=1e-4, atol=3e-5), (H, D)
    assert any("rows_kernel" in k for k in kernels) and any("heads_kernel" in k for k in kernels), kernels
    # isolated destination: no in-edges -> act(addend * scale + shift)
    g0 = bot_amd.Graph(torch.tensor([0, 1]), torch.tensor([1, 1]), 3).to(DEV)
    x = torch.randn(3, 2, 8, device=DEV)
    ad = torch.randn(3, 2, 8, device=DEV)
    o = _C.gat_infer(g0.csc, x, torch.randn(3, 2, device=DEV), None, None, None, addend=ad)
    assert torch.equal(o[0], ad[0]) and torch.equal(o[2], ad[2])
    # determinism
    xb, lb = torch.randn(n, 3, 250, device=DEV), torch.randn(n, 3, device=DEV)
    assert torch.equal(_C.gat_infer(big.csc, xb, lb), _C.gat_infer(big.csc, xb, lb))


def test_evaluate_inference_path_full_size():
    """f3 at BASELINE config 2's size: `evaluate()` (run.py:290-322) with 1 label-reuse iteration through the inference-only
    layers against the generic eval-mode forward of the same model (fuse_layers off: modular layers + fused BN kernel), and the
    label-reuse cache of the first projection against recomputation."""
    from bot_amd import nn as bnn, synth, train as T
    from bot_amd.nn import fused
    import torch.nn.functional as F
    ds = synth.make_dataset("arxiv", device="cpu", seed=0)
    g = ds.graph.to(DEV)
    g.create_formats_()
    C = ds.n_classes
    torch.manual_seed(0)
    model = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, n_layers=3, n_heads=3, n_hidden=250,
                    norm="batch", dropout=0.75, input_drop=0.25, attn_drop=0.1, linear=True).to(DEV)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.3)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
    feat, labels = ds.feat.to(DEV), ds.labels.to(DEV)
    tr, va, te = ds.train_idx.to(DEV), ds.val_idx.to(DEV), ds.test_idx.to(DEV)
    n0 = fused.INFER_CALLS
    ev = T.evaluate(model, g, feat, labels, tr, va, te, use_labels=True, n_label_iters=1, loss="loge", n_classes=C)
    assert fused.INFER_CALLS - n0 == 6
    model.fuse_layers = False
    ev_ref = T.evaluate(model, g, feat, labels, tr, va, te, use_labels=True, n_label_iters=1, loss="loge", n_classes=C)
    assert fused.INFER_CALLS - n0 == 6
    assert float((ev[6] - ev_ref[6]).abs().max()) <= 1e-4
    assert all(abs(float(a) - float(b)) <= 1e-4 for a, b in zip(ev[3:6], ev_ref[3:6]))
    assert all(abs(a - b) <= 2e-4 for a, b in zip(ev[:3], ev_ref[:3]))  # accuracies: a tie in an argmax may flip a node
    # not only against the HIP generic path — the same evaluate() (run.py:290-322: all training labels as inputs,
    # eval-mode BatchNorm, one label-reuse iteration run.py:304-308) restated on the oracle's C kernels at FULL size, every logit
    # of the 169 343 nodes within 1e-4 absolute and the three losses within 1e-4
    from oracle import c_ops
    from oracle import ref_models as RM
    from tests import full_size as FS
    torch.set_num_threads(FS._oracle_threads(32))
    c_ops.set_num_threads(FS._oracle_threads(32))
    s_, d_ = ds.graph.edges()
    cg = c_ops.CGraph(s_, d_, ds.graph.number_of_nodes())
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = dict(n_layers=3, n_heads=3, n_hidden=250, n_classes=C, norm="batch", non_interactive_attn=False, use_symmetric_norm=False,
               linear=True, residual=False, training=False)
    with torch.no_grad():
        x = RM.add_labels(ds.feat, ds.labels, ds.train_idx, C)
        pred = RM.gat_forward(cg, x, sd, **cfg)
        unl = torch.cat([ds.val_idx, ds.test_idx])
        x[unl, -C:] = torch.softmax(pred[unl], dim=-1)
        pred = RM.gat_forward(cg, x, sd, **cfg)
        losses = [float(RM.compute_loss(pred[i], ds.labels[i], "loge")) for i in (ds.train_idx, ds.val_idx, ds.test_idx)]
    for tag, e in (("inference-only layers", ev), ("generic eval path", ev_ref)):
        err = float((e[6].cpu() - pred).abs().max())
        print(f"evaluate() full size vs the C oracle, {tag}: max |logit diff| {err:.2e} (logit scale {float(pred.abs().max()):.1f})")
        assert err <= PC.FWD_ATOL, (tag, err)
        assert all(abs(float(a) - b) <= 1e-4 for a, b in zip(e[3:6], losses)), (tag, e[3:6], losses)


# ---------------------------------------------------------------------------------------------- hipGraph-captured train step
@pytest.mark.isolated
def test_captured_train_step_bit_identical_and_fresh_masks():
    """the whole train step (forward, loge loss, backward, RMSprop) captured into a hipGraph.  (a) With every drop
    rate 0 the replayed step is BIT-identical to the eager step from the same state (same kernels, same order, no atomics):
    parameters, BatchNorm buffers and logits after 3 steps.  (b) With the reference's drop rates every replay draws fresh
    masks (torch's graph-safe generators + the device seed word of the fused dropout kernels) and training proceeds."""
    import copy
    import torch.nn.functional as F
    from bot_amd import nn as bnn, synth, train as T
    ds = synth.make_dataset("arxiv", device=DEV, seed=0, scale=0.2)
    g, C = ds.graph, ds.n_classes
    g.create_formats_()
    mask = torch.rand(ds.train_idx.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(3)) < 0.5
    kw = dict(use_labels=True, loss="loge", n_classes=C)

    def make(drop):
        torch.manual_seed(0)
        k = 1.0 if drop else 0.0
        m = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, n_layers=3, n_heads=3, n_hidden=64,
                    norm="batch", dropout=0.75 * k, input_drop=0.25 * k, attn_drop=0.1 * k, linear=True).to(DEV)
        return m, torch.optim.RMSprop(m.parameters(), lr=0.002, capturable=True)

    # (a) bit-identity without dropout, fixed mask
    m1, o1 = make(False)
    m2, o2 = make(False)
    m2.load_state_dict(copy.deepcopy(m1.state_dict()))
    cap = T.captured_train_step(m2, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o2, warmup=3, mask=mask, **kw)
    for _ in range(3 + 1):          # the capture ran 3 warm-up steps + 1 capture pass (capture itself does not execute)
        pass
    for _ in range(3):               # eager: the same number of optimizer steps as warm-up, then 3 more
        T.train_step(m1, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o1, mask=mask, **kw)
    for _ in range(3):
        le, pe = T.train_step(m1, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o1, mask=mask, **kw)
        lc, pc = cap()
    torch.cuda.synchronize()
    assert torch.equal(lc, le) and torch.equal(pc, pe)
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    # (a') ADVICE r2: evaluate() between replays.  The inference path caches the merged weights and the eval-mode BatchNorm
    # (scale, shift) per parameter version; a replay moves parameters and running statistics INSIDE the graph, where no
    # version counter sees it — the captured model must still evaluate like the eager one after every further step.
    ekw = dict(use_labels=True, n_label_iters=0, loss="loge", n_classes=C)
    prev = None
    for _ in range(2):
        ev_c = T.evaluate(m2, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, **ekw)
        ev_e = T.evaluate(m1, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, **ekw)
        assert torch.equal(ev_c[6], ev_e[6]) and ev_c[:3] == ev_e[:3]
        assert prev is None or not torch.equal(prev, ev_c[6])                    # the step in between did change the model
        prev = ev_c[6].clone()
        T.train_step(m1, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o1, mask=mask, **kw)
        cap()
    # forward-only passes in training mode move only the running statistics (raw-pointer writes of bot_bn_stats_*): seen too
    ev0 = T.evaluate(m1, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, **ekw)[6].clone()
    m1.train()
    with torch.no_grad():
        m1(g, T.add_labels(ds.feat, ds.labels, ds.train_idx, C))
    ev1 = T.evaluate(m1, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, **ekw)[6]
    m1.fuse_layers = False
    ev1_generic = T.evaluate(m1, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, **ekw)[6]
    m1.fuse_layers = True
    assert not torch.equal(ev0, ev1) and torch.allclose(ev1, ev1_generic, atol=1e-4)
    # (b) dropout on: fresh masks per replay, finite decreasing-ish loss
    m3, o3 = make(True)
    cap3 = T.captured_train_step(m3, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o3, warmup=3, mask_rate=0.5, **kw)
    losses, preds = [], []
    for _ in range(4):
        l, p = cap3()
        losses.append(float(l))
        preds.append(p.clone())
    assert all(torch.isfinite(torch.tensor(losses))) and len(set(losses)) == 4, losses
    assert not torch.equal(preds[0], preds[1])
    from bot_amd import _C
    assert int(_C.SEED_OFFSET) >= 4
    _C.SEED_OFFSET = None


def test_attention_dropout_in_kernel(golden):
    """nn.Dropout on the attention weights (models.py:544) fused into the attention kernels: a_drop = a * keep / (1 - p) with a
    Philox mask that the backward regenerates.  Keep rate ~ 1 - p, kept entries scaled exactly, `a` itself untouched, a fresh
    mask per seed, the device seed word changes it, and the backward equals autograd of softmax * (the forward's mask)."""
    n = 20000
    rs, rd = _powerlaw(n, 150000, 5)
    s, d = R.preprocess_edges(rs, rd, n)
    g = bot_amd.Graph(s, d, n).to(DEV)
    E = g.number_of_edges()
    for H in (3, 1, 6):
        el, er = torch.randn(n, H, device=DEV), torch.randn(n, H, device=DEV)
        p, seed = 0.3, 0x0123456789ABCDEF + H
        a0 = _C.gat_attn_fwd(g.csc, el, er, None, None, None, 0.2, H, None)
        a, ad = _C.gat_attn_fwd(g.csc, el, er, None, None, None, 0.2, H, None, drop=(p, seed))
        assert torch.equal(a, a0)
        kept = ad != 0
        rate = float(kept.sum()) / float((a0 != 0).sum())
        assert abs(rate - (1 - p)) < 5e-3, rate
        assert torch.allclose(ad[kept], (a0 / (1 - p))[kept], rtol=1e-6, atol=0)
        _, ad2 = _C.gat_attn_fwd(g.csc, el, er, None, None, None, 0.2, H, None, drop=(p, seed))
        _, ad3 = _C.gat_attn_fwd(g.csc, el, er, None, None, None, 0.2, H, None, drop=(p, seed + 1))
        assert torch.equal(ad, ad2) and not torch.equal(ad3 != 0, kept)
        _C.SEED_OFFSET = torch.ones(1, dtype=torch.int64, device=DEV)
        try:
            _, ad4 = _C.gat_attn_fwd(g.csc, el, er, None, None, None, 0.2, H, None, drop=(p, seed))
            assert not torch.equal(ad4 != 0, kept)
        finally:
            _C.SEED_OFFSET = None
        # backward: d(a_drop) in, dz out == autograd through softmax(leaky(z)) * factor with the forward's mask
        gd = torch.randn(E, H, device=DEV)
        dz, der = _C.gat_attn_bwd(g.csc, el, er, None, None, 0.2, H, a, gd, None, None, True, None, drop=(p, seed))
        factor = torch.where(kept, torch.full_like(ad, 1 / (1 - p)), torch.zeros_like(ad))
        dz_ref, der_ref = _C.gat_attn_bwd(g.csc, el, er, None, None, 0.2, H, a, gd * factor, None, None, True, None)
        assert torch.allclose(dz, dz_ref, rtol=1e-5, atol=1e-7) and torch.allclose(der, der_ref, rtol=1e-4, atol=1e-6)




def test_merged_weight_kernels_match_tensor_ops():
    """merge.hip: the layer's merged projection weight [W_fc^T | W_res^T | wl | wr | 0] and the gradients of fc.weight,
    res_fc.weight, attn_l, attn_r through it, against the tensor-op definition (fused.cat_weight / cat_weight_aggfirst + autograd)."""
    from bot_amd import nn as bnn
    from bot_amd.nn import fused
    gen = torch.Generator().manual_seed(12)
    for fin, H, D, linear, attn_r in ((168, 3, 250, True, False), (750, 3, 250, True, True), (750, 1, 40, True, False), (33, 2, 7, False, True)):
        for with_fc in (True, False):
            conv = bnn.GATConv(fin, D, num_heads=H, linear=linear, non_interactive_attn=attn_r).to(DEV)
            ref = fused._kp(fused.cat_weight(conv) if with_fc else fused.cat_weight_aggfirst(conv))
            g = torch.randn(ref.shape, generator=gen).to(DEV)
            params = [p for p in conv.parameters()]
            ref_grads = torch.autograd.grad((ref * g).sum(), params, allow_unused=True)
            out = fused.merged_weight(conv, with_fc=with_fc)
            assert out.shape == ref.shape and torch.allclose(out, ref, rtol=1e-5, atol=1e-6)
            grads = torch.autograd.grad((out * g).sum(), params, allow_unused=True)
            for p, a, b in zip(conv.named_parameters(), grads, ref_grads):
                if b is None:
                    assert a is None or float(a.abs().max()) == 0.0, p[0]
                else:
                    assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max())), p[0]




def test_take_rows_edge_sized_gather():
    """bot_amd.graph.take_rows at index counts where torch-ROCm's own row gather breaks (2^26 indices, 16-byte rows): equal to
    gathers done piecewise on small slices."""
    from bot_amd.graph import take_rows
    gen = torch.Generator(device=DEV).manual_seed(0)
    E, n = (1 << 26) + 12345, 100_003
    idx = torch.randint(0, n, (E,), device=DEV, generator=gen)
    for shape in ((n,), (n, 4), (n, 8)):
        x = torch.randn(shape, device=DEV, generator=gen)
        z = take_rows(x, idx)
        assert z.shape == (E,) + shape[1:]
        for a in range(0, E, 1 << 22):
            assert torch.equal(z[a:a + (1 << 22)], x[idx[a:a + (1 << 22)]])
    i32 = idx[:1000].int()
    assert torch.equal(take_rows(x, i32), x[i32.long()])


def test_colsum_cancellation():
    """bot_colsum_f32 (bias gradients): column sums of 2.4 M rows whose true sum is ~0 (the gradient in front of a training-mode
    BatchNorm) stay within 1 % of ONE row's magnitude of the fp64 sum; ragged widths and strided rows."""
    gen = torch.Generator(device=DEV).manual_seed(5)
    for n, F, ld in ((2_400_000, 47, 47), (300_001, 968, 968), (1000, 5, 8), (1, 3, 3)):
        buf = torch.randn(n, ld, device=DEV, generator=gen) * 1e-6
        x = buf[:, :F]
        if n > 1:
            x -= x.mean(0)
        ref = x.double().sum(0)
        got = _C.colsum(x)
        assert (got.double() - ref).abs().max() <= 1e-8, (n, F, float((got.double() - ref).abs().max()))
    x = torch.randn(100_000, 33, device=DEV, generator=gen) + 3.0
    assert torch.allclose(_C.colsum(x).double(), x.double().sum(0), rtol=1e-6)


def test_weight_grad_row_chunks():
    """bot_amd.ops.weight_grad / ops.linear: the chunked weight gradient equals the plain product (fp64 reference) for row counts
    around the chunking threshold, including a ragged tail; gradients of ops.linear match F.linear's."""
    gen = torch.Generator(device=DEV).manual_seed(1)
    for n in (1000, ops.SPLITK_MIN_ROWS - 1, ops.SPLITK_MIN_ROWS + 12345):
        x = torch.randn(n, 40, device=DEV, generator=gen)
        dy = torch.randn(n, 24, device=DEV, generator=gen)
        dy -= dy.mean(0)
        ref = dy.double().t() @ x.double()
        got = ops.weight_grad(dy, x)
        assert (got.double() - ref).abs().max() <= 2e-5 * ref.abs().max()
        w = torch.randn(24, 40, device=DEV, generator=gen).requires_grad_()
        b = torch.randn(24, device=DEV, generator=gen).requires_grad_()
        xl = x.clone().requires_grad_()
        ops.linear(xl, w, b).backward(dy)
        w2, b2, x2 = w.detach().clone().requires_grad_(), b.detach().clone().requires_grad_(), x.clone().requires_grad_()
        torch.nn.functional.linear(x2, w2, b2).backward(dy)
        assert torch.allclose(xl.grad, x2.grad, atol=1e-5, rtol=1e-5)
        assert (w.grad - w2.grad).abs().max() <= 1e-4 * w2.grad.abs().max()
        assert (b.grad.double() - dy.double().sum(0)).abs().max() <= 1e-5 * max(1.0, n ** 0.5)




def test_gemm_halves_against_fp64():
    """bot_amd.gemm (halves_scale / halves_split / gemm_halves through the C ABI): the fp16-halves GEMMs against an fp64 product,
    next to the stock fp32 GEMM's own error — forward (x w^T), input gradient and the chunked weight gradient, at operand
    magnitudes from 1e-9 to 1e3 and ragged sizes."""
    from bot_amd import gemm
    gen = torch.Generator(device=DEV).manual_seed(2)
    for n, K, P, sx, sd in ((20000, 750, 1536, 4.0, 1e-7), (9000, 100, 47, 1e3, 1e-9), (8192 * 2 + 5, 168, 250, 1.0, 1.0)):
        x = torch.randn(n, K, device=DEV, generator=gen) * sx
        x[:, ::7] = 0
        w = torch.randn(P, K, device=DEV, generator=gen) * 0.05
        d = torch.randn(n, P, device=DEV, generator=gen) * sd
        xs = gemm.split(x, 0)
        assert xs.scale.shape == (2,) and float(xs.scale[0] * xs.scale[1]) == 1.0
        top = float(x.abs().max() * xs.scale[0])
        assert 2 ** 13 < top <= 2 ** 14
        # the halves reproduce the operand to fp32's own rounding
        # left layouts: [h1 | h1 | 2^11 h2], or (wide operands, v15) [h1 | 2^11 h2] without the duplicate
        h1, h2 = xs.buf[:, :K].double(), xs.buf[:, xs.h2_off:xs.h2_off + K].double() / gemm.SHIFT
        assert xs.order == (2 if xs.piece >= gemm.NODUP_MIN_PIECE else 0) and xs.buf.shape[1] == (2 if xs.order == 2 else 3) * xs.piece
        if xs.order == 0:
            assert torch.equal(xs.buf[:, :K], xs.buf[:, xs.piece:xs.piece + K])
        assert ((h1 + h2) * float(xs.scale[1]) - x.double()).abs().max() <= 2.0 ** -22 * x.abs().max()
        assert not xs.buf[:, K:xs.piece].any()
        for name, got, ref, stock in (
                ("fwd", gemm.mm_nt(xs, gemm.split(w, 1)), x.double() @ w.double().t(), x @ w.t()),
                ("dx", gemm.mm_nt(gemm.split(d, 0), gemm.split(w.t().contiguous(), 1)), d.double() @ w.double(), d @ w),
                ("dw", gemm.tn(xs, gemm.split(d, 0)), x.double().t() @ d.double(), x.t() @ d)):
            scale = ref.abs().max()
            e, e32 = float((got.double() - ref).abs().max() / scale), float((stock.double() - ref).abs().max() / scale)
            print(f"gemm_halves {name} n={n} K={K} P={P}: err {e:.2e} (stock fp32 {e32:.2e})")
            assert got.shape == ref.shape and e <= max(4e-6, 3 * e32), (name, e, e32)
    # VERDICT r2 #3 — dynamic range.  Left operands whose ROWS and COLUMNS are log-uniform over 2^-12 .. 2^12 (entries spread over
    # 48 binades) under ONE power-of-two scale per matrix: the error of every output row relative to THAT ROW's largest entry,
    # next to the stock fp32 GEMM's.  The second half of a left operand is stored times 2^11 (csrc/halves.hip "Dynamic range"), so
    # rows down to 2^-28 of the matrix maximum keep fp32-GEMM accuracy; without it the smallest rows here were 1e-4 off.
    n, K, P = 20000, 750, 1536
    for which in ("rows", "cols", "rows+cols"):
        x = torch.randn(n, K, device=DEV, generator=gen)
        if "rows" in which:
            x = x * torch.exp2((torch.rand(n, 1, device=DEV, generator=gen) * 2 - 1) * 12)
        if "cols" in which:
            x = x * torch.exp2((torch.rand(1, K, device=DEV, generator=gen) * 2 - 1) * 12)
        w = torch.randn(P, K, device=DEV, generator=gen) * 0.05
        d = torch.randn(n, P, device=DEV, generator=gen) * torch.exp2((torch.rand(n, 1, device=DEV, generator=gen) * 2 - 1) * 12)
        xs, ds, wr, wtr = gemm.split(x, 0), gemm.split(d, 0), gemm.split(w, 1), gemm.split(w.t().contiguous(), 1)
        for name, got, ref, stock in (("fwd", gemm.mm_nt(xs, wr), x.double() @ w.double().t(), x @ w.t()),
                                      ("dx", gemm.mm_nt(ds, wtr), d.double() @ w.double(), d @ w)):
            rowmax = ref.abs().amax(1).clamp_min(1e-300)
            e = ((got.double() - ref).abs().amax(1) / rowmax)
            e32 = ((stock.double() - ref).abs().amax(1) / rowmax)
            print(f"gemm_halves {name} {which} 2^-12..2^12: worst row {float(e.max()):.2e} median row {float(e.median()):.2e} "
                  f"(stock fp32 {float(e32.max()):.2e} / {float(e32.median()):.2e})")
            assert float(e.max()) <= 4 * float(e32.max()) and float(e.median()) <= 2 * float(e32.median()), (name, which)
        # the weight gradient sums over the rows: measured against its own largest entry (small rows cannot matter)
        got, ref, stock = gemm.tn(xs, ds), x.double().t() @ d.double(), x.t() @ d
        e, e32 = float((got.double() - ref).abs().max() / ref.abs().max()), float((stock.double() - ref).abs().max() / ref.abs().max())
        print(f"gemm_halves dw {which}: err {e:.2e} (stock fp32 {e32:.2e})")
        assert e <= max(4e-6, 3 * e32)
    # zeros, magnitudes beyond the clamp of the scale's exponent, non-finite entries
    z = gemm.split(torch.zeros(9000, 64, device=DEV), 0)
    assert float(z.scale[0]) == 1.0 and not z.buf.any()
    for mag, want in ((1e-30, 2.0 ** 60), (1e30, 2.0 ** -86)):
        t = gemm.split(torch.full((9000, 64), mag, device=DEV), 0)
        assert float(t.scale[0]) == want and float(t.scale[0] * t.scale[1]) == 1.0 and bool(torch.isfinite(t.buf.float()).all())
    bad = torch.ones(9000, 64, device=DEV)
    bad[5, 5] = float("inf")
    assert float(gemm.split(bad, 0).scale[0]) == 1.0
    w = torch.randn(64, 64, device=DEV)
    assert not bool(torch.isfinite(gemm.mm_nt(gemm.split(bad, 0), gemm.split(w, 1))[5]).any())   # poisons its row like fp32 would


def test_bn_epilogue_writes_halves():
    """bot_bn_stats_halves_f32 + bot_bn_act_fwd_halves_f32: the same y as the plain epilogue (bit for bit), the scale bounds max|y|
    (derived from the column extremes, no pass over y), and the halves buffer equals halves_split of y at that scale, padding
    zeroed; strided input rows (pitch 752) and a ragged last quad (F = 750)."""
    gen = torch.Generator(device=DEV).manual_seed(11)
    for n, F, ld, p in ((20001, 750, 752, 0.75), (9000, 256, 256, 0.0), (8200, 120, 120, 0.5)):
        x = torch.randn(n, ld, device=DEV, generator=gen)[:, :F] * 3 + 0.5
        w = torch.randn(F, device=DEV, generator=gen)
        b = torch.randn(F, device=DEV, generator=gen)
        rm, rv, nbt = torch.zeros(F, device=DEV), torch.ones(F, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
        mean, invstd, hscale = _C.bn_stats_halves(x, 1e-5, 0.1, rm, rv, nbt, w, b, p)
        mean2, invstd2 = _C.bn_stats(x, 1e-5, 0.0)
        assert torch.equal(mean, mean2) and torch.equal(invstd, invstd2) and int(nbt) == 1
        piece = (F + 63) // 64 * 64
        y, buf = _C.bn_act_fwd(x, mean, invstd, w, b, True, p, 1234, halves=(hscale, piece))
        y2 = _C.bn_act_fwd(x, mean, invstd, w, b, True, p, 1234)
        assert torch.equal(y, y2)
        s = float(hscale[0])
        assert float(hscale[0] * hscale[1]) == 1.0 and float(y.abs().max()) * s <= 2.0 ** 14 * (1 + 1e-6)
        assert float(y.abs().max()) * s > 2.0 ** 5                     # the bound is not absurdly loose
        ref = _C.halves_split(y, hscale, 0, piece)
        assert torch.equal(buf, ref)
        _, buf2 = _C.bn_act_fwd(x, mean, invstd, w, b, True, p, 1234, halves=(hscale, piece, 2), want_y=False)      # v15: without the duplicate
        assert torch.equal(buf2, _C.halves_split(y, hscale, 2, piece)) and buf2.shape == (n, 2 * piece)
        assert torch.equal(buf2[:, :piece], ref[:, :piece]) and torch.equal(buf2[:, piece:], ref[:, 2 * piece:])


def test_skinny_gemm_against_fp64():
    """bot_skinny_gemm_f32 (bf16x6 MFMA products of fp32 operands split in registers) against fp64, next to the stock fp32 GEMM's
    error: both B layouts, accumulate, strided batches writing side by side, ragged m / n / k, tiny and huge magnitudes."""
    gen = torch.Generator(device=DEV).manual_seed(4)
    for m, k, n, mag in ((20000, 168, 250, 1.0), (4099, 250, 168, 1e-8), (1000, 256, 768, 1e4), (130, 7, 5, 1.0), (33, 40, 129, 1.0)):
        a = torch.randn(m, k, device=DEV, generator=gen) * mag
        w = torch.randn(n, k, device=DEV, generator=gen) * 0.1           # [n, k]: nn.Linear's layout
        ref = a.double() @ w.double().t()
        e32 = float(((a @ w.t()).double() - ref).abs().max() / ref.abs().max())
        for b_is_kn in (False, True):
            out = torch.full((m, n), float("nan"), device=DEV)
            _C.skinny_gemm(a, w.t().contiguous() if b_is_kn else w, b_is_kn=b_is_kn, out=out)
            e = float((out.double() - ref).abs().max() / ref.abs().max())
            print(f"skinny_gemm m={m} k={k} n={n} b_is_kn={b_is_kn}: err {e:.2e} (stock fp32 {e32:.2e})")
            assert e <= max(2e-6, 3 * e32)
        base = torch.randn(m, n, device=DEV, generator=gen) * mag
        out = base.clone()
        _C.skinny_gemm(a, w, b_is_kn=False, out=out, accumulate=True)
        assert float((out.double() - (base.double() + ref)).abs().max() / (ref.abs().max() + base.abs().max())) <= 3e-6
    # three heads: z [H, N, Fin] x W [H, D, Fin]^T written side by side into the columns of one [N, 768] matrix, on top of what is there
    H, N, Fin, D, P = 3, 5000, 168, 250, 768
    z = torch.randn(H, N, Fin, device=DEV, generator=gen)
    W = torch.randn(H, D, Fin, device=DEV, generator=gen) * 0.1
    out = torch.randn(N, P, device=DEV, generator=gen)
    want = out.double().clone()
    for i in range(H):
        want[:, i * D:(i + 1) * D] += z[i].double() @ W[i].double().t()
    _C.skinny_gemm(z, W, b_is_kn=False, out=out, accumulate=True, batch=H, strides=(N * Fin, D * Fin, D), m=N, n=D, k=Fin)
    assert float((out.double() - want).abs().max()) <= 2e-5 and torch.equal(out[:, H * D:].double(), want[:, H * D:])
    # row-strided A (a column slice of a wider matrix) whose rows are only 8-byte aligned, B stored [k, n]
    dx = torch.randn(N, P, device=DEV, generator=gen)
    dz = torch.empty(H, N, Fin, device=DEV)
    _C.skinny_gemm(dx, W, b_is_kn=True, out=dz, batch=H, strides=(D, D * Fin, N * Fin), m=N, n=Fin, k=D)
    for i in range(H):
        ref = dx[:, i * D:(i + 1) * D].double() @ W[i].double()
        assert float((dz[i].double() - ref).abs().max() / ref.abs().max()) <= 2e-6


def test_tn_gemm_against_fp64():
    """bot_tn_gemm_f32 (exact fp32 MFMA, reduction over the node rows, per-chunk partials added in order) against fp64: the two
    weight-gradient shapes of the aggregate-first layer incl. the strided per-head batch, ragged sizes, determinism."""
    gen = torch.Generator(device=DEV).manual_seed(6)
    for n, kx, ky in ((50001, 168, 768), (20000, 250, 168), (300, 5, 7), (4097, 33, 65)):
        x = torch.randn(n, kx, device=DEV, generator=gen)
        y = torch.randn(n, ky, device=DEV, generator=gen) * 1e-3
        ref = x.double().t() @ y.double()
        got = _C.tn_gemm(x, y)
        e, e32 = float((got.double() - ref).abs().max() / ref.abs().max()), float(((x.t() @ y).double() - ref).abs().max() / ref.abs().max())
        print(f"tn_gemm n={n} kx={kx} ky={ky}: err {e:.2e} (stock fp32 {e32:.2e})")
        assert got.shape == (kx, ky) and e <= max(2e-6, 3 * e32)
        assert torch.equal(got, _C.tn_gemm(x, y))
        assert torch.equal(_C.tn_gemm(y, x, transpose_out=True), _C.tn_gemm(y, x).t())
    H, N, Fin, D, P = 3, 30000, 168, 250, 768
    dx = torch.randn(N, P, device=DEV, generator=gen)
    z = torch.randn(H, N, Fin, device=DEV, generator=gen)
    got = _C.tn_gemm(dx, z, batch=H, strides=(D, N * Fin, 0), n=N, kx=D, ky=Fin)
    for i in range(H):
        ref = dx[:, i * D:(i + 1) * D].double().t() @ z[i].double()
        assert float((got[i].double() - ref).abs().max() / ref.abs().max()) <= 2e-6


def test_bench_line_contract():
    """`python bench.py` (small scale, few steps) prints ONE JSON object as the last stdout line with the driver's keys, the roofline
    and cpu_baseline objects, and — new in round 2 — the stock-fp32-GEMM timing of the same steps beside `value`."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--scale", "0.2",
                          "--cpu-steps", "1"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["unit"] == "edges/s" and line["dtype"] == "f32"
    assert line["value"] > 0 and line["ms_per_step"] > 0 and line["higher_is_better"] is True
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "kernel" in r
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["value"] > 0
    assert "workload" in line["config"] and "gemm" in line["config"]
    s = line["stock_fp32_gemm"]
    assert s is not None and s["ms_per_step"] > 0 and s["unit"] == "edges/s"
    assert line["parity"]["criterion"] == "abs" and line["parity"]["ok"] is True and "criterion_text" in line["parity"]
    # the other BASELINE configs carry cpu_baseline + parity too (round 3), computed on a bounded sample that the line names
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "cora", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["roofline"] is not None and line["cpu_baseline"]["value"] > 0 and "S-cora" in line["cpu_baseline"]["sample"]
    assert line["parity"]["ok"] is True and line["parity"]["criterion"] == "abs" and "sample" in line["parity"]


def test_bench_self_launch_chain_on_one_gpu():
    """`python bench.py --gpus N` starts its own ranks.  On the 1-GPU box the same chain — this process launches
    `python -m torch.distributed.run ... bench.py <same arguments>` as a child BEFORE touching the GPU, the rank initialises RCCL
    (watchdogged), runs the partitioned step, and the relay keeps rank 0's JSON line last — is exercised with `--self-launch
    --force-partitioned` at N = 1; and `--gpus 2` ends with a message and rc 2 instead of an assertion or a hang."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--self-launch", "--force-partitioned", "--steps", "2",
                          "--warmup", "1", "--scale", "0.1", "--cpu-baseline", "off"], capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "torch.distributed.run" in out.stderr                      # the launcher announced its child
    line = json.loads(out.stdout.strip().splitlines()[-1])
    part = line["config"]["partition"]
    assert line["n_gpus"] == 1 and part["n_ranks"] == 1 and part["rccl_version"] not in (None, "unknown") and len(part["ranks"]) == 1
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                             cwd=root, env=env)
        assert out.returncode == 2 and "needs 2 GPUs" in out.stderr and out.stdout.strip() == ""


# ---------------------------------------------------------------------------------------------- replays, at length
def _replay_case(kind, part_group=None, capture=True):
    """(eager step fn, captured step object, eager model, captured model) of one stack on a small graph of its BASELINE shape, every
    drop rate 0 and a FIXED label mask, both models from the same state.  `capture=False`: no hipGraph is built (the captured step object
    is None) - for callers that only want the eager step."""
    import copy
    import torch.nn.functional as F
    from bot_amd import nn as bnn, synth, train as T
    name, scale = {"arxiv": ("arxiv", 0.2), "cora": ("cora", 1.0), "reddit": ("reddit", 0.04), "arxiv-1rank": ("arxiv", 0.2)}[kind]
    ds = synth.make_dataset(name, device=DEV, seed=0, scale=scale)
    g, C = ds.graph, ds.n_classes
    g.create_formats_()
    mask = torch.rand(ds.train_idx.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(3)) < 0.5

    def make():
        torch.manual_seed(0)
        if name == "arxiv":
            m = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, n_layers=3, n_heads=3, n_hidden=64,
                        norm="batch", dropout=0.0, input_drop=0.0, attn_drop=0.0, linear=True).to(DEV)
            return m, torch.optim.RMSprop(m.parameters(), lr=0.002, capturable=True), dict(use_labels=True, loss="loge", n_classes=C)
        hid, layers = (16, 2) if name == "cora" else (256, 3)
        m = bnn.GCN(in_feats=ds.feat.shape[1], n_classes=C, n_hidden=hid, n_layers=layers, activation=F.relu,
                    norm="none" if name == "cora" else "batch", norm_adj="symm", dropout=0.0).to(DEV)
        return m, torch.optim.Adam(m.parameters(), lr=0.01, capturable=True), dict(use_labels=False, loss="logit", n_classes=C)

    m1, o1, kw = make()
    m2, o2, _ = make()
    m2.load_state_dict(copy.deepcopy(m1.state_dict()))
    if kind == "arxiv-1rank":
        from bot_amd import dist as bdist
        part = bdist.partition_dataset(ds, 0, 1, DEV, part_group)
        m1, m2 = bdist.wrap_model(m1, part_group), bdist.wrap_model(m2, part_group)

        def eager():
            return bdist.train_step(m1, part, o1, mask=mask, group=part_group, **kw)

        def body():
            m2.train()
            o2.zero_grad(set_to_none=True)
            loss, pred = bdist.forward_backward(m2, part, mask=mask, group=part_group, **kw)
            o2.step()
            return loss, pred
        cap = T.CapturedTrainStep(body, DEV, warmup=3) if capture else None
    else:
        def eager():
            return T.train_step(m1, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o1, mask=mask, **kw)
        cap = T.captured_train_step(m2, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o2, warmup=3, mask=mask, **kw) if capture else None
    return eager, cap, m1, m2


def init_one_rank_rccl():
    """A 1-rank RCCL process group over a TCP store on a free local port.  A port found free by bind(0) + close can be taken again before
    init_process_group listens on it (EADDRINUSE once in 32 tight-loop runs, round 6): retried on another port.  (A file store avoids the
    port but made the same tests take 50 - 180 s instead of 4: FileStore polls.)"""
    import socket
    import torch.distributed as dist
    last = None
    for _ in range(5):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        try:
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
            return
        except Exception as e:      # DistNetworkError (a RuntimeError): address already in use
            if "in use" not in str(e).lower() and "EADDRINUSE" not in str(e):
                raise
            last = e
    raise last


@pytest.fixture
def one_rank_rccl():
    import torch.distributed as dist
    init_one_rank_rccl()
    try:
        yield None
    finally:
        dist.destroy_process_group()


@pytest.mark.isolated
@pytest.mark.parametrize("kind", ["arxiv", "cora", "reddit", "arxiv-1rank"])
def test_captured_step_twelve_replays_bitwise(kind, request):
    """round 3 found torch's multi-workgroup reductions returning wrong sums from the FOURTH replay of a captured
    step on, and rerouted the three call sites that showed it.  This replays each capturable stack — the config-2 GAT, the config-1 and
    config-3 GCNs and the 1-rank partitioned form (RCCL collectives captured) — TWELVE times and compares the loss, the logits and
    EVERY parameter gradient with the eager step from the same state after every replay, bit for bit, then the parameters and
    buffers at the end: any reduction (or anything else) that goes wrong under replay shows up as the first differing tensor."""
    group = request.getfixturevalue("one_rank_rccl") if kind == "arxiv-1rank" else None
    eager, cap, m1, m2 = _replay_case(kind, group)
    assert cap.drained is (kind == "arxiv-1rank")       # a live RCCL group: the watchdog's list was emptied before the capture
    for _ in range(3):                    # the capture ran 3 warm-up steps on m2 (the capture pass itself does not execute)
        eager()
    for it in range(12):
        le, pe = eager()
        lc, pc = cap()
        torch.cuda.synchronize()
        assert torch.equal(lc, le), (kind, it, float(lc), float(le))
        assert torch.equal(pc, pe), (kind, it)
        for (k, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
            assert torch.equal(a.grad, b.grad), (kind, it, k, float((a.grad - b.grad).abs().max()))
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), (kind, k)
    from bot_amd import _C
    _C.SEED_OFFSET = None


@pytest.mark.parametrize("opt", ["bot", "torch"])
def test_side_stream_bitwise(opt):
    """bot_amd.side: the weight-gradient products on a second stream (event fork behind the input gradient's launch, joined when the
    backward pass ends) - same kernels, same operands, separate outputs: loss, logits, every gradient and every parameter after the
    optimizer step are bit for bit those of the serial order, over several steps (a race would show as a differing tensor: the steps
    keep reusing the freed operand blocks), on the config-2 stack at its real width (layer 0 aggregate-first on the grouped kernels)."""
    import copy
    import torch.nn.functional as F
    from bot_amd import nn as bnn, side, synth, train as T, optim as bopt
    from bot_amd.nn import fused
    ds = synth.make_dataset("arxiv", device=DEV, seed=0, scale=0.2)
    g, C = ds.graph, ds.n_classes
    g.create_formats_()

    def make():
        torch.manual_seed(0)
        m = bnn.GAT(dim_node=ds.feat.shape[1] + C, dim_edge=0, dim_output=C, activation=F.relu, n_layers=3, n_heads=3, n_hidden=250,
                    norm="batch", dropout=0.0, input_drop=0.0, attn_drop=0.0, linear=True).to(DEV)
        return m, (bopt.RMSprop if opt == "bot" else torch.optim.RMSprop)(m.parameters(), lr=0.002)
    m1, o1 = make()
    m2, o2 = make()
    m2.load_state_dict(copy.deepcopy(m1.state_dict()))
    kw = dict(use_labels=True, loss="loge", n_classes=C)
    was = side.ENABLED
    try:
        for it in range(6):
            mask = torch.rand(ds.train_idx.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(it)) < 0.5
            side.ENABLED = False
            f0 = side.FORKS
            l1, p1 = T.train_step(m1, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o1, mask=mask, **kw)
            assert side.FORKS == f0
            g1 = {k: v.grad.clone() for k, v in m1.named_parameters()}
            side.ENABLED = True
            l0c = fused.L0_CALLS
            l2, p2 = T.train_step(m2, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, o2, mask=mask, **kw)
            assert side.FORKS >= f0 + 3, "the side stream was not used"
            assert fused.L0_CALLS > l0c, "layer 0 did not take the grouped-halves form"
            torch.cuda.synchronize()
            assert torch.equal(l1, l2) and torch.equal(p1, p2), it
            # the split of a side-produced merged gradient ran on the side stream and autograd STOLE its results (a clone would have
            # been a main-stream kernel racing the side stream: what the first build of bot_amd.side did by holding a second reference)
            stolen = [k for k, v in m2.named_parameters() if v.grad is not None and v.grad.untyped_storage().data_ptr() in side.LAST_MARKS]
            assert any(k.startswith("convs.1.") for k in stolen), stolen
            for k, v in m2.named_parameters():
                assert torch.equal(g1[k], v.grad), (it, k, float((g1[k] - v.grad).abs().max()))
        for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
            assert torch.equal(a, b), k
    finally:
        side.ENABLED = was


def test_capture_beside_a_live_rccl_watchdog():
    """ROOT CAUSE of round 5's SIGABRT, as a test (bot_amd.train.drain_rccl_watchdog; tools/exp_capture_watchdog.py): eager collectives
    leave Works in ProcessGroupNCCL's watchdog list; on this HIP, hipEventQuery of their (eager, completed) end events raises
    hipErrorCapturedEvent while RCCL's stream is inside a capture, and the watchdog thread takes the process down.  The tool captures an
    async all-reduce whose overlap window is stretched across several watchdog periods right behind three eager all-reduces, each trial in
    its own process: with the product's drain in front of the capture every trial must finish; without it the trial is EXPECTED to die of
    SIGABRT (printed, not asserted: a later ROCm may stop refusing such events, which only makes the drain unnecessary)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "exp_capture_watchdog.py")
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    for trial in range(3):
        out = subprocess.run([sys.executable, tool, "--part", "pg", "--drain", "1"], capture_output=True, text=True, timeout=600, cwd=root)
        assert out.returncode == 0 and "finished without an abort" in out.stdout, (trial, out.returncode, out.stderr[-3000:])
    # the leg without the remedy is information, not a criterion (whatever it does, the product never runs it)
    out = subprocess.run([sys.executable, tool, "--part", "pg", "--drain", "0"], capture_output=True, text=True, timeout=600, cwd=root)
    if out.returncode == 0:
        print("without the drain the same capture survived: this stack no longer refuses eagerly recorded events")
    else:
        print("without the drain the same capture ended with rc %d (%s)" % (out.returncode, "hipErrorCapturedEvent in RCCL's watchdog: the hazard is present "
              "on this stack" if "last recorded in a capturing stream" in out.stderr else "another reason: " + out.stderr[-300:]))


@pytest.mark.isolated
@pytest.mark.parametrize("kind", ["arxiv", "cora", "reddit", "arxiv-1rank"])
def test_no_multi_workgroup_torch_reduction_in_capturable_steps(kind, request, tmp_path):
    """The other half of the same ADVICE item: whatever the root cause of the wrong replayed sums is, no capturable step may contain
    a torch reduction that spans several workgroups (the kind that stages partials behind a memset-zeroed semaphore): the kernel
    trace of one eager step of each stack — same launches as its captured form — must show `at::native::reduce_kernel` with a
    one workgroup per output (grid.y == 1) only; every N-sized sum goes through the library's fixed-order kernels (bot_colsum_f32 and friends)."""
    import json
    from torch.profiler import ProfilerActivity, profile
    group = request.getfixturevalue("one_rank_rccl") if kind == "arxiv-1rank" else None
    eager, _, m1, m2 = _replay_case(kind, group, capture=False)       # (round 5 built a capture here and threw it away: VERDICT r5 #1b)
    eager()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        eager()
        torch.cuda.synchronize()
    path = str(tmp_path / "trace.json")
    prof.export_chrome_trace(path)
    evs = [e for e in json.load(open(path))["traceEvents"] if e.get("cat") == "kernel"]
    assert len(evs) > 10, "no kernel records in the trace"
    red = [e for e in evs if "at::native::reduce_kernel" in e["name"]]
    with_grid = [e for e in red if "grid" in e.get("args", {})]
    # ATen/native/cuda/Reduce.cuh: grid = (output blocks, ctas_per_output); ctas_per_output > 1 <=> should_global_reduce() <=> partials in
    # a staging buffer + the memset-zeroed semaphore.  (grid.x > 1 alone is one workgroup per slice of outputs: no cross-workgroup sum.)
    multi = [(e["name"][:120], e["args"]["grid"]) for e in with_grid if int(e["args"]["grid"][1]) > 1]
    print(f"{kind}: {len(evs)} kernels, {len(red)} torch reduce_kernel launches ({len(with_grid)} with grid info), multi-workgroup: {multi}")
    assert len(with_grid) == len(red), "the trace carries no grid sizes: cannot tell single- from multi-workgroup reductions"
    assert not multi, multi
    from bot_amd import _C
    _C.SEED_OFFSET = None


def test_gemm_halves3_nt_kernel():
    """csrc/halves3.hip (round 4): the hand-written NT product of two halves operands — each operand half staged once, three MFMAs per
    fragment pair — against fp64 next to the hipBLASLt formulation of the same three products, on ragged shapes (row / column / k
    remainders, outputs narrower than a tile, odd output pitches incl. the [N, 750] input gradient's 8-byte rows and a strided `out`),
    bitwise run to run and bitwise equal to its own plain-loop build (same terms, same order), and refusing bad arguments."""
    from bot_amd import gemm

    def left(x, order):                        # a left operand in a GIVEN layout (gemm.split chooses by piece width)
        piece = (x.shape[1] + gemm.PIECE_ALIGN - 1) // gemm.PIECE_ALIGN * gemm.PIECE_ALIGN
        scale = _C.halves_scale(x)
        return gemm.Halves(_C.halves_split(x, scale, order, piece), scale, x.shape[0], x.shape[1], piece, order)

    gen = torch.Generator(device=DEV).manual_seed(11)
    for (m, K, P, ldc) in ((1000, 96, 300, None), (513, 750, 1536, None), (20000, 1536, 750, None), (4099, 64, 40, None), (777, 250, 257, 301),
                           (300, 33, 5, None), (9000, 750, 1536, 1540)):
        x = torch.randn(m, K, device=DEV, generator=gen) * 3
        x[:, ::5] = 0
        w = torch.randn(P, K, device=DEV, generator=gen) * 0.05
        xs, ws = left(x, 0), gemm.split(w, 1)
        ref = x.double() @ w.double().t()
        lib = _C.gemm_halves(xs.buf, ws.buf, gemm._alpha(xs, ws, P), trans_b=True)
        out = None
        if ldc is not None:
            backing = torch.full((m, ldc), 7.0, device=DEV)
            out = backing[:, :P]
        got = _C.gemm_halves3_nt(xs.buf, ws.buf, xs.scale, ws.scale, xs.piece, ws.piece, xs.piece, out=out)
        if ldc is not None:
            assert bool((backing[:, P:] == 7.0).all()), "wrote outside the output columns"
        again = _C.gemm_halves3_nt(xs.buf, ws.buf, xs.scale, ws.scale, xs.piece, ws.piece, xs.piece)
        plain = _C.gemm_halves3_nt(xs.buf, ws.buf, xs.scale, ws.scale, xs.piece, ws.piece, xs.piece, mode=32)
        sc = float(ref.abs().max())
        e, el = float((got.double() - ref).abs().max()) / sc, float((lib.double() - ref).abs().max()) / sc
        print(f"gemm_halves3_nt m={m} K={K} P={P}: err {e:.2e} (hipBLASLt, same operands: {el:.2e})")
        assert e <= max(4e-6, 1.5 * el), (m, K, P, e, el)
        assert torch.equal(got, again) and torch.equal(got, plain)
        # the same left operand WITHOUT its duplicate piece (order 2, v15): the same halves at another offset -> the same bits; and
        # gemm.split / gemm.mm_nt choose that layout by themselves for wide operands
        x2 = left(x, 2)
        assert x2.buf.shape == (m, 2 * xs.piece) and torch.equal(x2.buf[:, :xs.piece], xs.buf[:, :xs.piece])
        assert torch.equal(x2.buf[:, xs.piece:], xs.buf[:, 2 * xs.piece:])
        assert torch.equal(_C.gemm_halves3_nt(x2.buf, ws.buf, x2.scale, ws.scale, x2.piece, ws.piece, x2.piece, a2_off=x2.piece), got)
        auto = gemm.split(x, 0)
        assert auto.order == (2 if xs.piece >= gemm.NODUP_MIN_PIECE else 0) and gemm.LEFT_NODUP
        if auto.order == 2 or P >= gemm.NT_MIN_COLS:
            assert torch.equal(gemm.mm_nt(auto, ws), got)
    assert gemm.NT_KERNEL == "halves3"          # the default route of gemm.mm_nt (forward + input gradient of the merged projections)
    # --- TN: the weight gradient x^T d of two LEFT operands (192 x 192 tiles, transposing LDS reads, split-K, three LDS stages), against
    # fp64 next to the library formulation (batched chunk products + combine): ragged row counts (the last step of the last split is
    # zero-filled by the buffer descriptor), piece widths that are not multiples of the tile, narrow and wide results, one and eight splits
    assert gemm.TN_KERNEL == "halves3"
    for (n, K, P) in ((4099, 750, 1536), (20000, 1536, 750), (33, 64, 40), (50001, 168, 250), (40000, 100, 968), (169343 // 4, 750, 1536), (30011, 750, 240)):
        x = torch.randn(n, K, device=DEV, generator=gen) * 3
        d = torch.randn(n, P, device=DEV, generator=gen) * 1e-3
        d[:, ::3] = 0
        xs, ds = left(x, 0), left(d, 0)
        ref = x.double().t() @ d.double()
        got = _C.gemm_halves3_tn(xs.buf, ds.buf, xs.scale, ds.scale, xs.piece, ds.piece, K, P)
        again = _C.gemm_halves3_tn(xs.buf, ds.buf, xs.scale, ds.scale, xs.piece, ds.piece, K, P)
        gemm.TN_KERNEL = "lib"
        try:
            lib = gemm.tn(xs, ds)
        finally:
            gemm.TN_KERNEL = "halves3"
        # gemm.tn's routes: wide results on the plain kernel, narrow ones of >= 8192 rows on its grouped form (the same tile grid as a
        # list, 192 x 128 tiles, more row splits: other partial sums, so held to fp64 instead of bitwise), the rest on the library
        # formulation; operands with or without the duplicate piece give the same bits on every route
        big = xs.piece * ds.piece >= gemm.TN_MIN_OUT
        routed = gemm.tn(xs, ds)
        if big:
            assert torch.equal(routed, got)
        elif n >= 8192 and gemm._tn_tiles(K, P) is not None:
            er = float((routed.double() - ref).abs().max() / ref.abs().max())
            print(f"gemm.tn n={n} K={K} P={P} on the grouped kernel ({len(gemm._tn_tiles(K, P))} tiles): err {er:.2e}")
            assert er <= 4e-6 and not torch.equal(routed, lib)
        else:
            assert torch.equal(routed, lib)
        x2, d2 = left(x, 2), left(d, 2)
        for (a, b) in ((x2, d2), (x2, ds), (xs, d2)):
            assert torch.equal(gemm.tn(a, b), routed)
        sc = float(ref.abs().max())
        e, el = float((got.double() - ref).abs().max()) / sc, float((lib.double() - ref).abs().max()) / sc
        print(f"gemm_halves3_tn n={n} K={K} P={P}: err {e:.2e} (library formulation: {el:.2e})")
        assert got.shape == (K, P) and e <= max(4e-6, 3 * el), (n, K, P, e, el)
        assert torch.equal(got, again)
    with pytest.raises(_C.BotKernelError):
        _C.gemm_halves3_nt(xs.buf, ws.buf, xs.scale, ws.scale, xs.piece, ws.piece, 40)       # k not a multiple of 32


def test_stats_byproduct_of_grouped_nt():
    """ABI 17 / 19: BatchNorm's column partials as a by-product of the grouped NT launch that writes the layer output
    (bot_gemm_halves3_nt_grouped2_f32 `stats_*`) + bot_bn_stats_halves_partials_f32, against the pass form on the same output
    (bot_bn_stats_halves_f32): the output itself bit for bit the launch without statistics; per 256-row tile the pivot IS the tile's first
    stored row, the partial sums / extremes against torch on the stored values; mean / invstd / running statistics / the halves scale
    against the pass form and against fp64 - also for columns whose mean is 1e3 standard deviations from zero (round 5's zero
    pivot cancelled there), produced with the launch's own column-shift epilogue."""
    from bot_amd.nn import fused
    gen = torch.Generator(device=DEV).manual_seed(37)
    for (N, H, D, Fin) in ((20011, 3, 250, 168), (700, 2, 70, 40)):
        P2 = (H * D + 2 * H + 127) // 128 * 128
        FP, DP, g_fwd, _, _ = fused._l0_tables(H, D, Fin, P2, True, N)
        HD, KA = H * D, (1 + H) * FP
        A = (torch.randn(N, 2 * KA, device=DEV, generator=gen) * 50).half()
        A[:, KA:] *= 0.01
        B = (torch.randn(HD, 6 * FP, device=DEV, generator=gen) * 30 + 3).half()
        sa, sb = torch.tensor([4.0, 0.25], device=DEV), torch.tensor([8.0, 0.125], device=DEV)
        tiles = (N + 255) // 256
        plain = torch.zeros(N, P2, device=DEV)
        _C.gemm_halves3_nt_grouped(A, B, sa, sb, KA, 2 * FP, plain, g_fwd, FP // 32)
        sd0 = float(plain[:, :HD].double().std(0).mean())
        for far in (0.0, 1e3):              # column means `far` standard deviations away from zero
            shift = torch.zeros(P2, device=DEV)
            shift[:HD] = far * sd0 * (1 + torch.rand(HD, device=DEV, generator=gen))
            ref = torch.zeros(N, P2, device=DEV)
            _C.gemm_halves3_nt_grouped(A, B, sa, sb, KA, 2 * FP, ref, g_fwd, FP // 32, col_shift=shift)
            part, minmax = torch.full((tiles, 2, HD), 7.0, device=DEV), torch.full((tiles, 2, HD), 7.0, device=DEV)
            pivot = torch.full((tiles, HD), 7.0, device=DEV)
            out = torch.zeros(N, P2, device=DEV)
            _C.gemm_halves3_nt_grouped(A, B, sa, sb, KA, 2 * FP, out, g_fwd, FP // 32, col_shift=shift, stats=(part, minmax, pivot))
            assert torch.equal(out, ref)
            x = out[:, :HD]
            assert torch.equal(pivot, x[0::256])                # the tile's first stored row
            pad = torch.cat([x, x.new_full((tiles * 256 - N, HD), float("nan"))]).view(tiles, 256, HD)
            d64 = pad.double() - pivot.double().unsqueeze(1)
            s_ref, q_ref = torch.nansum(d64, 1), torch.nansum(d64 * d64, 1)
            big = q_ref.sqrt().max() * 16                      # a tile's sum of 256 terms: errors relative to the tile's 2-norm
            assert float((part[:, 0].double() - s_ref).abs().max() / big) < 1e-6
            assert float(((part[:, 1].double() - q_ref).abs() / q_ref.abs().clamp(min=1e-30)).max()) < 1e-5
            mn = torch.where(torch.isnan(pad), torch.full_like(pad, float("inf")), pad).min(1).values
            mx = torch.where(torch.isnan(pad), torch.full_like(pad, float("-inf")), pad).max(1).values
            assert torch.equal(minmax[:, 0], mn) and torch.equal(minmax[:, 1], mx)
            # finished statistics against the pass form and against fp64
            w, b = torch.randn(HD, device=DEV, generator=gen), torch.randn(HD, device=DEV, generator=gen)
            rm1, rv1, nb1 = torch.zeros(HD, device=DEV), torch.ones(HD, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
            rm2, rv2, nb2 = rm1.clone(), rv1.clone(), nb1.clone()
            m1, i1, h1 = _C.bn_stats_halves(x, 1e-5, 0.1, rm1, rv1, nb1, w, b, 0.25)
            m2, i2, h2 = _C.bn_stats_halves_partials(part, minmax, pivot, N, 1e-5, 0.1, rm2, rv2, nb2, w, b, 0.25)
            x64 = x.double()
            mu64, is64 = x64.mean(0), torch.rsqrt(x64.var(0, unbiased=False) + 1e-5)
            sd = (1.0 / i1).max()
            ulp = float(x.abs().max()) * 2.0 ** -23            # a mean cannot be closer to fp64 than the fp32 spacing at its magnitude
            assert float((m2.double() - mu64).abs().max()) <= max(2e-6 * float(sd), ulp), (far, float((m2.double() - mu64).abs().max()), ulp)
            assert float(((i2.double() - is64) / is64).abs().max()) < 2e-5, (far, float(((i2.double() - is64) / is64).abs().max()))
            assert float((m1 - m2).abs().max()) <= max(2e-6 * float(sd), 2 * ulp) and float(((i1 - i2) / i1).abs().max()) < 4e-5
            assert float((rm1 - rm2).abs().max()) <= max(2e-6 * float(sd), ulp) and float(((rv1 - rv2) / rv1).abs().max()) < 4e-5 and int(nb2) == 1
            assert torch.equal(h1, h2), (far, h1, h2)


def test_nt64_kernel_and_fragment_major_operand():
    """Round 5: the 128-byte-line form of the NT halves GEMM (csrc/halves3.hip gemm_halves3_nt64_kernel: two k-steps per iteration, wave tile
    256 x 32, the wave's weight fragments straight into registers) is bit for bit the 128 x 64-wave-tile kernel (mode bit 1024 forces that one)
    on ragged m / n / k, strided outputs, both left layouts and with the second scale; a fragment-major right operand
    (bot_halves_split_frag_f16) gives bit for bit the result of the row-major one, and matches its restatement (tests/_oracle_backend.py)."""
    from tests import _oracle_backend as OB
    from bot_amd import gemm
    gen = torch.Generator(device=DEV).manual_seed(31)
    OLD = 1024
    for (m, K, P) in ((1000, 96, 300), (513, 750, 1536), (20000, 1536, 750), (4099, 64, 40), (257, 128, 17), (3000, 250, 193)):
        x = torch.randn(m, K, device=DEV, generator=gen) * 3
        w = torch.randn(P, K, device=DEV, generator=gen) * 0.05
        ws = gemm.split(w, 1)
        frag = _C.halves_split_frag(w, ws.scale, ws.piece)
        assert torch.equal(frag.cpu(), OB.halves_split_frag(w.cpu(), ws.scale.cpu(), ws.piece)), (m, K, P)
        for order in (0, 2):
            piece = ws.piece
            sc = _C.halves_scale(x)
            xb = _C.halves_split(x, sc, order, piece)
            a2 = piece if order == 2 else 2 * piece
            old = _C.gemm_halves3_nt(xb, ws.buf, sc, ws.scale, piece, piece, piece, a2_off=a2, mode=OLD)
            new = _C.gemm_halves3_nt(xb, ws.buf, sc, ws.scale, piece, piece, piece, a2_off=a2)
            fr = _C.gemm_halves3_nt(xb, frag, sc, ws.scale, piece, piece, piece, a2_off=a2, b_frag=True, n=P)
            assert torch.equal(old, new) and torch.equal(old, fr), (m, K, P, order)
            ref = x.double() @ w.double().t()
            assert float((new.double() - ref).abs().max() / ref.abs().max()) < 3e-6
            if piece >= 64:
                s2 = torch.tensor([float(sc[0]) * 128, float(sc[1]) / 128], device=DEV)
                for split in (32, piece - 32):
                    d1 = _C.gemm_halves3_nt(xb, ws.buf, sc, ws.scale, piece, piece, piece, a2_off=a2, mode=OLD, scale_a2=s2, k_split=split)
                    d2 = _C.gemm_halves3_nt(xb, frag, sc, ws.scale, piece, piece, piece, a2_off=a2, scale_a2=s2, k_split=split, b_frag=True, n=P)
                    assert torch.equal(d1, d2), (m, K, P, order, split)
        big = torch.zeros(m, P + 6, device=DEV)
        o = big[:, 2:2 + P]
        _C.gemm_halves3_nt(xb, frag, sc, ws.scale, piece, piece, piece, a2_off=a2, out=o, b_frag=True, n=P)
        assert torch.equal(o, old) and bool((big[:, :2] == 0).all()) and bool((big[:, 2 + P:] == 0).all())
    assert gemm.split_right(torch.randn(1536, 750, device=DEV)).order == 3 and gemm.split_right(torch.randn(40, 100, device=DEV)).order == 1


def test_nt64_random_shapes():
    """Randomised differential test of the NT halves GEMM family (round 5): for 24 random (m, K, P, left layout, output pitch, second scale,
    right-operand layout) the 128-byte-line kernel - row-major and fragment-major weights - is bit for bit the 128 x 64 kernel and within
    3e-6 of the fp64 product of the same fp32 operands; nothing outside the output's columns is touched."""
    import random
    from bot_amd import gemm
    rng = random.Random(5)
    gen = torch.Generator(device=DEV).manual_seed(41)
    OLD = 1024
    for case in range(24):
        m = rng.choice([1, 15, 255, 256, 257, 1000, 4099, 9001])
        K = rng.choice([1, 31, 64, 100, 250, 500, 750])
        P = rng.choice([1, 17, 40, 191, 192, 256, 300, 750])
        order = rng.choice([0, 2])
        x = torch.randn(m, K, device=DEV, generator=gen) * 10 ** rng.uniform(-3, 3)
        x[:, ::3] *= 1e-3
        w = torch.randn(P, K, device=DEV, generator=gen) * 10 ** rng.uniform(-3, 1)
        ws = gemm.split(w, 1)
        piece = ws.piece
        sc = _C.halves_scale(x)
        xb = _C.halves_split(x, sc, order, piece)
        a2 = piece if order == 2 else 2 * piece
        frag = _C.halves_split_frag(w, ws.scale, piece)
        pad = rng.choice([0, 1, 2, 6])
        kw = {}
        if piece >= 64 and rng.random() < 0.5:
            f = 2.0 ** rng.randint(-12, 12)
            kw = dict(scale_a2=torch.tensor([float(sc[0]) * f, float(sc[1]) / f], device=DEV), k_split=32 * rng.randint(1, piece // 32 - 1))
        outs = []
        for mode, b, bf in ((OLD, ws.buf, False), (0, ws.buf, False), (0, frag, True)):
            big = torch.full((m, P + pad), 3.0, device=DEV)
            o = big[:, pad // 2:pad // 2 + P]
            _C.gemm_halves3_nt(xb, b, sc, ws.scale, piece, piece, piece, a2_off=a2, out=o, mode=mode, b_frag=bf, n=P, **kw)
            assert bool((big[:, :pad // 2] == 3.0).all()) and bool((big[:, pad // 2 + P:] == 3.0).all()), (case, mode)
            outs.append(o.clone())
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (case, m, K, P, order, kw.get("k_split"))
        if not kw:
            ref = x.double() @ w.double().t()
            assert float((outs[1].double() - ref).abs().max() / ref.abs().max().clamp(min=1e-300)) < 3e-6, (case, m, K, P)


def test_grouped_halves_kernels(golden):
    """v16, the aggregate-first layer's dense products (csrc/halves3.hip grouped forms, csrc/spmm.hip halves epilogue) each against the
    definition in include/bot_gnn.h evaluated in fp64 from the SAME fp16 operands (tests/_oracle_backend.py's restatement, accumulating in float64):
    (1) bot_spmm_bcast_halves_f16 == halves_split(order 2) of bot_spmm_bcast_f32's slab, bit for bit, zero padding included, long rows too;
    (2) nt_grouped in the forward arrangement ([x | z_h] [Wres_h | W_h]^T, 250-column groups on an [N, 768] output, two A column ranges)
        and the d z arrangement (per-head A columns, per-head B rows, per-head output slabs);
    (3) tn_grouped: per-head and merged weight gradients, plain and transposed blocks, ragged k_valid / p_valid, many splits."""
    from tests import _oracle_backend as OB
    from bot_amd import gemm
    from bot_amd.nn import fused
    gen = torch.Generator(device=DEV).manual_seed(23)
    # --- (1)
    s_, d_, n = golden.graph("g300")
    for chunk, (H, Fin, FP) in ((8, (3, 22, 64)), (4096, (3, 22, 64)), (8, (1, 7, 64)), (4096, (4, 64, 64)), (16, (2, 168, 192)), (4096, (3, 250, 256))):
        g = bot_amd.Graph(s_, d_, n, chunk=chunk).to(DEV)
        x = torch.randn(n, Fin, device=DEV, generator=gen)
        w = torch.rand(g.number_of_edges(), H, device=DEV, generator=gen)
        scale = _C.halves_scale(x * 40)
        KA = (1 + H) * FP
        A = torch.full((n, 2 * KA), 7.0, dtype=torch.float16, device=DEV)
        _C.spmm_bcast_halves(g.csc, x, w, None, scale, A, FP, FP, KA, FP)
        z = _C.spmm_bcast(g.csc, x, w, None, head_outer=True)
        for h in range(H):
            ref = _C.halves_split(z[h], scale, 2, FP)
            assert torch.equal(A[:, FP * (1 + h):FP * (2 + h)], ref[:, :FP]) and torch.equal(A[:, KA + FP * (1 + h):KA + FP * (2 + h)], ref[:, FP:]), (chunk, H, Fin)
        assert bool((A[:, :FP] == 7.0).all()) and bool((A[:, KA:KA + FP] == 7.0).all())       # x's columns are not the SpMM's
    # --- (2) + (3) at a shape with every raggedness of config 2: H = 3, D = 250, Fin = 168, N not a multiple of anything
    for (N, H, D, Fin, kp) in ((20011, 3, 250, 168, True), (5000, 2, 70, 40, False)):
        P2 = (H * D + 2 * H + 127) // 128 * 128
        FP, DP, g_fwd, g_dz, t_tn = fused._l0_tables(H, D, Fin, P2, kp, N)
        HD, KA = H * D, (1 + H) * FP
        A = (torch.randn(N, 2 * KA, device=DEV, generator=gen) * 50).half()
        A[:, KA:] *= 0.01
        B = torch.randn(HD, 6 * FP, device=DEV, generator=gen) * 30
        B[:, 2 * FP:] /= 2048                                  # a right operand's second half is a rounding remainder (the product drops a2 b2)
        B = B.half()
        sa, sb = torch.tensor([4.0, 0.25], device=DEV), torch.tensor([8.0, 0.125], device=DEV)
        out = torch.full((N, P2), 3.0, device=DEV)
        _C.gemm_halves3_nt_grouped(A, B, sa, sb, KA, 2 * FP, out, g_fwd, FP // 32)
        assert bool((out[:, HD:] == 3.0).all()), "wrote outside the groups' columns"
        ref = torch.zeros(N, P2, dtype=torch.float64)
        OB.gemm_halves3_nt_grouped(A.cpu(), B.cpu(), sa.cpu(), sb.cpu(), KA, 2 * FP, ref, g_fwd, FP // 32)
        e = float((out[:, :HD].cpu() - ref[:, :HD]).abs().max() / ref[:, :HD].abs().max())
        print(f"nt_grouped forward N={N} H={H} D={D} Fin={Fin}: vs restatement {e:.2e}")
        assert e < 2e-6
        # fp64 from the same halves, head 1
        a64 = lambda c0, k: A[:, c0:c0 + k].double() + A[:, KA + c0:KA + c0 + k].double() / 2048
        b64 = lambda r, c0, k: B[r, c0:c0 + k].double() + B[r, 2 * FP + c0:2 * FP + c0 + k].double()
        rows = slice(D, 2 * D)
        r64 = (a64(0, FP) @ b64(rows, 0, FP).t() + a64(2 * FP, FP) @ b64(rows, FP, FP).t()) * (0.25 * 0.125)
        assert float((out[:, D:2 * D].double() - r64).abs().max() / r64.abs().max()) < 3e-6       # (the dropped a2 b2 term: 2^-22)
        again = torch.empty_like(out)
        _C.gemm_halves3_nt_grouped(A, B, sa, sb, KA, 2 * FP, again, g_fwd, FP // 32)
        assert torch.equal(again[:, :HD], out[:, :HD])
        # d z
        Dh = (torch.randn(N, 2 * H * DP, device=DEV, generator=gen) * 20).half()
        Wt = torch.randn(H * Fin, 3 * DP, device=DEV, generator=gen) * 10
        Wt[:, DP:] /= 2048
        Wt = Wt.half()
        dz = torch.full((H, N, Fin), 5.0, device=DEV)
        _C.gemm_halves3_nt_grouped(Dh, Wt, sa, sb, H * DP, DP, dz[0], g_dz, 0)
        ref = torch.zeros(H, N, Fin, dtype=torch.float64)
        OB.gemm_halves3_nt_grouped(Dh.cpu(), Wt.cpu(), sa.cpu(), sb.cpu(), H * DP, DP, ref[0], g_dz, 0)
        e = float((dz.cpu() - ref).abs().max() / ref.abs().max())
        print(f"nt_grouped d z: vs restatement {e:.2e}")
        assert e < 2e-6
        # weight gradients
        flat = torch.full((HD * Fin + P2 * Fin,), 9.0, device=DEV)
        _C.gemm_halves3_tn_grouped(A, Dh, sb, sa, KA, H * DP, flat, t_tn)
        ref = torch.full((HD * Fin + P2 * Fin,), 9.0, dtype=torch.float64)
        OB.gemm_halves3_tn_grouped(A.cpu(), Dh.cpu(), sb.cpu(), sa.cpu(), KA, H * DP, ref, t_tn)
        dW, dWr = flat[:HD * Fin], flat[HD * Fin:].view((Fin, P2) if kp else (P2, Fin))
        rW, rWr = ref[:HD * Fin], ref[HD * Fin:].view((Fin, P2) if kp else (P2, Fin))
        e1 = float((dW.cpu() - rW).abs().max() / rW.abs().max())
        res = (dWr[:, :HD] if kp else dWr[:HD]).cpu(), (rWr[:, :HD] if kp else rWr[:HD])
        e2 = float((res[0] - res[1]).abs().max() / res[1].abs().max())
        print(f"tn_grouped: d W {e1:.2e}  d Wres {e2:.2e} vs restatement")
        assert e1 < 3e-6 and e2 < 3e-6
        tail = (dWr[:, HD:] if kp else dWr[HD:])
        assert bool((tail == 9.0).all()), "wrote outside the tiles' blocks"
        flat2 = torch.empty_like(flat)
        _C.gemm_halves3_tn_grouped(A, Dh, sb, sa, KA, H * DP, flat2, t_tn)
        assert torch.equal(flat2[:HD * Fin], dW)
        # the 192 x 192 tile form of the same launch (a tile list with a d block wider than 128 columns selects it): x-role = the gradient
        # operand's head blocks in 192-column tiles, d-role = [x | z_h]
        t192 = []
        for h in range(H):
            for j in range((D + 191) // 192):
                kv = min(192, D - 192 * j)
                t192.append((h * DP + 192 * j, kv, FP * (1 + h), Fin, (h * D + 192 * j) * Fin, Fin, 0))
        big = torch.full((HD * Fin,), 9.0, device=DEV)
        _C.gemm_halves3_tn_grouped(Dh, A, sa, sb, H * DP, KA, big, t192)
        e3 = float((big.cpu() - rW).abs().max() / rW.abs().max())
        print(f"tn_grouped, 192 x 192 tiles: d W {e3:.2e}")
        assert e3 < 3e-6
    with pytest.raises(_C.BotKernelError):
        _C.gemm_halves3_nt_grouped(A, B, sa, sb, KA, 2 * FP, out, [(0, 300, 0, 0, 1, 0)], 0)        # a group wider than a tile
    # --- bot_halves_split_heads_f16 == H calls of halves_split_cols(order 2) on column slices (a row-strided source, odd D too)
    for (N, H, D, DP, ld) in ((5003, 3, 250, 256, 768), (1000, 2, 7, 64, 20), (300, 4, 64, 64, 256)):
        src = torch.randn(N, ld, device=DEV, generator=gen)[:, :H * D]
        sc = _C.halves_scale(src)
        got = _C.halves_split_heads(src, sc, H, D, DP)
        ref = torch.full((N, 2 * H * DP), 1.0, dtype=torch.float16, device=DEV)
        for h in range(H):
            _C.halves_split_cols(src[:, h * D:(h + 1) * D], sc, 2, ref, H * DP, h * DP, DP)
        assert torch.equal(got, ref)
    # --- the BatchNorm backward that writes dx as the head-padded halves operand under a BOUNDED scale (reduce_max -> bound -> apply_halves):
    # the bound holds (and is within 64x of max|dx|), the fp32 dx next to it is bit for bit bn_act_bwd_apply's, the halves are
    # halves_split_heads of that dx at that scale, the padding columns stay as the caller left them; training and eval statistics
    for (N, H, D, DP, p_drop, train) in ((20011, 3, 250, 256, 0.5, True), (5000, 2, 6, 64, 0.0, True), (3000, 3, 250, 256, 0.25, False)):
        Fc = H * D
        x = torch.randn(N, Fc, device=DEV, generator=gen) * 2 + 0.3
        dy = torch.randn(N, Fc, device=DEV, generator=gen) * 1e-3
        w, b = torch.randn(Fc, device=DEV, generator=gen), torch.randn(Fc, device=DEV, generator=gen)
        mean, invstd = _C.bn_stats(x, 1e-5, 0.0)
        sg, sgx, ws = _C.bn_act_bwd_reduce(dy, x, mean, invstd, w, b, True, p_drop, 77, want_max=True)
        sg2, sgx2 = _C.bn_act_bwd_reduce(dy, x, mean, invstd, w, b, True, p_drop, 77)
        assert torch.equal(sg, sg2) and torch.equal(sgx, sgx2)
        slots = _C.absmax_slots(DEV)
        _C.bn_bwd_bound(ws, N, sg if train else None, sgx if train else None, N, w, invstd, slots)
        ref = _C.bn_act_bwd_apply(dy, x, mean, invstd, w, b, True, p_drop, 77, sg if train else None, sgx if train else None, N)
        bound, true = float(slots.max().reshape(1).view(torch.float32)[0]), float(ref.abs().max())
        print(f"bn_bwd_bound N={N} F={Fc} train={train}: bound {bound:.3e} max|dx| {true:.3e} (x{bound / true:.1f})")
        assert true <= bound <= 64 * true
        sc = _C.halves_scale_from_slots(slots)
        hout = torch.full((N, 2 * H * DP), 3.0, dtype=torch.float16, device=DEV)
        dx32 = torch.empty(N, Fc, device=DEV)
        _C.bn_act_bwd_apply_halves(dy, x, mean, invstd, w, b, True, p_drop, 77, sg if train else None, sgx if train else None, N, sc, hout, D, DP, out=dx32)
        assert torch.equal(dx32, ref)
        want = _C.halves_split_heads(ref, sc, H, D, DP)
        for h in range(H):
            for off in (0, H * DP):
                blk = slice(off + h * DP, off + h * DP + D)
                assert torch.equal(hout[:, blk], want[:, blk])
                assert bool((hout[:, off + h * DP + D:off + (h + 1) * DP] == 3.0).all())
        only = torch.full_like(hout, 3.0)
        _C.bn_act_bwd_apply_halves(dy, x, mean, invstd, w, b, True, p_drop, 77, sg if train else None, sgx if train else None, N, sc, only, D, DP)
        assert torch.equal(only, hout)
    # --- bot_tn_narrow_f32: the attention columns of the merged gradient (a column slice of the gradient buffer) against the input
    for (N, kx, ky, tr) in ((20011, 18, 168, True), (777, 1, 5, False), (5000, 32, 256, False)):
        buf = torch.randn(N, kx + 7, device=DEV, generator=gen)
        y = torch.randn(N, ky, device=DEV, generator=gen)
        dst = torch.full((ky + 2, kx + 3) if tr else (kx + 2, ky + 3), 4.0, device=DEV)
        view = dst[:ky, 3:] if tr else dst[:kx, :ky]
        _C.tn_narrow(buf[:, 7:], y, view, transpose_out=tr)
        ref = buf[:, 7:].double().t() @ y.double()
        e = float((view.double() - (ref.t() if tr else ref)).abs().max() / ref.abs().max())
        print(f"tn_narrow n={N} kx={kx} ky={ky}: err {e:.2e}")
        assert e < 2e-6 and int((dst == 4.0).sum()) == dst.numel() - kx * ky
        again = torch.empty_like(view)
        _C.tn_narrow(buf[:, 7:], y, again, transpose_out=tr)
        assert torch.equal(again, view)


def test_abi17_kernels(golden):
    """ABI 17, each new entry point against its definition in include/bot_gnn.h (tests/_oracle_backend.py's restatement):
    (1) bot_halves_scale_from_slots2_f32: multiplier and cap; (2) bot_halves_tail_f16: segments with sources and zeros, bit for bit;
    (3) bot_spmm_dot_halves_f16 == halves_split(order 2) of bot_spmm_dot_f32's slab, bit for bit, and the same `dot`, long rows included;
    (4) bot_gemm_halves3_nt2_f32 / _tn2_f32: equal to the one-scale products when both scales agree (bit for bit), and against fp64 from
        the same fp16 operands when the tail's scale differs by 2^7 / 2^-9 (the accumulators rescaled by the exact ratio)."""
    from tests import _oracle_backend as OB
    from bot_amd import gemm
    gen = torch.Generator(device=DEV).manual_seed(29)
    # --- (1)
    slots = _C.absmax_slots(DEV)
    _C.absmax_into(torch.tensor([[0.3, -5.0, 1.0]], device=DEV), slots)
    cap = torch.tensor([2.0 ** 3, 2.0 ** -3], device=DEV)
    for mult, cp, ratio in ((None, None, 1.0), (700.0, None, 1.0), (1.0, cap, 4.0), (1e-6, cap, 2.0 ** 20), (3.0, cap, 2.0 ** 20)):
        got = _C.halves_scale_from_slots(slots, mult=mult, cap=cp, cap_ratio=ratio).cpu()
        ref = OB.halves_scale_from_slots(slots.cpu(), mult=mult, cap=None if cp is None else cp.cpu(), cap_ratio=ratio)
        assert torch.equal(got, ref), (mult, ratio, got, ref)
    # --- (2)
    n, ld, h2 = 1237, 2 * 96, 96
    out = torch.full((n, ld), 3.0, dtype=torch.float16, device=DEV)
    a, b = torch.randn(n, 3, device=DEV, generator=gen) * 5, torch.randn(n, 8, device=DEV, generator=gen)[:, 1:4]
    sc = torch.tensor([64.0, 1 / 64.0], device=DEV)
    segs = [(64, 3, a), (67, 3, b), (30, 2, None), (70, 26, None)]
    _C.halves_tail(segs, sc, out, h2)
    ref = torch.full((n, ld), 3.0, dtype=torch.float16)
    OB.halves_tail([(c, w, None if t is None else t.cpu()) for c, w, t in segs], sc.cpu(), ref, h2)
    assert torch.equal(out.cpu(), ref)
    # --- (3)
    s_, d_, nn = golden.graph("g300")
    for chunk, (H, D) in ((8, (3, 250)), (4096, (3, 250)), (8, (2, 64)), (4096, (4, 40)), (16, (3, 48))):
        g = bot_amd.Graph(s_, d_, nn, chunk=chunk).to(DEV)
        E = g.number_of_edges()
        x = torch.randn(nn, H, D, device=DEV, generator=gen)
        y = torch.randn(nn, H, D, device=DEV, generator=gen)
        w = torch.rand(E, H, device=DEV, generator=gen)
        piece = (H * D + 63) // 64 * 64
        buf = torch.full((nn, 2 * piece), 7.0, dtype=torch.float16, device=DEV)
        scale = _C.halves_scale(x.reshape(nn, -1) * 300)
        assert _C.spmm_dot_halves_fits(x, y, buf, D, piece)
        dot = _C.spmm_dot_halves(g.csr, x, w, g.csr2csc, y, scale, buf, D, piece)
        o32, dot32 = _C.spmm_dot(g.csr, x, w, g.csr2csc, y)
        ref = _C.halves_split(o32.reshape(nn, H * D), scale, 2, piece)
        assert torch.equal(dot, dot32), (chunk, H, D)
        assert torch.equal(buf[:, :H * D], ref[:, :H * D]) and torch.equal(buf[:, piece:piece + H * D], ref[:, piece:piece + H * D]), (chunk, H, D)
        assert bool((buf[:, H * D:piece] == 7.0).all())
    assert not _C.spmm_dot_halves_fits(x[:, :1], y[:, :1], buf, 16, piece)        # one head: the head-major kernel's shape
    # --- (4)
    for (N, K, P, split) in ((20011, 1536, 750, 1504), (5003, 384, 128, 256)):
        A = (torch.randn(N, 2 * K, device=DEV, generator=gen) * 50).half()
        A[:, K:] *= 0.01
        Bm = torch.randn(P, 3 * K, device=DEV, generator=gen) * 30
        Bm[:, K:2 * K] /= 2048
        Bm = Bm.half()
        X = (torch.randn(N, 2 * 192, device=DEV, generator=gen) * 20).half()
        sa, sb, sx = (torch.tensor([v, 1 / v], device=DEV) for v in (4.0, 8.0, 2.0))
        one = _C.gemm_halves3_nt(A, Bm, sa, sb, K, K, K, a2_off=K)
        same = _C.gemm_halves3_nt(A, Bm, sa, sb, K, K, K, a2_off=K, scale_a2=sa, k_split=split)
        assert torch.equal(one, same)
        t_one = _C.gemm_halves3_tn(X, A, sx, sa, 192, K, 168, K - 3, x2_off=192, d2_off=K)
        t_same = _C.gemm_halves3_tn(X, A, sx, sa, 192, K, 168, K - 3, x2_off=192, d2_off=K, scale_d2=sa, p_split=split)
        assert torch.equal(t_one, t_same)
        for v2 in (4.0 * 2 ** 7, 4.0 * 2 ** -9):
            s2 = torch.tensor([v2, 1 / v2], device=DEV)
            got = _C.gemm_halves3_nt(A, Bm, sa, sb, K, K, K, a2_off=K, scale_a2=s2, k_split=split)
            a64 = (A[:, :K].double() + A[:, K:].double() / 2048)
            a64[:, :split] /= 4.0
            a64[:, split:] /= v2
            b64 = (Bm[:, :K].double() + Bm[:, K:2 * K].double()) / 8.0
            r64 = a64 @ b64.t()
            e = float((got.double() - r64).abs().max() / r64.abs().max())
            assert e < 3e-6, (N, v2, e)
            assert torch.equal(got, _C.gemm_halves3_nt(A, Bm, sa, sb, K, K, K, a2_off=K, scale_a2=s2, k_split=split))
            tg = _C.gemm_halves3_tn(X, A, sx, sa, 192, K, 168, K - 3, x2_off=192, d2_off=K, scale_d2=s2, p_split=split)
            x64 = (X[:, :168].double() + X[:, 192:192 + 168].double() / 2048) / 2.0
            t64 = x64.t() @ a64[:, :K - 3]
            te = float((tg.double() - t64).abs().max() / t64.abs().max())
            assert te < 3e-6, (N, v2, te)
            # the restatement agrees (it is what the CPU suite runs the layer on)
            ro = OB.gemm_halves3_nt(A.cpu(), Bm.cpu(), sa.cpu(), sb.cpu(), K, K, K, a2_off=K, scale_a2=s2.cpu(), k_split=split)
            assert float((got.cpu() - ro).abs().max() / ro.abs().max()) < 3e-6


def test_grouped_halves_kernels_random_lists():
    """Random group / tile lists for the two grouped halves kernels (the layer only ever builds the regular ones of fused._l0_tables and
    gemm._tn_tiles): ragged n_valid / k_valid / p_valid, odd output offsets (the float4 / float2 / scalar store paths), every k_seg
    position, one to twelve groups, one to sixteen tiles of both tile widths, few and many rows (one split / many) — each against
    tests/_oracle_backend.py's float64 restatement from the same fp16 operands, and untouched output around the blocks."""
    import random
    from tests import _oracle_backend as OB
    rnd = random.Random(5)
    gen = torch.Generator(device=DEV).manual_seed(29)
    sa, sb = torch.tensor([2.0, 0.5], device=DEV), torch.tensor([0.25, 4.0], device=DEV)
    for trial in range(24):
        m = rnd.choice([1, 37, 256, 300, 1000, 4099])
        ka, kb = rnd.choice([64, 192, 384]), rnd.choice([64, 128, 384])
        a2, b2 = ka + rnd.choice([0, 64]), kb + rnd.choice([0, 32])
        A = (torch.randn(m, a2 + ka, device=DEV, generator=gen) * 9).half()
        nb = rnd.choice([5, 100, 256, 700])
        B = torch.randn(nb, b2 + kb, device=DEV, generator=gen) * 5
        B[:, b2:] /= 2048
        B = B.half()
        ld = rnd.choice([1030, 1031, 1032])
        out = torch.full((m, ld), 6.0, device=DEV)
        groups, col = [], rnd.choice([0, 1, 2, 3])
        k_seg = rnd.choice([0, 1, 2])
        for _ in range(rnd.randint(1, 12)):
            steps = rnd.randint(1, min(ka, kb) // 32)
            nv = rnd.choice([1, 3, 4, 60, 250, 256])
            if col + nv > ld or len(groups) == 12:
                break
            a0 = 8 * rnd.randint(0, (ka - 32 * min(k_seg, steps)) // 8) if k_seg else 0
            a1 = 8 * rnd.randint(0, (ka - 32 * steps) // 8)
            groups.append((rnd.randint(0, nb - 1), nv, a0, a1, steps, col))
            col += nv + rnd.choice([0, 1, 5])
        if not groups:
            continue
        _C.gemm_halves3_nt_grouped(A, B, sa, sb, a2, b2, out, groups, k_seg)
        ref = torch.full((m, ld), 6.0, dtype=torch.float64)
        # the restatement reads B rows b_row0 .. b_row0 + n_valid - 1; rows past the end of B repeat its last row (the kernel clamps)
        Bc = torch.cat([B.cpu(), B.cpu()[-1:].expand(256, -1)])
        OB.gemm_halves3_nt_grouped(A.cpu(), Bc, sa.cpu(), sb.cpu(), a2, b2, ref, groups, k_seg)
        err = float((out.cpu().double() - ref).abs().max() / max(1e-30, float(ref.abs().max())))
        assert err < 3e-6, (trial, err, groups, k_seg)
        # TN
        n = rnd.choice([33, 1000, 5000, 20011])
        kx, kd = rnd.choice([192, 448]), rnd.choice([128, 192, 320])
        x2, d2 = kx + rnd.choice([0, 64]), kd + rnd.choice([0, 64])
        X = (torch.randn(n, x2 + kx, device=DEV, generator=gen) * 7).half()
        Dm = (torch.randn(n, d2 + kd, device=DEV, generator=gen) * 3).half()
        pt = rnd.choice([128, 192])
        tiles, off = [], rnd.choice([0, 1, 2])
        for _ in range(rnd.randint(1, 16)):
            kv, pv = rnd.choice([1, 58, 168, 192]), rnd.choice([1, 4, 122, 128] if pt == 128 else [5, 168, 192])
            if kv > kx or pv > kd:
                continue
            tr = rnd.choice([0, 1])
            ldo = (kv if tr else pv) + rnd.choice([0, 3])
            tiles.append((8 * rnd.randint(0, (kx - kv) // 8), kv, 8 * rnd.randint(0, (kd - pv) // 8), pv, off, ldo, tr))
            off += (pv if tr else kv) * ldo + rnd.choice([0, 7])
        if not tiles:
            continue
        flat = torch.full((off + 8,), 2.0, device=DEV)
        _C.gemm_halves3_tn_grouped(X, Dm, sa, sb, x2, d2, flat, tiles)
        ref = torch.full((off + 8,), 2.0, dtype=torch.float64)
        OB.gemm_halves3_tn_grouped(X.cpu(), Dm.cpu(), sa.cpu(), sb.cpu(), x2, d2, ref, tiles)
        err = float((flat.cpu().double() - ref).abs().max() / max(1e-30, float(ref.abs().max())))
        assert err < 3e-6, (trial, err, tiles)


def test_step_glue_kernels():
    """include/bot_gnn.h v14 "the train step's glue" (csrc/step.hip), each kernel against the tensor ops it replaces (run.py:229-267,
    :240-243, models.py:711, torch.optim.RMSprop), then one whole fused train step against the tensor-op form of the same step."""
    import math
    import torch.nn.functional as F
    from bot_amd import nn as bnn, optim as boptim, synth, train as T
    gen = torch.Generator(device=DEV).manual_seed(5)
    n, Fin, C = 5003, 7, 5
    feat = torch.randn(n, Fin, device=DEV, generator=gen)
    labels = torch.randint(0, C, (n, 1), device=DEV, generator=gen)
    tr = torch.randperm(n, device=DEV, generator=gen)[:2700]
    mask = torch.rand(2700, device=DEV, generator=gen) < 0.5
    # --- label_split with a given mask, both modes
    for use_labels in (True, False):
        code = torch.full((n,), -1, dtype=torch.int32, device=DEV)
        wn = torch.zeros(n, device=DEV)
        cnt = _C.label_split(tr, labels, mask, 0.5, 0, use_labels, code if use_labels else None, wn)
        w_ref = torch.zeros(n, device=DEV)
        w_ref[tr] = ((~mask) if use_labels else mask).float()
        assert torch.equal(wn, w_ref) and float(cnt) == float(w_ref.sum())
        if use_labels:
            c_ref = torch.full((n,), -1, dtype=torch.int32, device=DEV)
            c_ref[tr[mask]] = labels[tr[mask], 0].int()
            assert torch.equal(code, c_ref)
    # --- its own random split: rate, reproducibility, a different stream per seed
    outs = []
    for seed in (11, 11, 12):
        code = torch.full((n,), -1, dtype=torch.int32, device=DEV)
        wn = torch.zeros(n, device=DEV)
        cnt = _C.label_split(tr, labels, None, 0.3, seed, True, code, wn)
        outs.append((code.clone(), float(cnt)))
    assert torch.equal(outs[0][0], outs[1][0]) and not torch.equal(outs[0][0], outs[2][0])
    assert abs(outs[0][1] / 2700 - 0.7) < 0.04 and int((outs[0][0] >= 0).sum()) + int(outs[0][1]) == 2700
    # --- build_input: exact without dropout; dropout rate / scaling / reproducibility with it
    code = torch.full((n,), -1, dtype=torch.int32, device=DEV)
    code[tr[mask]] = labels[tr[mask], 0].int()
    ref = T.add_labels(feat, labels, tr[mask], C)
    assert torch.equal(_C.build_input(feat, code, C, 0.0, 0), ref)
    assert torch.equal(_C.build_input(feat, None, 0, 0.0, 0), feat)
    a, b, c = (_C.build_input(feat, code, C, 0.25, s) for s in (3, 3, 4))
    assert torch.equal(a, b) and not torch.equal(a, c)
    kept = a != 0
    assert abs(float(kept[:, :Fin].float().mean()) - 0.75) < 0.01
    assert torch.allclose(a[kept], (ref / 0.75)[kept], rtol=1e-6, atol=0)
    # --- node_loss (three kinds) against autograd of the tensor-op form; labels outside the prediction set may be placeholders
    wn = torch.zeros(n, device=DEV)
    wn[tr[~mask]] = 1.0
    cnt = wn.sum().reshape(1)
    bad = labels.clone()
    bad[wn == 0] = -1
    for kind in ("logit", "loge", "savage"):
        x = (torch.randn(n, C, device=DEV, generator=gen) * 3).requires_grad_()
        y_ref = T.per_node_loss(x, labels, kind)
        (torch.where(wn > 0, y_ref, torch.zeros_like(y_ref)).sum() / cnt[0]).backward()
        y, dx = _C.node_loss(x.detach(), bad, wn, cnt, kind, T.EPSILON)
        assert y.numel() % 64 == 0 and not y[n:].any()
        assert torch.allclose(y[:n], torch.where(wn > 0, y_ref.detach(), torch.zeros_like(y_ref)), rtol=2e-5, atol=2e-6), kind
        assert float((dx - x.grad).abs().max()) <= 1e-5 * float(x.grad.abs().max()) + 1e-12, kind
    # --- rmsprop_step against torch.optim.RMSprop (ragged sizes, weight decay, warm-up learning rates)
    sizes = [(1,), (1023,), (1025,), (40, 7), (3, 250, 168), (64,)]
    for wd in (0.0, 0.01):
        p1 = [torch.randn(*s, device=DEV, generator=gen).requires_grad_() for s in sizes]
        p2 = [p.detach().clone().requires_grad_() for p in p1]
        o1, o2 = torch.optim.RMSprop(p1, lr=0.002, weight_decay=wd), boptim.RMSprop(p2, lr=0.002, weight_decay=wd)
        for epoch in range(1, 6):
            T.adjust_learning_rate(o1, 0.002, epoch), T.adjust_learning_rate(o2, 0.002, epoch)
            for a_, b_ in zip(p1, p2):
                g = torch.randn(a_.shape, device=DEV, generator=gen)
                a_.grad, b_.grad = g.clone(), g.clone()
            o1.step(), o2.step()
        for a_, b_ in zip(p1, p2):
            assert torch.allclose(a_, b_, rtol=1e-6, atol=1e-7)
        assert set(o2.state_dict()["state"][0]) == {"step", "square_avg"}
    with pytest.raises(NotImplementedError):
        boptim.RMSprop(p2, momentum=0.9)
    # --- one whole step: fused glue vs tensor ops (drop rates 0, fixed mask), GAT with labels and GCN without
    ds = synth.make_dataset("arxiv", device=DEV, seed=0, scale=0.05)
    g, Cc = ds.graph, ds.n_classes
    g.create_formats_()
    m_ = torch.rand(ds.train_idx.shape, device=DEV, generator=gen) < 0.5
    for kind in ("gat", "gcn"):
        res = []
        for fusedstep in (True, False):
            torch.manual_seed(0)
            if kind == "gat":
                model = bnn.GAT(dim_node=ds.feat.shape[1] + Cc, dim_edge=0, dim_output=Cc, activation=F.relu, n_layers=3, n_heads=3, n_hidden=32,
                                norm="batch", linear=True).to(DEV)
                kw = dict(use_labels=True, loss="loge", n_classes=Cc)
            else:
                model = bnn.GCN(in_feats=ds.feat.shape[1], n_classes=Cc, n_hidden=32, n_layers=2, activation=F.relu, norm="batch").to(DEV)
                kw = dict(use_labels=False, loss="logit", n_classes=Cc)
            opt = boptim.RMSprop(model.parameters(), lr=0.002)
            T.FUSED_STEP = fusedstep
            try:
                loss, pred = T.train_step(model, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, opt, mask=m_, **kw)
            finally:
                T.FUSED_STEP = True
            res.append((loss.detach(), pred.detach(), [p.grad.clone() for p in model.parameters()]))
        assert abs(float(res[0][0]) - float(res[1][0])) <= 2e-6 * max(1.0, abs(float(res[1][0]))), kind
        assert torch.equal(res[0][1], res[1][1]), kind                           # same forward: same operand, no dropout
        for ga, gb in zip(res[0][2], res[1][2]):
            assert float((ga - gb).abs().max()) <= 1e-5 * float(gb.abs().max()) + 1e-12, kind
        # (post-step parameters are not compared here: the FIRST RMSprop update is lr * g / (0.1 |g| + eps), a sign function of
        # entries near zero; the update kernel is held to torch.optim.RMSprop on identical gradients above)
    # with the reference's input dropout the fused step draws its own (Philox) mask: finite loss, a different input every step
    torch.manual_seed(1)
    model = bnn.GAT(dim_node=ds.feat.shape[1] + Cc, dim_edge=0, dim_output=Cc, activation=F.relu, n_layers=2, n_heads=2, n_hidden=16, norm="batch",
                    input_drop=0.25, dropout=0.5, linear=True).to(DEV)
    opt = boptim.RMSprop(model.parameters(), lr=0.002)
    ls = [float(T.train_step(model, g, ds.feat, ds.labels, ds.train_idx, ds.val_idx, ds.test_idx, opt, use_labels=True, loss="loge", n_classes=Cc)[0])
          for _ in range(3)]
    assert all(math.isfinite(v) for v in ls) and len(set(ls)) == 3
