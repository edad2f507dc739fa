"""World-size-2 (and 3) gloo tests of the 1-D vertex-partitioned mode on CPU: partition plan invariants, halo
exchange forward/backward, SyncBatchNorm, and a full GAT / GCN train step giving the same logits and
parameter gradients as the single-process step.  Kernels are emulated (tests/_oracle_backend.py); the
collectives, the partitioning and the autograd plumbing are the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests._golden import Golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_plan_invariants(golden):
    from bot_amd import dist as bdist
    s, d, n = golden.graph("g300")
    for world in (1, 2, 3, 8):
        bounds = bdist.partition_bounds(torch.bincount(d, minlength=n), world)
        assert bounds[0] == 0 and bounds[-1] == n and all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
        parts = [bdist.build_partition(s, d, n, r, world) for r in range(world)]
        assert sum(p.n_edges for p in parts) == s.numel()  # every in-edge lives on exactly one rank
        if world <= 3:  # balanced by in-edges (heavy-tailed tiny graph: loose bound)
            assert max(p.n_edges for p in parts) <= 1.5 * s.numel() / world + 64
        for p in parts:
            g = p.graph
            assert g.number_of_dst_nodes() == p.n_owned and g.number_of_nodes() == p.n_owned + p.halo_global.numel()
            ls, ld = g.edges()
            glob = torch.cat([torch.arange(p.lo, p.hi), p.halo_global])
            m = (d >= p.lo) & (d < p.hi)
            assert torch.equal(glob[ls], s[m]) and torch.equal(ld + p.lo, d[m])  # same edges, same order
            assert torch.all(p.halo_global[1:] > p.halo_global[:-1])
            assert sum(g.halo.recv_splits) == p.halo_global.numel() and g.halo.recv_splits[p.rank] == 0
        for p in parts:  # what p sends to q is exactly what q's halo expects from p, in the same order
            off = 0
            for q, cnt in enumerate(p.graph.halo.send_splits):
                rows = p.graph.halo.send_rows[off:off + cnt].long() + p.lo
                hq = parts[q].halo_global
                exp = hq[(hq >= p.lo) & (hq < p.hi)]
                assert torch.equal(rows, exp)
                off += cnt


def _worker(rank, world, port, kind, tmp, fuse=False, halves=False, overlap=True):
    import torch.distributed as dist
    import torch.nn.functional as F
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from tests import _oracle_backend
        _oracle_backend.install_direct()
        import bot_amd
        from bot_amd.nn import fused
        fused.FORCE = fuse
        from bot_amd import halo
        halo.OVERLAP = overlap                               # exchange overlapped with the owned-source sweeps, or the one-exchange form
        from bot_amd import gemm
        gemm.FORCE = halves                                  # the projections on the fp16-halves path (emulated), both runs
        from bot_amd import dist as bdist
        from bot_amd import nn as bnn
        from bot_amd import train as T
        s, d, n = Golden().graph("g300")
        C, fin = 5, 9
        gen = torch.Generator().manual_seed(7)
        feat = torch.randn(n, fin, generator=gen)
        labels = torch.randint(0, C, (n, 1), generator=gen)
        perm = torch.randperm(n, generator=gen)
        tr, va, te = perm[: n // 2], perm[n // 2: 3 * n // 4], perm[3 * n // 4:]
        mask_full = torch.rand(n, generator=gen) < 0.5  # per-node coin, so every rank draws the same split

        def make():
            torch.manual_seed(3)
            if kind == "gat":
                return bnn.GAT(dim_node=fin + C, dim_edge=0, dim_output=C, n_hidden=6, n_layers=3, n_heads=2,
                               activation=F.relu, norm="batch", non_interactive_attn=True, use_symmetric_norm=True, linear=True)
            if kind == "gat_plain":  # BASELINE config-2 options (no symmetric norm): eligible for the fused layer node
                return bnn.GAT(dim_node=fin + C, dim_edge=0, dim_output=C, n_hidden=16, n_layers=3, n_heads=3,
                               activation=F.relu, norm="batch", non_interactive_attn=True, linear=True)  # layer 0: 14 <= 16 -> aggregate-first
            return bnn.GCN(in_feats=fin + C, n_classes=C, n_hidden=16, n_layers=3, activation=F.relu, norm="batch",
                           norm_adj="symm", use_linear=True)

        # single-process reference step
        ref = make().train()
        g = bot_amd.Graph(s, d, n)
        loss_ref, pred_ref, _ = T.forward_backward(ref, g, feat, labels, tr, va, te, use_labels=True, loss="loge",
                                                   n_classes=C, mask=mask_full[tr])
        # partitioned step
        model = bdist.wrap_model(make().train())
        part = bdist.build_partition(s, d, n, rank, world)
        part.feat, part.labels = feat[part.lo:part.hi], labels[part.lo:part.hi]
        tr_own = tr[(tr >= part.lo) & (tr < part.hi)]
        part.train_idx = tr_own - part.lo
        calls0 = fused.CALLS
        loss, pred = bdist.forward_backward(model, part, use_labels=True, loss="loge", n_classes=C, mask=mask_full[tr_own])
        assert (fused.CALLS > calls0) == (fuse and kind == "gat_plain"), (fused.CALLS, calls0)
        assert (fused.AGG_CALLS > 0) == (fuse and kind == "gat_plain")
        # ... and that node ran its grouped-halves form (v16: halo rows in the gather table, sync-BatchNorm sums reduced across ranks before
        # the bound of its direct gradient operand)
        assert (fused.L0_CALLS > 0) == (fuse and kind == "gat_plain")
        assert (fused.OVERLAP_CALLS > 0) == (fuse and kind == "gat_plain" and overlap), fused.OVERLAP_CALLS
        # the modular layers (GraphConv, GATConv outside the fused node) take bot_amd.halo's overlapped sums
        modular = kind in ("gat", "gcn") or not fuse
        assert (halo.CALLS > 0) == (modular and overlap), (halo.CALLS, kind, fuse, overlap)
        assert not halves or gemm.STATS["split"] > 0
        assert abs(loss.item() - loss_ref.item()) < 1e-5, (loss.item(), loss_ref.item())
        np.testing.assert_allclose(pred.detach().numpy(), pred_ref.detach()[part.lo:part.hi].numpy(), rtol=1e-4, atol=1e-5)
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            np.testing.assert_allclose(p.grad.numpy(), q.grad.numpy(), rtol=2e-4, atol=2e-5 * max(1.0, q.grad.abs().max().item()), err_msg=k)
        for (k, b), (_, c) in zip(model.named_buffers(), ref.named_buffers()):
            np.testing.assert_allclose(b.numpy(), c.numpy(), rtol=1e-4, atol=1e-5, err_msg=k)  # BN running stats
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,world,fuse,halves,overlap", [("gat", 2, False, False, True), ("gcn", 2, False, False, True),
                                                            ("gat", 3, False, False, True), ("gat_plain", 2, True, False, True),
                                                            ("gat_plain", 3, True, False, True), ("gat_plain", 2, False, False, True),
                                                            ("gat_plain", 2, True, True, True), ("gat_plain", 3, True, False, False),
                                                            ("gat_plain", 1, True, False, True),    # one rank: no halo rows at all
                                                            ("gcn", 3, False, False, False), ("gat", 2, False, False, False),
                                                            ("gcn", 1, False, False, True)])
def test_partitioned_step_matches_single_process(kind, world, fuse, halves, overlap, tmp_path):
    """`overlap`: the merged-GEMM layers ship `el` first and the projected rows asynchronously, sweeping the owned-source edges
    meanwhile (bot_amd/halo.py; the modular layers through halo.copy_u_sum / halo.u_mul_e_sum) — per-destination sums then run owned-source edges first: same values to rounding;
    False: the one-exchange form."""
    mp.spawn(_worker, args=(world, _free_port(), kind, str(tmp_path), fuse, halves, overlap), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _worker_edge(rank, world, port, kind, tmp, overlap=True):
    """The edge-feature GAT stacks of BASELINE configs 4 / 5 (ogbn-proteins / ogbn-products models) on a 1-D partition:
    logits of the owned rows and the all-reduced parameter gradients equal the single-process ones."""
    import torch.distributed as dist
    import torch.nn.functional as F
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from tests import _oracle_backend
        _oracle_backend.install_direct()
        import bot_amd
        from bot_amd import dist as bdist, halo
        from bot_amd.nn import edge_gat
        halo.OVERLAP = overlap
        s, d, n = Golden().graph("g300")
        E = s.numel()
        gen = torch.Generator().manual_seed(11)
        nf = torch.randn(n, 9, generator=gen)
        ef = torch.rand(E, 8, generator=gen)
        gout = torch.randn(n, 6, generator=gen)

        def make():
            torch.manual_seed(5)
            if kind == "proteins":
                return edge_gat.ProteinsGAT(node_feats=9, edge_feats=8, n_classes=6, n_layers=3, n_heads=2, n_hidden=5, edge_emb=16,
                                            activation=F.relu, dropout=0.0, input_drop=0.0, attn_drop=0.0, edge_drop=0.0)
            return edge_gat.ProductsGAT(node_feats=9, edge_feats=0, n_classes=6, n_layers=3, n_heads=2, n_hidden=5, edge_emb=0,
                                        activation=F.relu, dropout=0.0, input_drop=0.0, attn_drop=0.0, edge_drop=0.0, residual=True)

        ref = make().train()
        g = bot_amd.Graph(s, d, n)
        g.ndata["feat"] = nf
        if kind == "proteins":
            g.edata["feat"] = ef
        logits_ref = ref(g)
        (logits_ref * gout).sum().backward()

        model = bdist.wrap_model(make().train())
        part = bdist.build_partition(s, d, n, rank, world)
        assert torch.equal(s[part.edge_ids] >= 0, torch.ones(part.n_edges, dtype=torch.bool)) and part.edge_ids.numel() == part.n_edges
        pg = part.graph
        pg.ndata["feat"] = nf[part.lo:part.hi]
        if kind == "proteins":
            pg.edata["feat"] = ef[part.edge_ids]
        logits = model(pg)
        assert (halo.CALLS == 3) == overlap, halo.CALLS       # one overlapped aggregation per layer
        (logits * gout[part.lo:part.hi]).sum().backward()
        bdist.all_reduce_grads(model)
        np.testing.assert_allclose(logits.detach().numpy(), logits_ref.detach()[part.lo:part.hi].numpy(), rtol=1e-4, atol=1e-5)
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            if q.grad is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
                continue
            np.testing.assert_allclose(p.grad.numpy(), q.grad.numpy(), rtol=2e-4, atol=2e-5 * max(1.0, q.grad.abs().max().item()), err_msg=k)
        for (k, b), (_, c) in zip(model.named_buffers(), ref.named_buffers()):
            np.testing.assert_allclose(b.numpy(), c.numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,world,overlap", [("proteins", 2, True), ("products", 2, True), ("proteins", 3, True), ("products", 3, False)])
def test_partitioned_edge_gat_matches_single_process(kind, world, overlap, tmp_path):
    mp.spawn(_worker_edge, args=(world, _free_port(), kind, str(tmp_path), overlap), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _worker_extras(rank, world, port, partitioner, tmp):
    """Label reuse in the partitioned train step (run.py:274-279), partitioned evaluate() (run.py:290-322) and the
    community partitioner (renumbered ranges, results mapped back through Partition.node_ids) against one process."""
    import types
    import torch.distributed as dist
    import torch.nn.functional as F
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from tests import _oracle_backend
        _oracle_backend.install_direct()
        import bot_amd
        from bot_amd.nn import fused
        fused.FORCE = True
        from bot_amd import dist as bdist, synth
        from bot_amd import nn as bnn
        from bot_amd import train as T
        from oracle import ref_ops as R
        n, C, fin = 1500, 5, 9
        cs, cd = synth.community_edges(n, 9000, 3, n_blocks=6, p_in=0.9)
        s, d = R.preprocess_edges(cs, cd, n)
        gen = torch.Generator().manual_seed(7)
        feat = torch.randn(n, fin, generator=gen)
        labels = torch.randint(0, C, (n, 1), generator=gen)
        perm = torch.randperm(n, generator=gen)
        tr, va, te = perm[: n // 2], perm[n // 2: 3 * n // 4], perm[3 * n // 4:]
        mask_full = torch.rand(n, generator=gen) < 0.5

        def make():
            torch.manual_seed(3)
            return bnn.GAT(dim_node=fin + C, dim_edge=0, dim_output=C, n_hidden=16, n_layers=3, n_heads=3, activation=F.relu,
                           norm="batch", linear=True)

        g = bot_amd.Graph(s, d, n)
        ref = make().train()
        loss_ref, pred_ref, _ = T.forward_backward(ref, g, feat, labels, tr, va, te, use_labels=True, n_label_iters=1, loss="loge",
                                                   n_classes=C, mask=mask_full[tr])
        ds = types.SimpleNamespace(graph=g, feat=feat, labels=labels, train_idx=tr, val_idx=va, test_idx=te)
        part = bdist.partition_dataset(ds, rank, world, "cpu", partitioner=partitioner)
        ids = part.node_ids if part.node_ids is not None else torch.arange(part.lo, part.hi)
        assert (part.node_ids is not None) == (partitioner == "community")
        assert torch.equal(part.feat, feat[ids]) and torch.equal(part.labels, labels[ids])
        model = bdist.wrap_model(make().train())
        own_tr = ids[part.train_idx]                      # original ids of the owned training nodes, in local order
        loss, pred = bdist.forward_backward(model, part, use_labels=True, n_label_iters=1, loss="loge", n_classes=C,
                                            mask=mask_full[own_tr])
        assert abs(loss.item() - loss_ref.item()) < 1e-5, (loss.item(), loss_ref.item())
        np.testing.assert_allclose(pred.detach().numpy(), pred_ref.detach()[ids].numpy(), rtol=1e-4, atol=2e-5)
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            np.testing.assert_allclose(p.grad.numpy(), q.grad.numpy(), rtol=2e-4, atol=2e-5 * max(1.0, q.grad.abs().max().item()), err_msg=k)
        # evaluate(): same post-forward state on both sides (running statistics were updated identically above)
        ev_ref = T.evaluate(ref, g, feat, labels, tr, va, te, use_labels=True, n_label_iters=1, loss="loge", n_classes=C)
        n0 = fused.INFER_CALLS
        ev = bdist.evaluate(model, part, use_labels=True, n_label_iters=1, loss="loge", n_classes=C)
        assert fused.INFER_CALLS - n0 == 6                # the inference-only layers run partitioned too
        np.testing.assert_allclose(np.array(ev[:6], dtype=np.float64), np.array([float(v) for v in ev_ref[:6]]), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ev[6].numpy(), ev_ref[6][ids].numpy(), rtol=1e-4, atol=2e-5)
        if rank == 0:  # the edge-cut-aware partitioner: far fewer halo rows than contiguous ranges of the random numbering
            from bot_amd.graph import reorder_permutation
            before = bdist.halo_statistics(s, d, n, world)
            p2, _ = reorder_permutation(g, "community")
            inv = torch.empty_like(p2)
            inv[p2] = torch.arange(n)
            after = bdist.halo_statistics(inv[s], inv[d], n, world)
            assert sum(after["halo_rows_per_rank"]) < 0.6 * sum(before["halo_rows_per_rank"]), (before, after)
            assert after["cut_edges"] < 0.5 * before["cut_edges"]
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("partitioner,world", [("contiguous", 2), ("community", 2), ("community", 3)])
def test_partitioned_label_reuse_evaluate_and_partitioner(partitioner, world, tmp_path):
    mp.spawn(_worker_extras, args=(world, _free_port(), partitioner, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


SCALES = {"cora": 0.3, "arxiv": 0.004, "reddit": 0.0001, "proteins": 0.00004, "products": 0.00003}     # (reddit: 23 nodes - at 8 nodes a step whose random split leaves NO prediction node has a NaN mean, as in the reference)


def _worker_workloads(rank, world, port, tmp):
    """bench.py's five workloads (bot_amd/workloads.py), tiny and without dropout, partitioned over the ranks: two steps each run,
    the loss is finite, equal on all ranks, and equal to the single-process step of the same workload (same seed, same init)."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from tests import _oracle_backend
        _oracle_backend.install_direct()
        from bot_amd.nn import fused
        fused.FORCE = True
        from bot_amd import workloads
        for name in workloads.NAMES:
            single = workloads.build(name, "cpu", scale=SCALES[name], drop=False)
            torch.manual_seed(11)
            l0 = [float(single.step()[0]) for _ in range(2)]
            part = workloads.build(name, "cpu", rank=rank, world=world, partitioned=True, scale=SCALES[name], drop=False)
            torch.manual_seed(11)                      # the same label-mask draws as the single-process run need the same node order:
            l1 = [float(part.step()[0]) for _ in range(2)]   # not the case -> only edge-GAT / no-mask losses are compared exactly
            assert all(np.isfinite(l0)) and all(np.isfinite(l1)), (name, l0, l1)
            t = torch.tensor(l1)
            dist.all_reduce(t)
            assert torch.allclose(t / world, torch.tensor(l1), atol=1e-6), name      # every rank reports the global loss
            if name in ("proteins", "products"):       # no random mask in their step: comparable with one process
                np.testing.assert_allclose(l1, l0, rtol=2e-4, atol=1e-5, err_msg=name)
            assert part.n_edges == single.n_edges and part.e_local < part.n_edges
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_bench_workloads_single_and_partitioned(tmp_path):
    mp.spawn(_worker_workloads, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(2))


def _worker_world8(rank, world, port, tmp):
    """The WHOLE partitioned train step of the config-2 stack shape (3 layers, 3 heads, aggregate-first input layer, fused layer nodes,
    overlapped halo exchange) on a 20 000-node power-law graph over 8 ranks: logits, loss and every gradient against one process, and
    the halo byte counter (bot_amd.halo.BYTES, what bench.py reports as exchange_bytes_per_rank_per_step) against (a) the bytes the
    collective itself saw and (b) rows x widths of the partition plan."""
    import torch.distributed as dist
    import torch.nn.functional as F
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from tests import _oracle_backend
        _oracle_backend.install_direct()
        import bot_amd
        from bot_amd.nn import fused
        from bot_amd import dist as bdist, halo, nn as bnn, synth, train as T
        fused.FORCE = True
        halo.OVERLAP = True
        n, C, fin, H, D = 20000, 5, 9, 3, 16
        rs, rd = synth.powerlaw_edges(n, 120000, 11)
        g = bot_amd.preprocess(bot_amd.Graph(rs, rd, n))
        s, d = g.edges()
        gen = torch.Generator().manual_seed(7)
        feat = torch.randn(n, fin, generator=gen)
        labels = torch.randint(0, C, (n, 1), generator=gen)
        perm = torch.randperm(n, generator=gen)
        tr, va, te = perm[: n // 2], perm[n // 2: 3 * n // 4], perm[3 * n // 4:]
        mask_full = torch.rand(n, generator=gen) < 0.5

        def make():
            torch.manual_seed(3)
            return bnn.GAT(dim_node=fin + C, dim_edge=0, dim_output=C, n_hidden=D, n_layers=3, n_heads=H, activation=F.relu, norm="batch",
                           non_interactive_attn=True, linear=True)
        ref = make().train()
        loss_ref, pred_ref, _ = T.forward_backward(ref, g, feat, labels, tr, va, te, use_labels=True, loss="loge", n_classes=C, mask=mask_full[tr])
        model = bdist.wrap_model(make().train())
        part = bdist.build_partition(s, d, n, rank, world)
        part.feat, part.labels = feat[part.lo:part.hi], labels[part.lo:part.hi]
        tr_own = tr[(tr >= part.lo) & (tr < part.hi)]
        part.train_idx = tr_own - part.lo
        plan = part.graph.halo
        assert plan.n_halo > 0 and plan.n_send > 0 and sum(1 for c in plan.recv_splits if c) >= 4     # a power-law graph: most peers are neighbours
        # (a) what the collective itself sees
        seen = {"sent": 0, "received": 0, "calls": 0}
        real = dist.all_to_all_single

        def counting(out, inp, *a, **k):
            seen["sent"] += inp.numel() * inp.element_size()
            seen["received"] += out.numel() * out.element_size()
            seen["calls"] += 1
            return real(out, inp, *a, **k)
        dist.all_to_all_single = counting
        b0 = dict(halo.BYTES)
        try:
            loss, pred = bdist.forward_backward(model, part, use_labels=True, loss="loge", n_classes=C, mask=mask_full[tr_own])
        finally:
            dist.all_to_all_single = real
        moved = {k: halo.BYTES[k] - b0[k] for k in b0}
        assert moved == seen, (moved, seen)
        # (b) rows x widths: layer 0 (aggregate-first, its input is data) ships [x | el] forward (fin + C + 4 columns) and the attention
        # columns back (4); layers 1 and 2 ship el then the projected rows forward, the rows' and el's gradients back (overlapped form)
        w0 = fin + C + 4
        fwd = [w0, H + H * D, 1 + C]
        bwd = [4, H * D + H, C + 1]
        assert moved["sent"] == 4 * (plan.n_send * sum(fwd) + plan.n_halo * sum(bwd)), (moved, plan.n_send, plan.n_halo)
        assert moved["received"] == 4 * (plan.n_halo * sum(fwd) + plan.n_send * sum(bwd))
        assert fused.L0_CALLS > 0 and fused.OVERLAP_CALLS > 0
        assert abs(loss.item() - loss_ref.item()) < 1e-5, (loss.item(), loss_ref.item())
        np.testing.assert_allclose(pred.detach().numpy(), pred_ref.detach()[part.lo:part.hi].numpy(), rtol=1e-4, atol=1e-5)
        for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            np.testing.assert_allclose(p.grad.numpy(), q.grad.numpy(), rtol=2e-4, atol=2e-5 * max(1.0, q.grad.abs().max().item()), err_msg=k)
        for (k, b), (_, c) in zip(model.named_buffers(), ref.named_buffers()):
            np.testing.assert_allclose(b.numpy(), c.numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
        open(os.path.join(tmp, f"ok{rank}"), "w").write("%d %d" % (moved["sent"], moved["received"]))
    finally:
        dist.destroy_process_group()


def test_world8_whole_step_and_exchange_bytes(tmp_path):
    """world 8 was covered by partition-plan invariants only."""
    world = 8
    mp.spawn(_worker_world8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    vals = [tuple(int(v) for v in (tmp_path / f"ok{r}").read_text().split()) for r in range(world)]
    assert sum(v[0] for v in vals) == sum(v[1] for v in vals)          # every byte sent is received by someone
