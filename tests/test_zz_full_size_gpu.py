"""GPU parity at FULL size for BASELINE configs 3, 4 and 5 (config 4's fp32 + fp64 oracle steps take 300 s of host time, side by side; no
environment knob shrinks them: VERDICT r5 #6) — one whole train step each against the oracle's C kernels.  tests/conftest.py orders the
suite by evidence value: configs 3 and 5 run with the other oracle comparisons, in front of the property / self-comparison tests and the
isolated capture tests; config 4's 300 s leg runs last.  Config 2's full-size tests are short and live in test_gpu_parity.py."""
import pytest
import torch

from tests import parity_cases as PC

pytestmark = pytest.mark.gpu
DEV = "cuda"
# config 4 at full size: the number of parameter gradients that are NOT within 1e-4 of the fp64 step and pass only because the HIP run is at
# most twice as far from it as the fp32 oracle is (round 5, measured: profiles/r05_config4_full.log)
# Round 5 at full size: 53 of 64 within 1e-4; 10 over 1e-4 but CLOSER to the fp64 step than the fp32 oracle is (5x .. 300x closer: the
# attention / edge-encoder gradients, sums over 600 in-edges of cancelling terms); 1 (edge_encoder.0.bias, 3.9e-4 vs the oracle's 3.0e-4)
# farther than the oracle, inside the factor 2.
CONFIG4_RANKED_MAX = 11
CONFIG4_WORSE_THAN_ORACLE_MAX = 1


def test_full_size_config3_reddit_gcn_against_c_oracle():
    """BASELINE config 3 at its full synthetic size — S-reddit, 232 965 nodes / 113.7 M edges, GCN 3 x 256
    with BatchNorm — one train step (dropout 0) on the HIP path (L2-blocked SpMM + hub rows, W-first and aggregate-first
    GraphConv) against the oracle's C kernels on the host cores: every logit within 1e-4, every gradient entry within 1e-4 of
    its gradient's largest entry (oracle at the HIP run's ReLU gates, tests/full_size.py:KinkGates)."""
    from tests import full_size as FS
    r, cpu = FS.workload_parity("reddit", DEV)
    print("full-size parity S-reddit GCN", r, "oracle step %.1f s" % cpu["seconds"])
    assert r["n"] == 232965 and r["edges"] > 100_000_000
    assert r["criterion"] == "abs" and r["max_abs_logit_diff"] <= PC.FWD_ATOL, r       # 1e-4 absolute (logits up to 13)
    assert r["max_rel_grad_err"] <= PC.GRAD_RTOL, r
    assert r["max_abs_preact_at_differing_gate"] <= 1e-4 and r["ok"], r


def test_full_size_config5_products_gat_against_c_oracle():
    """BASELINE config 5 at its full synthetic size — S-products, 2 449 029 nodes / 126 M edges, GAT 3 layers x 4 heads x 120
    (src/ogbn-products/models.py, full-graph branch) — one train step (drop rates 0, loge loss of gat.py:107-118) on the HIP path
    against the oracle's C kernels on the host cores: logits within 1e-4 (relative to their scale beyond 10), every gradient entry
    within 1e-4 of its gradient's largest entry, the oracle at the HIP run's ReLU / leaky-ReLU gates.  Two things differ from the
    config-2 test, both measured (tests/diag_products_dw.py): the oracle accumulates its Linear weight gradients in fp64 (its
    fp32 sgemm is 2.1e-4 off the fp64 product of its own operands over 2.45 M rows, the HIP run's GEMM 5e-5), and the dst_fc
    biases — in front of a training-mode BatchNorm, gradient identically zero in exact arithmetic, 1e-10 of noise in both runs —
    are measured against the dst_fc weight gradient's scale."""
    from tests import full_size as FS
    r, cpu = FS.workload_parity("products", DEV)
    print("full-size parity S-products GAT", r, "oracle step %.1f s" % cpu["seconds"])
    assert r["n"] == 2449029 and r["edges"] > 120_000_000
    assert r["criterion"] == "abs" and r["max_abs_logit_diff"] <= PC.FWD_ATOL, r       # 1e-4 absolute (logits up to 2.6)
    assert r["max_rel_grad_err"] <= PC.GRAD_RTOL, r
    assert r["max_abs_preact_at_differing_gate"] <= 1e-4 and r["ok"], r


def test_full_size_config4_proteins_gat_against_c_oracle():
    """BASELINE config 4 — S-proteins at FULL size (132 534 nodes / 79 M edges, mean in-degree 600: the stack is badly conditioned)
    with 8 edge features, GAT 6 layers x
    6 heads x 80 (src/ogbn-proteins/models.py, full-graph branch: node encoder, per-layer edge encoders, inter-layer residual)
    — one train step (drop rates 0, BCE-with-logits over 112 tasks, gat.py:203-207) on the HIP path against the oracle's C
    kernels, the oracle at the HIP run's gates.  Logits: within 1e-4 relative to their scale.  Gradients: this stack is badly
    conditioned in fp32 (mean in-degree 600, logits up to 125, attn_dst_fc's gradient is a sum over in-edges of softmax
    gradients that cancel), so two fp32 runs differ by more than 1e-4 on some parameters no matter how they are written.
    The criterion is therefore against the SAME step in fp64 (liboracle_f64.so): every HIP gradient is within 1e-4 of the exact
    one, or at most twice as far from it as the reference-order fp32 CPU run is."""
    import os
    from tests import full_size as FS
    scale = 1.0
    r, cpu = FS.workload_parity("proteins", DEV, scale=scale)
    rank = r.pop("rank")
    print("parity S-proteins GAT at scale %g" % scale, r, "fp32 oracle step %.1f s" % cpu["seconds"])
    if os.environ.get("BOT_PARITY_TABLE"):
        for k, (eh, eo) in rank.items():
            print("  %-28s HIP vs fp64 %.3e   fp32 oracle vs fp64 %.3e" % (k, eh, eo))
    assert abs(r["n"] - 132534 * scale) <= 1 and r["edges"] > 70_000_000 * scale
    # logits up to 125: not "within 1e-4 absolute of the fp32 oracle" (3.1e-4) — ranked against the fp64 step instead (FS.CRITERIA)
    assert r["criterion"] == "fp64-ranked" and r["logit_err_vs_fp64"] <= max(PC.FWD_ATOL, 2 * r["oracle_logit_err_vs_fp64"]), r
    by_abs = [k for k, (eh, eo) in rank.items() if eh <= PC.GRAD_RTOL]
    by_rank = {k: (eh, eo) for k, (eh, eo) in rank.items() if PC.GRAD_RTOL < eh <= 2 * eo}
    failed = {k: (eh, eo) for k, (eh, eo) in rank.items() if eh > max(PC.GRAD_RTOL, 2 * eo)}
    worse = {k: v for k, v in by_rank.items() if v[0] > v[1]}        # over 1e-4 AND farther from the fp64 step than the fp32 oracle
    msg = (f"{len(by_abs)} of {len(rank)} gradients within {PC.GRAD_RTOL:g} of the fp64 step; {len(by_rank)} only through the 2x clause, of which "
           f"{len(by_rank) - len(worse)} are closer to fp64 than the fp32 oracle and {len(worse)} farther {sorted(worse)}: "
           f"{ {k: ('%.2e' % a, '%.2e' % b) for k, (a, b) in by_rank.items()} }; {len(failed)} fail {failed}")
    print(msg)
    assert not failed, msg
    # how many gradients need the second clause is part of the bar: neither count may grow (measured at full size, see the constants)
    assert len(by_rank) <= CONFIG4_RANKED_MAX and len(worse) <= CONFIG4_WORSE_THAN_ORACLE_MAX, msg
    assert r["ok"], r
