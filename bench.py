#!/usr/bin/env python3
"""Headline benchmark: edges/sec of one full-batch train step (forward + loss + backward + optimizer step) of a BASELINE.json
configuration on a synthetic graph of its shape — by default config 2, the ogbn-arxiv-shaped 3 x 3 x 250 GAT — plus the HBM
roofline of the step's dominant SpMM, a CPU baseline timed in the same run and (config 2) full-size parity against the oracle.

    python bench.py --gpus 1 --steps 20 --warmup 5 [--workload cora|arxiv|reddit|proteins|products]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--workload ...]

Prints ONE JSON line on rank 0.  `value` = preprocessed edges of the whole graph / wall time per step (max over ranks), inputs
resident in HBM.  For N > 1 the same graph is 1-D vertex-partitioned over the ranks with a halo all-to-all per layer on RCCL
("scaling": "strong"); every rank builds the seeded dataset in its own HBM and cuts its block out there.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F16_PEAK_TFLOPS = 2500.0   # dense fp16 MFMA peak of one MI355X (/opt/skills/guides/MI355X_MICROARCH.md; not the 2:1-sparsity figure)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, MI355X_MICROARCH.md "Chip-level parameters"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="arxiv", choices=["arxiv", "cora", "products", "reddit", "proteins"],
                    help="BASELINE.json configs 2 (default: the one `metric` is quoted on), 1, 5, 3, 4")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the graph (debug only; reported in config)")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--parity-scale", type=float, default=None,
                    help="configs 1, 3, 4, 5: size of the bounded sample (a graph of the same generator, same density) on which the "
                         "CPU baseline is timed and the parity object computed; default tests/full_size.py:PARITY_SCALE "
                         "(cora and reddit 1.0 = the bench graph itself, proteins / products 0.125); the full-size comparison is the GPU test-suite's")
    ap.add_argument("--cpu-cap", type=float, default=180.0,
                    help="proteins / products: seconds one oracle step of the bench graph itself may take (estimated from the 1/8 sample) "
                         "for `cpu_baseline` to be timed on it instead of on the sample")
    ap.add_argument("--gemm-tuning", default="file", choices=["file", "off", "tune"],
                    help="file: hipBLASLt/rocBLAS kernel selections from bot_amd/tuning (TunableOp, read-only); "
                         "tune: also time shapes missing from the file and write them to gpurun_out/ (maintenance)")
    ap.add_argument("--gemm", default=None, choices=["halves", "f32"],
                    help="dense projections: fp32 operands as two fp16 halves each on the fp16 matrix cores (default, "
                         "bot_amd/gemm.py) or the stock fp32 GEMM; default: $BOT_GEMM or halves")
    ap.add_argument("--norm-adj", default="rw", choices=["rw", "symm"],
                    help="arxiv only. rw: BASELINE config 2 (run.py:1011-1013); symm: the --norm-adj=symm variant of the same "
                         "command (run.py:1023-1025), reported in config")
    ap.add_argument("--partitioner", default="contiguous", choices=["contiguous", "community"],
                    help="N > 1: contiguous id ranges, or ranges of the label-propagation community order (bot_amd/dist.py)")
    ap.add_argument("--ignore-hbm-budget", action="store_true",
                    help="start even if bot_amd.workloads.hbm_budget (a coarse estimate) exceeds the free HBM of some rank")
    ap.add_argument("--capture", default="off", choices=["on", "off"],
                    help="on: replay the train step as ONE hipGraph (bot_amd.train.CapturedTrainStep) instead of ~300 eager launches "
                         "(configs 1-3; works partitioned too, RCCL collectives are captured).  Default off: the roofline object "
                         "needs HIP events around individual launches inside the timed region, which a replay has none of, and "
                         "the single-GPU step is GPU-bound (capture pays off where a rank's GPU work drops to a few ms)")
    ap.add_argument("--launch-timeout", type=float, default=3600.0,
                    help="N > 1 started without a launcher: seconds after which the self-started torch.distributed.run child is killed")
    ap.add_argument("--self-launch", action="store_true",
                    help="start the ranks through torch.distributed.run even for N = 1 (what N > 1 does by itself; lets a 1-GPU box "
                         "exercise the launcher, the relay and - with --force-partitioned - RCCL initialisation end to end)")
    ap.add_argument("--collective-timeout", type=float, default=120.0,
                    help="N > 1: seconds a rank waits for init_process_group and for the first all-to-all before it exits non-zero")
    ap.add_argument("--force-partitioned", action="store_true",
                    help="run the 1-D partitioned code path even with one rank (exercises the RCCL plumbing on a 1-GPU box)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ self-launch (N > 1)
def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def is_result_line(line: str) -> bool:
    """The rank-0 JSON line of this script (held back by the relay so that it is the LAST line of the parent's stdout)."""
    t = line.strip()
    if not (t.startswith("{") and t.endswith("}") and '"metric"' in t):
        return False
    try:
        return "metric" in json.loads(t)
    except ValueError:
        return False


def relay_child(cmd, env=None, timeout=None, out=None) -> int:
    """Run `cmd` as a fresh CHILD process (never an exec: a process that replaces itself after touching the GPU takes the box down),
    pass its stdout through line by line, hold result lines back and print the last one at the very end; stderr is inherited.
    Returns the child's return code (124 if `timeout` seconds passed: the child's own process group — the one started here — is
    killed)."""
    out = out if out is not None else sys.stdout
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1, start_new_session=True)
    timed_out = []

    def expire():
        timed_out.append(True)
        try:
            os.killpg(p.pid, 9)      # the exact group started above (start_new_session), nothing matched by name
        except OSError:
            pass

    timer = threading.Timer(timeout, expire) if timeout else None
    if timer:
        timer.daemon = True
        timer.start()
    result = None
    try:
        for line in p.stdout:
            if is_result_line(line):
                result = line
                continue
            out.write(line)
            out.flush()
        rc = p.wait()
    finally:
        if timer:
            timer.cancel()
    if result is not None:
        out.write(result if result.endswith("\n") else result + "\n")
        out.flush()
    if timed_out:
        print(f"bench.py: the {len(cmd)}-word launch command did not finish in {timeout} s; its process group was killed", file=sys.stderr)
        return 124
    return rc


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks ourselves, exactly as the
    driver would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    <same arguments>`), as a child process, BEFORE anything in this process touches the GPU.  (The reference pins one device,
    src/no-sampling/run.py:524-527; nothing to mirror there.)"""
    have = torch.cuda.device_count()      # counting devices does not initialise the GPU on this image
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this node has {have}", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool: RCCL's intra-node transport needs it
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    rc = 1
    for attempt in range(3):      # (a port found free by bind(0) + close can be taken again before the launcher listens on it: a fast failure is retried)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *argv]
        print("bench.py: launching " + " ".join(cmd), file=sys.stderr)
        t0 = time.time()
        rc = relay_child(cmd, env=env, timeout=args.launch_timeout)
        if rc in (0, 2, 4, 124) or time.time() - t0 > 20.0:
            break
        print(f"bench.py: the launch ended with rc {rc} after {time.time() - t0:.1f} s; trying another port", file=sys.stderr)
    return rc


class Watchdog:
    """`with Watchdog(120, "init_process_group"):` — if the block has not finished in time the rank prints what hung and exits
    non-zero (os._exit: a stuck collective cannot be interrupted from Python; it never re-execs anything), so that a wedged
    rendezvous or a dead xGMI link ends the run with a message instead of a hang."""

    def __init__(self, seconds, what, code=3):
        self.seconds, self.what, self.code = seconds, what, code
        self.done = threading.Event()

    def _watch(self):
        if not self.done.wait(self.seconds):
            print(f"bench.py rank {os.environ.get('RANK', '0')}: {self.what} did not complete in {self.seconds} s - exiting", file=sys.stderr,
                  flush=True)
            os._exit(self.code)

    def __enter__(self):
        threading.Thread(target=self._watch, daemon=True).start()
        return self

    def __exit__(self, *exc):
        self.done.set()
        return False


def spmm_alg_bytes(n, e, H, D, weighted):
    """compulsory bytes of one SpMM pass, int32 indices."""
    return 4 * (2 * n * H * D + e + (n + 1) + (e * H if weighted else 0))


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_and_parity(ds, n_classes, steps, dev, fuse=True):
    """Config 2.  `cpu_baseline`: the oracle's C restatement of DGL's CPU kernels (oracle/c_ops.py + oracle/ref_models.py) driving
    the same 3-layer GAT train step (forward + loge loss + backward; dropout 0, fixed label mask) on the host cores of this box:
    1 warm-up + `steps` timed steps, median.  `parity`: the SAME step (same weights, same mask, dropout 0) on the HIP path, every
    logit and every parameter gradient compared with the oracle's (tests/full_size.py) — outside the timed region."""
    from bot_amd import workloads
    from tests import full_size as FS
    cfg = {k: workloads.ARXIV_GAT[k] for k in FS.GAT_ARXIV}
    C = n_classes
    feat, labels, train_idx = ds.feat.cpu(), ds.labels.cpu(), ds.train_idx.cpu()
    sd = FS.init_state(cfg, feat.shape[1] + C, C, seed=0)
    mask = torch.rand(train_idx.shape, generator=torch.Generator().manual_seed(7)) < 0.5
    s, d = (t.cpu() for t in ds.graph.edges())
    n = ds.graph.number_of_nodes()
    # 32 threads: measured best on the 256-thread host of the GPU box (tools/exp_cpu_threads.py, 1/4-scale step:
    # 8/16/32/64/128/256 threads -> 1.56/1.21/1.07/1.53/2.90/19.5 s); more threads only add contention.
    pred, grads, times, threads, _ = FS.oracle_step(s, d, n, feat, labels, train_idx, mask, sd, cfg, C, steps=steps + 1)
    timed = sorted(times[1:])
    t = timed[len(timed) // 2]
    cpu = {"value": s.numel() / t, "unit": "edges/s", "cores": threads, "kind": "port",
           "sample": f"{steps} full train steps (fwd+loss+bwd, dropout 0) of the same graph after 1 warm-up; median step "
                     f"{t:.3f} s (all: {', '.join(f'{x:.2f}' for x in times[1:])}); OpenMP C restatement of DGL's CPU "
                     f"SpMM/SDDMM/edge_softmax + torch CPU GEMMs",
           "comparable_to_value": True, "cpu_model": cpu_model_name(), "host_threads": os.cpu_count()}
    g = ds.graph.to(dev)
    g.create_formats_()
    hp, hg, gates = FS.hip_step(g, ds.feat.to(dev), ds.labels.to(dev), ds.train_idx.to(dev), mask, sd, cfg, C, fuse=fuse)
    # gradients are compared with the oracle evaluated at the HIP run's ReLU / leaky-ReLU gates (tests/full_size.py:KinkGates);
    # the logits with the oracle's own gates (the plain step timed above) as well
    gp, gg, _, _, gstats = FS.oracle_step(s, d, n, feat, labels, train_idx, mask, sd, cfg, C, gates=gates)
    parity = FS.compare(hp, hg, gp, gg, gstats)
    parity["max_abs_logit_diff"] = max(parity["max_abs_logit_diff"], float((hp.cpu().double() - pred.double()).abs().max()))
    ok = (parity["max_abs_logit_diff"] <= 1e-4 and parity["max_rel_grad_err"] <= 1e-4 and parity["max_abs_preact_at_differing_gate"] <= 1e-4)
    parity.update(criterion="abs", criterion_text=FS.CRITERIA["abs"], ok=bool(ok), sample="the bench graph itself (full size)")
    parity["against"] = ("oracle/c_ops.py (C restatement of DGL's CPU kernels): same weights + label mask, dropout 0, training-mode "
                         "BatchNorm; logits vs the plain oracle step, gradients vs the oracle at the HIP run's ReLU / leaky-ReLU gates")
    return cpu, parity


def cpu_baseline_and_parity_sample(name, dev, scale, full_scale=1.0, cap_s=180.0):
    """Configs 1, 3, 4, 5.  One train step (drop rates 0) of the same configuration on the HIP path and on the oracle's C kernels
    (tests/full_size.py:workload_parity): `parity` = every logit and every parameter gradient of that step with the criterion that
    was applied, `cpu_baseline` = the oracle step's edges/s on the host cores.
    * `scale` == `full_scale` (cora, reddit): everything on the bench graph itself; the CPU timing is the SECOND oracle step.
    * `scale` < `full_scale` (proteins, products: 1/8): parity (incl. the fp64 leg of config 4) on the bounded sample — a graph of the
      same generator and density; the sample's CPU step (second of two) is timed, and if eight^-1-extrapolated to the bench graph it
      fits `cap_s` seconds, ONE oracle step of the bench graph itself is timed as well and becomes `cpu_baseline`;
      otherwise the sample's rate is reported with `comparable_to_value: false` and the GPU's own rate on the sample beside it.
    The full-size parity comparison of configs 3-5 is tests/test_zz_full_size_gpu.py (minutes of CPU time each)."""
    from tests import full_size as FS
    r, t = FS.workload_parity(name, dev, scale=scale, timed=True, warm=True, gpu_steps=5)
    r.pop("rank", None)
    what = f"S-{name} generator at scale {scale}: N={t['nodes']} E={t['edges']}"
    text = "OpenMP C restatement of DGL's CPU SpMM/SDDMM/edge_softmax + torch CPU GEMMs"
    on_bench_graph = scale == full_scale
    cpu = {"value": t["edges"] / t["seconds"], "unit": "edges/s", "cores": t["threads"], "kind": "port",
           "sample": f"the SECOND of two train steps (fwd+loss+bwd, drop rates 0) of the {what}; {t['seconds']:.2f} s (first: "
                     f"{t['first_step_seconds']:.2f} s); {text}",
           "comparable_to_value": on_bench_graph, "gpu_value_on_sample": t["edges"] / t["gpu_seconds_per_step"],
           "gpu_value_on_sample_note": "edges/s of the HIP path on the SAME graph as `value` of this object (5 steps after 2 warm-ups, drop rates 0)",
           "cpu_model": cpu_model_name(), "host_threads": os.cpu_count()}
    if not on_bench_graph:
        est = t["seconds"] * full_scale / scale
        if est <= cap_s:
            _, tf = FS.workload_parity(name, dev, scale=full_scale, exact=False, timed=True)
            cpu.update(value=tf["edges"] / tf["seconds"], comparable_to_value=True, sample_value=cpu["value"],
                       sample=f"ONE train step (fwd+loss+bwd, drop rates 0; un-warmed: a second one would double {tf['seconds']:.0f} s) of the BENCH "
                              f"GRAPH ITSELF (N={tf['nodes']} E={tf['edges']}); {tf['seconds']:.2f} s; {text}.  `sample_value`: the second of two "
                              f"steps of the {what} ({t['seconds']:.2f} s), the graph `parity` and `gpu_value_on_sample` are from")
        else:
            cpu["sample"] += f"; a step of the bench graph itself would take ~{est:.0f} s > the {cap_s:.0f} s cap (--cpu-cap)"
    r["sample"] = what + (" (the bench graph itself)" if on_bench_graph else " (bounded sample; full size: tests/test_zz_full_size_gpu.py)")
    r["against"] = ("oracle/c_ops.py (C restatement of DGL's CPU kernels) + oracle/ref_models.py: same weights, drop rates 0, "
                    "training-mode BatchNorm, the oracle at the HIP run's ReLU / leaky-ReLU gates")
    return cpu, r


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if (args.gpus > 1 or args.self_launch) and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process only launches the ranks and relays their output (no GPU call before this)
        sys.exit(launch_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if torch.cuda.device_count() <= local:
        sys.exit(f"bench.py rank {rank}: needs {max(args.gpus, local + 1)} GPUs, this node has {torch.cuda.device_count()}")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    partitioned = world > 1 or args.force_partitioned
    rccl = None
    if partitioned:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        with Watchdog(args.collective_timeout, "init_process_group(nccl)"):
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        with Watchdog(args.collective_timeout, "the first all-to-all (RCCL over xGMI)"):
            # every pair of ranks exchanges one row before any real work: a dead link or a wedged peer shows up here, by name
            probe = torch.full((world, 4), float(rank), device=dev)
            got = torch.empty_like(probe)
            dist.all_to_all_single(got, probe)
            torch.cuda.synchronize()
            assert got[:, 0].tolist() == [float(r) for r in range(world)], f"rank {rank}: all-to-all probe returned {got[:, 0].tolist()}"
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            rccl = "unknown"

    from bot_amd import _C, gemm, tuning, workloads
    if args.gemm is not None:
        gemm.MODE = args.gemm
    tuned = tuning.enable(tune_missing=args.gemm_tuning == "tune") if args.gemm_tuning != "off" else False
    if args.gemm_tuning == "tune":
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        torch.cuda.tunable.set_filename(os.path.join(ROOT, "gpurun_out", f"tunableop_new_rank{rank}.csv"))

    capture = args.capture == "on"
    # a coarse HBM budget BEFORE anything is allocated (every rank, stderr): a run that cannot fit says so here, rc 4, instead of dying in
    # the allocator on the first multi-GPU box it sees
    budget = workloads.hbm_budget(args.workload, world if partitioned else 1, args.scale)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    print(f"[bench] rank {rank}: HBM budget (estimate) {budget['total'] / 2**30:.1f} GiB = whole-graph build {budget['whole_graph_build'] / 2**30:.1f} "
          f"+ step {budget['step'] / 2**30:.1f}; free {free_b / 2**30:.1f} of {total_b / 2**30:.1f} GiB", file=sys.stderr, flush=True)
    fits = budget["total"] <= free_b
    if partitioned:         # one verdict for the whole job: a rank that left alone would strand the others in the build's collectives
        import torch.distributed as dist
        flag = torch.tensor([1.0 if fits else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        fits = bool(flag.item() > 0.5)
    if not fits and args.ignore_hbm_budget:
        print(f"[bench] rank {rank}: the estimate exceeds the free memory; --ignore-hbm-budget: starting anyway", file=sys.stderr, flush=True)
    elif not fits:
        print(f"[bench] rank {rank}: --workload {args.workload} at scale {args.scale} on {world} rank(s) needs ~{budget['total'] / 2**30:.0f} GiB per "
              f"rank, {free_b / 2**30:.0f} GiB are free: not starting (lower --scale or raise --gpus)", file=sys.stderr, flush=True)
        sys.exit(4)
    torch.cuda.reset_peak_memory_stats(dev)
    wl = workloads.build(args.workload, dev, rank=rank, world=world, partitioned=partitioned, seed=0, scale=args.scale,
                         norm_adj=args.norm_adj, partitioner=args.partitioner, capture=capture)
    barrier = torch.distributed.barrier if partitioned else (lambda: None)

    for _ in range(args.warmup):
        wl.step()
    # ---- timed region: exactly K steps between barrier + synchronize
    _C.PROFILE, _C.PROFILE_SKIP = [], ("gemm_halves",)    # the sparse sweeps are timed live; the GEMMs in three extra steps below
    prof = _C.PROFILE
    barrier()
    torch.cuda.synchronize()
    from bot_amd import halo
    halo_bytes0 = dict(halo.BYTES)              # (host counters of the halo all-to-alls: config.partition reports what the timed steps moved)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    halo_moved = {k: (halo.BYTES[k] - halo_bytes0[k]) // max(1, args.steps) for k in halo.BYTES}
    _C.PROFILE, _C.PROFILE_SKIP = None, ()
    gprof, gsteps = [], 3
    if gemm.MODE == "halves" and not wl.captured:      # the dense projections' launches, HIP events on the launch stream, 3 more steps
        # (with the weight-gradient products INLINE for these three steps: on the side stream they run beside other kernels and an event
        # pair around one measures the shared interval, not the kernel - the timed region above keeps the side stream)
        from bot_amd import side
        side_was, side.ENABLED = side.ENABLED, False
        # (and without the BatchNorm-backward by-product, ABI 18: with it two of the launches also do a reduce pass's work in their epilogue)
        bnb_was, gemm.BN_BYPRODUCT = gemm.BN_BYPRODUCT, False
        _C.PROFILE = gprof
        for _ in range(gsteps):
            wl.step()
        torch.cuda.synchronize()
        _C.PROFILE = None
        side.ENABLED, gemm.BN_BYPRODUCT = side_was, bnb_was
        gprof = [r for r in gprof if r[0] == "gemm_halves"]
    # "t_step ... optimizer excluded and reported separately".  `value` keeps the optimizer INSIDE (the conservative
    # number); three more steps with HIP events around optimizer.step() on its stream (torch's current stream) give its share
    opt_ms = None
    if not wl.captured and getattr(wl, "optimizer", None) is not None:
        opt, evs = wl.optimizer, []
        inner = opt.step

        def timed_step(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = inner(*a, **k)
            e1.record()
            evs.append((e0, e1))
            return r

        opt.step = timed_step
        try:
            for _ in range(3):
                wl.step()
            torch.cuda.synchronize()
        finally:
            del opt.step          # the instance attribute: the class's (hooked) method is back
        opt_ms = sum(a.elapsed_time(b) for a, b in evs) / max(1, len(evs))
    if partitioned:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3

    # ---- the same K steps once more with the stock fp32 GEMMs (outside the timed region above; reported beside `value` so that
    # both numbers come from one process, one graph, one set of weights)
    stock = None
    if gemm.MODE == "halves" and args.gemm is None and not wl.captured:
        gemm.MODE = "f32"
        for _ in range(max(2, args.warmup)):
            wl.step()
        barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            wl.step()
        torch.cuda.synchronize()
        barrier()
        dt1 = time.perf_counter() - t1
        gemm.MODE = "halves"
        if partitioned:
            t = torch.tensor([dt1], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt1 = float(t.item())
        stock = {"ms_per_step": dt1 / args.steps * 1e3, "value": wl.n_edges / (dt1 / args.steps), "unit": "edges/s",
                 "what": "the same steps with every projection on the stock fp32 GEMM (--gemm f32)"}

    # ---- roofline of the dominant kernel: the SpMM of the hidden layers' shape (forward CSC sweep; for the dense graphs the
    # L2-blocked form), HIP events recorded around each launch on the launch stream inside the timed region.
    H, D, weighted = wl.dominant_shape
    sel = [r for r in prof if r[0] in ("spmm", "spmm_blocked") and r[1] == (H, D, weighted)]
    fam = max({r[0] for r in sel}, key=lambda f: sum(r[2].elapsed_time(r[3]) for r in sel if r[0] == f)) if sel else None
    sel = [r for r in sel if r[0] == fam]
    roof = None
    if sel:
        durs = [r[2].elapsed_time(r[3]) * 1e-3 for r in sel]
        kernel = sel[0][4]   # the template instance the launch function dispatched for this shape (bot_last_kernel)
        alg = spmm_alg_bytes(wl.n_local, wl.e_local, H, D, weighted)
        avg = sum(durs) / len(durs)
        ach = alg / avg / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "spmm_traffic.json")
        tj = (json.load(open(tf)) if os.path.exists(tf) else {}).get(args.workload, {})
        if world == 1 and args.scale == 1.0 and tj.get("kernel") == kernel:
            # separate rocprofv3 --pmc passes over THIS command (tools/pmc_bench.sh <workload>), per launch of the same kernel
            # instance on the same graph; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md "HBM"
            traffic = tj.get("bytes_per_launch")
        roof = {"bound": "hbm", "kernel": f"{kernel} (SpMM forward, H={H} D={D}{', weighted' if weighted else ''}, hidden layers)",
                "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": traffic, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": round(avg * 1e3, 4),
                "launches_timed": len(durs),
                # measured L2<->fabric bytes (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/spmm_traffic.json) over the same
                # launch time: what the memory system actually delivered (Infinity-Cache hits included)
                "traffic_GBs": round(traffic / avg / 1e9, 1) if traffic else None,
                "traffic_frac_of_peak": round(traffic / avg / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                "l2_hit_rate": tj.get("l2_hit_rate") if traffic else None, "traffic_source": tj.get("source") if traffic else None}

    # the other HIP sweeps of the step, same live events, same byte accounting (SURVEY §8d formulas), for context
    others = []
    fam_bytes = {
        "spmm": lambda h, d, w: spmm_alg_bytes(wl.n_local, wl.e_local, h, d, w),
        "spmm_blocked": lambda h, d, w: spmm_alg_bytes(wl.n_local, wl.e_local, h, d, w),
        "spmm_dot": lambda h, d: 4 * (3 * wl.n_local * h * d + wl.e_local + wl.n_local + 1 + 2 * wl.e_local * h),
        "spmm_bcast": lambda h, d: 4 * (wl.n_local * d * (1 + h) + wl.e_local + wl.e_local * h),
        "spmm_dot_bcast": lambda h, d: 4 * (wl.n_local * d * (2 + h) + wl.e_local + 2 * wl.e_local * h),
        "sddmm_dot_bcast": lambda h, d: 4 * (wl.n_local * d * (1 + h) + wl.e_local + wl.e_local * h),
    }
    groups = {}
    for r in prof:
        groups.setdefault((r[0], r[1]), []).append(r)
    for (fam_, key), rs in sorted(groups.items(), key=lambda kv: -sum(r[2].elapsed_time(r[3]) for r in kv[1])):
        if fam_ not in fam_bytes:
            continue
        avg_ms = sum(r[2].elapsed_time(r[3]) for r in rs) / len(rs)
        b = fam_bytes[fam_](*key)
        others.append({"op": fam_, "shape": list(key), "kernel": rs[0][4], "launches": len(rs), "avg_launch_ms": round(avg_ms, 4),
                       "algorithmic_GBs": round(b / avg_ms / 1e6, 1), "frac": round(b / avg_ms / 1e6 / HBM_PEAK_GBS, 4)})
    if roof is not None:
        roof["all_sparse_sweeps"] = others
        # the dense projections on the fp16 matrix cores, same live events: fp16 MFMA flops actually issued (three products per
        # fp32 product, zero padding included) against gfx950's dense fp16 peak
        gs = gprof
        if gs:
            g_ms = sum(r[2].elapsed_time(r[3]) for r in gs)
            g_fl = sum(2.0 * r[1][0] * r[1][1] * r[1][2] * r[1][3] for r in gs)
            # The same launches against the COMBINED roofline: a projection with a short reduction axis (K = 100 .. 480 at configs 3-5: two to
            # eight k-steps per tile) is bound by writing its fp32 output, not by the matrix cores.  Per launch t_roof = max(flops / MFMA peak,
            # bytes / HBM peak) with bytes = 4 (m k + n k + m n) of the fp32-sized operands and result (plain launches: profile key (m, n, 3 k, 1);
            # grouped launches carry only their flops and count as MFMA-bound)
            t_roof = hbm_bound = g_by = 0.0
            for r in gs:
                m_, n_, k3, bt = r[1]
                fl = 2.0 * m_ * n_ * k3 * bt
                by = 4.0 * bt * (m_ * (k3 / 3.0) + n_ * (k3 / 3.0) + m_ * n_) if k3 > 1 else 0.0
                tm, th = fl / (MFMA_F16_PEAK_TFLOPS * 1e12), by / (HBM_PEAK_GBS * 1e9)
                t_roof += max(tm, th)
                g_by += by
                hbm_bound += 1 if th > tm else 0
            roof["dense_projections"] = {"bound": "mfma", "what": "the halves-GEMM launches (bot_gemm_halves3_nt_f32 / _tn_f32 and their grouped forms: hand-written NT / TN products; bot_gemm_halves_f32, "
                                                 "hipBLASLt fp16 -> fp32, where a shape is left to it: none in config 2) of three more steps after the timed region (HIP events; weight-gradient products inline for these steps, not on the side stream, and the input-gradient products without the BatchNorm-backward by-product in their epilogue); flops = "
                                                 "fp16 MFMA flops of the valid output columns, three products per fp32 product",
                                         "launches_per_step": len(gs) / gsteps, "ms_per_step": round(g_ms / gsteps, 3),
                                         "achieved": round(g_fl / g_ms / 1e9, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                         "frac": round(g_fl / g_ms / 1e9 / MFMA_F16_PEAK_TFLOPS, 4),
                                         "fp32_equivalent_TFLOPs": round(g_fl / 3 / g_ms / 1e9, 1),
                                         "combined_roofline": {"frac": round(t_roof * 1e3 / g_ms, 4), "launches_bound_by_hbm": int(hbm_bound),
                                                               "launches": len(gs), "bytes_per_step": int(g_by / gsteps),
                                                               "what": "sum over the launches of max(flops / 2.5 PFLOP/s, 4 (m k + n k + m n) B / 8 TB/s) / measured time"}}

    cpu = parity = None
    if rank == 0 and world == 1 and args.cpu_baseline != "off" and args.workload == "arxiv" and args.norm_adj == "rw":
        cpu, parity = cpu_baseline_and_parity(wl.dataset, wl.dataset.n_classes, args.cpu_steps, dev)
    elif rank == 0 and world == 1 and args.cpu_baseline != "off" and args.workload != "arxiv":
        from tests import full_size as FS
        sample_scale = (args.parity_scale if args.parity_scale is not None else FS.PARITY_SCALE[args.workload]) * args.scale
        cpu, parity = cpu_baseline_and_parity_sample(args.workload, dev, sample_scale, full_scale=args.scale, cap_s=args.cpu_cap)

    part_info = None
    if partitioned:
        # what a SCALE line needs to explain itself: per rank the owned rows, the in-edges it sweeps, the halo rows it receives and
        # the rows it ships per hidden layer and direction, and the bytes those exchanges move per step (fp32; layer 0 of the GAT
        # ships the narrow inputs, bot_amd/nn/fused.py)
        p = wl.dataset.part
        mine = torch.tensor([rank, int(p.n_owned), int(p.n_edges), int(p.graph.halo.n_halo), int(p.graph.halo.n_send),
                             int(sum(1 for c in p.graph.halo.recv_splits if c)), int((p.graph.edges()[0] >= p.n_owned).sum()),
                             int(halo_moved["sent"]), int(halo_moved["received"]), int(halo_moved["calls"])],
                            dtype=torch.int64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)          # plain tensors: the same collective path as the step itself
        if rank == 0:
            keys = ("rank", "owned_rows", "edges", "halo_rows", "send_rows", "peers_recv", "cut_edges", "halo_bytes_sent_per_step",
                    "halo_bytes_received_per_step", "halo_all_to_alls_per_step")
            gathered = [dict(zip(keys, t.tolist())) for t in allr]
            assert len(gathered) == world and [g["rank"] for g in gathered] == list(range(world))
            part_info = {"n_ranks": len(gathered), "rccl_version": rccl, "ranks": gathered,
                         "exchange_bytes_per_rank_per_step": [g["halo_bytes_sent_per_step"] + g["halo_bytes_received_per_step"] for g in gathered],
                         "note": "MEASURED: bytes this rank handed to + received from the halo all-to-alls of one timed step (bot_amd.halo.a2a counts "
                                 "every one: forward rows other ranks need, reverse their gradients; layer 0 of the GAT ships its narrow inputs forward "
                                 "and only the attention columns back) - tests/test_dist_gloo.py holds the counter to the bytes the collective saw and "
                                 "to rows x widths of the partition plan"}
    if rank == 0:
        out = {
            "metric": "edges/sec full-batch GAT fwd+bwd on ogbn-arxiv; achieved HBM GB/s vs peak",
            "value": wl.n_edges / (ms * 1e-3), "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            # SURVEY §8d companions of `value` (which counts PREPROCESSED edges and keeps the optimizer step inside the time)
            "value_raw_edges": wl.raw_edges / (ms * 1e-3), "raw_edges": wl.raw_edges, "edges": wl.n_edges,
            "optimizer_ms": None if opt_ms is None else round(opt_ms, 4),
            "step_ms_without_optimizer": None if opt_ms is None else round(ms - opt_ms, 4),
            "value_without_optimizer": None if opt_ms is None else wl.n_edges / ((ms - opt_ms) * 1e-3),
            "config": {"workload": wl.describe,
                       "gemm": ("fp32 operands as two fp16 halves each (h1 + h2 = 22-23 of the 24 significand bits, power-of-two scale found "
                                "on the device), a1 b1 + a1 b2 + a2 b1 on the fp16 MFMAs with fp32 accumulation (hand-written gfx950 kernels, csrc/halves3.hip: each "
                                "operand half staged once, three MFMAs per fragment pair); error against fp64 equal to the stock fp32 GEMM's "
                                "(tests/test_gpu_parity.py::test_gemm_halves_against_fp64); --gemm f32 = stock fp32")
                       if gemm.MODE == "halves" else "stock fp32 GEMM (hipBLASLt / rocBLAS)",
                       "gemm_kernel_selection": "TunableOp file" if tuned else "library default",
                       "scale": args.scale, "launch": "one hipGraph replay per step" if wl.captured else "eager",
                       "parallelism": "single GPU" if world == 1 else f"1-D vertex partition x{world} ({args.partitioner} ranges)",
                       "partition": part_info},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "stock_fp32_gemm": stock,
            "hbm": {"estimate_GiB_per_rank": round(budget["total"] / 2**30, 2), "peak_allocated_GiB_rank0": round(torch.cuda.max_memory_allocated(dev) / 2**30, 2),
                    "what": "bot_amd.workloads.hbm_budget printed before allocation (rc 4 if it exceeds the free memory) / torch's peak over build + all legs"},
        }
    if partitioned:
        torch.distributed.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio; flush it first so that the JSON line is the LAST line of stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
