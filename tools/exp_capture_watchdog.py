"""Round 6: what ended round 5's GPU suite.  The native trace (tools/r06_abort_hunt.sh, profiles/r06_abort_trace.txt) names
the caller: ProcessGroupNCCL's WATCHDOG thread polls WorkNCCL::isCompleted() -> hipEventQuery on the end event of a collective that ran
EAGERLY (the capture's warm-up steps) and gets hipErrorCapturedEvent ("operation not permitted on an event last recorded in a capturing
stream") once RCCL's stream has JOINED a capture; the watchdog rethrows, std::terminate, SIGABRT.  This script pins the mechanism:

  --part hip      no process group: an event recorded eagerly on a stream, the stream then captures - does hipEventQuery refuse it?
  --part pg       1-rank RCCL group: eager all-reduces, then a capture that holds RCCL's stream for longer than the watchdog's 100 ms poll;
                  (async all-reduce, a sleep, wait: the halo exchange's overlap window, stretched); --drain 1: bot_amd.train.drain_rccl_watchdog()
                  before the capture (what CapturedTrainStep does) - synchronize + several poll periods of sleep
Each --part pg trial runs in its own process (an abort ends it); `--loop N` starts N children and counts exit codes."""
import argparse
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def init_one_rank():
    """1-rank RCCL group over a TCP store on a free local port; a port lost between bind(0) + close and the listen is retried."""
    import socket
    import torch
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
        except Exception as e:
            if "in use" not in str(e).lower() and "EADDRINUSE" not in str(e):
                raise
            last = e
    raise last


def part_hip():
    import torch
    x = torch.zeros(1 << 20, device="cuda")
    for joined in (False, True):
        s, n = torch.cuda.Stream(), torch.cuda.Stream()
        e = torch.cuda.Event()
        with torch.cuda.stream(n if joined else s):
            x.add_(1)
            e.record()
        torch.cuda.synchronize()
        before = e.query()
        res = {}

        def poll(key):
            try:
                res[key] = e.query()
            except Exception as ex:
                res[key] = "RAISED " + str(ex).splitlines()[0]
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                x.add_(1)
                t = threading.Thread(target=poll, args=("capturing, event's stream not yet in the capture" if joined else "capturing on the event's stream",))
                t.start(), t.join()
                if joined:
                    n.wait_stream(s)
                    with torch.cuda.stream(n):
                        x.add_(1)
                    s.wait_stream(n)
                    t = threading.Thread(target=poll, args=("capturing, event's stream has joined the capture",))
                    t.start(), t.join()
        except Exception as ex:
            res["capture_end"] = "RAISED " + str(ex).splitlines()[0]
        poll("after the capture")
        print("event recorded EAGERLY on %s; query before the capture: %s; %s" % ("a second stream" if joined else "the capture stream", before, res), flush=True)


def part_pg(drain, hold, eager):
    import torch
    import torch.distributed as dist
    init_one_rank()
    t = torch.ones(1 << 16, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    time.sleep(0.5)
    side = torch.cuda.Stream()
    for it in range(5):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(eager):
                dist.all_reduce(t)              # eager: each Work sits in the watchdog's list until its next poll (every 100 ms)
            if drain:
                from bot_amd import train as T
                assert T.drain_rccl_watchdog() is True      # the product's remedy (bot_amd.train.CapturedTrainStep calls it before every capture)
            # capture_begin by hand: torch.cuda.graph.__enter__ runs synchronize + gc.collect + empty_cache first, ~100 ms in which the watchdog
            # usually retires the eager Works by itself (why the suite's abort was a 1-in-10 event and not a certainty)
            g = torch.cuda.CUDAGraph()
            g.capture_begin(capture_error_mode="thread_local")
            t.add_(1)
            w = dist.all_reduce(t, async_op=True)       # RCCL's stream joins the capture here ...
            time.sleep(hold)                            # ... stays in it across at least one watchdog poll (the halo exchange's overlap window, stretched) ...
            w.wait()                                    # ... and is joined back here
            t.add_(1)
            g.capture_end()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
    print("pg trial finished without an abort (drain=%d hold=%.2f eager=%d)" % (drain, hold, eager), flush=True)
    dist.destroy_process_group()


def part_step(captures):
    """The suite's own case: the 1-rank partitioned config-2 step (halo all-to-alls with async_op=True, gradient / BatchNorm / loss all-reduces),
    `captures` times {CapturedTrainStep (3 eager warm-up steps, then the capture), eager steps, replays}."""
    import torch
    import torch.distributed as dist
    from tests.test_gpu_parity import _replay_case
    init_one_rank()
    for c in range(captures):
        eager, cap, m1, m2 = _replay_case("arxiv-1rank", None)
        for _ in range(4):
            eager()
            cap()
        torch.cuda.synchronize()
        time.sleep(0.3)
    print("step trial finished without an abort (%d captures)" % captures, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--part", choices=["hip", "pg", "step"], required=True)
    ap.add_argument("--drain", type=int, default=0)
    ap.add_argument("--hold", type=float, default=0.25)
    ap.add_argument("--eager", type=int, default=3)
    ap.add_argument("--captures", type=int, default=2)
    ap.add_argument("--loop", type=int, default=0)
    a = ap.parse_args()
    if a.loop:
        rcs = []
        for i in range(a.loop):
            out = subprocess.run([sys.executable, __file__, "--part", a.part, "--drain", str(a.drain), "--hold", str(a.hold), "--eager", str(a.eager), "--captures", str(a.captures)],
                                 capture_output=True, text=True, timeout=600)
            rcs.append(out.returncode)
            if i == 0 or out.returncode != 0 and rcs.count(out.returncode) == 1:
                msg = [l for l in (out.stdout + out.stderr).splitlines() if "HIP error" in l or "finished" in l or "terminate called" in l]
                print("  trial %d rc %d: %s" % (i, out.returncode, " | ".join(msg[:3])), flush=True)
        print("part=%s drain=%d hold=%.2f eager=%d captures=%d TORCH_NCCL_CUDA_EVENT_CACHE=%s: %d trials, exit codes %s"
              % (a.part, a.drain, a.hold, a.eager, a.captures, os.environ.get("TORCH_NCCL_CUDA_EVENT_CACHE", "(unset)"), a.loop,
                 {r: rcs.count(r) for r in sorted(set(rcs))}), flush=True)
    elif a.part == "hip":
        part_hip()
    elif a.part == "step":
        part_step(a.captures)
    else:
        part_pg(a.drain, a.hold, a.eager)
