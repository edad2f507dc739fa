"""bench.py's self-launch (`python bench.py --gpus N` with no launcher around it), the relay that keeps the rank-0 JSON line last,
the per-rank watchdog, and the halo exchange with peers that share no rows (zero-length splits).  CPU only."""
import io
import os
import socket
import subprocess
import sys
import textwrap
import time

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_relay_keeps_result_line_last_and_returns_child_code(tmp_path):
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(textwrap.dedent('''
        import json, sys
        print("rank banner")
        print(json.dumps({"metric": "m", "value": 1.5, "n_gpus": 2}))
        print("RCCL teardown noise after the result")
        print(json.dumps({"not": "a result"}))
        sys.exit(int(sys.argv[1]))
    '''))
    for code in (0, 7):
        out = io.StringIO()
        rc = bench.relay_child([sys.executable, str(stub), str(code)], out=out)
        lines = out.getvalue().splitlines()
        assert rc == code
        assert lines[-1] == '{"metric": "m", "value": 1.5, "n_gpus": 2}'
        assert lines[:3] == ["rank banner", "RCCL teardown noise after the result", '{"not": "a result"}']


def test_relay_timeout_kills_only_its_child(tmp_path):
    import bench
    stub = tmp_path / "hang.py"
    stub.write_text("import time\nprint('started', flush=True)\ntime.sleep(600)\n")
    out = io.StringIO()
    t0 = time.time()
    rc = bench.relay_child([sys.executable, str(stub)], timeout=2.0, out=out)
    assert rc == 124 and time.time() - t0 < 30 and out.getvalue().splitlines() == ["started"]


def test_gpus_2_without_gpus_fails_cleanly():
    """On a box with fewer GPUs than asked for: a message, rc != 0, no hang, nothing launched."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has 2 GPUs")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert p.returncode == 2 and "needs 2 GPUs" in p.stderr and p.stdout.strip() == ""


def test_launch_command_is_the_drivers(monkeypatch):
    import bench
    seen = {}
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(bench, "relay_child", lambda cmd, env=None, timeout=None, out=None: seen.update(cmd=cmd, env=env, timeout=timeout) or 0)
    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1", "--workload", "reddit"]
    assert bench.launch_ranks(bench.parse(argv), argv) == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["timeout"] == 3600.0


def test_watchdog_exits_nonzero_on_a_hang(tmp_path):
    stub = tmp_path / "wd.py"
    stub.write_text(textwrap.dedent(f'''
        import sys, time
        sys.path.insert(0, {ROOT!r})
        import bench
        with bench.Watchdog(0.5, "quick block"):
            pass
        with bench.Watchdog(1.0, "the stuck collective"):
            time.sleep(600)
    '''))
    t0 = time.time()
    p = subprocess.run([sys.executable, str(stub)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 3 and "the stuck collective did not complete" in p.stderr and time.time() - t0 < 120


def _worker_zero_peers(rank, world, port, tmp):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from tests import _oracle_backend
        _oracle_backend.install_direct()
        from bot_amd import dist as bdist
        # a path graph 0 - 1 - ... - 11 cut into three ranges: ranks 0 and 2 share no edge (zero-length splits both ways), and
        # rank 1's vertex 5..6 region talks to both; plus an isolated tail so one rank ships nothing at all when world == 3
        n = 12
        s = torch.arange(0, n - 1)
        d = torch.arange(1, n)
        s, d = torch.cat([s, d, torch.arange(n)]), torch.cat([d, s, torch.arange(n)])
        part = bdist.build_partition(s, d, n, rank, world, bounds=[n * k // world for k in range(world + 1)])
        plan = part.graph.halo
        assert plan.recv_splits[rank] == 0 and plan.send_splits[rank] == 0
        if world == 3:
            far = 2 - rank if rank != 1 else None
            if far is not None:
                assert plan.send_splits[far] == 0 and plan.recv_splits[far] == 0
        x = (torch.arange(part.lo, part.hi, dtype=torch.float32)[:, None] * torch.tensor([1.0, 10.0])).requires_grad_()
        ext = plan.extend(x)
        glob = torch.cat([torch.arange(part.lo, part.hi), part.halo_global]).float()
        assert torch.equal(ext.detach()[:, 0], glob)                                  # every halo row came from its owner
        w = torch.arange(1, ext.shape[0] + 1, dtype=torch.float32)[:, None]
        (ext * w).sum().backward()
        torch.save({"grad": x.grad, "lo": part.lo, "w_halo": w[part.n_owned:, 0], "halo": part.halo_global}, os.path.join(tmp, f"z{rank}.pt"))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_with_zero_length_peers(world, tmp_path):
    mp.spawn(_worker_zero_peers, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(os.path.join(tmp_path, f"z{r}.pt")) for r in range(world)]
    # gradient of an owned row = its own weight + the weights every other rank put on its halo copy
    n = 12
    expect = torch.zeros(n)
    for r in rs:
        k = r["grad"].shape[0]
        expect[r["lo"]:r["lo"] + k] += torch.arange(1, k + 1, dtype=torch.float32)
        expect[r["halo"]] += r["w_halo"]
    got = torch.cat([r["grad"][:, 0] for r in rs])
    assert torch.equal(got, expect)
