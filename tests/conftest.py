import os
import subprocess
import sys
import tempfile
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHILD_ENV = "BOT_TEST_ISOLATED_CHILD"       # set in the child pytest process an `isolated` test runs in
TRACE_ENV = "BOT_ABORT_TRACE_FILE"          # where libbot_gnn's SIGABRT / std::terminate handler appends the aborting thread's native frames


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "isolated: the test body runs in a fresh child pytest process (hipGraph capture, RCCL process groups, "
                                       "profiler sessions: a child that aborts is ONE red test, the suite goes on)")
    # Round 5's driver run ended with SIGABRT inside a hipGraph capture and only Python frames in the log.  Arm the library's abort trace
    # (include/bot_gnn.h bot_debug_abort_trace) in every test process: to the file the parent named, else to the process's REAL stderr
    # (pytest_configure runs while pytest's fd capture is suspended, so fd 2 is the log the driver keeps).
    try:
        from bot_amd import _C
        path = os.environ.get(TRACE_ENV)
        if not path:
            path = "/proc/self/fd/%d" % os.dup(2)
        _C.debug_abort_trace(path)
    except Exception as e:      # CPU suite on a tree without the built library: the ABI tests report that themselves
        print("tests/conftest.py: abort trace not armed (%s)" % e, file=sys.stderr)


def pytest_collection_modifyitems(config, items):
    """Order by evidence value: comparisons with the oracle / the golden vectors first (tests/test_zz_full_size_gpu.py's
    configs 3 and 5 among them), then property / self-comparison tests, then the isolated capture / process-group / profiler tests, and
    the 300 s config-4 oracle leg last — so that whatever ends a run early (an abort, the driver's wall clock) costs the least evidence."""
    def rank(item):
        name = item.name
        if name.startswith("test_native_library_is_loaded"):
            return -1
        if "config4" in name:
            return 4
        if item.get_closest_marker("isolated"):
            return 3
        if any(w in name for w in ("oracle", "golden", "bit_exact", "fp64", "against_torch", "matches_torch")):
            return 0
        return 2
    if any(i.get_closest_marker("gpu") for i in items):
        gpu = [i for i in items if i.get_closest_marker("gpu")]
        order = sorted(range(len(gpu)), key=lambda k: (rank(gpu[k]), k))          # stable: file order inside a rank
        it = iter([gpu[k] for k in order])
        items[:] = [next(it) if i.get_closest_marker("gpu") else i for i in items]


@pytest.hookimpl(tryfirst=True)
def pytest_pyfunc_call(pyfuncitem):
    """An `isolated` test: the parent starts `python -m pytest <this node id>` as a CHILD process (fork + exec of a new interpreter: the
    parent, which has touched the GPU, is never replaced) and passes / fails with the child's exit code; the child (CHILD_ENV set) runs the body."""
    if pyfuncitem.get_closest_marker("isolated") is None or os.environ.get(CHILD_ENV):
        return None
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()            # the child shares this GPU: give the cached blocks back first
    except Exception:
        pass
    fd, trace = tempfile.mkstemp(prefix="bot_abort_trace_", suffix=".txt")
    os.close(fd)
    env = dict(os.environ)
    env[CHILD_ENV] = "1"
    env[TRACE_ENV] = trace
    env.setdefault("TORCH_SHOW_CPP_STACKTRACES", "1")
    t0 = time.time()
    # -s: nothing the runtime prints on the way down (HIP, RCCL, libstdc++'s terminate) is lost in a capture file of an aborted process
    out = subprocess.run([sys.executable, "-m", "pytest", pyfuncitem.nodeid, "-x", "-q", "-s", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=float(os.environ.get("BOT_TEST_CHILD_TIMEOUT", "900")))
    native = open(trace).read() if os.path.exists(trace) else ""
    os.unlink(trace)
    tail = (out.stdout[-3000:] + "\n--- stderr ---\n" + out.stderr[-6000:]).strip()
    print("isolated child of %s: rc %d in %.1f s\n%s" % (pyfuncitem.nodeid, out.returncode, time.time() - t0, out.stdout[-1500:]))
    if out.returncode != 0:
        pytest.fail("child pytest process of %s ended with rc %d\n%s\n--- native abort trace ---\n%s"
                    % (pyfuncitem.nodeid, out.returncode, tail, native or "(none)"), pytrace=False)
    assert " passed" in out.stdout and " skipped" not in out.stdout.splitlines()[-1], tail      # the child really ran the body
    return True


@pytest.fixture(scope="session")
def golden():
    from tests._golden import Golden
    return Golden()
