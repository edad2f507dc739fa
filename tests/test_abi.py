"""The C-ABI library loads, exports every symbol include/bot_gnn.h declares, validates arguments
without touching a GPU, and its host-side row plan is correct."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from bot_amd import _C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "bot_gnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bot_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    syms = header_symbols()
    assert len(syms) >= 15
    lib = ctypes.CDLL(_C.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in bot_gnn.h but not exported"
    assert set(syms) == set(_C.EXPORTED)  # the binding covers the whole header
    assert lib.bot_abi_version() == 19


_SCALAR = {"int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "float": ctypes.c_float,
           "double": ctypes.c_double}


def header_prototypes():
    """{name: (result ctype, [argument ctypes])} parsed from include/bot_gnn.h: every pointer (and bot_stream_t) is c_void_p, scalars by
    their C type."""
    text = open(os.path.join(ROOT, "include", "bot_gnn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)

    def ctype(decl):
        d = decl.strip()
        if "*" in d or re.match(r"(const\s+)?bot_stream_t\b", d):
            return ctypes.c_void_p
        return _SCALAR[re.sub(r"\bconst\b", "", d).split()[0]]
    out = {}
    for ret, name, params in re.findall(r"([A-Za-z_][\w\s\*]*?)\b(bot_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        args = [] if params.strip() in ("", "void") else [ctype(p) for p in params.split(",")]
        out[name] = (ctype(ret), args)
    return out


def signature_mismatches(sigs, protos):
    """Differences between a ctypes table and the parsed prototypes: argument COUNT, and per position the class pointer / int32 / int64 /
    uint64 / float / double (int == int32_t on this ABI; a char* is a pointer)."""
    def norm(t):
        return {ctypes.c_char_p: ctypes.c_void_p, ctypes.c_int: ctypes.c_int32}.get(t, t)
    bad = []
    for name, (res, args) in sigs.items():
        pres, pargs = protos[name]
        if norm(res) is not norm(pres):
            bad.append((name, "result", res, pres))
        if len(args) != len(pargs):
            bad.append((name, "argument count", len(args), len(pargs)))
            continue
        bad += [(name, i, a, b) for i, (a, b) in enumerate(zip(args, pargs)) if norm(a) is not norm(b)]
    return bad


def test_binding_signatures_match_the_header_prototypes():
    """bot_amd/_C.py's _SIGS is a hand-kept table of positional c_void_p / c_int64 lists - an argument-count or int / pointer /
    float slip against include/bot_gnn.h is silent memory corruption.  Every prototype of the header is parsed and compared with the table:
    same set of names, same result class, same argument count, same class at every position; the check is shown to bite on three planted
    slips (one argument dropped, an int64 for an int32, a float for a pointer)."""
    protos = header_prototypes()
    assert set(protos) == set(_C._SIGS) == set(header_symbols()) and len(protos) >= 80
    assert signature_mismatches(_C._SIGS, protos) == []
    # the functions really carry the table (argtypes are what ctypes converts by)
    for name, (res, args) in _C._SIGS.items():
        fn = getattr(_C._lib, name)
        assert fn.restype is res and list(fn.argtypes) == list(args), name
    res, args = _C._SIGS["bot_spmm_f32"]
    planted = dict(_C._SIGS)
    planted["bot_spmm_f32"] = (res, args[:-1])
    assert signature_mismatches(planted, protos) == [("bot_spmm_f32", "argument count", len(args) - 1, len(args))]
    i32 = args.index(ctypes.c_int32)
    planted["bot_spmm_f32"] = (res, args[:i32] + [ctypes.c_int64] + args[i32 + 1:])
    assert signature_mismatches(planted, protos) == [("bot_spmm_f32", i32, ctypes.c_int64, ctypes.c_int32)]
    planted["bot_spmm_f32"] = (res, [ctypes.c_float] + args[1:])
    assert signature_mismatches(planted, protos) == [("bot_spmm_f32", 0, ctypes.c_float, ctypes.c_void_p)]


def test_argument_validation_without_gpu():
    lib = _C._lib
    # H = 0 is rejected before any launch
    rc = lib.bot_spmm_f32(None, None, 4, 0, None, 4, None, None, 0, None, 4, 4, None, None, 0, 4, None, 4, 4, None, 0, 0, None, None)
    assert rc == -2 and b"H=0" in lib.bot_last_error()
    rc = lib.bot_spmm_f32(None, None, 4, 0, None, 4, None, None, 0, None, 4, 4, None, None, 1, 4, None, 4, 4, None, 0, 0, None, None)
    assert rc == -1 and b"NULL" in lib.bot_last_error()
    rc = lib.bot_segment_sum_f32(None, 3, 0, None, 0, 0, None, None, 1, None, None)
    assert rc == -2
    rc = lib.bot_gat_attn_bwd_f32(None, None, -1, 0, None, 0, 8, None, None, None, None, 0.2, 1, None, None, None, None, None, None, None, 0.0, 0, None, None)
    assert rc == -2
    # empty problems are no-ops
    assert lib.bot_degrees_i64(None, 0, None, None) == 0
    assert lib.bot_sddmm_u_add_v_f32(None, None, 0, None, None, 1, None, None) == 0


def plan_reference(indptr, chunk):
    items, longs, ptr, slot = [], [], [0], 0
    for r in range(len(indptr) - 1):
        b, e = indptr[r], indptr[r + 1]
        if e - b > chunk:
            longs.append(r)
            for s in range(b, e, chunk):
                items.append((r, s, min(s + chunk, e), slot))
                slot += 1
            ptr.append(slot)
    rest = [(r, indptr[r], indptr[r + 1], -1) for r in range(len(indptr) - 1) if indptr[r + 1] - indptr[r] <= chunk]
    rest.sort(key=lambda t: -(t[2] - t[1]))  # stable: longest first, row order within a degree
    return items + rest, longs, ptr


@pytest.mark.parametrize("chunk", [1, 4, 64])
def test_row_plan(chunk):
    rng = np.random.default_rng(0)
    deg = np.concatenate([rng.integers(0, 10, 200), [0, 0, 300, 65, 64, 1000]])
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    items, long_rows, long_ptr, n_slots = _C.row_plan(torch.from_numpy(indptr), chunk)
    ref_items, ref_long, ref_ptr = plan_reference(indptr.tolist(), chunk)
    assert items.tolist() == [list(t) for t in ref_items]
    assert long_rows.tolist() == ref_long and long_ptr.tolist() == ref_ptr and n_slots == ref_ptr[-1]
    # every position is covered exactly once
    cover = np.zeros(indptr[-1], dtype=np.int32)
    for r, b, e, s in items.tolist():
        cover[b:e] += 1
    assert np.all(cover == 1)
    assert _C.default_chunk(2_484_941) in (64, 128, 256, 512)


def test_row_plan_rejects_bad_indptr():
    bad = torch.tensor([0, 5, 3], dtype=torch.int32)
    with pytest.raises(_C.BotKernelError):
        _C.row_plan(bad, 4)


def test_tn_split_count_fills_the_xcds():
    """bot_gemm_halves3_tn_workspace_floats = splits x kp x pp: one split per XCD when a split's 192 x 192 tiles are the XCD's 32 CUs (config 2:
    4 x 8 tiles), q <= 8 per XCD where they are not - the q with the best (q tiles) / (32 ceil(q tiles / 32)), smallest on ties (S-products'
    [480, N] x [N, 968]: 3 x 6 = 18 tiles -> q = 7: 126 workgroups per XCD, 98 % of four rounds) - never a split under 4096 rows.  Host
    arithmetic only, against a restatement of the rule."""
    from bot_amd import _C
    f = _C._lib.bot_gemm_halves3_tn_workspace_floats

    def rule(n, kp, pp):
        tiles = -(-kp // 192) * -(-pp // 192)
        by_rows = n // 4096
        if by_rows < 8:
            return max(1, by_rows)
        best_q, best = 1, 0.0
        for q in range(1, 9):
            if 8 * q > by_rows:
                break
            eff = q * tiles / (32.0 * -(-(q * tiles) // 32))
            if eff > best + 1e-9:
                best, best_q = eff, q
        return 8 * best_q
    for n, kp, pp in ((169343, 768, 1536), (2449029, 512, 1024), (132534, 512, 1024), (232965, 640, 256), (20000, 768, 1536), (3000, 768, 1536)):
        assert f(n, kp, pp) // (kp * pp) == rule(n, kp, pp), (n, kp, pp)
    assert rule(169343, 768, 1536) == 8 and rule(2449029, 512, 1024) == 56 and rule(3000, 768, 1536) == 1


def test_abort_trace_names_the_aborting_thread(tmp_path):
    """include/bot_gnn.h bot_debug_abort_trace (v19): abort() from a thread that has no Python frame of its own — how a runtime's watchdog or
    the HIP runtime would end the process — leaves that thread's native frames in the file, then Python's faulthandler prints its part and
    the process dies of SIGABRT as before."""
    import subprocess
    import sys
    trace = tmp_path / "trace.txt"
    code = ("import faulthandler, ctypes, threading; faulthandler.enable(); from bot_amd import _C; _C.debug_abort_trace(%r); "
            "t = threading.Thread(target=ctypes.CDLL(None).abort); t.start(); t.join()" % str(trace))
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == -6, (out.returncode, out.stderr[-500:])
    text = trace.read_text()
    assert "libbot_gnn abort trace: SIGABRT" in text and "abort" in text.split("===")[2] and "end of trace" in text
    assert "Fatal Python error: Aborted" in out.stderr            # the handler installed before still ran


def test_isolated_marker_contains_an_abort():
    """tests/conftest.py `isolated`: the body runs in a child pytest process; a child that ABORTS is one red test whose report carries the
    native trace, the tests after it still run, and the parent process never executes a body (VERDICT r5 #1c)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("BOT_TEST_ISOLATED_CHILD", "BOT_ABORT_TRACE_FILE")}
    out = subprocess.run([sys.executable, "-m", "pytest", "tests/isolation_cases.py", "-q", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 1, out.stdout[-3000:]
    assert "1 failed, 4 passed" in out.stdout, out.stdout[-3000:]
    assert "test_second_aborts ended with rc -6" in out.stdout or "ended with rc 134" in out.stdout, out.stdout[-3000:]
    assert "libbot_gnn abort trace: SIGABRT" in out.stdout and "Fatal Python error: Aborted" in out.stdout, out.stdout[-3000:]
