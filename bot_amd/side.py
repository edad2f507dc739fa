"""A second HIP stream for the weight-gradient products of the backward pass (round 5).

Every weight gradient of the GAT stack (`x^T d out`, a reduction over the node rows on the matrix cores: csrc/halves3.hip
`gemm_halves3_tn*`) is needed only by the optimizer, while the kernels that FOLLOW it in the serial order of the backward - the
previous layer's BatchNorm backward (HBM-bound), its transposed sparse sweep and attention backward (fabric-bound) - leave the matrix
cores idle.  `run(fn, *keep)` launches `fn()` on a side stream ordered behind everything launched so far on the current stream
(event fork, no host synchronisation) and returns at once; the current stream is joined to the side stream when the autograd engine
finishes the backward pass (engine callback), or earlier by `join()`.  Nothing is computed differently - same kernels, same operands,
separate outputs, no atomics - so every gradient is bit for bit the serial one (tests/test_gpu_parity.py::test_side_stream_bitwise).

Rules that keep this safe:
  * a tensor the side stream reads or writes is kept alive until the join (`keep` + the results): the caching allocator reuses a freed
    block in HOST order on the stream that allocated it, which the side stream's kernels are not part of;
  * a consumer of a side-produced tensor either runs on the side stream too (`produced(t)`: bot_amd.nn.fused._MergeWeight.backward) or
    joins first; autograd's own leaf accumulation only STEALS such a tensor (`.grad is None`, one contribution - checked by the caller);
  * inside a hipGraph capture the fork / join are captured as graph dependencies (event wait / record only).

`BOT_SIDE_STREAM=0` runs everything inline on the current stream.
"""
from __future__ import annotations

import os

import torch

ENABLED = os.environ.get("BOT_SIDE_STREAM", "1") != "0"
PRIORITY = int(os.environ.get("BOT_SIDE_PRIORITY", "0"))   # 0: the device's LOWEST stream priority (below torch's default-priority streams: the main
# stream's workgroups go first when both have some ready); -1: the highest (the side stream's go first).  bot_stream_create has no "default priority" form.
MERGE_ON_SIDE = os.environ.get("BOT_SIDE_MERGE", "1") != "0"
MIN_OUT = int(os.environ.get("BOT_SIDE_MIN_OUT", "0"))   # weight gradients with fewer entries (piece x piece) run inline (0: all on the side stream; the output layer's narrow
# one measured the same either way, profiles/r05_side_step_ab.txt)
FORKS = 0           # side launches since import (tests assert the path was taken)
_STREAMS = {}       # device index -> torch.cuda.Stream
_KEEP = []          # tensors alive until the join
_MARKS = set()      # storage addresses of tensors the side stream produced since the last join
LAST_MARKS = set()  # ... of the pass joined last (tests: a stolen gradient's storage is in here, a cloned one's is not)
_PENDING = [None]   # (device, the stream that forked) of the open fork, None when joined
_CB = [-1]          # id of the backward pass (graph task) an engine callback is queued for


def usable(t) -> bool:
    """Fork only inside a backward pass run by the autograd engine (the join rides on its completion callback) - and not while the
    stream is being captured into a hipGraph: a replayed multi-branch graph ran 3.7 ms SLOWER per config-2 step than the one-branch graph
    (14.4 vs 10.7 ms, profiles/r05_prefetch_and_capture.txt), which already has what the fork buys in eager mode."""
    return ENABLED and t.is_cuda and torch._C._current_graph_task_id() != -1 and not torch.cuda.is_current_stream_capturing()


def _stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _STREAMS.get(idx)
    if s is None:
        from . import _C
        # a stream of the library's own, not one of torch's 32 pooled streams (round-robin: a pooled stream held for the process lifetime
        # sooner or later IS the stream a later torch.cuda.Stream() / graph capture / process group gets)
        s = _STREAMS[idx] = _C.stream_create(torch.device("cuda", idx), high_priority=PRIORITY < 0)
    return s


def _storages(x, out):
    if isinstance(x, torch.Tensor):
        out.append(x)
    elif isinstance(x, (tuple, list)):
        for y in x:
            _storages(y, out)
    elif hasattr(x, "buf"):             # bot_amd.gemm.Halves
        out.append(x.buf)
        out.append(x.scale)


def run(fn, *keep):
    """`fn()` (kernel launches only: no host reads of device data) on the side stream, behind everything launched so far on the current
    stream.  Returns fn()'s result; its tensors must not be touched on the current stream before the join except by `run` itself."""
    global FORKS
    ts = []
    _storages(keep, ts)
    dev = ts[0].device
    cur = torch.cuda.current_stream(dev)
    if _PENDING[0] is not None and _CB[0] != torch._C._current_graph_task_id():
        join()                          # a backward pass that died with an exception never reached its join: start clean
    s = _stream(dev)
    s.wait_stream(cur)
    with torch.cuda.stream(s):
        out = fn()
    res = []
    _storages(out, res)
    _KEEP.append(ts)
    # (the RESULTS are not referenced here: they were allocated on the side stream, so a block freed early is reused only by the side
    # stream's own later allocations - and a second reference would stop autograd's leaf accumulation from stealing a gradient: it
    # would CLONE it on the main stream, beside the kernel still writing it)
    for t in res:
        _MARKS.add(t.untyped_storage().data_ptr())
    del res
    _PENDING[0] = (dev, cur)
    FORKS += 1
    task = torch._C._current_graph_task_id()
    if task != -1 and _CB[0] != task:       # (keyed by the pass: a backward that died with an exception leaves no stale flag)
        _CB[0] = task
        torch.autograd.Variable._execution_engine.queue_callback(_on_backward_end)
    return out


def unhooked(p) -> bool:
    """No tensor hook and no post-accumulate-grad hook on parameter `p`: autograd's leaf accumulation will only take the gradient over."""
    return not getattr(p, "_backward_hooks", None) and not getattr(p, "_post_accumulate_grad_hooks", None)


def produced(t) -> bool:
    """Was `t` written by the side stream since the last join (so that a consumer must run there too, or join first)?"""
    return _PENDING[0] is not None and t is not None and t.is_cuda and t.untyped_storage().data_ptr() in _MARKS


def join():
    """The current stream waits for the side stream (event only); side-produced tensors are ordinary tensors from here on."""
    if _PENDING[0] is not None:
        dev, cur = _PENDING[0]          # the stream that forked (the engine runs its callbacks on the caller's streams; the fork's own is exact)
        cur.wait_stream(_stream(dev))
        _PENDING[0] = None
    _KEEP.clear()
    if _MARKS:
        LAST_MARKS.clear()
        LAST_MARKS.update(_MARKS)
    _MARKS.clear()


def _on_backward_end():
    _CB[0] = -1
    join()
