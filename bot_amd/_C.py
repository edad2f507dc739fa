"""ctypes binding of libbot_gnn.so (include/bot_gnn.h) — the only compute backend of bot_amd.

There is no CPU or pure-PyTorch fallback: if the shared library is missing the import fails, and
every wrapper refuses non-HIP tensors.  Tensors cross the boundary as raw device pointers plus
sizes/strides; launches go to torch's current HIP stream and are not synchronised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int32, c_int64, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BOT_AMD_LIB") or os.path.join(_HERE, "lib", "libbot_gnn.so")  # override: A/B builds of the kernels
ABI_VERSION = 19

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build the gfx950 kernels first "
        "(`make -C bot_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
        "bot_amd has no CPU / PyTorch fallback."
    )
_lib = ctypes.CDLL(LIB_PATH)

_P = c_void_p
_SIGS = {
    "bot_abi_version": (ctypes.c_int, []),
    "bot_last_error": (c_char_p, []),
    "bot_last_kernel": (c_char_p, []),
    "bot_debug_abort_trace": (ctypes.c_int, [c_char_p]),
    "bot_row_plan_default_chunk": (c_int32, [c_int64]),
    "bot_row_plan_size_host": (ctypes.c_int, [_P, c_int64, c_int32, _P, _P, _P]),
    "bot_row_plan_fill_host": (ctypes.c_int, [_P, c_int64, c_int32, _P, _P, _P]),
    "bot_degrees_i64": (ctypes.c_int, [_P, c_int64, _P, _P]),
    "bot_spmm_workspace_floats": (c_int64, [c_int64, c_int32, c_int32]),
    "bot_spmm_set_layout": (ctypes.c_int, [c_int32]),
    "bot_spmm_f32": (ctypes.c_int, [_P, _P, c_int64, c_int64, _P, c_int64, _P, _P, c_int64, _P, c_int64, c_int64, _P, _P,
                                    c_int32, c_int32, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, _P]),
    "bot_spmm_blocked_f32": (ctypes.c_int, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, c_int64, _P,
                                            c_int32, c_int32, _P, c_int64, _P, c_int64, _P]),
    "bot_spmm_dot_f32": (ctypes.c_int, [_P, _P, c_int64, c_int64, _P, c_int64, _P, _P, c_int64, _P, c_int64, c_int64, _P, _P,
                                        _P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, c_int64, _P, _P, _P, _P]),
    "bot_spmm_bcast_f32": (ctypes.c_int, [_P, _P, c_int64, c_int64, _P, c_int64, _P, _P, c_int64, _P, c_int64, _P, _P, c_int32,
                                          c_int32, _P, c_int64, c_int64, _P, _P]),
    "bot_spmm_bcast_halves_f16": (ctypes.c_int, [_P, _P, c_int64, c_int64, _P, c_int64, _P, _P, c_int64, _P, c_int64, _P, _P, c_int32, c_int32, _P, _P,
                                                 c_int64, c_int64, c_int32, c_int32, _P, _P]),
    "bot_spmm_dot_bcast_f32": (ctypes.c_int, [_P, _P, c_int64, c_int64, _P, c_int64, _P, _P, c_int64, _P, c_int64, c_int64, _P,
                                              _P, _P, c_int64, c_int32, c_int32, _P, c_int64, _P, _P, _P]),
    "bot_sddmm_dot_f32": (ctypes.c_int, [_P, _P, c_int64, c_int64, _P, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64,
                                         c_int32, c_int32, _P, _P, c_int32, _P]),
    "bot_sddmm_dot_bcast_f32": (ctypes.c_int, [_P, _P, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int32,
                                               c_int32, _P, _P, _P]),
    "bot_sddmm_u_add_v_f32": (ctypes.c_int, [_P, _P, c_int64, _P, _P, c_int32, _P, _P]),
    "bot_gat_attn_fwd_f32": (ctypes.c_int, [_P, _P, c_int64, c_int64, _P, c_int64, c_int32, _P, _P, _P, _P, _P, c_float,
                                            c_int32, _P, _P, _P, c_float, c_uint64, _P, _P, _P]),
    "bot_gat_attn_bwd_f32": (ctypes.c_int, [_P, _P, c_int64, c_int64, _P, c_int64, c_int32, _P, _P, _P, _P, c_float, c_int32,
                                            _P, _P, _P, _P, _P, _P, _P, c_float, c_uint64, _P, _P]),
    "bot_gat_infer_workspace_floats": (c_int64, [c_int64, c_int32, c_int32]),
    "bot_gat_infer_f32": (ctypes.c_int, [_P, _P, c_int64, c_int64, _P, c_int64, _P, _P, c_int64, c_int64, _P, c_int64, c_int64,
                                         _P, c_int64, _P, c_int64, _P, _P, c_float, c_int32, c_int32, _P, c_int64, c_int64, _P, _P,
                                         c_int32, _P, c_int64, c_int64, _P, _P, _P]),
    "bot_segment_sum_f32": (ctypes.c_int, [_P, c_int64, c_int64, _P, c_int64, c_int32, _P, _P, c_int32, _P, _P]),
    "bot_gather_rows_f32": (ctypes.c_int, [_P, c_int64, _P, c_int64, c_int32, _P, c_int64, _P]),
    "bot_scatter_add_rows_f32": (ctypes.c_int, [_P, c_int64, _P, c_int64, c_int32, _P, c_int64, _P]),
    "bot_merge_weight_fwd_f32": (ctypes.c_int, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "bot_merge_weight_bwd_f32": (ctypes.c_int, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P]),
    "bot_edge_mlp_workspace_floats": (c_int64, []),
    "bot_edge_mlp_fwd_f32": (ctypes.c_int, [_P, c_int32, _P, _P, c_int32, _P, c_int32, c_int64, _P, _P]),
    "bot_edge_mlp_bwd_f32": (ctypes.c_int, [_P, c_int32, _P, _P, c_int32, _P, c_int32, _P, c_int64, _P, _P, _P, _P, _P]),
    "bot_random_keep_workspace_bytes": (c_int64, []),
    "bot_random_keep_u8": (ctypes.c_int, [c_int64, c_int64, c_uint64, _P, _P, _P]),
    "bot_halves_workspace_floats": (c_int64, []),
    "bot_halves_scale_f32": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, _P, _P, _P]),
    "bot_halves_split_f16": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, _P, c_int32, _P, c_int64, c_int32, _P]),
    "bot_halves_split_cols_f16": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, _P, c_int32, _P, c_int64, c_int32, c_int32, _P]),
    "bot_halves_split_heads_f16": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, c_int32, _P, _P, c_int64, c_int32, c_int32, _P]),
    "bot_gemm_halves_f32": (ctypes.c_int, [c_int32, c_int32, c_int64, c_int64, c_int64, _P, _P, c_int64, _P, c_int64, _P, c_int64,
                                           c_int32, c_int64, c_int64, c_int64, c_float, _P, c_int64, c_int32, c_int32, _P]),
    "bot_gemm_halves3_nt_f32": (ctypes.c_int, [c_int64, c_int64, c_int64, _P, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64,
                                               c_int32, _P]),
    "bot_gemm_halves3_tn_workspace_floats": (c_int64, [c_int64, c_int64, c_int64]),
    "bot_gemm_halves3_tn_f32": (ctypes.c_int, [c_int64, c_int64, c_int64, c_int64, c_int64, _P, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, _P,
                                               c_int64, _P, c_int32, _P]),
    "bot_gemm_halves3_nt_grouped_f32": (ctypes.c_int, [c_int64, c_int64, _P, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int32, _P,
                                                       c_int32, _P, _P, c_int32, _P, c_int32, _P]),
    "bot_gemm_halves3_tn_grouped_workspace_floats": (c_int64, [c_int64, c_int32]),
    "bot_gemm_halves3_tn_grouped_f32": (ctypes.c_int, [c_int64, _P, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int32, _P, _P, c_int32, _P]),
    "bot_tn_narrow_workspace_floats": (c_int64, [c_int64, c_int32, c_int32]),
    "bot_tn_narrow_f32": (ctypes.c_int, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, c_int32, _P, _P]),
    "bot_bn_act_bwd_reduce_max_f32": (ctypes.c_int, [_P, c_int64, _P, c_int64, c_int64, c_int32, _P, _P, _P, _P, c_int32, c_float, c_uint64, _P, _P, _P, _P, _P]),
    "bot_bn_bwd_bound_f32": (ctypes.c_int, [c_int32, c_int64, _P, _P, _P, c_double, _P, _P, _P, _P]),
    "bot_bn_act_bwd_apply_halves_f32": (ctypes.c_int, [_P, c_int64, _P, c_int64, c_int64, c_int32, _P, _P, _P, _P, c_int32, c_float, c_uint64, _P, _P, _P, c_double,
                                                       _P, c_int64, _P, _P, c_int64, c_int32, c_int32, c_int32, _P]),
    "bot_label_split_f32": (ctypes.c_int, [_P, c_int64, _P, c_int64, _P, c_float, c_uint64, _P, c_int32, _P, _P, _P, _P, _P]),
    "bot_build_input_f32": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, c_int32, _P, c_float, c_uint64, _P, _P, c_int64, _P]),
    "bot_node_loss_f32": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, _P, c_int64, _P, _P, c_int32, c_float, _P, c_int64, _P, c_int64, _P]),
    "bot_rmsprop_step_f32": (ctypes.c_int, [c_int32, _P, _P, _P, _P, c_float, _P, c_float, c_float, c_float, _P]),
    "bot_gemm_halves_last_algo": (ctypes.c_int, [_P, _P]),
    "bot_gemm_halves_library_version": (ctypes.c_int, [_P, _P]),
    "bot_tn_gemm_workspace_floats": (c_int64, [c_int64, c_int32, c_int32, c_int32]),
    "bot_tn_gemm_f32": (ctypes.c_int, [_P, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, c_int32, c_int32, c_int64, c_int64,
                                       c_int64, _P, _P]),
    "bot_skinny_gemm_f32": (ctypes.c_int, [_P, c_int64, _P, c_int64, c_int32, _P, c_int64, c_int64, c_int32, c_int32, c_int32, c_int32,
                                           c_int64, c_int64, c_int64, _P]),
    "bot_bn_workspace_floats": (c_int64, [c_int32]),
    "bot_colstats_f32": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, _P, _P, _P, _P]),
    "bot_colsum_f32": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, _P, _P, _P]),
    "bot_bn_stats_f32": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, c_float, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "bot_bn_stats_halves_f32": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P]),
    "bot_bn_act_fwd_halves_f32": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, _P, _P, _P, _P, c_int32, c_float, c_uint64, _P, _P, c_int64,
                                                 _P, _P, c_int64, c_int32, c_int32, _P]),
    "bot_bn_act_fwd_f32": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, _P, _P, _P, _P, c_int32, c_float, c_uint64, _P, _P, c_int64, _P]),
    "bot_bn_act_bwd_reduce_f32": (ctypes.c_int, [_P, c_int64, _P, c_int64, c_int64, c_int32, _P, _P, _P, _P, c_int32, c_float,
                                                 c_uint64, _P, _P, _P, _P, _P]),
    "bot_bn_act_bwd_apply_f32": (ctypes.c_int, [_P, c_int64, _P, c_int64, c_int64, c_int32, _P, _P, _P, _P, c_int32, c_float,
                                                c_uint64, _P, _P, _P, c_double, _P, c_int64, _P, _P]),
    "bot_halves_tn_combine_f32": (ctypes.c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P, _P, _P, c_int64, _P]),
    "bot_halves_scale_from_slots2_f32": (ctypes.c_int, [_P, c_float, _P, c_float, _P, _P]),
    "bot_halves_tail_f16": (ctypes.c_int, [c_int64, c_int32, _P, _P, _P, _P, _P, c_int64, c_int32, _P]),
    "bot_spmm_dot_halves_fits": (ctypes.c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, c_int64, c_int32]),
    "bot_spmm_dot_halves_f16": (ctypes.c_int, [_P, _P, c_int64, c_int64, _P, c_int64, _P, _P, c_int64, _P, c_int64, c_int64, _P, _P, _P, c_int64, c_int64,
                                               c_int32, c_int32, _P, _P, c_int64, c_int64, c_int32, _P, _P, _P]),
    "bot_gemm_halves3_nt_bn_rows": (c_int32, [c_int64]),
    "bot_gemm_halves3_nt3_f32": (ctypes.c_int, [c_int64, c_int64, c_int64, _P, _P, c_int64, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, c_int32, _P, c_int64,
                                                _P, c_int32, _P]),
    "bot_bn_act_bwd_reduce_partials_f32": (ctypes.c_int, [_P, c_int32, c_int32, _P, _P, _P]),
    "bot_bn_bwd_bound_partials_f32": (ctypes.c_int, [c_int32, _P, c_int32, _P, _P, c_double, _P, _P, _P, _P]),
    "bot_bn_bwd_partials_finish_f32": (ctypes.c_int, [_P, _P, c_int32, c_int32, _P, _P, c_int32, c_double, _P, _P, _P, _P]),
    "bot_gemm_halves3_nt2_f32": (ctypes.c_int, [c_int64, c_int64, c_int64, _P, _P, c_int64, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, c_int32, _P, c_int64,
                                                c_int32, _P]),
    "bot_halves_split_frag_f16": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, _P, _P, c_int32, _P]),
    "bot_gemm_halves3_tn2_f32": (ctypes.c_int, [c_int64, c_int64, c_int64, c_int64, c_int64, _P, _P, _P, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64, _P,
                                                c_int64, _P, c_int32, _P]),
    "bot_stream_create": (ctypes.c_int, [c_int32, _P]),
    "bot_gemm_halves3_nt_grouped2_f32": (ctypes.c_int, [c_int64, c_int64, _P, _P, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int32, _P,
                                                        c_int32, _P, _P, c_int32, _P, _P, _P, _P, c_int32, c_int32, _P]),
    "bot_bn_stats_halves_partials_f32": (ctypes.c_int, [_P, _P, c_int32, _P, c_int64, c_int32, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P]),
    "bot_absmax_slots": (c_int32, []),
    "bot_absmax_slots_f32": (ctypes.c_int, [_P, c_int64, c_int64, c_int32, _P, _P]),
    "bot_halves_scale_from_slots_f32": (ctypes.c_int, [_P, _P, _P]),
}
for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(_lib, _name)  # AttributeError here = header and library disagree
    _fn.restype, _fn.argtypes = _res, _args
EXPORTED = tuple(_SIGS)

if _lib.bot_abi_version() != ABI_VERSION:
    raise ImportError(f"libbot_gnn.so ABI {_lib.bot_abi_version()} != binding ABI {ABI_VERSION}; rebuild")


class BotKernelError(RuntimeError):
    pass


# bench.py sets this to a list to time individual launches with HIP events recorded on the launch stream
# (torch's current stream); entries are (kernel family, shape key, start event, end event).
PROFILE = None


PROFILE_SKIP = ()   # kernel families not to time while PROFILE is a list


def _timed(name, key, launch):
    if PROFILE is None or name in PROFILE_SKIP:
        return launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = launch()
    e1.record()
    PROFILE.append((name, key, e0, e1, _lib.bot_last_kernel().decode()))
    return rc


def _check(rc: int, what: str):
    if rc != 0:
        raise BotKernelError(f"{what} failed (rc={rc}): {_lib.bot_last_error().decode()}")


def _ptr(t):
    return None if t is None else t.data_ptr()


def _ld(t):
    """Row pitch of a row-major matrix view; torch leaves the stride of a size-1 dimension arbitrary, so a one-row matrix reports
    its width."""
    return max(int(t.stride(-2)), int(t.shape[-1])) if t.shape[-2] == 1 else int(t.stride(-2))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise BotKernelError(
                "bot_amd kernels run on MI355X only: got a CPU tensor and there is no CPU fallback "
                "(move the graph and features to the GPU)")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise BotKernelError(f"operands live on different devices ({dev} and {t.device}): graph structures and features "
                                 "must be on the same GPU")


def _f32(t, name):
    if t is not None and t.dtype != torch.float32:
        raise BotKernelError(f"{name} must be float32, got {t.dtype}")
    return t


def _i32(t, name):
    if t is not None and (t.dtype != torch.int32 or not t.is_contiguous()):
        raise BotKernelError(f"{name} must be contiguous int32")
    return t


# ------------------------------------------------------------------------------------------------ host plan
def default_chunk(nnz: int) -> int:
    return int(_lib.bot_row_plan_default_chunk(int(nnz)))


def row_plan(indptr_cpu: torch.Tensor, chunk: int):
    """Host-side work plan for one compressed direction (include/bot_gnn.h "Row plan").

    indptr_cpu: CPU int32 [n_rows+1].  Returns CPU int32 tensors (items [n_items,4], long_rows, long_ptr)."""
    assert indptr_cpu.device.type == "cpu" and indptr_cpu.dtype == torch.int32 and indptr_cpu.is_contiguous()
    n_rows = indptr_cpu.numel() - 1
    if n_rows == 0:     # a direction without rows (the halo rows of a 1-rank partition): an empty plan
        return torch.empty((0, 4), dtype=torch.int32), torch.empty((0,), dtype=torch.int32), torch.zeros((1,), dtype=torch.int32), 0
    n_items, n_long, n_slots = c_int64(), c_int64(), c_int64()
    _check(_lib.bot_row_plan_size_host(indptr_cpu.data_ptr(), n_rows, chunk, ctypes.addressof(n_items),
                                       ctypes.addressof(n_long), ctypes.addressof(n_slots)), "row_plan_size")
    items = torch.empty((n_items.value, 4), dtype=torch.int32)
    long_rows = torch.empty((n_long.value,), dtype=torch.int32)
    long_ptr = torch.empty((n_long.value + 1,), dtype=torch.int32)
    _check(_lib.bot_row_plan_fill_host(indptr_cpu.data_ptr(), n_rows, chunk, items.data_ptr(), _ptr(long_rows) if n_long.value else None,
                                       long_ptr.data_ptr()), "row_plan_fill")
    return items, long_rows, long_ptr, int(n_slots.value)


# ------------------------------------------------------------------------------------------------ kernels
# `d` is a bot_amd.graph.Direction: indptr/indices (int32, device), items/long_rows/long_ptr, sizes.

def degrees(d) -> torch.Tensor:
    _dev(d.indptr)
    out = torch.empty(d.n_rows, dtype=torch.int64, device=d.indptr.device)
    _check(_lib.bot_degrees_i64(d.indptr.data_ptr(), d.n_rows, out.data_ptr(), _stream()), "degrees")
    return out


def _slab(x, name):
    """[n,H,D] float32 view with unit inner stride -> (tensor, ld, hs)."""
    _f32(x, name)
    if x.dim() != 3:
        raise BotKernelError(f"{name} must be [n,H,D]")
    if x.stride(2) != 1 and x.shape[2] != 1:
        x = x.contiguous()
    if x.shape[1] == 1:
        return x, x.stride(0), max(x.shape[2], 1)
    return x, x.stride(0), x.stride(1)


def spmm_blocked(bp, x, w, out, addend=None):
    """L2-blocked SpMM over the tiles of `bp` (bot_amd.blocked.BlockedPlan); writes only the rows bp covers.  x / out /
    addend: [n,H,D] with contiguous rows (any row stride)."""
    H, D = x.shape[1], x.shape[2]
    _check(_timed("spmm_blocked", (H, D, w is not None), lambda: _lib.bot_spmm_blocked_f32(
        bp.tile_rows.data_ptr(), bp.ptr.data_ptr(), bp.b_src.data_ptr(), bp.b_lrow.data_ptr(), bp.b_pos.data_ptr(), bp.n_tiles,
        bp.nblk, bp.block_rows, bp.T, bp.epi, bp.round_tiles, x.data_ptr(), x.stride(0), _ptr(w), H, D, out.data_ptr(), out.stride(0),
        _ptr(addend), addend.stride(0) if addend is not None else 0, _stream())),
        "spmm_blocked")
    return out


def _rows_contiguous(t):
    """[n,H,D] whose rows are H*D contiguous floats (row stride free)."""
    return t.stride(2) == 1 and (t.shape[1] == 1 or t.stride(1) == t.shape[2])


SPMM_LAYOUT = None   # None: per direction (flat 16-byte lanes when its plan is in XCD order = the numbering has locality); "flat" / "rows": force


def spmm(d, x, w=None, wperm=None, out=None, addend=None):
    """out[r,h,:] = sum_k w[wperm[k],h] * x[indices[k],h,:] (+ addend[r,h,:])   (w None: plain sum).  x: [n_src,H,D]."""
    _dev(x, w, d.indptr)
    x, ldx, hsx = _slab(x, "x")
    if wperm is None and out is None and _rows_contiguous(x) and (
            addend is None or (addend.dim() == 3 and addend.dtype == torch.float32 and _rows_contiguous(addend))):
        from . import blocked
        bp = blocked.plan_for(d, x.shape[0], x.shape[1], x.shape[2])
        if bp is not None:
            # the plan's LDS layout assumes the vector width blocked.layout() derives from D alone; the C side re-derives it
            # from the operands' alignment too — a slab whose base or row stride is not a multiple of it takes the row kernel
            vec = blocked.layout(x.shape[1], x.shape[2])[0]
            ok = lambda t: t is None or (t.data_ptr() % (4 * vec) == 0 and t.stride(0) % vec == 0)
            if not (ok(x) and ok(addend)):
                bp = None
        if bp is not None:  # dense graph: L2-blocked sweep for the regular rows, row-per-group kernel for the hubs
            if w is not None:
                w = _f32(w, "w").contiguous()
            out = torch.empty((d.n_rows, x.shape[1], x.shape[2]), dtype=torch.float32, device=x.device)
            spmm_blocked(bp, x, w, out, addend)
            if bp.heavy is not None:
                spmm(bp.heavy, x, w, None, out=out, addend=addend)
            return out
    lda = hsa = 0
    if addend is not None:
        addend, lda, hsa = _slab(addend, "addend")
    H, D = x.shape[1], x.shape[2]
    if w is not None:
        _f32(w, "w")
        w = w.contiguous()
        if w.dim() != 2 or w.shape[1] != H:
            raise BotKernelError(f"w must be [nnz,{H}], got {tuple(w.shape)}")
    if out is None:
        out = torch.empty((d.n_rows, H, D), dtype=torch.float32, device=x.device)
    out_, ldo, hso = _slab(out, "out")
    assert out_ is out
    partial = None
    if d.n_long:
        partial = torch.empty(int(_lib.bot_spmm_workspace_floats(d.n_slots, H, D)), dtype=torch.float32, device=x.device)
    flat = SPMM_LAYOUT == "flat" or (SPMM_LAYOUT is None and getattr(d, "plan_order", "degree") == "xcd")
    _lib.bot_spmm_set_layout(1 if flat else 0)     # thread-local hint; identical results either way (include/bot_gnn.h)
    _check(_timed("spmm", (H, D, w is not None), lambda: _lib.bot_spmm_f32(
        d.indptr.data_ptr(), d.indices.data_ptr(), d.n_rows, d.nnz, d.items.data_ptr(), d.n_items, _ptr(d.long_rows),
        _ptr(d.long_ptr), d.n_long, x.data_ptr(), ldx, hsx, _ptr(w), _ptr(_i32(wperm, "wperm")), H, D, out.data_ptr(), ldo, hso,
        _ptr(addend), lda, hsa, _ptr(partial), _stream())), "spmm")
    return out


def spmm_dot_max_d(x):
    """Largest D the fused spmm_dot launch covers for this slab's alignment."""
    return 1024 if (x.shape[2] % 4 == 0 and x.stride(0) % 4 == 0) else (512 if x.shape[2] % 2 == 0 and x.stride(0) % 2 == 0 else 256)


def spmm_dot(d, x, w, wperm, y, out=None, dot=None, absmax=None):
    """Fused backward of u_mul_e_sum on direction `d`:  out[r,h,:] = sum_k w[wperm[k],h] x[indices[k],h,:]  and
    dot[wperm[k],h] = <y[r,h,:], x[indices[k],h,:]>.  Returns (out [n_rows,H,D], dot [nnz,H]).
    `dot`: write into this [E,H] array instead of a fresh one — `d` may be a PART of a larger direction (Graph.halo_split) whose
    entries are reached through `wperm`; the parts of one direction fill disjoint rows of the same array.
    `absmax`: `absmax_slots()` words that receive max|out| as a by-product (include/bot_gnn.h "Maxima as by-products")."""
    _dev(x, w, y, d.indptr)
    x, ldx, hsx = _slab(x, "x")
    y, ldy, hsy = _slab(y, "y")
    H, D = x.shape[1], x.shape[2]
    w = _f32(w, "w").contiguous()
    if out is None:
        out = torch.empty((d.n_rows, H, D), dtype=torch.float32, device=x.device)
    out_, ldo, hso = _slab(out, "out")
    assert out_ is out
    if dot is None:
        dot = torch.empty((d.nnz, H), dtype=torch.float32, device=x.device)
    elif dot.dtype != torch.float32 or not dot.is_contiguous() or dot.dim() != 2 or dot.shape[1] != H:
        raise BotKernelError(f"spmm_dot: dot must be a contiguous float32 [E,{H}] array")
    partial = None
    if d.n_long:
        partial = torch.empty(int(_lib.bot_spmm_workspace_floats(d.n_slots, H, D)), dtype=torch.float32, device=x.device)
    _check(_timed("spmm_dot", (H, D), lambda: _lib.bot_spmm_dot_f32(
        d.indptr.data_ptr(), d.indices.data_ptr(), d.n_rows, d.nnz, d.items.data_ptr(), d.n_items, _ptr(d.long_rows),
        _ptr(d.long_ptr), d.n_long, x.data_ptr(), ldx, hsx, w.data_ptr(), _ptr(_i32(wperm, "wperm")), y.data_ptr(), ldy, hsy,
        H, D, out.data_ptr(), ldo, hso, dot.data_ptr(), _ptr(partial), _ptr(absmax), _stream())), "spmm_dot")
    return out, dot


def spmm_dot_halves_fits(x, y, hout, hsh, h2_off) -> bool:
    """Would `spmm_dot_halves` take these operands (x, y: [n, H, D] slabs; hout: the operand's [n, >= h2_off + H hsh] fp16 view)?"""
    x, ldx, hsx = _slab(x, "x")
    y, ldy, hsy = _slab(y, "y")
    return bool(_lib.bot_spmm_dot_halves_fits(x.data_ptr(), ldx, hsx, y.data_ptr(), ldy, hsy, x.shape[1], x.shape[2], hout.data_ptr(), hout.stride(0),
                                              int(hsh), int(h2_off)))


def spmm_dot_halves(d, x, w, wperm, y, hscale, hout, hsh, h2_off, dot=None):
    """`spmm_dot` whose first result leaves the kernel as a LEFT halves operand: hout[r, h * hsh + e] = h1, hout[r, h2_off + h * hsh + e] =
    2^11 h2 of hscale[0] * out[r,h,e] (bot_spmm_dot_halves_f16; all-heads layout only: ask spmm_dot_halves_fits first).  Returns dot [nnz, H]."""
    _dev(x, w, y, d.indptr, hout, hscale)
    x, ldx, hsx = _slab(x, "x")
    y, ldy, hsy = _slab(y, "y")
    H, D = x.shape[1], x.shape[2]
    w = _f32(w, "w").contiguous()
    assert hout.dtype == torch.float16 and hout.stride(1) == 1 and hout.shape[0] == d.n_rows
    if dot is None:
        dot = torch.empty((d.nnz, H), dtype=torch.float32, device=x.device)
    elif dot.dtype != torch.float32 or not dot.is_contiguous() or dot.dim() != 2 or dot.shape[1] != H:
        raise BotKernelError(f"spmm_dot_halves: dot must be a contiguous float32 [E,{H}] array")
    partial = None
    if d.n_long:
        partial = torch.empty(int(_lib.bot_spmm_workspace_floats(d.n_slots, H, D)), dtype=torch.float32, device=x.device)
    _check(_timed("spmm_dot", (H, D), lambda: _lib.bot_spmm_dot_halves_f16(
        d.indptr.data_ptr(), d.indices.data_ptr(), d.n_rows, d.nnz, d.items.data_ptr(), d.n_items, _ptr(d.long_rows),
        _ptr(d.long_ptr), d.n_long, x.data_ptr(), ldx, hsx, w.data_ptr(), _ptr(_i32(wperm, "wperm")), y.data_ptr(), ldy, hsy,
        H, D, hscale.data_ptr(), hout.data_ptr(), hout.stride(0), int(hsh), int(h2_off), dot.data_ptr(), _ptr(partial), _stream())), "spmm_dot_halves")
    return dot


def spmm_bcast(d, x, w, wperm=None, head_outer=True):
    """out[r,h,:] = sum_k w[wperm[k],h] * x[indices[k],:]   x: [n_src, D] (row stride allowed), w: [nnz, H<=4].
    Returns [H, n_rows, D] (head_outer, the batched-GEMM layout) or [n_rows, H, D]."""
    _dev(x, w, d.indptr)
    _f32(x, "x")
    if x.stride(1) != 1:
        x = x.contiguous()
    w = _f32(w, "w").contiguous()
    H, D = w.shape[1], x.shape[1]
    if head_outer:
        out = torch.empty((H, d.n_rows, D), dtype=torch.float32, device=x.device)
        ldo, hso = D, d.n_rows * D
    else:
        out = torch.empty((d.n_rows, H, D), dtype=torch.float32, device=x.device)
        ldo, hso = H * D, D
    partial = None
    if d.n_long:
        partial = torch.empty(int(_lib.bot_spmm_workspace_floats(d.n_slots, H, D)), dtype=torch.float32, device=x.device)
    _check(_timed("spmm_bcast", (H, D), lambda: _lib.bot_spmm_bcast_f32(
        d.indptr.data_ptr(), d.indices.data_ptr(), d.n_rows, d.nnz, d.items.data_ptr(), d.n_items, _ptr(d.long_rows),
        _ptr(d.long_ptr), d.n_long, x.data_ptr(), x.stride(0), w.data_ptr(), _ptr(_i32(wperm, "wperm")), H, D, out.data_ptr(),
        ldo, hso, _ptr(partial), _stream())), "spmm_bcast")
    return out


def spmm_bcast_halves(d, x, w, wperm, hscale, hout, col0, hsh, h2_off, hpiece):
    """spmm_bcast with the result written as fp16 halves [h1 | 2^11 h2] of hscale[0] * out into the operand buffer `hout` [n_rows, ld]:
    head h's row block at columns col0 + h hsh .. + hpiece - 1 (zeros behind the D = x.shape[1] columns), the second half h2_off columns
    behind the first — the aggregated slab of the aggregate-first GAT layer as a GEMM operand, no fp32 copy (bot_spmm_bcast_halves_f16)."""
    _dev(x, w, d.indptr, hout, hscale)
    _f32(x, "x")
    if x.stride(1) != 1:
        x = x.contiguous()
    w = _f32(w, "w").contiguous()
    H, D = w.shape[1], x.shape[1]
    assert hout.dtype == torch.float16 and hout.dim() == 2 and hout.shape[0] == d.n_rows and hout.stride(1) == 1 and col0 % 4 == 0
    assert col0 + h2_off + (H - 1) * hsh + hpiece <= hout.shape[1]
    partial = None
    if d.n_long:
        partial = torch.empty(int(_lib.bot_spmm_workspace_floats(d.n_slots, H, D)), dtype=torch.float32, device=x.device)
    _check(_timed("spmm_bcast", (H, D), lambda: _lib.bot_spmm_bcast_halves_f16(
        d.indptr.data_ptr(), d.indices.data_ptr(), d.n_rows, d.nnz, d.items.data_ptr(), d.n_items, _ptr(d.long_rows),
        _ptr(d.long_ptr), d.n_long, x.data_ptr(), x.stride(0), w.data_ptr(), _ptr(_i32(wperm, "wperm")), H, D, hscale.data_ptr(),
        hout.data_ptr() + 2 * col0, hout.stride(0), hsh, h2_off, hpiece, _ptr(partial), _stream())), "spmm_bcast_halves")
    return hout


def spmm_dot_bcast(d, x, w, wperm, y, out=None):
    """Backward of spmm_bcast on direction `d`: x [H, n, D] head-outer (gradient of the aggregated slab), y [n_rows, D] the
    layer input.  Returns (out [n_rows, D] = sum_k sum_h w x, dot [nnz, H] = <y[r], x[indices[k],h]>); `out` may be a
    row-strided [n_rows, D] view."""
    _dev(x, w, y, d.indptr)
    _f32(x, "x"), _f32(y, "y")
    assert x.dim() == 3 and x.is_contiguous()
    if y.stride(1) != 1:
        y = y.contiguous()
    H, n, D = x.shape
    w = _f32(w, "w").contiguous()
    if out is None:
        out = torch.empty((d.n_rows, D), dtype=torch.float32, device=x.device)
    assert out.stride(1) == 1 and out.shape == (d.n_rows, D)
    dot = torch.empty((d.nnz, H), dtype=torch.float32, device=x.device)
    partial = None
    if d.n_long:
        partial = torch.empty(int(_lib.bot_spmm_workspace_floats(d.n_slots, 1, D)), dtype=torch.float32, device=x.device)
    _check(_timed("spmm_dot_bcast", (H, D), lambda: _lib.bot_spmm_dot_bcast_f32(
        d.indptr.data_ptr(), d.indices.data_ptr(), d.n_rows, d.nnz, d.items.data_ptr(), d.n_items, _ptr(d.long_rows),
        _ptr(d.long_ptr), d.n_long, x.data_ptr(), D, n * D, w.data_ptr(), _ptr(_i32(wperm, "wperm")), y.data_ptr(), y.stride(0),
        H, D, out.data_ptr(), out.stride(0), dot.data_ptr(), _ptr(partial), _stream())), "spmm_dot_bcast")
    return out, dot


def sddmm_dot(d, x, y, operm=None, out=None):
    """out[operm[k],h] = <x[indices[k],h,:], y[r,h,:]> for every position k of every row r.  -> [nnz,H]"""
    _dev(x, y, d.indptr)
    x, ldx, hsx = _slab(x, "x")
    y, ldy, hsy = _slab(y, "y")
    H, D = x.shape[1], x.shape[2]
    if out is None:
        out = torch.empty((d.nnz, H), dtype=torch.float32, device=x.device)
    _check(_timed("sddmm_dot", (H, D), lambda: _lib.bot_sddmm_dot_f32(
        d.indptr.data_ptr(), d.indices.data_ptr(), d.n_rows, d.nnz, d.items.data_ptr(), d.n_items, x.data_ptr(), ldx, hsx,
        y.data_ptr(), ldy, hsy, H, D, out.data_ptr(), _ptr(_i32(operm, "operm")), 0, _stream())), "sddmm_dot")
    return out


def sddmm_dot_bcast(d, x, y, operm=None):
    """out[operm[k],h] = <x[indices[k],:], y[h,r,:]> — x [n_src, D] (row stride allowed), y [H, n_rows, D] head-outer.  -> [nnz,H]"""
    _dev(x, y, d.indptr)
    _f32(x, "x"), _f32(y, "y")
    if x.stride(1) != 1:
        x = x.contiguous()
    assert y.dim() == 3 and y.is_contiguous()
    H, n, D = y.shape
    out = torch.empty((d.nnz, H), dtype=torch.float32, device=x.device)
    _check(_timed("sddmm_dot_bcast", (H, D), lambda: _lib.bot_sddmm_dot_bcast_f32(
        d.indptr.data_ptr(), d.indices.data_ptr(), d.n_rows, d.nnz, d.items.data_ptr(), d.n_items, x.data_ptr(), x.stride(0),
        y.data_ptr(), D, n * D, H, D, out.data_ptr(), _ptr(_i32(operm, "operm")), _stream())), "sddmm_dot_bcast")
    return out


def u_add_v(src, dst, x, y=None):
    """out[e,:] = x[src[e],:] (+ y[dst[e],:]).  x, y: [n,W] float32; src/dst: int32 [E]."""
    _dev(x, y, src)
    x = _f32(x, "x").contiguous()
    y = None if y is None else _f32(y, "y").contiguous()
    E, W = src.numel(), x.shape[1]
    out = torch.empty((E, W), dtype=torch.float32, device=x.device)
    _check(_lib.bot_sddmm_u_add_v_f32(_i32(src, "src").data_ptr(), _ptr(_i32(dst, "dst")), E, x.data_ptr(), _ptr(y), W,
                                      out.data_ptr(), _stream()), "u_add_v")
    return out


def zsign_buffer(d, H, slope):
    """uint8 [nnz] for the forward to record [z > 0] per head (H <= 8, slope != 1), else None: the backward then skips the
    el[src] gather and the ee / edge-id reads."""
    if os.environ.get("BOT_NO_ZSIGN"):  # measurements: the backward re-derives the signs from el / er / ee
        return None
    return torch.empty(d.nnz, dtype=torch.uint8, device=d.indptr.device) if (H <= 8 and slope != 1.0 and d.indptr.is_cuda) else None


def gat_attn_fwd(d, el, er, ee, eperm, keep, slope, H, aperm, zsign=None, drop=None):
    """Fused logits + leaky-ReLU + per-row softmax.  el/er: [n,H]; ee: [nnz,H] via eperm; -> a [nnz,H].
    `zsign`: optional uint8 [nnz] output (see zsign_buffer).  `drop` = (p, seed) with p > 0: also returns
    a_drop = a * keep / (1 - p) (the attention dropout of models.py:544, Philox mask) -> (a, a_drop)."""
    _dev(el, er, ee, d.indptr)
    el = None if el is None else _f32(el, "el").contiguous()
    er = None if er is None else _f32(er, "er").contiguous()
    ee = None if ee is None else _f32(ee, "ee").contiguous()
    if keep is not None and (keep.dtype != torch.uint8 or not keep.is_contiguous()):
        raise BotKernelError("keep must be contiguous uint8")
    a = torch.empty((d.nnz, H), dtype=torch.float32, device=d.indptr.device)
    p, seed = drop if drop is not None else (0.0, 0)
    a_drop = torch.empty_like(a) if p > 0 else None
    _check(_lib.bot_gat_attn_fwd_f32(d.indptr.data_ptr(), d.indices.data_ptr(), d.n_rows, d.nnz, _ptr(d.long_rows), d.n_long,
                                     d.chunk, _ptr(el), _ptr(er), _ptr(ee), _ptr(_i32(eperm, "eperm")), _ptr(keep),
                                     float(slope), H, a.data_ptr(), _ptr(_i32(aperm, "aperm")), _ptr(zsign), float(p),
                                     int(seed) & 0xFFFFFFFFFFFFFFFF, _seed_off(p), _ptr(a_drop), _stream()), "gat_attn_fwd")
    return a if drop is None else (a, a_drop if a_drop is not None else a)


def gat_attn_bwd(d, el, er, ee, eperm, slope, H, a, da, aperm, zperm, want_der, zsign=None, drop=None):
    """Backward of gat_attn_fwd -> (dz [nnz,H] at zperm, der [n_rows,H] or None).  `zsign`: what the forward recorded.
    `drop` = the forward's (p, seed): `da` is then the gradient of a_drop."""
    _dev(a, da, d.indptr)
    el = None if el is None else _f32(el, "el").contiguous()
    er = None if er is None else _f32(er, "er").contiguous()
    ee = None if ee is None else _f32(ee, "ee").contiguous()
    a = _f32(a, "a").contiguous()
    da = _f32(da, "da").contiguous()
    dz = torch.empty((d.nnz, H), dtype=torch.float32, device=a.device)
    der = torch.empty((d.n_rows, H), dtype=torch.float32, device=a.device) if want_der else None
    _check(_lib.bot_gat_attn_bwd_f32(d.indptr.data_ptr(), d.indices.data_ptr(), d.n_rows, d.nnz, _ptr(d.long_rows), d.n_long,
                                     d.chunk, _ptr(el), _ptr(er), _ptr(ee), _ptr(_i32(eperm, "eperm")), float(slope), H,
                                     a.data_ptr(), da.data_ptr(), _ptr(_i32(aperm, "aperm")), dz.data_ptr(),
                                     _ptr(_i32(zperm, "zperm")), _ptr(der), _ptr(zsign), float(drop[0]) if drop else 0.0,
                                     (int(drop[1]) & 0xFFFFFFFFFFFFFFFF) if drop else 0, _seed_off(drop[0]) if drop else None,
                                     _stream()), "gat_attn_bwd")
    return dz, der


def gat_infer(d, x, el=None, er=None, ee=None, ew=None, slope=0.2, addend=None, scale=None, shift=None, relu=False, out=None, absmax=None):
    """Inference-only GAT layer in one sweep (include/bot_gnn.h bot_gat_infer_f32):
        out[r,h,:] = act((sum_k softmax_k(leaky(el[src] + er[r] + ee[k]))[h] * ew[k] * x[src,h,:] + addend[r,h,:]) * scale + shift)
    x: [n_src,H,D] (strided slab); el [n_src,H] / er [n_rows,H]: row-strided views allowed; ee [nnz,H], ew [nnz]: position order;
    addend [n_rows,H,D]; scale / shift [H*D].  el None: plain ew-weighted sum (no softmax).  Returns [n_rows,H,D]."""
    _dev(x, el, er, ee, ew, addend, scale, shift, d.indptr)
    x, ldx, hsx = _slab(x, "x")
    H, D = x.shape[1], x.shape[2]

    def node2(t, name):
        if t is None:
            return None, 0
        _f32(t, name)
        t = t.reshape(t.shape[0], -1) if t.dim() != 2 else t
        if t.shape[1] != H:
            raise BotKernelError(f"{name} must be [n,{H}], got {tuple(t.shape)}")
        if t.stride(1) != 1 and H > 1:
            t = t.contiguous()
        return t, (t.stride(0) if t.shape[0] > 1 else H)
    el, ldel = node2(el, "el")
    er, lder = node2(er, "er")
    if ee is not None:
        ee = _f32(ee, "ee").reshape(-1, H).contiguous()
    if ew is not None:
        ew = _f32(ew, "ew").reshape(-1).contiguous()
    lda = hsa = 0
    if addend is not None:
        addend, lda, hsa = _slab(addend, "addend")
    scale = None if scale is None else _f32(scale, "scale").reshape(-1).contiguous()
    shift = None if shift is None else _f32(shift, "shift").reshape(-1).contiguous()
    if out is None:
        out = torch.empty((d.n_rows, H, D), dtype=torch.float32, device=x.device)
    out_, ldo, hso = _slab(out, "out")
    assert out_ is out
    ws = None
    if d.n_long:
        ws = torch.empty(int(_lib.bot_gat_infer_workspace_floats(d.n_slots, H, D)), dtype=torch.float32, device=x.device)
    _check(_timed("gat_infer", (H, D, el is not None), lambda: _lib.bot_gat_infer_f32(
        d.indptr.data_ptr(), d.indices.data_ptr(), d.n_rows, d.nnz, d.items.data_ptr(), d.n_items, _ptr(d.long_rows),
        _ptr(d.long_ptr), d.n_long, d.n_slots, x.data_ptr(), ldx, hsx, _ptr(el), ldel, _ptr(er), lder, _ptr(ee), _ptr(ew),
        float(slope), H, D, _ptr(addend), lda, hsa, _ptr(scale), _ptr(shift), int(bool(relu)), out.data_ptr(), ldo, hso, _ptr(ws),
        _ptr(absmax), _stream())), "gat_infer")
    return out


def segment_sum(d, vals, perm=None):
    """out[r,:] = sum_k vals[perm[k],:].  vals: [nnz,W] -> [n_rows,W]"""
    _dev(vals, d.indptr)
    vals = _f32(vals, "vals").contiguous()
    W = vals.shape[1]
    out = torch.empty((d.n_rows, W), dtype=torch.float32, device=vals.device)
    _check(_lib.bot_segment_sum_f32(d.indptr.data_ptr(), d.n_rows, d.nnz, _ptr(d.long_rows), d.n_long, d.chunk, vals.data_ptr(),
                                    _ptr(_i32(perm, "perm")), W, out.data_ptr(), _stream()), "segment_sum")
    return out


def gather_rows(x, rows, out=None):
    """out[i,:] = x[rows[i],:]   x: [n,F] float32 (row stride allowed), rows: int32; `out` may be row-strided."""
    _dev(x, rows)
    _f32(x, "x")
    if x.stride(1) != 1:
        x = x.contiguous()
    if out is None:
        out = torch.empty((rows.numel(), x.shape[1]), dtype=torch.float32, device=x.device)
    assert out.stride(1) == 1 and out.shape == (rows.numel(), x.shape[1])
    _check(_lib.bot_gather_rows_f32(x.data_ptr(), x.stride(0), _i32(rows, "rows").data_ptr(), rows.numel(), x.shape[1],
                                    out.data_ptr(), out.stride(0), _stream()), "gather_rows")
    return out


def scatter_add_rows(x, rows, vals):
    """x[rows[i],:] += vals[i,:] in place; rows sorted-unique int32."""
    _dev(x, rows, vals)
    _f32(x, "x"), _f32(vals, "vals")
    assert x.stride(1) == 1
    vals = vals.contiguous()
    _check(_lib.bot_scatter_add_rows_f32(x.data_ptr(), x.stride(0), _i32(rows, "rows").data_ptr(), rows.numel(), x.shape[1],
                                         vals.data_ptr(), vals.stride(0), _stream()), "scatter_add_rows")
    return x


# ------------------------------------------------------------------------------------------------ BatchNorm + ReLU + dropout
# Device word mixed into every fused-dropout Philox key (include/bot_gnn.h `seed_offset`).  None = not used.  A hipGraph-captured
# train step (bot_amd.train.CapturedTrainStep) allocates it and bumps it once per replay, so that replays draw fresh masks.
SEED_OFFSET = None


def _seed_off(p):
    return SEED_OFFSET.data_ptr() if (SEED_OFFSET is not None and p > 0) else None


def _mat(x, name):
    _f32(x, name)
    if x.dim() != 2:
        raise BotKernelError(f"{name} must be [n,F]")
    return x if x.stride(1) == 1 else x.contiguous()


def _bn_ws(F, device):
    return torch.empty(int(_lib.bot_bn_workspace_floats(F)), dtype=torch.float32, device=device)


def colstats(x):
    """Per-column mean and M2 = sum (x - mean)^2 of x [n,F] -> (mean [F], m2 [F])."""
    _dev(x)
    x = _mat(x, "x")
    n, F = x.shape
    mean = torch.empty(F, dtype=torch.float32, device=x.device)
    m2 = torch.empty(F, dtype=torch.float32, device=x.device)
    _check(_lib.bot_colstats_f32(x.data_ptr(), x.stride(0), n, F, mean.data_ptr(), m2.data_ptr(), _bn_ws(F, x.device).data_ptr(),
                                 _stream()), "colstats")
    return mean, m2


def halves_scale(x):
    """scale [2] = (s, 1/s), s = 2^(14 - ceil(log2 max|x|)) of x [n,F] — stays on the device."""
    _dev(x)
    x = _mat(x, "x")
    scale = torch.empty(2, dtype=torch.float32, device=x.device)
    ws = torch.empty(int(_lib.bot_halves_workspace_floats()), dtype=torch.float32, device=x.device)
    _check(_lib.bot_halves_scale_f32(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], scale.data_ptr(), ws.data_ptr(), _stream()),
           "halves_scale")
    return scale


_SLOT_POOL = {}          # device -> [zeroed block [_SLOT_SETS, slots], next free set]
_SLOT_SETS = 512


def absmax_slots(device):
    """Zeroed words for the producers' max|value| by-products (include/bot_gnn.h "Maxima as by-products").  Handed out from a block zeroed
    by ONE fill per _SLOT_SETS requests (a set is used once: a fill launch per request was 5.7 us of the step each, seven times a step);
    a block lives until its last set is dropped.  Inside a hipGraph capture every request is its own captured fill (a replay must find
    zeros again)."""
    n = int(_lib.bot_absmax_slots())
    dev = torch.device(device)
    if dev.type != "cuda" or torch.cuda.is_current_stream_capturing():
        return torch.zeros(n, dtype=torch.int32, device=dev)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ent = _SLOT_POOL.get(key)
    if ent is None or ent[1] >= _SLOT_SETS:
        ent = _SLOT_POOL[key] = [torch.zeros((_SLOT_SETS, n), dtype=torch.int32, device=dev), 0]
    ent[1] += 1
    return ent[0][ent[1] - 1]


def absmax_into(x, slots):
    """Fold max|x| of a (row-strided) [n,F] float32 matrix into the slots — for the columns no producer kernel covers."""
    _dev(x, slots)
    x = _mat(x, "x")
    _check(_lib.bot_absmax_slots_f32(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], slots.data_ptr(), _stream()), "absmax_slots")
    return slots


def halves_scale_from_slots(slots, mult=None, cap=None, cap_ratio=1.0):
    """scale [2] = (s, 1/s) from the by-product slots: what halves_scale finds with a pass over the matrix.  `mult`: the scale of mult x
    the slots' maximum (a bound assembled by the caller); `cap`: another (s, 1/s) pair - the result is at most cap[0] * cap_ratio
    (bot_halves_scale_from_slots2_f32)."""
    scale = torch.empty(2, dtype=torch.float32, device=slots.device)
    if mult is None and cap is None:
        _check(_lib.bot_halves_scale_from_slots_f32(slots.data_ptr(), scale.data_ptr(), _stream()), "halves_scale_from_slots")
    else:
        _dev(cap)
        _check(_lib.bot_halves_scale_from_slots2_f32(slots.data_ptr(), float(1.0 if mult is None else mult), _ptr(cap), float(cap_ratio), scale.data_ptr(),
                                                     _stream()), "halves_scale_from_slots2")
    return scale


def halves_tail(segments, scale, out, h2_off):
    """Column segments of the LEFT halves operand `out` [n, ld] (h1 at column c, 2^11 h2 at c + h2_off) from small fp32 sources, or zeroed:
    segments = [(col, width, src [n, width] float32 (row-strided) or None), ...]  (bot_halves_tail_f16)."""
    _dev(out, scale)
    assert out.dtype == torch.float16 and out.stride(1) == 1 and 1 <= len(segments) <= 8
    n = out.shape[0]
    cols = (c_int64 * (2 * len(segments)))()
    srcs = (c_void_p * len(segments))()
    lds = (c_int64 * len(segments))()
    for g, (col, width, src) in enumerate(segments):
        cols[2 * g], cols[2 * g + 1] = int(col), int(width)
        if src is not None:
            _dev(src)
            _f32(src, "segment source")
            if src.shape != (n, width) or src.stride(1) != 1:
                raise BotKernelError(f"halves_tail: segment {g}: source shape {tuple(src.shape)} / stride {tuple(src.stride())} for width {width}")
            srcs[g], lds[g] = src.data_ptr(), src.stride(0)
        else:
            srcs[g], lds[g] = None, 0
    _check(_lib.bot_halves_tail_f16(n, len(segments), cols, srcs, lds, scale.data_ptr(), out.data_ptr(), out.stride(0), int(h2_off), _stream()), "halves_tail")
    return out


def halves_tn_combine(a, b, P, rem_a=None, rem_b=None):
    """out [K, P] = sum_c a[c][:, :P] + (sum_c a[c][:, PP:PP+P] + sum_c b[c][:, :P]) * 2^-11 for a [C, K, 2 PP], b [C, K, PP] contiguous
    (+ remainder chunks rem_a [K, 2 PP] / rem_b [K, PP]) — include/bot_gnn.h bot_halves_tn_combine_f32."""
    _dev(a, b)
    a, b = _f32(a, "a").contiguous(), _f32(b, "b").contiguous()
    if a.dim() == 2:
        a, b = a.unsqueeze(0), b.unsqueeze(0)
    C, K, PP2 = a.shape
    PP = PP2 // 2
    assert b.shape == (C, K, PP) and P <= PP
    if rem_a is not None:
        rem_a, rem_b = rem_a.contiguous(), rem_b.contiguous()
        assert rem_a.shape == (K, PP2) and rem_b.shape == (K, PP)
    out = torch.empty((K, P), dtype=torch.float32, device=a.device)
    _check(_lib.bot_halves_tn_combine_f32(a.data_ptr(), b.data_ptr(), C, K, PP, P, _ptr(rem_a), _ptr(rem_b), out.data_ptr(), P, _stream()),
           "halves_tn_combine")
    return out


def halves_split(x, scale, order, piece, out=None):
    """fp16 halves of x [n,F] scaled by scale[0]: [h1|h1|h2] (order 0), [h1|h2|h1] (order 1) or [h1|h2] (order 2: a left operand without
    the duplicate piece), pieces `piece` columns wide."""
    _dev(x, scale)
    x = _mat(x, "x")
    n, F = x.shape
    if out is None:
        out = torch.empty((n, (2 if order == 2 else 3) * piece), dtype=torch.float16, device=x.device)
    _check(_lib.bot_halves_split_f16(x.data_ptr(), x.stride(0), n, F, _ptr(scale), order, out.data_ptr(), out.stride(0), piece, _stream()),
           "halves_split")
    return out


def halves_split_cols(x, scale, order, buf, piece, col, width):
    """Split x [n,F] into columns [col, col + width) of the three pieces of the halves operand `buf` [n, 3 * piece] (zeros behind
    the F columns of x): several matrices side by side as ONE operand under one scale — bot_halves_split_cols_f16.  order 2: the two
    pieces [h1 | 2^11 h2] of a buffer [n, >= 2 * piece], `piece` = the distance of the second half from the first."""
    _dev(x, scale, buf)
    x = _mat(x, "x")
    n, F = x.shape
    assert buf.dtype == torch.float16 and buf.shape[0] == n and buf.shape[1] >= (2 if order == 2 else 3) * piece and buf.stride(1) == 1
    assert col % 2 == 0 and col + width <= piece
    _check(_lib.bot_halves_split_cols_f16(x.data_ptr(), x.stride(0), n, F, _ptr(scale), order, buf.data_ptr() + 2 * col, buf.stride(0), piece,
                                          width, _stream()), "halves_split_cols")
    return buf


def halves_split_heads(x, scale, H, D, DP, out=None):
    """x [n, H * D] (row-strided view allowed) -> the LEFT operand [n, 2 * H * DP] = [h1 | 2^11 h2] with every head's D columns in a block of
    DP (zero padded) — bot_halves_split_heads_f16."""
    _dev(x, scale)
    _f32(x, "x")
    n = x.shape[0]
    assert x.shape[1] == H * D and x.stride(1) == 1
    if out is None:
        out = torch.empty((n, 2 * H * DP), dtype=torch.float16, device=x.device)
    _check(_lib.bot_halves_split_heads_f16(x.data_ptr(), x.stride(0), n, H, D, _ptr(scale), out.data_ptr(), out.stride(0), H * DP, DP, _stream()),
           "halves_split_heads")
    return out


_GEMM_WS = {}
# Kernel choice for shapes without a recorded selection.  0 (default): hipBLASLt's first heuristic choice — no timing runs, no
# synchronisation, the same kernel in every run and on every rank.  Opt-in (maintenance / exploration, NOT bitwise reproducible across
# runs, synchronises on the first call per shape): 1 = fastest of 16 heuristic candidates timed on the caller's buffers, 2 = of all
# solutions (tools/tune_halves_gemm.py, which records the winners in bot_amd/tuning/halves_gemm.json).
GEMM_TUNE = int(os.environ.get("BOT_GEMM_TUNE", "0"))
GEMM_ALGOS = None      # {shape key: hipBLASLt solution index} recorded by tools/tune_halves_gemm.py (bot_amd/tuning/halves_gemm.json)
GEMM_SEEN = None       # tools set this to a dict to collect {shape key: (solution index, ms in the search)} of the launches


def _gemm_algos():
    global GEMM_ALGOS
    if GEMM_ALGOS is None:
        GEMM_ALGOS = {}
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "halves_gemm.json")
        if os.environ.get("BOT_GEMM_ALGOS", "1") != "0" and os.path.exists(path):
            import json
            with open(path) as f:
                rec = json.load(f)
            # indices are only meaningful for the library build they were recorded on
            if rec.get("hipblaslt") == _hipblaslt_tag():
                GEMM_ALGOS = {k: int(v["index"]) for k, v in rec.get("shapes", {}).items()}
    return GEMM_ALGOS


def _hipblaslt_tag():
    """Identifies the hipBLASLt build whose solution indices the tuning file holds: torch's version string (it ships the library)."""
    return f"torch {torch.__version__} hip {torch.version.hip}"


def hipblaslt_versions():
    """(compiled-against, running-on) hipBLASLt versions as major * 100000 + minor * 100 + patch; running-on is 0 before the first
    gemm_halves call.  The library is built with /opt/rocm's headers and binds whichever libhipblaslt.so the process loaded first
    — torch's bundled copy when torch is imported (same soname).  A different MAJOR means another struct layout: refuse."""
    c, r = c_int32(0), c_int32(0)
    _lib.bot_gemm_halves_library_version(ctypes.byref(c), ctypes.byref(r))
    return c.value, r.value


_LT_CHECKED = False


def gemm_halves(a, b, alpha, *, trans_a=False, trans_b=False, out=None, batch=1, strides=(0, 0, 0), m=None, n=None, k=None, beta=0.0,
                ldc=None):
    """C[m,n] = alpha[j] * op(a) op(b): a, b fp16 matrices (row-major views, unit column stride), C fp32; `alpha` a device
    vector of n floats (a 1-element tensor is expanded).  beta != 0 accumulates onto `out`; `ldc` overrides out's row pitch
    (batched products written side by side into the columns of one matrix: strides[2] = the column offset per batch).  m / n / k default to the operands' shapes; batch > 1 with element strides (a, b, c) for strided batches."""
    _dev(a, b, alpha)
    if a.dtype != torch.float16 or b.dtype != torch.float16 or a.stride(-1) != 1 or b.stride(-1) != 1:
        raise BotKernelError("gemm_halves: operands must be fp16 with unit column stride")
    if m is None:
        m, k = (a.shape[-1], a.shape[-2]) if trans_a else (a.shape[-2], a.shape[-1])
    if n is None:
        n = b.shape[-2] if trans_b else b.shape[-1]
    if out is None:
        out = torch.empty((m, n) if batch == 1 else (batch, m, n), dtype=torch.float32, device=a.device)
    if alpha.numel() == 1:
        alpha = alpha.reshape(1).expand(n).contiguous()
    if alpha.numel() != n or alpha.dtype != torch.float32:
        raise BotKernelError("gemm_halves: alpha must hold n floats")
    ws = _GEMM_WS.get(a.device)
    if ws is None:
        ws = _GEMM_WS[a.device] = torch.empty(64 << 20, dtype=torch.uint8, device=a.device)
    sa, sb, sc = strides
    if batch > 1 and sc == 0:
        sc = out.stride(0)
    ldc = _ld(out) if ldc is None else ldc
    lda, ldb = _ld(a), _ld(b)
    key = f"{int(trans_a)}{int(trans_b)} m{m} n{n} k{k} lda{lda} ldb{ldb} ldc{ldc} b{batch} s{sa},{sb},{sc}"
    index = _gemm_algos().get(key, -1) if beta == 0.0 else -1
    _check(_timed("gemm_halves", (m, n, k, batch), lambda: _lib.bot_gemm_halves_f32(
        int(trans_a), int(trans_b), m, n, k, alpha.data_ptr(), a.data_ptr(), lda, b.data_ptr(), ldb, out.data_ptr(),
        ldc, batch, sa, sb, sc, float(beta), ws.data_ptr(), ws.numel(), int(GEMM_TUNE), index, _stream())), "gemm_halves")
    global _LT_CHECKED
    if not _LT_CHECKED:
        _LT_CHECKED = True
        c, r = hipblaslt_versions()
        if r and c // 100000 != r // 100000:
            raise BotKernelError(f"libbot_gnn.so was built against hipBLASLt {c} but the process runs on hipBLASLt {r}: rebuild against "
                                 "the headers of the library that is loaded")
    if GEMM_SEEN is not None and key not in GEMM_SEEN:
        idx, ms = ctypes.c_int32(-1), ctypes.c_float(0.0)
        _lib.bot_gemm_halves_last_algo(ctypes.byref(idx), ctypes.byref(ms))
        GEMM_SEEN[key] = (idx.value, ms.value)
    return out


def stream_create(device, high_priority=False):
    """A torch.cuda.ExternalStream over a HIP stream of the library's own (include/bot_gnn.h bot_stream_create): NOT one of torch's pooled
    streams."""
    h = c_void_p()
    with torch.cuda.device(device):
        _check(_lib.bot_stream_create(int(bool(high_priority)), ctypes.byref(h)), "stream_create")
    return torch.cuda.ExternalStream(h.value, device=device)


def debug_abort_trace(path):
    """include/bot_gnn.h bot_debug_abort_trace: a SIGABRT / std::terminate in this process leaves the aborting thread's native frames in
    `path` (diagnostic; armed by tests/conftest.py, never by the product)."""
    _check(_lib.bot_debug_abort_trace(os.fsencode(path)), "debug_abort_trace")


class _BnBwdStats(ctypes.Structure):
    """include/bot_gnn.h bot_bn_bwd_stats_t"""
    _fields_ = [("x", c_void_p), ("ldx", c_int64), ("mean", c_void_p), ("invstd", c_void_p), ("weight", c_void_p), ("bias", c_void_p), ("relu", c_int32),
                ("p", c_float), ("seed", c_uint64), ("seed_offset", c_void_p), ("part", c_void_p), ("pmax", c_void_p)]


class BnBwdStats:
    """The reduce pass of a fused BatchNorm / ReLU / dropout epilogue's backward as a by-product of the NT product that writes the epilogue's
    incoming gradient (include/bot_gnn.h "v18"): x [m, F] the epilogue's input, the statistics / affine vectors, the dropout (p, seed) of its
    forward.  `gemm_halves3_nt(..., bn=this)` fills part / pmax [ceil(m / 256), 2, F]; `sums()` / `bound()` finish them."""

    def __init__(self, x, mean, invstd, weight, bias, relu, p, seed, want_max=True):
        _dev(x, mean, invstd, weight, bias)
        self.x = _mat(x, "x")
        self.mean, self.invstd, self.weight, self.bias = _f32(mean, "mean"), _f32(invstd, "invstd"), weight, bias
        self.relu, self.p, self.seed = bool(relu), float(p), int(seed)
        n, F = self.x.shape
        self.n, self.F, self.nblk = n, F, (n + 255) // 256
        self.part = torch.empty((self.nblk, 2, F), dtype=torch.float32, device=x.device)
        self.pmax = torch.empty((self.nblk, 2, F), dtype=torch.float32, device=x.device) if want_max else None

    def fits(self, m, n, k) -> bool:
        """The product [m, n] of piece width k can carry the by-product: the 256 x 32 form, n = F even, 8-byte aligned x rows."""
        return (m == self.n and n == self.F and self.F % 2 == 0 and self.x.stride(0) % 2 == 0 and self.x.data_ptr() % 8 == 0 and
                _lib.bot_gemm_halves3_nt_bn_rows(int(k)) == 256)

    def struct(self):
        return _BnBwdStats(self.x.data_ptr(), self.x.stride(0), self.mean.data_ptr(), self.invstd.data_ptr(), _ptr(self.weight), _ptr(self.bias),
                           int(self.relu), self.p, self.seed, _seed_off(self.p), self.part.data_ptr(), _ptr(self.pmax))

    def sums(self):
        sg = torch.empty(self.F, dtype=torch.float32, device=self.x.device)
        sgx = torch.empty(self.F, dtype=torch.float32, device=self.x.device)
        _check(_lib.bot_bn_act_bwd_reduce_partials_f32(self.part.data_ptr(), self.nblk, self.F, sg.data_ptr(), sgx.data_ptr(), _stream()),
               "bn_act_bwd_reduce_partials")
        return sg, sgx

    def finish(self, batch_stats, total_count, slots):
        """sums() and bound() in one launch (one rank: the local sums are the final ones) -> (sum_g, sum_gx)"""
        assert self.pmax is not None
        _dev(slots)
        sg = torch.empty(self.F, dtype=torch.float32, device=self.x.device)
        sgx = torch.empty(self.F, dtype=torch.float32, device=self.x.device)
        _check(_lib.bot_bn_bwd_partials_finish_f32(self.part.data_ptr(), self.pmax.data_ptr(), self.nblk, self.F, sg.data_ptr(), sgx.data_ptr(),
                                                   int(bool(batch_stats)), float(total_count), _ptr(self.weight), self.invstd.data_ptr(), slots.data_ptr(),
                                                   _stream()), "bn_bwd_partials_finish")
        return sg, sgx

    def bound(self, sum_g, sum_gx, total_count, slots):
        assert self.pmax is not None
        _dev(slots)
        _check(_lib.bot_bn_bwd_bound_partials_f32(self.F, self.pmax.data_ptr(), self.nblk, _ptr(sum_g), _ptr(sum_gx), float(total_count), _ptr(self.weight),
                                                  self.invstd.data_ptr(), slots.data_ptr(), _stream()), "bn_bwd_bound_partials")
        return slots


def gemm_halves3_nt(a, b, scale_a, scale_b, piece_a, piece_b, k, out=None, mode=0, a2_off=None, scale_a2=None, k_split=0, b_frag=False, n=None, bn=None):
    """out[m, n] = scale_a[1] scale_b[1] (a1 b1^T + a1 b2^T + a2 b1^T) from a LEFT operand buffer a [m, 3 piece_a] (or [m, 2 piece_a]
    without the duplicate piece: a2_off = piece_a) and a RIGHT operand buffer b [n, 3 piece_b] (bot_amd.gemm.Halves.buf / .scale), k =
    the common piece width used (bot_gemm_halves3_nt_f32).  scale_a2 / k_split: a's columns from k_split (a multiple of 32) on were
    written under a second scale (bot_gemm_halves3_nt2_f32).  b_frag: b is a fragment-major right operand (halves_split_frag) of `n` rows."""
    _dev(a, b, scale_a, scale_b, scale_a2)
    m = a.shape[0]
    if n is None:
        assert not b_frag
        n = b.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    # bn (BnBwdStats): out is the gradient arriving at that epilogue - its reduce pass leaves with the tiles (bot_gemm_halves3_nt3_f32)
    st = None
    if bn is not None:
        if not bn.fits(m, n, k):
            raise BotKernelError("gemm_halves3_nt: this product cannot carry the BatchNorm-backward by-product (BnBwdStats.fits)")
        st = bn.struct()
    _check(_timed("gemm_halves", (m, n, 3 * k, 1), lambda: _lib.bot_gemm_halves3_nt3_f32(
        m, n, k, scale_a.data_ptr(), _ptr(scale_a2), int(k_split), scale_b.data_ptr(), a.data_ptr(), _ld(a), 2 * piece_a if a2_off is None else a2_off,
        b.data_ptr(), _ld(b), piece_b, int(bool(b_frag)), out.data_ptr(), _ld(out), ctypes.addressof(st) if st is not None else None, int(mode), _stream())),
        "gemm_halves3_nt")
    return out


def halves_split_frag(x, scale, piece):
    """x [n, F] scaled by scale[0] as a RIGHT operand in fragment-major layout (include/bot_gnn.h bot_halves_split_frag_f16): a fp16 buffer
    [2 * ceil(n / 16) * piece / 32, 512] - one row per KB fragment block, h1 blocks first, h2 blocks behind them."""
    _dev(x, scale)
    x = _mat(x, "x")
    n, F = x.shape
    assert piece % 64 == 0 and piece >= F
    out = torch.empty((2 * ((n + 15) // 16) * (piece // 32), 512), dtype=torch.float16, device=x.device)
    _check(_lib.bot_halves_split_frag_f16(x.data_ptr(), x.stride(0), n, F, _ptr(scale), out.data_ptr(), piece, _stream()), "halves_split_frag")
    return out


def _table(rows, width):
    flat = [int(v) for r in rows for v in r]
    assert len(flat) == width * len(rows)
    return (ctypes.c_int64 * len(flat))(*flat)


def gemm_halves3_nt_grouped(a, b, scale_a, scale_b, a2_off, b2_off, out, groups, k_seg, mode=0, col_scale=None, col_shift=None, relu=False, absmax=None,
                            stats=None):
    """The grouped NT product (bot_gemm_halves3_nt_grouped_f32): for every group (b_row0, n_valid, a_col0, a_col1, k_steps, c_off)
        out.flat[r * ld + c_off + j] = scale_a[1] scale_b[1] * sum_{t < k_steps} sum_{i < 32} A3[r, (a_col0 if t < k_seg else a_col1) + 32 t + i] . B3[b_row0 + j, 32 t + i]
    for j < n_valid <= 256, over the rows of the left operand buffer a ([h1 at column c, 2^11 h2 at column c + a2_off]) and the right
    operand buffer b ([h1 | h2 at + b2_off]); `out` is a row-major fp32 matrix view whose storage the offsets c_off address (ld = its
    row pitch).  The per-head products of the aggregate-first GAT layer in one launch.  Optional epilogue: out = relu?(out * col_scale[c] +
    col_shift[c]) with c = c_off + j (the eval-mode BatchNorm / bias behind the layer) and max|out| into `absmax` slots."""
    _dev(a, b, out, scale_a, scale_b, col_scale, col_shift, absmax)
    assert a.dtype == torch.float16 and b.dtype == torch.float16 and a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    assert out.dtype == torch.float32 and out.stride(-1) == 1
    tab = _table(groups, 6)
    # (profile key: m, n, k, batch with 2 m n k batch = the fp16 MFMA flops of the valid output columns, three products each)
    # stats = (part [tiles, 2, F], minmax [tiles, 2, F], pivot [tiles, F]): column statistics of the stored values as a by-product (tiles =
    # ceil(m / 256) row blocks, F = the width the groups' output columns c_off + j index; all three WRITTEN: the pivot of a tile is its first
    # stored value - bot_gemm_halves3_nt_grouped2_f32; finished by bn_stats_halves_partials)
    sp = sm = sv = None
    sF = 0
    if stats is not None:
        sp, sm, sv = stats
        _dev(sp, sm, sv)
        sF = sv.shape[1]
        tiles = (a.shape[0] + 255) // 256
        assert sp.shape == (tiles, 2, sF) and sm.shape == (tiles, 2, sF) and sv.shape == (tiles, sF)
        assert sp.is_contiguous() and sm.is_contiguous() and sv.is_contiguous()
        assert sp.dtype == sm.dtype == sv.dtype == torch.float32
    _check(_timed("gemm_halves", (a.shape[0], sum(int(g[1]) * 96 * int(g[4]) for g in groups), 1, 1), lambda: _lib.bot_gemm_halves3_nt_grouped2_f32(
        a.shape[0], b.shape[0], scale_a.data_ptr(), scale_b.data_ptr(), a.data_ptr(), _ld(a), a2_off, b.data_ptr(), _ld(b), b2_off, out.data_ptr(),
        int(out.stride(-2)), len(groups), tab, int(k_seg), _ptr(col_scale), _ptr(col_shift), int(bool(relu)), _ptr(absmax), _ptr(sp), _ptr(sm), _ptr(sv),
        int(sF), int(mode), _stream())), "gemm_halves3_nt_grouped")
    return out


def gemm_halves3_tn_grouped(x, d, scale_x, scale_d, x2_off, d2_off, out, tiles, mode=0):
    """The grouped TN product (bot_gemm_halves3_tn_grouped_f32): for every tile (x_col0, k_valid, d_col0, p_valid, out_off, ldo, transposed)
        out.flat[out_off + k * ldo + p] = scale_x[1] scale_d[1] * sum_n X3[n, x_col0 + k] . D3[n, d_col0 + p],  k < k_valid <= 192, p < p_valid <= 192
    (transposed: out.flat[out_off + p * ldo + k])
    over the rows of two LEFT operand buffers (h1 at column c, 2^11 h2 at c + x2_off / d2_off); `out` is one contiguous fp32 buffer that
    the offsets address.  The weight gradients of the aggregate-first GAT layer (per head, and of its merged projection) in one launch."""
    _dev(x, d, out, scale_x, scale_d)
    assert x.dtype == torch.float16 and d.dtype == torch.float16 and x.stride(1) == 1 and d.stride(1) == 1 and x.shape[0] == d.shape[0]
    assert out.dtype == torch.float32 and out.is_contiguous()
    n = x.shape[0]
    ws = torch.empty(int(_lib.bot_gemm_halves3_tn_grouped_workspace_floats(n, len(tiles))), dtype=torch.float32, device=x.device)
    tab = _table(tiles, 7)
    _check(_timed("gemm_halves", (n, sum(3 * int(t[1]) * int(t[3]) for t in tiles), 1, 1), lambda: _lib.bot_gemm_halves3_tn_grouped_f32(
        n, scale_x.data_ptr(), scale_d.data_ptr(), x.data_ptr(), _ld(x), x2_off, d.data_ptr(), _ld(d), d2_off, out.data_ptr(), len(tiles), tab,
        ws.data_ptr(), int(mode), _stream())), "gemm_halves3_tn_grouped")
    return out


def gemm_halves3_tn(x, d, scale_x, scale_d, piece_x, piece_d, k, p, mode=0, x2_off=None, d2_off=None, scale_d2=None, p_split=0):
    """out[k, p] = scale_x[1] scale_d[1] (x1^T d1 + x1^T d2 + x2^T d1) from two LEFT operand buffers x [n, 3 piece_x], d [n, 3 piece_d]
    (include/bot_gnn.h bot_gemm_halves3_tn_f32): the weight gradient of a projection, reduced over the n rows.  scale_d2 / p_split: d's
    columns from p_split (a multiple of 4) on were written under a second scale (bot_gemm_halves3_tn2_f32)."""
    _dev(x, d, scale_x, scale_d, scale_d2)
    n = x.shape[0]
    out = torch.empty((k, p), dtype=torch.float32, device=x.device)
    ws = torch.empty(int(_lib.bot_gemm_halves3_tn_workspace_floats(n, piece_x, piece_d)), dtype=torch.float32, device=x.device)
    _check(_timed("gemm_halves", (k, p, 3 * n, 1), lambda: _lib.bot_gemm_halves3_tn2_f32(
        n, k, p, piece_x, piece_d, scale_x.data_ptr(), scale_d.data_ptr(), _ptr(scale_d2), int(p_split), x.data_ptr(), _ld(x),
        2 * piece_x if x2_off is None else x2_off, d.data_ptr(), _ld(d), 2 * piece_d if d2_off is None else d2_off,
        out.data_ptr(), _ld(out), ws.data_ptr(), int(mode), _stream())), "gemm_halves3_tn")
    return out


def _idx64(t, name):
    if t.dtype != torch.int64 or not t.is_contiguous() or t.dim() != 1:
        raise BotKernelError(f"{name} must be a contiguous 1-D int64 tensor (got {t.dtype}, stride {tuple(t.stride())})")


def _labels64(labels):
    """[N] or [N, 1..] int64 labels as a 2-D view with unit column stride (the kernels read column 0 at a row stride)."""
    if labels.dtype != torch.int64:
        raise BotKernelError(f"labels must be int64 (got {labels.dtype})")
    lab = labels.reshape(labels.shape[0], -1)
    if lab.shape[1] > 1 and lab.stride(1) != 1:
        lab = lab.contiguous()
    return lab


def label_split(train_idx, labels, mask, mask_rate, seed, use_labels, code, wn):
    """include/bot_gnn.h bot_label_split_f32: writes code / wn at the training nodes, returns count (1-element float tensor on the device)."""
    _dev(train_idx, labels, wn)
    _idx64(train_idx, "train_idx")
    count = torch.empty(1, dtype=torch.float32, device=wn.device)
    ws = torch.empty(128, dtype=torch.int32, device=wn.device)
    lab = _labels64(labels)
    if wn.dtype != torch.float32 or not wn.is_contiguous() or (code is not None and (code.dtype != torch.int32 or not code.is_contiguous())):
        raise BotKernelError("label_split: wn must be contiguous float32, code contiguous int32")
    m = None
    if mask is not None:            # one byte per training node, nonzero = an input-label node (a float 0/1 mask is converted, not reinterpreted)
        _dev(mask)
        m = mask.to(torch.bool).contiguous().view(torch.uint8)
        if m.numel() != train_idx.numel():
            raise BotKernelError("label_split: mask must have one entry per training node")
    _check(_lib.bot_label_split_f32(train_idx.data_ptr(), train_idx.numel(), lab.data_ptr(), lab.stride(0), _ptr(m), float(mask_rate), int(seed),
                                    _seed_off(1.0), int(bool(use_labels)), _ptr(code), wn.data_ptr(), count.data_ptr(), ws.data_ptr(), _stream()), "label_split")
    return count


def build_input(feat, code, n_classes, p, seed):
    """include/bot_gnn.h bot_build_input_f32: dropout_p([feat | onehot(code)]) as a new [N, F + C] tensor."""
    _dev(feat)
    _f32(feat, "feat")
    n, F = feat.shape
    out = torch.empty((n, F + n_classes), dtype=torch.float32, device=feat.device)
    _check(_lib.bot_build_input_f32(feat.data_ptr(), _ld(feat), n, F, n_classes, _ptr(code), float(p), int(seed), _seed_off(p), out.data_ptr(), _ld(out),
                                    _stream()), "build_input")
    return out


LOSS_KINDS = {"logit": 0, "loge": 1, "savage": 2}


def node_loss(x, labels, wn, count, kind, eps, want_grad=True):
    """include/bot_gnn.h bot_node_loss_f32: (y [n_pad] with n_pad = n rounded up to 64, zero beyond n and where wn == 0; dx [n, C] or None)."""
    _dev(x, labels, wn, count)
    _f32(x, "x")
    n, C = x.shape
    n_pad = (n + 63) // 64 * 64
    y = torch.empty(n_pad, dtype=torch.float32, device=x.device)
    dx = torch.empty((n, C), dtype=torch.float32, device=x.device) if want_grad else None
    lab = _labels64(labels)
    if wn.dtype != torch.float32 or not wn.is_contiguous() or count.dtype != torch.float32:
        raise BotKernelError("node_loss: wn / count must be contiguous float32")
    _check(_lib.bot_node_loss_f32(x.data_ptr(), _ld(x), n, C, lab.data_ptr(), lab.stride(0), wn.data_ptr(), count.data_ptr(), LOSS_KINDS[kind], float(eps),
                                  y.data_ptr(), n_pad, _ptr(dx), C, _stream()), "node_loss")
    return y, dx


def rmsprop_step(params, grads, square_avgs, lr, alpha, eps, weight_decay, lr_dev=None):
    """include/bot_gnn.h bot_rmsprop_step_f32 over lists of contiguous float32 tensors (48 per launch)."""
    for i in range(0, len(params), 48):
        ps, gs, sq = params[i:i + 48], grads[i:i + 48], square_avgs[i:i + 48]
        k = len(ps)
        for t in (*ps, *gs, *sq):
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise BotKernelError("rmsprop_step: contiguous float32 tensors only")
        P = (c_void_p * k)(*[t.data_ptr() for t in ps])
        G = (c_void_p * k)(*[t.data_ptr() for t in gs])
        S = (c_void_p * k)(*[t.data_ptr() for t in sq])
        Nn = (c_int64 * k)(*[t.numel() for t in ps])
        if lr_dev is not None and (lr_dev.dtype != torch.float32 or lr_dev.device != ps[0].device):
            raise BotKernelError("rmsprop_step: a tensor learning rate must be float32 on the parameters' device")
        _check(_lib.bot_rmsprop_step_f32(k, P, G, S, Nn, float(lr), _ptr(lr_dev), float(alpha), float(eps), float(weight_decay), _stream()), "rmsprop_step")


def skinny_gemm(a, b, *, b_is_kn, out, accumulate=False, batch=1, strides=(0, 0, 0), m=None, n=None, k=None, ldc=None):
    """out[m,n] (+)= a[m,k] @ (b if b_is_kn else b^T), k <= 256 (include/bot_gnn.h bot_skinny_gemm_f32): fp32 in and out, bf16x6 MFMA
    products inside.  a, b, out: fp32 row-major views with unit column stride; batch > 1: element strides (a, b, out)."""
    _dev(a, b, out)
    for t, name in ((a, "a"), (b, "b"), (out, "out")):
        _f32(t, name)
        if t.stride(-1) != 1 and t.shape[-1] != 1:
            raise BotKernelError(f"skinny_gemm: {name} must have unit column stride")
    if m is None:
        m, k = a.shape[-2], a.shape[-1]
    if n is None:
        n = b.shape[-1] if b_is_kn else b.shape[-2]
    sa, sb, sc = strides
    _check(_timed("skinny_gemm", (m, n, k, batch), lambda: _lib.bot_skinny_gemm_f32(
        a.data_ptr(), _ld(a), b.data_ptr(), _ld(b), int(b_is_kn), out.data_ptr(), _ld(out) if ldc is None else ldc, m, n, k,
        int(accumulate), batch, sa, sb, sc, _stream())), "skinny_gemm")
    return out


def tn_gemm(x, y, *, out=None, batch=1, strides=(0, 0, 0), n=None, kx=None, ky=None, transpose_out=False):
    """out[kx, ky] = x[n, kx]^T y[n, ky] (include/bot_gnn.h bot_tn_gemm_f32: exact fp32 MFMA, reduction over the n rows);
    transpose_out: out[ky, kx] (workgroup blocks are 256 columns of x by 192 of y: pass the wider operand as x).
    x, y: fp32 row-major views with unit column stride; batch > 1: element strides (x, y, out), out [batch, kx, ky]."""
    _dev(x, y)
    _f32(x, "x"), _f32(y, "y")
    if (x.stride(-1) != 1 and x.shape[-1] != 1) or (y.stride(-1) != 1 and y.shape[-1] != 1):
        raise BotKernelError("tn_gemm: operands must have unit column stride")
    if n is None:
        n, kx, ky = x.shape[-2], x.shape[-1], y.shape[-1]
    if out is None:
        shp = (ky, kx) if transpose_out else (kx, ky)
        out = torch.empty(shp if batch == 1 else (batch,) + shp, dtype=torch.float32, device=x.device)
    sx, sy, so = strides
    if batch > 1 and so == 0:
        so = out.stride(0)
    ws = torch.empty(int(_lib.bot_tn_gemm_workspace_floats(n, kx, ky, batch)), dtype=torch.float32, device=x.device)
    _check(_timed("tn_gemm", (n, kx, ky, batch), lambda: _lib.bot_tn_gemm_f32(
        x.data_ptr(), _ld(x), y.data_ptr(), _ld(y), n, kx, ky, out.data_ptr(), _ld(out), int(transpose_out), batch, sx, sy, so,
        ws.data_ptr(),
        _stream())), "tn_gemm")
    return out


def tn_narrow(x, y, out, transpose_out=False):
    """out[kx, ky] = x[n, kx]^T y[n, ky] for a handful of x columns (kx <= 32, ky <= 256), transpose_out: out[ky, kx] (include/bot_gnn.h
    bot_tn_narrow_f32: fp32 FMAs, deterministic).  x, y, out: fp32 row-major views with unit column stride."""
    _dev(x, y, out)
    _f32(x, "x"), _f32(y, "y"), _f32(out, "out")
    n, kx, ky = x.shape[0], x.shape[1], y.shape[1]
    assert y.shape[0] == n and out.shape == ((ky, kx) if transpose_out else (kx, ky))
    if x.stride(1) != 1 or y.stride(1) != 1 or (out.stride(1) != 1 and out.shape[1] != 1):
        raise BotKernelError("tn_narrow: operands must have unit column stride")
    ws = torch.empty(int(_lib.bot_tn_narrow_workspace_floats(n, kx, ky)), dtype=torch.float32, device=x.device)
    _check(_timed("tn_narrow", (n, kx, ky), lambda: _lib.bot_tn_narrow_f32(
        x.data_ptr(), _ld(x), y.data_ptr(), _ld(y), n, kx, ky, out.data_ptr(), _ld(out), int(transpose_out), ws.data_ptr(), _stream())), "tn_narrow")
    return out


def colsum(x):
    """Per-column sum of x [n,F] -> [F] (two-stage, finished in double)."""
    _dev(x)
    x = _mat(x, "x")
    n, F = x.shape
    out = torch.empty(F, dtype=torch.float32, device=x.device)
    _check(_lib.bot_colsum_f32(x.data_ptr(), x.stride(0), n, F, out.data_ptr(), _bn_ws(F, x.device).data_ptr(), _stream()), "colsum")
    return out


def _written(*ts):
    """The library wrote these tensors through raw pointers: move their autograd version counters, as an in-place torch op
    would, so that anything keyed on `_version` (the inference-path caches of bot_amd.nn.fused, saved-tensor checks) sees it."""
    ts = [t for t in ts if t is not None]
    if ts:
        torch._C._increment_version(ts)


def bn_stats(x, eps, momentum, running_mean=None, running_var=None, num_batches_tracked=None):
    """Training-mode statistics step of nn.BatchNorm1d over the rows of x [n,F] in one call: returns (mean, invstd) and updates
    the running statistics / step counter in place (include/bot_gnn.h bot_bn_stats_f32)."""
    _dev(x, running_mean, running_var)
    x = _mat(x, "x")
    n, F = x.shape
    mean = torch.empty(F, dtype=torch.float32, device=x.device)
    invstd = torch.empty(F, dtype=torch.float32, device=x.device)
    _check(_lib.bot_bn_stats_f32(x.data_ptr(), x.stride(0), n, F, float(eps), float(momentum), mean.data_ptr(), invstd.data_ptr(),
                                 _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked), _bn_ws(F, x.device).data_ptr(),
                                 _stream()), "bn_stats")
    _written(running_mean, running_var, num_batches_tracked)
    return mean, invstd


def bn_stats_halves(x, eps, momentum, running_mean, running_var, num_batches_tracked, weight, bias, p):
    """bn_stats plus hscale [2] = (s, 1/s) bounding the epilogue's output (include/bot_gnn.h bot_bn_stats_halves_f32)."""
    _dev(x, running_mean, running_var, weight, bias)
    x = _mat(x, "x")
    n, F = x.shape
    mean = torch.empty(F, dtype=torch.float32, device=x.device)
    invstd = torch.empty(F, dtype=torch.float32, device=x.device)
    hscale = torch.empty(2, dtype=torch.float32, device=x.device)
    _check(_lib.bot_bn_stats_halves_f32(x.data_ptr(), x.stride(0), n, F, float(eps), float(momentum), mean.data_ptr(), invstd.data_ptr(),
                                        _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked), _ptr(weight), _ptr(bias),
                                        float(p), hscale.data_ptr(), _bn_ws(F, x.device).data_ptr(), _stream()), "bn_stats_halves")
    _written(running_mean, running_var, num_batches_tracked)
    return mean, invstd, hscale


def bn_stats_halves_partials(part, minmax, pivot, n, eps, momentum, running_mean, running_var, num_batches_tracked, weight, bias, p):
    """bn_stats_halves from column partials a producer of x delivered (gemm_halves3_nt_grouped `stats`) instead of a pass over x
    (include/bot_gnn.h bot_bn_stats_halves_partials_f32): -> (mean, invstd, hscale)."""
    _dev(part, minmax, pivot, running_mean, running_var, weight, bias)
    nblk, _, F = part.shape
    assert pivot.shape == (nblk, F) and minmax.shape == part.shape and part.is_contiguous() and minmax.is_contiguous() and pivot.is_contiguous()
    mean = torch.empty(F, dtype=torch.float32, device=part.device)
    invstd = torch.empty(F, dtype=torch.float32, device=part.device)
    hscale = torch.empty(2, dtype=torch.float32, device=part.device)
    ws = torch.empty(F, dtype=torch.float32, device=part.device)
    _check(_lib.bot_bn_stats_halves_partials_f32(part.data_ptr(), minmax.data_ptr(), int(nblk), pivot.data_ptr(), int(n), int(F), float(eps), float(momentum),
                                                 mean.data_ptr(), invstd.data_ptr(), _ptr(running_mean), _ptr(running_var), _ptr(num_batches_tracked),
                                                 _ptr(weight), _ptr(bias), float(p), hscale.data_ptr(), ws.data_ptr(), _stream()), "bn_stats_halves_partials")
    _written(running_mean, running_var, num_batches_tracked)
    return mean, invstd, hscale


def bn_act_fwd(x, mean, invstd, weight, bias, relu, p, seed, halves=None, want_y=True):
    """y = dropout_p(relu?((x - mean) * invstd * weight + bias)); Philox mask from `seed`.
    halves = (hscale [2], piece[, pieces]): also returns y's fp16 halves [n, pieces * piece] scaled by hscale[0] -> (y, buf); pieces 3 (default):
    [h1 | h1 | 2^11 h2], 2: [h1 | 2^11 h2].
    want_y=False (with halves): only the halves are written; the returned y is None."""
    _dev(x, mean, invstd)
    x = _mat(x, "x")
    n, F = x.shape
    if halves is not None:
        hscale, piece = halves[:2]
        pieces = halves[2] if len(halves) > 2 else 3
        y = torch.empty((n, F), dtype=torch.float32, device=x.device) if want_y else None
        buf = torch.empty((n, pieces * piece), dtype=torch.float16, device=x.device)
        _check(_timed("bn_act_fwd", (F,), lambda: _lib.bot_bn_act_fwd_halves_f32(
            x.data_ptr(), x.stride(0), n, F, mean.data_ptr(), invstd.data_ptr(), _ptr(weight), _ptr(bias), int(relu), float(p),
            int(seed), _seed_off(p), _ptr(y), F, hscale.data_ptr(), buf.data_ptr(), buf.stride(0), piece, pieces, _stream())),
            "bn_act_fwd_halves")
        return y, buf
    y = torch.empty((n, F), dtype=torch.float32, device=x.device)
    _check(_timed("bn_act_fwd", (F,), lambda: _lib.bot_bn_act_fwd_f32(
        x.data_ptr(), x.stride(0), n, F, mean.data_ptr(), invstd.data_ptr(), _ptr(weight), _ptr(bias), int(relu), float(p),
        int(seed), _seed_off(p), y.data_ptr(), y.stride(0), _stream())), "bn_act_fwd")
    return y


def bn_act_bwd_reduce(dy, x, mean, invstd, weight, bias, relu, p, seed, want_max=False):
    """Column sums (sum_g, sum_gx) of the masked upstream gradient and of g * xhat.  want_max: also the column maxima of |g| and |xhat|,
    left in the returned workspace for bn_bwd_bound -> (sum_g, sum_gx, workspace)."""
    _dev(dy, x)
    dy, x = _mat(dy, "dy"), _mat(x, "x")
    n, F = x.shape
    sg = torch.empty(F, dtype=torch.float32, device=x.device)
    sgx = torch.empty(F, dtype=torch.float32, device=x.device)
    ws = _bn_ws(F, x.device)
    fn = _lib.bot_bn_act_bwd_reduce_max_f32 if want_max else _lib.bot_bn_act_bwd_reduce_f32
    _check(fn(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), n, F, mean.data_ptr(), invstd.data_ptr(), _ptr(weight), _ptr(bias), int(relu),
              float(p), int(seed), _seed_off(p), sg.data_ptr(), sgx.data_ptr(), ws.data_ptr(), _stream()), "bn_act_bwd_reduce")
    return (sg, sgx, ws) if want_max else (sg, sgx)


def bn_bwd_bound(ws, n, sum_g, sum_gx, total_count, weight, invstd, slots):
    """max |dx| of the BatchNorm backward bounded from the reduce pass's column maxima (`ws` of bn_act_bwd_reduce(want_max=True)) and the
    FINAL column sums (None: eval statistics), folded into the by-product `slots` (bot_bn_bwd_bound_f32)."""
    _dev(ws, invstd, slots)
    F = invstd.shape[0]
    _check(_lib.bot_bn_bwd_bound_f32(F, n, ws.data_ptr(), _ptr(sum_g), _ptr(sum_gx), float(total_count), _ptr(weight), invstd.data_ptr(), slots.data_ptr(),
                                     _stream()), "bn_bwd_bound")
    return slots


def bn_act_bwd_apply_halves(dy, x, mean, invstd, weight, bias, relu, p, seed, sum_g, sum_gx, total_count, hscale, hout, hD, hDP, out=None, h2_off=None):
    """bn_act_bwd_apply writing dx as the LEFT halves operand `hout` [n, 2 * (F / hD) * hDP] = [h1 | 2^11 h2] of hscale[0] * dx, the columns in
    blocks of hD every hDP (padding columns untouched); `out`: optionally dx in fp32 as well (bot_bn_act_bwd_apply_halves_f32)."""
    _dev(dy, x, hout, hscale)
    dy, x = _mat(dy, "dy"), _mat(x, "x")
    n, F = x.shape
    if h2_off is None:
        h2_off = (F // hD) * hDP
        assert hout.shape == (n, 2 * h2_off)
    else:       # `hout`: a column range of a wider operand (its first column = the block's first), the second half h2_off columns behind
        assert hout.shape[0] == n and hout.stride(0) >= h2_off + (F // hD) * hDP
    assert hout.dtype == torch.float16 and hout.stride(1) == 1
    assert out is None or (out.stride(1) == 1 and out.dtype == torch.float32)
    _check(_lib.bot_bn_act_bwd_apply_halves_f32(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), n, F, mean.data_ptr(), invstd.data_ptr(), _ptr(weight),
                                                _ptr(bias), int(relu), float(p), int(seed), _seed_off(p), _ptr(sum_g), _ptr(sum_gx), float(total_count),
                                                _ptr(out), out.stride(0) if out is not None else 0, hscale.data_ptr(), hout.data_ptr(), hout.stride(0), h2_off,
                                                hD, hDP, _stream()), "bn_act_bwd_apply_halves")
    return hout


def bn_act_bwd_apply(dy, x, mean, invstd, weight, bias, relu, p, seed, sum_g, sum_gx, total_count, out=None, absmax=None):
    """dx of the fused BatchNorm+ReLU+dropout; sum_g/sum_gx None = statistics were constants (eval mode).
    `out` may be a row-strided [n,F] view (e.g. a column slice of a wider gradient buffer).  `absmax`: `absmax_slots()` words that
    receive max|dx| as a by-product."""
    _dev(dy, x)
    dy, x = _mat(dy, "dy"), _mat(x, "x")
    n, F = x.shape
    dx = out if out is not None else torch.empty((n, F), dtype=torch.float32, device=x.device)
    assert dx.stride(1) == 1 and dx.dtype == torch.float32
    _check(_lib.bot_bn_act_bwd_apply_f32(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), n, F, mean.data_ptr(),
                                         invstd.data_ptr(), _ptr(weight), _ptr(bias), int(relu), float(p), int(seed), _seed_off(p),
                                         _ptr(sum_g), _ptr(sum_gx), float(total_count), dx.data_ptr(), dx.stride(0), _ptr(absmax),
                                         _stream()),
           "bn_act_bwd_apply")
    return dx


# ------------------------------------------------------------------------------------------------ fused edge MLP (ogbn-proteins)
def edge_mlp_supported(I, J, H):
    return I == 8 and J == 16 and 1 <= H <= 8


def random_keep(n, n_keep, seed, device):
    """uint8 [n] mask of a uniformly random subset of exactly n_keep elements (the edge-drop mask), a pure function of
    (n, n_keep, seed)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise BotKernelError("random_keep: the HIP kernels need a GPU device")
    keep = torch.empty(n, dtype=torch.uint8, device=device)
    ws = torch.empty(int(_lib.bot_random_keep_workspace_bytes()), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        _check(_lib.bot_random_keep_u8(n, n_keep, seed & 0xFFFFFFFFFFFFFFFF, keep.data_ptr(), ws.data_ptr(), _stream()), "random_keep")
    return keep


def edge_mlp_fwd(ef, W1, b1, W2):
    """ee[e,:] = W2 . relu(W1 . ef[e,:] + b1)   ef [E,8], W1 [16,8], b1 [16], W2 [H,16] -> [E,H]"""
    _dev(ef, W1, b1, W2)
    ef, W1, b1, W2 = (_f32(t, "edge_mlp operand").contiguous() for t in (ef, W1, b1, W2))
    E, H = ef.shape[0], W2.shape[0]
    out = torch.empty((E, H), dtype=torch.float32, device=ef.device)
    _check(_lib.bot_edge_mlp_fwd_f32(ef.data_ptr(), ef.shape[1], W1.data_ptr(), b1.data_ptr(), W1.shape[0], W2.data_ptr(), H, E,
                                     out.data_ptr(), _stream()), "edge_mlp_fwd")
    return out


def edge_mlp_bwd(ef, W1, b1, W2, dz):
    """Weight gradients (dW1 [16,8], db1 [16], dW2 [H,16]) of edge_mlp_fwd for upstream dz [E,H]."""
    _dev(ef, W1, b1, W2, dz)
    ef, W1, b1, W2, dz = (_f32(t, "edge_mlp operand").contiguous() for t in (ef, W1, b1, W2, dz))
    E, H = ef.shape[0], W2.shape[0]
    dW1, db1, dW2 = torch.empty_like(W1), torch.empty_like(b1), torch.empty_like(W2)
    ws = torch.empty(int(_lib.bot_edge_mlp_workspace_floats()), dtype=torch.float32, device=ef.device)
    _check(_lib.bot_edge_mlp_bwd_f32(ef.data_ptr(), ef.shape[1], W1.data_ptr(), b1.data_ptr(), W1.shape[0], W2.data_ptr(), H,
                                     dz.data_ptr(), E, dW1.data_ptr(), db1.data_ptr(), dW2.data_ptr(), ws.data_ptr(), _stream()),
           "edge_mlp_bwd")
    return dW1, db1, dW2


# ------------------------------------------------------------------------------------------------ merged projection weight
def merge_weight_fwd(W, Wres, attn_l, attn_r, H, D, P, with_fc, block=None):
    """[K, P] = [W^T (with_fc) | Wres^T | wl | wr | 0] (include/bot_gnn.h bot_merge_weight_fwd_f32); `block`: column width of the two
    copied blocks (default H*D)."""
    block = H * D if block is None else int(block)
    _dev(W, Wres, attn_l, attn_r)
    W = _f32(W, "W").contiguous()
    Wres = None if Wres is None else _f32(Wres, "Wres").contiguous()
    al = _f32(attn_l, "attn_l").contiguous()
    ar = None if attn_r is None else _f32(attn_r, "attn_r").contiguous()
    K = W.shape[1]
    out = torch.empty((K, P), dtype=torch.float32, device=W.device)
    _check(_lib.bot_merge_weight_fwd_f32(W.data_ptr(), _ptr(Wres), al.data_ptr(), _ptr(ar), H, D, K, P, int(with_fc), block, out.data_ptr(),
                                         _stream()), "merge_weight_fwd")
    return out


def merge_weight_bwd(W, attn_l, attn_r, H, D, P, with_fc, has_res, dm, block=None):
    _dev(W, attn_l, attn_r, dm)
    block = H * D if block is None else int(block)
    W, al, dm = W.contiguous(), attn_l.contiguous(), _f32(dm, "d_merged").contiguous()
    ar = None if attn_r is None else attn_r.contiguous()
    K = W.shape[1]
    dW = torch.empty_like(W)
    dWres = torch.empty_like(W) if has_res else None
    dal = torch.empty_like(al)
    dar = None if ar is None else torch.empty_like(ar)
    _check(_lib.bot_merge_weight_bwd_f32(W.data_ptr(), al.data_ptr(), _ptr(ar), H, D, K, P, int(with_fc), block, dm.data_ptr(), dW.data_ptr(),
                                         _ptr(dWres), dal.data_ptr(), _ptr(dar), _stream()), "merge_weight_bwd")
    return dW, dWres, dal, dar
