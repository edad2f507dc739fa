"""Message / reduce builders — the slice of `dgl.function` the reference imports
(src/no-sampling/models.py:8, :374-381, :523-547; src/ogbn-proteins/gat.py:58).

They only describe the operation; `Graph.update_all` / `Graph.apply_edges` map each
(message, reduce) pair onto one HIP kernel family in `bot_amd.ops`.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Message:
    kind: str   # "copy_u" | "copy_e" | "u_add_v" | "u_mul_e"
    a: str      # first operand field
    b: str | None
    out: str


@dataclass(frozen=True)
class Reduce:
    kind: str   # "sum"
    msg: str
    out: str


def copy_u(u, out):
    return Message("copy_u", u, None, out)


def copy_src(src, out):
    """Old name of copy_u used at models.py:374,381."""
    return Message("copy_u", src, None, out)


def copy_e(e, out):
    return Message("copy_e", e, None, out)


def u_add_v(u, v, out):
    return Message("u_add_v", u, v, out)


def u_mul_e(u, e, out):
    return Message("u_mul_e", u, e, out)


def sum(msg, out):  # noqa: A001 - mirrors dgl.function.sum
    return Reduce("sum", msg, out)
