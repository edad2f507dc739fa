"""Host side of the L2-blocked SpMM (bot_amd/csrc/blocked.hip): builds, once per graph direction and row width, the
(tile, column block, wave)-sorted edge structure and the restricted plan for hub rows.  Integer work with torch on the
device the graph lives on."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass

import torch

ENABLED = True
MIN_MEAN_DEGREE = 96       # below this a (row, block) visit holds < 1 edge: nothing to reuse
L2_BLOCK_BYTES = 2 << 20   # source rows per column block * row bytes (half of the 4 MiB L2 of an XCD; 1 / 2 / 4 / 8 MiB: 7.80 / 6.54 / 6.44 / 7.95 ms on S-reddit)
# The fused backward (spmm_dot) has NO blocked form.  Round 2 built one with two [T, H*D] tiles in LDS (T = 32: 34.4 ... 22.7 ms with
# 1 ... 16 MiB column blocks against 21.9 ms for the row kernel at S-proteins, profiles/r02_blocked_dot_proteins.txt); round 4 built five
# register-resident forms (T = 64, accumulators and the tile's own rows in VGPRs, 8 or 16 waves, with and without a software pipeline):
# 22.2 - 26.5 ms at 2 - 4 MiB blocks against 22.0 ms (profiles/r04_blocked_dot_proteins.txt).  None won; both were removed.
TILE_ROWS = 128            # destination rows per workgroup (32 / 64 / 128; 256 when rows are gathered by lane groups)
TILE_LDS_BYTES = 128 * 1024  # LDS per workgroup: one 16-wave workgroup per CU
WAVES = 16                 # wavefronts per workgroup (bot_amd/csrc/blocked.hip kBWaves)
ROUND_WORKGROUPS = 256     # one resident 16-wave workgroup per CU
MIN_ROW_FLOATS = 16        # rows of <= 32 (16) vector lanes are gathered 2 (4) edges per instruction by lane groups
HUB_FACTOR = 8             # rows longer than HUB_FACTOR x mean stay on the row-per-group kernel


@dataclass
class BlockedPlan:
    tile_rows: torch.Tensor
    ptr: torch.Tensor
    b_src: torch.Tensor
    b_lrow: torch.Tensor
    b_pos: torch.Tensor
    n_tiles: int
    nblk: int
    T: int
    epi: int   # edges per gather instruction (lane groups of 64 / epi lanes)
    round_tiles: int
    heavy: object  # Direction restricted to the hub rows, or None
    block_rows: int = 0


def _heavy_direction(d, heavy_rows):
    """A copy of direction `d` whose plan covers only `heavy_rows` (every one is split into chunk-sized items)."""
    dev = d.indptr.device
    beg = d.indptr[heavy_rows.long()].long()
    end = d.indptr[heavy_rows.long() + 1].long()
    nchunk = (end - beg + d.chunk - 1) // d.chunk
    long_ptr = torch.zeros(heavy_rows.numel() + 1, dtype=torch.int64, device=dev)
    long_ptr[1:] = torch.cumsum(nchunk, 0)
    n_slots = int(long_ptr[-1])
    owner = torch.repeat_interleave(torch.arange(heavy_rows.numel(), device=dev), nchunk)
    j = torch.arange(n_slots, device=dev) - long_ptr[owner]
    ib = beg[owner] + j * d.chunk
    ie = torch.minimum(ib + d.chunk, end[owner])
    items = torch.stack([heavy_rows.long()[owner], ib, ie, torch.arange(n_slots, device=dev)], 1).to(torch.int32).contiguous()
    return dataclasses.replace(d, items=items, long_rows=heavy_rows.to(torch.int32).contiguous(),
                               long_ptr=long_ptr.to(torch.int32).contiguous(), n_items=n_slots,
                               n_long=int(heavy_rows.numel()), n_slots=n_slots, blocked={})


def layout(H: int, D: int):
    """(vec, epi, LDS row pitch in floats) the kernel uses for rows of H x D floats (contiguous, 16-byte aligned base)."""
    vec = 4 if D % 4 == 0 else (2 if D % 2 == 0 else 1)
    lanes = (H * D + vec - 1) // vec
    epi = 4 if lanes <= 16 else (2 if lanes <= 32 else 1)
    group = 64 // epi
    return vec, epi, (lanes + group - 1) // group * group * vec


def build(d, n_src: int, H: int, D: int) -> BlockedPlan:
    F = H * D
    dev = d.indptr.device
    deg = (d.indptr[1:] - d.indptr[:-1]).long()
    mean = max(1.0, d.nnz / max(1, d.n_rows))
    hub_thr = max(int(HUB_FACTOR * mean), d.chunk)
    vec, epi, Fp = layout(H, D)
    T = 256 if epi > 1 else TILE_ROWS
    while T > 32 and Fp * 4 * T > TILE_LDS_BYTES:
        T //= 2
    cb = max(64, L2_BLOCK_BYTES // (F * 4))
    cb = 1 << (cb.bit_length() - 1)
    nblk = (n_src + cb - 1) // cb
    regular = deg <= hub_thr
    reg_rows = torch.nonzero(regular).squeeze(1)
    order = torch.argsort(deg[reg_rows], descending=True, stable=True)
    reg_rows = reg_rows[order]                                    # heaviest first
    n_reg = int(reg_rows.numel())
    # Tiles of equal WORK, not equal height: sweepers that carry the same number of edges cross the column blocks at the
    # same pace (measured: with equal-height tiles the round of the heaviest rows ran at 34 % L2 hits, the uniform rounds at
    # 80 %).  A tile takes rows until it holds ~target edges or T rows.
    rdeg = deg[reg_rows]
    target = max(int(rdeg.sum()) // max(1, (n_reg + T - 1) // T), 1)      # edges per tile if all tiles were full-height
    heavy_part = rdeg * T > target                                          # rows whose tile fills up by edges first
    n_hp = int(heavy_part.sum())                                            # (a prefix: rows are sorted by degree)
    tile_hp = torch.div(torch.cumsum(rdeg[:n_hp], 0) - rdeg[:n_hp], target, rounding_mode="floor")
    if n_hp:                                                                # make ids dense and cap the height at T
        _, tile_hp = torch.unique_consecutive(tile_hp, return_inverse=True)
        first = torch.ones(n_hp, dtype=torch.bool, device=dev)
        first[1:] = tile_hp[1:] != tile_hp[:-1]
        start = torch.nonzero(first).squeeze(1)
        within = torch.arange(n_hp, device=dev) - start[tile_hp]
        within_hp = within                                                  # heights stay below T: every row here has deg > target / T
        n_t_hp = int(tile_hp.max()) + 1
    else:
        within_hp = torch.zeros(0, dtype=torch.int64, device=dev)
        n_t_hp = 0
    rest = torch.arange(n_reg - n_hp, device=dev)
    tile_of = torch.cat([tile_hp, n_t_hp + rest // T])
    slot_in_tile = torch.cat([within_hp, rest % T])
    assert int(slot_in_tile.max()) < T if n_reg else True
    n_tiles = int(tile_of.max()) + 1 if n_reg else 0
    tile_rows = torch.full((n_tiles * T,), -1, dtype=torch.int32, device=dev)
    tile_rows[tile_of * T + slot_in_tile] = reg_rows.to(torch.int32)
    slot_of_row = torch.full((d.n_rows,), -1, dtype=torch.int64, device=dev)
    slot_of_row[reg_rows] = tile_of * T + slot_in_tile
    row_of_pos = torch.repeat_interleave(torch.arange(d.n_rows, device=dev), deg)
    slot = slot_of_row[row_of_pos]
    pos = torch.nonzero(slot >= 0).squeeze(1)                     # positions of the blocked edges, ascending
    slot = slot[pos]
    src = d.indices[pos].long()
    tile, lrow = slot // T, slot % T
    stream = tile * WAVES + (lrow % WAVES)                        # (tile, wave): one contiguous edge stream per wave
    key = (stream * nblk + src // cb) * T + lrow                  # inside a stream: by column block, then row
    key, perm = torch.sort(key, stable=True)                      # ties keep position order = ascending edge id
    b_src, b_lrow, b_pos = src[perm], lrow[perm], pos[perm]
    counts = torch.bincount(stream, minlength=n_tiles * WAVES)
    if epi > 1 and key.numel():
        # lane groups: every (stream, block, row) run is padded to a multiple of `epi` slots (source -1), so the slots of
        # one gather instruction always share their destination row
        run_key, run_len = torch.unique_consecutive(key, return_counts=True)
        pad_len = (run_len + epi - 1) // epi * epi
        run_start = torch.cumsum(run_len, 0) - run_len
        pad_start = torch.cumsum(pad_len, 0) - pad_len
        run_of = torch.repeat_interleave(torch.arange(run_len.numel(), device=dev), run_len)
        dest = pad_start[run_of] + (torch.arange(key.numel(), device=dev) - run_start[run_of])
        total = int(pad_len.sum())
        run_lrow = run_key % T
        p_src = torch.full((total,), -1, dtype=torch.int64, device=dev)
        p_pos = torch.full((total,), -1, dtype=torch.int64, device=dev)
        p_src[dest], p_pos[dest] = b_src, b_pos
        b_src, b_pos = p_src, p_pos
        b_lrow = torch.repeat_interleave(run_lrow, pad_len)
        counts = torch.zeros(n_tiles * WAVES, dtype=torch.int64, device=dev)
        counts.index_add_(0, torch.div(run_key, nblk * T, rounding_mode="floor"), pad_len)
    ptr = torch.zeros(n_tiles * WAVES + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(counts, 0)
    assert int(ptr[-1]) < 2 ** 31, "blocked edge stream exceeds int32 offsets"
    heavy_rows = torch.nonzero(~regular).squeeze(1)
    heavy = _heavy_direction(d, heavy_rows) if heavy_rows.numel() else None
    return BlockedPlan(tile_rows, ptr.to(torch.int32).contiguous(), b_src.to(torch.int32).contiguous(),
                       b_lrow.to(torch.uint8).contiguous(), b_pos.to(torch.int32).contiguous(), n_tiles, nblk, T, epi,
                       ROUND_WORKGROUPS, heavy, cb)


def plan_for(d, n_src: int, H: int, D: int):
    """BlockedPlan for direction `d` and row width H*D, or None when the row-per-group kernel is the right one."""
    F = H * D
    vec = layout(H, D)[0]
    if not ENABLED or F > 256 * vec or F < MIN_ROW_FLOATS or d.n_rows == 0 or d.nnz < MIN_MEAN_DEGREE * d.n_rows or not d.indptr.is_cuda:
        return None
    cache = d.blocked
    if (H, D) not in cache:
        cache[(H, D)] = build(d, n_src, H, D)
    return cache[(H, D)]
