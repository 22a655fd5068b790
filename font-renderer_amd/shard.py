"""Glyph-batch sharding across the GPUs of one node.

Glyphs are independent (/root/reference/src/tools/render_glyph.zig:24-31 reads only
its own glyph), so rank r owns the contiguous glyph range [n*r/W, n*(r+1)/W) and
renders it into its own atlas rows: NO collective on the render path.  The optional
last step assembles the row bands in a preallocated atlas (RCCL over xGMI on GPUs; gloo in
the CPU tests) — equal bands, the last one trimmed, no concatenation: either onto ONE rank
(root=r: dist.gather into views of the root's atlas — with RCCL a group of point-to-point
transfers, each peer's bytes crossing its own xGMI link to the root once; SURVEY §5's
preferred form) or onto every rank (ONE all_gather_into_tensor: W times the ingress)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


def shard_ranges(n_items: int, world: int) -> List[Tuple[int, int]]:
    return [shard_range(n_items, r, world) for r in range(world)]


def band_rows(n_glyphs: int, cell: int, cols: int, world: int) -> List[int]:
    """pixel rows of every rank's atlas band: ceil(count_r / cols) cell rows"""
    return [((b - a + cols - 1) // cols) * cell for a, b in shard_ranges(n_glyphs, world)]


def gather_buffer(n_glyphs: int, cell: int, cols: int, world: int, device, dtype=torch.uint8) -> torch.Tensor:
    """the preallocated destination of gather_atlas: `world` equal bands of the largest band's height"""
    pad = max(band_rows(n_glyphs, cell, cols, world))
    return torch.empty((world * pad, cols * cell), dtype=dtype, device=device)


def gather_bands(band: torch.Tensor, out: torch.Tensor, root: Optional[int] = None, group=None) -> Optional[torch.Tensor]:
    """Equal row bands -> one buffer, for ANY workload (cells, pages, renderGlyph-sized images): `out` is
    (world * rows, width), `band` is this rank's rows x width — ideally the view out[rank * rows:(rank + 1) * rows] it
    rendered into (no staging copy).  root=None: all-gather, every rank gets `out`.  root=r: only rank r's `out` is
    filled (the others may pass any tensor of the right shape, or their own full buffer) -> `out` on the root, None elsewhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rows = band.shape[0]
    assert out.shape[0] == world * rows and out.shape[1:] == band.shape[1:] and out.is_contiguous()
    if root is None:
        src = band.contiguous()
        if dist.get_backend(group) != "nccl":
            src = src.clone()                     # (only RCCL documents the in-place form: input = own slot of the output)
        dist.all_gather_into_tensor(out, src, group=group)
        return out
    if rank == root:
        slots = [out[r * rows:(r + 1) * rows] for r in range(world)]
        mine = slots[rank]
        if band.data_ptr() != mine.data_ptr():
            mine.copy_(band)
        dist.gather(mine if dist.get_backend(group) == "nccl" else mine.clone(), slots, dst=root, group=group)
        return out
    dist.gather(band.contiguous(), None, dst=root, group=group)
    return None


def gather_atlas(local_rows: torch.Tensor, n_glyphs: int, cell: int, cols: int, group=None,
                 out: Optional[torch.Tensor] = None, compact: bool = True) -> torch.Tensor:
    """local_rows: this rank's band — either exactly its own rows_r x (cols*cell), or already a view of the
    padded band (pad x cols*cell).  One all_gather_into_tensor into `out` (gather_buffer(...); allocated here
    when None).  Bands are equal-sized slots of the largest band's height; ranks whose band is shorter leave
    the tail of their slot unspecified.  Returns the atlas: `out` itself when every band fills its slot (the
    usual case: n_glyphs a multiple of world * cols), else — with compact=True — the bands moved up against
    each other inside `out` (in place, rank by rank) and the view of the rows in use."""
    world = dist.get_world_size(group)
    rows = band_rows(n_glyphs, cell, cols, world)
    pad = max(rows)
    if out is None:
        out = gather_buffer(n_glyphs, cell, cols, world, local_rows.device, local_rows.dtype)
    assert out.shape == (world * pad, local_rows.shape[1]) and out.is_contiguous()
    src = local_rows
    if src.shape[0] != pad:                       # a short band: stage it in this rank's own slot of `out`
        rank = dist.get_rank(group)
        slot = out[rank * pad:(rank + 1) * pad]
        slot[:src.shape[0]].copy_(src)
        src = slot
    if dist.get_backend(group) != "nccl":
        src = src.clone()                         # (only RCCL documents the in-place form: input = own slot of the output)
    dist.all_gather_into_tensor(out, src.contiguous(), group=group)
    if not compact or all(r == pad for r in rows):
        return out
    dst = rows[0]
    for r in range(1, world):                     # move band r up against band r - 1 (overlap-safe: row chunks, top down)
        if dst != r * pad and rows[r]:
            band = out[r * pad:r * pad + rows[r]]
            if dst + rows[r] <= r * pad:
                out[dst:dst + rows[r]].copy_(band)
            else:
                out[dst:dst + rows[r]].copy_(band.clone())
        dst += rows[r]
    return out[:dst]
