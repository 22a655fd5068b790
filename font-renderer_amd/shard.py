"""Glyph-batch sharding across the GPUs of one node.

Glyphs are independent (/root/reference/src/tools/render_glyph.zig:24-31 reads only
its own glyph), so rank r owns the contiguous glyph range [n*r/W, n*(r+1)/W) and
renders it into its own atlas rows: NO collective on the render path.  The optional
last step gathers the row bands onto every rank with one all_gather (RCCL over xGMI
on GPUs; gloo in the CPU tests)."""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


def shard_ranges(n_items: int, world: int) -> List[Tuple[int, int]]:
    return [shard_range(n_items, r, world) for r in range(world)]


def gather_atlas(local_rows: torch.Tensor, n_glyphs: int, cell: int, cols: int, group=None) -> torch.Tensor:
    """local_rows: this rank's (rows_r*cell, cols*cell) band, rows_r = ceil(count_r/cols).
    Returns the concatenation of every rank's band (ranks may own different row counts:
    bands are padded to the largest, gathered in ONE collective, then trimmed)."""
    world = dist.get_world_size(group)
    counts = [b - a for a, b in shard_ranges(n_glyphs, world)]
    rows = [((c + cols - 1) // cols) * cell for c in counts]
    pad = max(rows)
    buf = local_rows
    if buf.shape[0] < pad:
        buf = torch.cat([buf, buf.new_zeros((pad - buf.shape[0], buf.shape[1]))], 0)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf.contiguous(), group=group)
    return torch.cat([o[:r] for o, r in zip(out, rows)], 0)
