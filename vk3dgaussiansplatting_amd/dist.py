"""Multi-GPU frames: shard by screen-tile rows, one process per GPU, one gather of RGBA8 strips.

The reference is single-GPU (no collective anywhere).  Tiles are independent
once each has its sorted list (RenderGaussians.comp:74-77 reads only its own range), so the frame
shards naturally: the gaussian arrays are replicated, rank r emits/sorts/renders only the tile rows
of its band (with GLOBAL tile ids, so keys, per-tile order and pixels equal the 1-GPU result) and
the only exchange is one gather of equal-size image strips to rank 0 per frame -- over RCCL/xGMI
when the process group's backend is "nccl", over gloo in the CPU tests.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import numpy as np


def tile_row_partition(tiles_y: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous bands of ceil(tiles_y / world_size) rows; trailing ranks may get fewer (or zero)
    rows.  Strips are padded to the same height for the collective."""
    per = (tiles_y + world_size - 1) // world_size
    out = []
    for r in range(world_size):
        b = min(r * per, tiles_y)
        e = min(b + per, tiles_y)
        out.append((b, e))
    return out


def balanced_row_partition(row_weights, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous bands whose WEIGHTS (e.g. sort elements per tile row of the last frames) are as equal as whole rows
    allow: band edge r sits at the row boundary whose weight prefix is nearest to r / world_size of the total.  Every
    rank gets at least one row while there are rows to give (a rank with no row would still pay the frame's fixed
    cost); with all-zero weights this is tile_row_partition's answer in spirit (equal row counts)."""
    wts = np.asarray(row_weights, dtype=np.float64).reshape(-1)
    tiles_y = int(wts.shape[0])
    if tiles_y == 0:
        return [(0, 0)] * world_size
    if not np.isfinite(wts).all() or wts.sum() <= 0.0:
        wts = np.ones(tiles_y)
    prefix = np.concatenate([[0.0], np.cumsum(wts)])          # prefix[k] = weight of rows [0, k)
    edges = [0]
    for r in range(1, world_size):
        target = prefix[-1] * r / world_size
        k = int(np.searchsorted(prefix, target))               # first boundary with prefix >= target
        if k > 0 and target - prefix[k - 1] <= prefix[min(k, tiles_y)] - target:
            k -= 1
        lo = min(edges[-1] + 1, tiles_y)                       # at least one row for the previous rank ...
        hi = max(lo, tiles_y - (world_size - r))               # ... and one left for each rank still to come
        edges.append(int(min(max(k, lo), hi)))
    edges.append(tiles_y)
    return [(edges[r], edges[r + 1]) for r in range(world_size)]


def strip_rows(tiles_y: int, world_size: int, tile: int = 16) -> int:
    """Pixel rows of one (padded) strip."""
    return ((tiles_y + world_size - 1) // world_size) * tile


def interleaved_rows(tiles_y: int, rank: int, world_size: int) -> List[int]:
    """Tile rows of `rank` when rows are dealt round-robin (gs_set_tile_rows_interleaved(rank, world_size))."""
    return list(range(rank, tiles_y, world_size))


class ShardedFrame:
    """Band render + gather.  n_strips strips (round-robin) so that the gather of frame f can run beside the
    compute of the following frames: `gather_async(k)` starts the collective on strip k, `wait(k)` orders the
    current stream (or the host, for gloo) behind it before strip k is rendered into again.

    `render_band(row_begin, row_end, strip)` (see `frame`) must write the band's pixel rows into the
    first rows of `strip` ([strip_rows, W, 4] uint8, torch tensor on the rank's device or CPU); it is
    the only thing that differs between the GPU path (Renderer.drawDevice into the strip's storage) and
    the CPU test (an injected checker)."""

    def __init__(self, width: int, height: int, rank: int, world_size: int, device="cpu", group=None,
                 host_gather: bool = False, n_strips: int = 2, interleaved: bool = False):
        import torch
        self.torch = torch
        self.width, self.height = width, height
        self.rank, self.world = rank, world_size
        self.tiles_y = (height + 15) // 16
        # interleaved: rank r owns tile rows r, r + world, ...; its strip holds them packed (owned row k at strip
        # rows [16 k, 16 k + 16)), which is what a context set up with gs_set_tile_rows_interleaved(r, world, 1) writes
        self.interleaved = interleaved
        self.bands = tile_row_partition(self.tiles_y, world_size)
        self.rows = strip_rows(self.tiles_y, world_size)
        self.group = group
        self.device = device
        self.n_strips = n_strips
        self.strips = [torch.zeros((self.rows, width, 4), dtype=torch.uint8, device=device) for _ in range(n_strips)]
        # host_gather: rehearsal of the N > 1 path on one GPU (gloo has no device gather)
        self.host_gather = host_gather
        gdev = "cpu" if host_gather else device
        self.gathered = [([torch.zeros((self.rows, width, 4), dtype=torch.uint8, device=gdev) for _ in range(world_size)]
                          if rank == 0 and world_size > 1 else None) for _ in range(n_strips)]
        self._pending = [None] * n_strips

    @property
    def strip(self):
        return self.strips[0]

    @property
    def band(self) -> Tuple[int, int]:
        return self.bands[self.rank]

    def gather_async(self, k: int = 0):
        """All ranks call, in the same order; starts the gather of strip k to rank 0."""
        if self.world == 1:
            return
        import torch.distributed as dist
        src = self.strips[k].cpu() if self.host_gather else self.strips[k]
        work = dist.gather(src, self.gathered[k], dst=0, group=self.group, async_op=True)
        self._pending[k] = (work, src)

    def wait(self, k: int = 0):
        """Order everything that follows (rendering into strip k again, reading gathered[k]) behind the
        pending gather of strip k."""
        if self._pending[k] is not None:
            self._pending[k][0].wait()
            self._pending[k] = None

    def wait_all(self):
        for k in range(self.n_strips):
            self.wait(k)

    def gather(self, k: int = 0):
        """Synchronous form: rank 0 gets the list of strips."""
        if self.world == 1:
            return [self.strips[k]]
        self.gather_async(k)
        self.wait(k)
        return self.gathered[k]

    def assemble(self, strips) -> "np.ndarray":
        """Rank 0: strips -> [H, W, 4] image (crops the padding of the last band)."""
        img = self.torch.zeros((self.height, self.width, 4), dtype=self.torch.uint8, device=strips[0].device)
        if self.interleaved:
            for r in range(self.world):
                for k, row in enumerate(interleaved_rows(self.tiles_y, r, self.world)):
                    y0, y1 = row * 16, min(row * 16 + 16, self.height)
                    img[y0:y1] = strips[r][k * 16: k * 16 + (y1 - y0)]
            return img
        for r, (b, e) in enumerate(self.bands):
            y0, y1 = b * 16, min(e * 16, self.height)
            if y1 > y0:
                img[y0:y1] = strips[r][: y1 - y0]
        return img

    def frame(self, render_band: Callable, k: int = 0) -> "np.ndarray | None":
        """contiguous: render_band(row_begin, row_end, strip); interleaved: render_band(rows, None, strip) with the
        list of owned tile rows, to be written packed."""
        b, e = (interleaved_rows(self.tiles_y, self.rank, self.world), None) if self.interleaved else self.band
        self.wait(k)
        render_band(b, e, self.strips[k])
        strips = self.gather(k)
        if self.rank == 0:
            return self.assemble(strips)
        return None
