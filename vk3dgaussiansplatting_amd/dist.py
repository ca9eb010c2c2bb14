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
    if not np.isfinite(wts).all() or (wts < 0.0).any() or wts.sum() <= 0.0:   # a negative weight: prefix not monotone
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


class RowBalancer:
    """Element-balanced contiguous bands that follow the scene: every K frames each rank contributes the sort-element
    counts of ITS tile rows (it has them: the frame's tile ranges) and, when it measures one, the GPU time of its share;
    one small all-reduce later every rank holds the same two vectors and derives the same new band edges.

    Model of a share's time: T_r = F + sum over its rows of weight(row).  F is what every share pays whatever it holds
    (the launches of a frame, the passes' latency floor: at R = 8 more than half of a share), estimated as the intercept of
    the least-squares line T = F + b E through the (elements, time) pairs of the last epochs; the rest of a rank's time is
    spread over its rows in proportion to their elements:

        weight(row) = elements(row) * (T_r - F) / E_r        (elements(row) when no times are known)

    i.e. a row costs what its elements cost on the rank that had it -- which folds what does not scale with E (long lists
    that saturate early, near-empty rows) into the next partition.  New edges sit at equal weight prefixes.  Hysteresis: the
    bands move only when the model says the slowest rank gets at least `min_gain` faster, so that noise does not make the
    bands (and the hipGraphs captured for them) flap.  Pure host arithmetic on identical inputs: every rank computes
    identical edges without a further exchange.  (csrc/gs_dist.cpp: gs_dist_rebalance is the same rule behind the C-ABI,
    statement for statement.)"""

    HISTORY_EPOCHS = 4

    def __init__(self, tiles_y: int, world_size: int, min_gain: float = 0.03):
        self.tiles_y, self.world, self.min_gain = int(tiles_y), int(world_size), float(min_gain)
        self.bands = tile_row_partition(self.tiles_y, self.world)
        self.history: List[Tuple[float, float]] = []      # (elements, ms) of every rank over the last epochs
        self.fixed_ms = 0.0

    @staticmethod
    def row_elements(ranges: np.ndarray, tiles_x: int, tiles_y: int) -> np.ndarray:
        """[tiles, 2] {start, end} per tile (GS_BUF_RANGES; zero for tiles of other ranks) -> elements per tile row."""
        r = np.asarray(ranges, dtype=np.int64).reshape(tiles_y, tiles_x, 2)
        return (r[:, :, 1] - r[:, :, 0]).sum(axis=1)

    @staticmethod
    def fixed_cost(history, newest_ms) -> float:
        """Intercept of the least-squares line through (elements, ms), kept inside [0, 0.8 min(newest times)]."""
        n = len(history)
        m_e = sum(h[0] for h in history) / n
        m_t = sum(h[1] for h in history) / n
        var = sum((h[0] - m_e) * (h[0] - m_e) for h in history)
        cov = sum((h[0] - m_e) * (h[1] - m_t) for h in history)
        f = 0.0
        if var > 0.0 and cov > 0.0:
            f = m_t - (cov / var) * m_e
        if not f > 0.0:
            f = 0.0
        return min(f, 0.8 * min(newest_ms))

    def update(self, row_elements, rank_ms=None) -> bool:
        """row_elements: the summed vector (every row from the rank that owns it); rank_ms: per-rank share times or None.
        Returns True when the bands changed."""
        elems = np.asarray(row_elements, dtype=np.float64).reshape(self.tiles_y)
        weights = elems.copy()
        fixed = 0.0
        if rank_ms is not None and all(t > 0.0 for t in rank_ms):
            rank_ms = [float(t) for t in rank_ms]
            totals = [float(elems[b:e].sum()) for b, e in self.bands]
            self.history = (self.history + list(zip(totals, rank_ms)))[-self.HISTORY_EPOCHS * self.world:]
            fixed = self.fixed_cost(self.history, rank_ms)
            for r, (b, e) in enumerate(self.bands):
                if e > b:
                    # a band without elements still took its time: spread it over the rows
                    weights[b:e] = elems[b:e] * ((rank_ms[r] - fixed) / totals[r]) if totals[r] > 0 else (rank_ms[r] - fixed) / (e - b)
        self.fixed_ms = fixed
        cost = lambda bands: fixed + max(weights[b:e].sum() for b, e in bands)
        new = balanced_row_partition(weights, self.world)
        if new != self.bands and cost(new) <= (1.0 - self.min_gain) * cost(self.bands):
            self.bands = new
            return True
        return False


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
                 host_gather: bool = False, n_strips: int = 2, interleaved: bool = False, bands=None):
        import torch
        self.torch = torch
        self.width, self.height = width, height
        self.rank, self.world = rank, world_size
        self.tiles_y = (height + 15) // 16
        # interleaved: rank r owns tile rows r, r + world, ...; its strip holds them packed (owned row k at strip
        # rows [16 k, 16 k + 16)), which is what a context set up with gs_set_tile_rows_interleaved(r, world, 1) writes
        self.interleaved = interleaved
        self.group = group
        self.device = device
        self.n_strips = n_strips
        # host_gather: rehearsal of the N > 1 path on one GPU (gloo has no device gather)
        self.host_gather = host_gather
        self._pending = [None] * n_strips
        self.rows = 0
        self.set_bands(bands if bands is not None else tile_row_partition(self.tiles_y, world_size))

    def set_bands(self, bands):
        """Contiguous bands of ANY heights (balanced_row_partition / RowBalancer): every rank passes the same list.  The
        strips are as tall as the tallest band (one gather of equal strips; the padding is cropped by assemble) and are
        re-allocated only when that height grows.  No gather may be pending."""
        torch = self.torch
        bands = [(int(b), int(e)) for b, e in bands]
        assert len(bands) == self.world and bands[0][0] == 0 and bands[-1][1] == self.tiles_y
        assert all(bands[r][1] == bands[r + 1][0] for r in range(self.world - 1)) and all(b <= e for b, e in bands)
        assert all(p is None for p in self._pending), "set_bands with a gather in flight"
        self.bands = bands
        rows = (self.tiles_y + self.world - 1) // self.world if self.interleaved else max(e - b for b, e in bands)
        rows = max(rows, 1) * 16
        if rows > self.rows:
            self.rows = rows
            self.strips = [torch.zeros((rows, self.width, 4), dtype=torch.uint8, device=self.device) for _ in range(self.n_strips)]
            gdev = "cpu" if self.host_gather else self.device
            self.gathered = [([torch.zeros((rows, self.width, 4), dtype=torch.uint8, device=gdev) for _ in range(self.world)]
                              if self.rank == 0 and self.world > 1 else None) for _ in range(self.n_strips)]

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
