"""world_size-2 gloo tests (CPU) of the multi-GPU path: tile-row partition, strip gather and frame
assembly.  The band renderer is injected: on the GPU it is Renderer.drawDevice into the strip, here it
is the oracle's band mode (tests may use the checker), so what is tested is exactly the N > 1 plumbing
of vk3dgaussiansplatting_amd/dist.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as tdist
import torch.multiprocessing as mp

from vk3dgaussiansplatting_amd import dist as gsdist


def test_tile_row_partition_covers_everything():
    for tiles_y in (1, 2, 7, 23, 68, 135):
        for world in (1, 2, 3, 4, 8):
            bands = gsdist.tile_row_partition(tiles_y, world)
            assert len(bands) == world
            assert bands[0][0] == 0 and bands[-1][1] == tiles_y or bands[-1][0] == bands[-1][1] == tiles_y
            flat = [r for b, e in bands for r in range(b, e)]
            assert flat == list(range(tiles_y))                       # disjoint, ordered, complete
            per = gsdist.strip_rows(tiles_y, world) // 16
            assert all(e - b <= per for b, e in bands)
    assert gsdist.tile_row_partition(135, 8) == [(0, 17), (17, 34), (34, 51), (51, 68), (68, 85), (85, 102), (102, 119), (119, 135)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, aos, view, proj, q):
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    sf = gsdist.ShardedFrame(w, h, rank, world, device="cpu")

    def render_band(rb, re, strip):
        p = oracle.make_params(w, h, view, proj, (0, 0, 0), row_begin=rb, row_end=re)
        r = oracle.full_pipeline(p, aos)
        rows = slice(rb * 16, min(re * 16, h))
        band = r["image"][rows]
        strip.zero_()
        strip[: band.shape[0]] = torch.from_numpy(np.ascontiguousarray(band))

    for _ in range(2):                                                # two frames through the same buffers
        img = sf.frame(render_band)
    if rank == 0:
        q.put(img.numpy())
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_equals_single_process(oracle_mod, world):
    from vk3dgaussiansplatting_amd import synth
    w, h = 200, 150                                                   # 10 tile rows, last one partial
    aos = synth.generate(1500, w, h, -2.2, seed=77)
    view, proj = oracle_mod.camera_matrices(np.zeros(3, np.float32), 0.0, 0.0, w / h)
    whole = oracle_mod.full_pipeline(oracle_mod.make_params(w, h, view, proj, (0, 0, 0)), aos)["image"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, w, h, aos, view, proj, q)) for r in range(world)]
    for p in procs:
        p.start()
    img = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert img.shape == (h, w, 4)
    assert np.array_equal(img, whole)


def _ring_worker(rank, world, port, w, h, q):
    """Three strips used round-robin with asynchronous gathers, as bench.py's frame slots do: frame f fills strip
    f % 3 with a value that identifies (frame, rank), starts its gather and moves on; before a strip is filled again
    its previous gather is waited for.  Rank 0 checks every assembled frame."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    sf = gsdist.ShardedFrame(w, h, rank, world, device="cpu", n_strips=3)
    b, e = sf.band
    ok, pending = True, {}
    frames = 8

    def check(k, f):
        sf.wait(k)
        if rank == 0:
            img = sf.assemble(sf.gathered[k])
            for r, (rb, re) in enumerate(sf.bands):
                rows = img[rb * 16:min(re * 16, h)]
                if rows.numel() and not bool((rows == (10 * f + r) % 251).all()):
                    return False
        return True

    for f in range(frames):
        k = f % 3
        if k in pending:
            ok = check(k, pending.pop(k)) and ok
        sf.strips[k].fill_((10 * f + rank) % 251)
        sf.gather_async(k)
        pending[k] = f
    for k, f in sorted(pending.items(), key=lambda kv: kv[1]):
        ok = check(k, f) and ok
    sf.wait_all()
    if rank == 0:
        q.put(ok)
    tdist.barrier()
    tdist.destroy_process_group()


def test_three_strips_in_flight():
    w, h, world = 96, 150, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ring_worker, args=(r, world, port, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    assert q.get(timeout=120) is True
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0


def test_balanced_row_partition_properties():
    rng = np.random.default_rng(3)
    for tiles_y in (1, 2, 7, 23, 68, 135):
        for world in (1, 2, 3, 4, 8):
            for kind in ("uniform", "ramp", "spike", "zeros", "random"):
                w = {"uniform": np.ones(tiles_y), "ramp": np.arange(1, tiles_y + 1.0), "zeros": np.zeros(tiles_y),
                     "spike": np.where(np.arange(tiles_y) == tiles_y // 2, 1e6, 1.0), "random": rng.integers(0, 500000, tiles_y)}[kind]
                bands = gsdist.balanced_row_partition(w, world)
                assert len(bands) == world and bands[0][0] == 0 and bands[-1][1] == tiles_y
                assert all(bands[r][1] == bands[r + 1][0] for r in range(world - 1))        # contiguous, ordered, complete
                assert all(b <= e for b, e in bands)
                if tiles_y >= world:
                    assert all(e > b for b, e in bands)                                     # nobody idles while there are rows
    # equal weights -> equal row counts (within one row); a ramp gives the heavy end fewer rows
    assert [e - b for b, e in gsdist.balanced_row_partition(np.ones(68), 4)] == [17, 17, 17, 17]
    ramp = [e - b for b, e in gsdist.balanced_row_partition(np.arange(1, 69.0), 4)]
    assert ramp[0] > ramp[1] > ramp[2] > ramp[3]
    # the split is at least as even (in weight) as equal row counts for a clustered profile
    w = np.r_[np.full(20, 50_000.0), np.full(10, 450_000.0), np.full(38, 120_000.0)]
    worst = lambda bands: max(w[b:e].sum() for b, e in bands)
    assert worst(gsdist.balanced_row_partition(w, 8)) < 0.8 * worst(gsdist.tile_row_partition(68, 8))


def test_row_balancer_is_deterministic_and_has_hysteresis():
    w = np.r_[np.full(20, 50_000), np.full(10, 450_000), np.full(38, 120_000)]
    a, b = gsdist.RowBalancer(68, 8), gsdist.RowBalancer(68, 8)
    assert a.update(w) is True and b.update(w) is True and a.bands == b.bands != gsdist.tile_row_partition(68, 8)
    before = list(a.bands)
    assert a.update(w * 1.01) is False and a.bands == before              # a percent of noise moves nothing
    # measured share times steer the weights: a rank that takes twice as long per element sheds rows
    ms = [1.0] * 8
    ms[3] = 2.0
    rows3 = a.bands[3][1] - a.bands[3][0]
    assert a.update(w, ms) is True and a.bands[3][1] - a.bands[3][0] < rows3
    # ranges -> elements per tile row
    ranges = np.zeros((3 * 4, 2), np.uint32)
    ranges[4:8, 0] = [10, 12, 15, 15]
    ranges[4:8, 1] = [12, 15, 15, 20]
    assert list(gsdist.RowBalancer.row_elements(ranges, 4, 3)) == [0, 10, 0]


def _rebalance_worker(rank, world, port, w, h, aos, view, proj, q):
    """Frames with the bands MOVING between them (equal rows -> element-balanced -> a lopsided split): every rank derives
    the new bands from the all-reduced per-row element counts, strips are re-sized, the assembled frame never changes."""
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    sf = gsdist.ShardedFrame(w, h, rank, world, device="cpu")
    tiles_y, tiles_x = sf.tiles_y, (w + 15) // 16
    bal = gsdist.RowBalancer(tiles_y, world, min_gain=0.0)
    mine = {}

    def render_band(rb, re, strip):
        p = oracle.make_params(w, h, view, proj, (0, 0, 0), row_begin=rb, row_end=re)
        r = oracle.full_pipeline(p, aos)
        mine["rows"] = gsdist.RowBalancer.row_elements(r["ranges"], tiles_x, tiles_y)
        band = r["image"][rb * 16:min(re * 16, h)]
        strip.zero_()
        strip[: band.shape[0]] = torch.from_numpy(np.ascontiguousarray(band))

    imgs, bands_seen = [], []
    for step in range(3):
        img = sf.frame(render_band)
        bands_seen.append(list(sf.bands))
        if rank == 0:
            imgs.append(img.numpy().copy())
        rows = torch.from_numpy(mine["rows"].astype(np.int64))
        tdist.all_reduce(rows)                                         # every row from the rank that owns it
        if step == 0:
            bal.update(rows.numpy())
            sf.set_bands(bal.bands)
        else:
            lop = [(0, 1)] + [(1 + (tiles_y - 1) * (r - 1) // (world - 1), 1 + (tiles_y - 1) * r // (world - 1)) for r in range(1, world)]
            sf.set_bands(lop)
    if rank == 0:
        q.put((imgs, bands_seen, int(rows.sum())))
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rebalanced_bands_give_the_same_frame(oracle_mod, world):
    from vk3dgaussiansplatting_amd import synth
    w, h = 200, 150
    aos = synth.generate(1500, w, h, -2.2, seed=78)
    aos[:, 1] = np.abs(aos[:, 1]) * 0.6 + 0.2 * aos[:, 2]              # crowd the splats into the upper rows
    view, proj = oracle_mod.camera_matrices(np.zeros(3, np.float32), 0.0, 0.0, w / h)
    ref = oracle_mod.full_pipeline(oracle_mod.make_params(w, h, view, proj, (0, 0, 0)), aos)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rebalance_worker, args=(r, world, port, w, h, aos, view, proj, q)) for r in range(world)]
    for p in procs:
        p.start()
    imgs, bands_seen, e_sum = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert e_sum == ref["e"]                                           # the summed row counts are the frame's elements
    assert bands_seen[0] == gsdist.tile_row_partition(10, world) and bands_seen[1] != bands_seen[0] and bands_seen[2] != bands_seen[1]
    for img in imgs:
        assert np.array_equal(img, ref["image"])
