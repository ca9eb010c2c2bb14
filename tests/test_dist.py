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
