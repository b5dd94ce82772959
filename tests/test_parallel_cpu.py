"""CPU tests of the Z-sharding plan and its collectives: world_size-2 (and 3) gloo process
groups on 127.0.0.1.  No GPU compute: slab/window/tile ownership logic plus the halo
exchange, all-gather and all-reduce plumbing of skoots_amd.parallel on CPU tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from skoots_amd import parallel as P
from skoots_amd.lib import cropper


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_slabs_and_windows_cover_the_volume():
    for Z, world in ((256, 1), (256, 2), (512, 8), (250, 3)):
        slabs = P.slab_bounds(Z, world)
        assert slabs[0][0] == 0 and slabs[-1][1] == Z
        assert all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]))
        for s in slabs:
            w = P.window_of(s, Z, world)
            assert w[0] <= s[0] and s[1] <= w[1] and 0 <= w[0] and w[1] <= Z


@pytest.mark.parametrize("shape,world", [((1024, 1024, 256), 2), ((2048, 2048, 512), 8), ((640, 333, 250), 3)])
def test_every_tile_runs_once_and_every_plane_is_delivered(shape, world):
    tile, ov = (300, 300, 20), (50, 50, 5)
    eff = list(tile)
    all_tiles = cropper.distinct_origins(shape, eff, ov)
    own_z = cropper.owner_table(shape[2], eff[2], ov[2])
    slabs = P.slab_bounds(shape[2], world)
    ext = P.straddle_extent(shape, tile, ov, slabs)
    union = []
    for r in range(world):
        mine, eff_r = P.tiles_for_slab(shape, tile, ov, slabs[r])
        assert eff_r == eff
        zs = {o[2] for o in mine}
        w = P.window_of(slabs[r], shape[2], world)
        assert all(w[0] <= o[2] and o[2] + eff[2] <= w[1] for o in mine)  # tiles fit the window
        assert mine == [o for o in all_tiles if o[2] in zs]  # reference order preserved
        # every owned plane is written by one of this rank's tiles or delivered by rank r-1
        prev = {o[2] for o in P.tiles_for_slab(shape, tile, ov, slabs[r - 1])[0]} if r else set()
        for z in range(*slabs[r]):
            if own_z[z] < 0:
                continue
            if own_z[z] in zs:
                continue
            assert own_z[z] in prev and z < slabs[r][0] + ext[r - 1], (r, z)
        assert slabs[r][1] + ext[r] <= w[1]
        union += mine
    assert sorted(union) == sorted(all_tiles)  # each tile exactly once


def _worker(rank, world, port, shape, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, Y, Z = shape
        g = torch.Generator().manual_seed(7)
        G = torch.randint(0, 1000, (X, Y, Z, 4), generator=g).to(torch.int32)  # same on every rank
        slabs = P.slab_bounds(Z, world)
        wins = [P.window_of(s, Z, world, halo=6) for s in slabs]
        (zlo, zhi), (wlo, whi) = slabs[rank], wins[rank]
        comm = P.Comm(rank, world)
        local = torch.full((X, Y, whi - wlo, 4), -1, dtype=torch.int32)
        local[:, :, zlo - wlo:zhi - wlo] = G[:, :, zlo:zhi]
        P.exchange_halo(local, slabs, wins, rank, comm)
        ok_halo = torch.equal(local, G[:, :, wlo:whi])
        parts = comm.all_gather(torch.tensor([rank * 10 + 1]))
        ok_gather = [int(p.item()) for p in parts] == [r * 10 + 1 for r in range(world)]
        m = comm.all_reduce_min(torch.tensor([5 + rank, 100 - rank], dtype=torch.int64))
        ok_min = m.tolist() == [5, 100 - (world - 1)]
        q.put((rank, ok_halo, ok_gather, ok_min))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(2, (5, 4, 24)), (3, (3, 5, 31))])
def test_halo_exchange_gloo(world, shape):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_halo, ok_gather, ok_min in results:
        assert ok_halo and ok_gather and ok_min, (rank, ok_halo, ok_gather, ok_min)


def test_halo_plan_is_symmetric():
    slabs = P.slab_bounds(512, 8)
    wins = [P.window_of(s, 512, 8) for s in slabs]
    for r in range(8):
        sends, recvs = P.halo_plan(slabs, wins, r)
        for q, lo, hi in sends:
            assert (r, lo, hi) in P.halo_plan(slabs, wins, q)[1]
        got = sorted((lo, hi) for _, lo, hi in recvs)
        need = [(wins[r][0], slabs[r][0]), (slabs[r][1], wins[r][1])]
        covered = sum(hi - lo for lo, hi in got)
        assert covered == sum(b - a for a, b in need)


def _grad_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from skoots_amd.train import sync_gradients
        g = torch.arange(10, dtype=torch.float32) * (rank + 1)
        sync_gradients(g)
        q.put((rank, g.tolist()))
    finally:
        dist.destroy_process_group()


def test_gradient_sync_gloo():
    """Data-parallel training step: the flat gradient buffer is averaged over the ranks."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = (torch.arange(10, dtype=torch.float32) * 1.5).tolist()
    for _, got in results:
        assert got == want
