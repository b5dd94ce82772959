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


def _global_writer_map(shape, eff, ov):
    """tile index (in distinct_origins order) that writes each voxel last, -1 = never written."""
    tiles = cropper.distinct_origins(shape, list(eff), ov)
    own = [cropper.owner_table(d, c, o) for d, c, o in zip(shape, eff, ov)]
    idx = {t: i for i, t in enumerate(tiles)}
    G = np.full(shape, -1, dtype=np.int32)
    for t in tiles:
        r = [np.nonzero(o == v)[0] for o, v in zip(own, t)]
        if all(len(a) for a in r):
            G[r[0][0]:r[0][-1] + 1, r[1][0]:r[1][-1] + 1, r[2][0]:r[2][-1] + 1] = idx[t]
    return G, tiles, idx


@pytest.mark.parametrize("shape,world", [((1024, 1024, 256), 2), ((2048, 1024, 256), 2), ((2048, 2048, 256), 4),
                                         ((2048, 2048, 512), 8), ((640, 333, 250), 3)])
def test_every_tile_runs_once_and_the_split_is_balanced(shape, world):
    tile, ov = (300, 300, 20), (50, 50, 5)
    plan, eff = P.tile_plan(shape, tile, ov, world)
    all_tiles = cropper.distinct_origins(shape, list(tile), ov)
    assert sorted(t for p in plan for t in p) == sorted(all_tiles)  # each tile exactly once
    n = [len(p) for p in plan]
    assert max(n) - min(n) <= 1, n  # 2048x2048x512 on 8 ranks: 637 or 638 tiles each, not 600 / 700
    slabs = P.slab_bounds(shape[2], world)
    for r, p in enumerate(plan):
        w = P.window_of(slabs[r], shape[2], world)
        assert all(w[0] <= o[2] and o[2] + eff[2] <= w[1] for o in p)  # tiles fit the window
    # every box a tile writes outside its rank's slab is scheduled exactly once, inside both windows
    for src, dst, (x0, x1, y0, y1, z0, z1) in P.block_plan(shape, eff, ov, plan):
        assert src != dst and slabs[dst][0] <= z0 < z1 <= slabs[dst][1]
        ws = P.window_of(slabs[src], shape[2], world)
        assert ws[0] <= z0 and z1 <= ws[1]


def test_thin_slabs_are_refused():
    with pytest.raises(ValueError, match="too thin"):
        P.tile_plan((300, 300, 64), (300, 300, 20), (50, 50, 5), 4, halo=4)


@pytest.mark.parametrize("shape,tile,ov,world", [((30, 26, 40), (12, 12, 8), (2, 2, 2), 2),
                                                 ((30, 26, 41), (12, 12, 8), (2, 2, 2), 3),
                                                 ((25, 12, 64), (12, 12, 8), (2, 3, 1), 4)])
def test_block_plan_delivers_every_slab_voxel(shape, tile, ov, world):
    """Simulated ranks (no process group): scatter own tiles into the window, apply the block plan,
    compare every slab with the single-rank writer map."""
    G, tiles, idx = _global_writer_map(shape, tile, ov)
    plan, eff = P.tile_plan(shape, tile, ov, world, halo=10)
    slabs = P.slab_bounds(shape[2], world)
    own = [cropper.owner_table(d, c, o) for d, c, o in zip(shape, eff, ov)]
    local = []
    for r in range(world):
        L = np.full(shape, -1, dtype=np.int32)  # full-size array, only the window is ever touched
        for t in plan[r]:
            rr = [np.nonzero(o == v)[0] for o, v in zip(own, t)]
            if all(len(a) for a in rr):
                L[rr[0][0]:rr[0][-1] + 1, rr[1][0]:rr[1][-1] + 1, rr[2][0]:rr[2][-1] + 1] = idx[t]
        local.append(L)
    for src, dst, (x0, x1, y0, y1, z0, z1) in P.block_plan(shape, eff, ov, plan):
        local[dst][x0:x1, y0:y1, z0:z1] = local[src][x0:x1, y0:y1, z0:z1]
    for r, (lo, hi) in enumerate(slabs):
        assert np.array_equal(local[r][:, :, lo:hi], G[:, :, lo:hi]), r


def _blocks_worker(rank, world, port, shape, tile, ov, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        G, tiles, idx = _global_writer_map(shape, tile, ov)
        plan, eff = P.tile_plan(shape, tile, ov, world, halo=10)
        slabs = P.slab_bounds(shape[2], world)
        wins = [P.window_of(s, shape[2], world, halo=10) for s in slabs]
        (zlo, zhi), (wlo, whi) = slabs[rank], wins[rank]
        own = [cropper.owner_table(d, c, o) for d, c, o in zip(shape, eff, ov)]
        a = torch.full((shape[0], shape[1], whi - wlo), -1, dtype=torch.int32)
        b = torch.full((shape[0], shape[1], whi - wlo, 4), -1, dtype=torch.int16)
        for t in plan[rank]:
            rr = [np.nonzero(o == v)[0] for o, v in zip(own, t)]
            if all(len(x) for x in rr):
                sl = (slice(rr[0][0], rr[0][-1] + 1), slice(rr[1][0], rr[1][-1] + 1),
                      slice(rr[2][0] - wlo, rr[2][-1] + 1 - wlo))
                a[sl] = idx[t]
                b[sl] = idx[t]
        P.exchange_blocks([a, b], P.block_plan(shape, eff, ov, plan), wins, rank, P.Comm(rank, world))
        want = torch.from_numpy(G[:, :, zlo:zhi])
        ok_a = torch.equal(a[:, :, zlo - wlo:zhi - wlo], want)
        ok_b = torch.equal(b[:, :, zlo - wlo:zhi - wlo], want.to(torch.int16).unsqueeze(-1).expand(-1, -1, -1, 4))
        q.put((rank, ok_a, ok_b))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape", [(2, (30, 26, 40)), (3, (30, 26, 41))])
def test_block_exchange_gloo(world, shape):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_blocks_worker, args=(r, world, port, shape, (12, 12, 8), (2, 2, 2), q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_a, ok_b in results:
        assert ok_a and ok_b, (rank, ok_a, ok_b)


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
        # trimmed exchange: only the planes each rank says it needs are filled
        needs = [(max(w[0], s_[0] - 2), min(w[1], s_[1] + 3)) for s_, w in zip(slabs, wins)]
        local2 = torch.full((X, Y, whi - wlo, 4), -1, dtype=torch.int32)
        local2[:, :, zlo - wlo:zhi - wlo] = G[:, :, zlo:zhi]
        P.exchange_halo(local2, slabs, wins, rank, comm, needs=needs)
        nlo, nhi = needs[rank]
        ok_halo = ok_halo and torch.equal(local2[:, :, nlo - wlo:nhi - wlo], G[:, :, nlo:nhi]) and \
            bool((local2[:, :, :nlo - wlo] == -1).all()) and bool((local2[:, :, nhi - wlo:] == -1).all())
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


def test_assign_reach_covers_the_owning_crops():
    from skoots_amd.lib.eval import ASSIGN_CROP, ASSIGN_OVERLAP
    for shape, world in (((2048, 2048, 512), 8), ((2048, 1024, 256), 2), ((320, 304, 180), 3), ((64, 64, 40), 2)):
        slabs = P.slab_bounds(shape[2], world)
        wins = [P.window_of(s_, shape[2], world) for s_ in slabs]
        eff = cropper.clamp_crop_(list(ASSIGN_CROP), shape)
        own = cropper.owner_table(shape[2], eff[2], ASSIGN_OVERLAP[2])
        for (lo, hi), (wl, wh), (a, b) in zip(slabs, wins, P.assign_reach(shape, ASSIGN_CROP, ASSIGN_OVERLAP, slabs, wins)):
            assert wl <= a <= lo and hi <= b <= wh
            for z in range(lo, hi):
                if own[z] >= 0:
                    assert a <= own[z] and own[z] + eff[2] <= b


def test_eight_rank_plan_at_2048x2048x512_and_its_byte_budget():
    """BASELINE configs[3] -- 2048x2048x512 on 8 ranks -- has never run on hardware (no 8-GPU node was available to any
    round).  This pins what the first run will be read against: the whole plan (tile plan, block plan, stage-3 reach,
    halo plan) at the real size and the bytes every rank sends per step under every exchange tag
    (parallel.comm_budget; DESIGN.md section 7 quotes these numbers).  xGMI moves ~153 GB/s per link and neighbour."""
    shape, world = (2048, 2048, 512), 8
    plan, eff = P.tile_plan(shape, (300, 300, 20), (50, 50, 5), world)
    assert eff == [300, 300, 20]
    assert sorted(len(p) for p in plan) == [637] * 4 + [638] * 4 and sum(len(p) for p in plan) == 5100
    assert len({o for p in plan for o in p}) == 5100                     # every distinct tile of the reference grid, once
    assert 5100 == len(cropper.distinct_origins(shape, [300, 300, 20], (50, 50, 5)))
    assert cropper.get_total_num_crops((1,) + shape, [300, 300, 20], (50, 50, 5)) == 6292                                       # what the reference's generator emits
    slabs = P.slab_bounds(512, world)
    windows = [P.window_of(s, 512, world) for s in slabs]
    for r, tiles in enumerate(plan):                                      # a rank's tiles lie inside its window
        assert all(windows[r][0] <= z and z + 20 <= windows[r][1] for (_, _, z) in tiles)
    budget = P.comm_budget(shape, world)
    MiB = 1 << 20
    assert [b["tiles"] for b in budget] == [len(p) for p in plan]
    # stage 1: a rank hands 36-176 MiB of (vector, skeleton) boxes to one or two NEIGHBOUR ranks only
    for r, b in enumerate(budget):
        assert 35 * MiB <= b["block_exchange"] <= 176 * MiB, (r, b["block_exchange"] / MiB)
        assert b["block_peers"] and all(abs(q - r) == 1 for q in b["block_peers"])
    total_blocks = sum(b["block_exchange"] for b in budget)
    assert 1.0 * 1024 * MiB < total_blocks < 1.2 * 1024 * MiB            # 1.09 GiB per step over the whole job
    # the blocks delivered + the voxels a rank keeps cover every written voxel of every slab exactly once
    blocks = P.block_plan(shape, eff, (50, 50, 5), plan)
    assert sum((b[2][1] - b[2][0]) * (b[2][3] - b[2][2]) * (b[2][5] - b[2][4]) for b in blocks) * 9 == total_blocks
    # stage 2: one 16 MiB int32 plane to the rank below, two fixed-size all-gathers
    assert [b["label_seam_planes"] for b in budget] == [0] + [16 * MiB] * 7
    assert all(b["label_meta"] == (4 + 2 * 16384) * 4 * 7 for b in budget)
    assert all(b["label_gather"] == (2048 * 2048 * 64 // 64) * 8 * 7 for b in budget)   # 32 MiB per receiver
    # stage 3: 8-42 planes of interleaved vectors per neighbour (32 MiB a plane), at most 2.4 GiB per rank
    for r, b in enumerate(budget):
        assert set(b["vector_halo_planes"]) == {q for q in (r - 1, r + 1) if 0 <= q < world}
        assert all(8 <= n <= 42 for n in b["vector_halo_planes"].values()), b["vector_halo_planes"]
        assert b["vector_halo"] == 32 * MiB * sum(b["vector_halo_planes"].values()) <= 2368 * MiB
    # per-rank total and the time it costs at one xGMI link per neighbour (153 GB/s): well under 3 % of a ~0.45 s stage 1
    worst = max(b["block_exchange"] + b["label_seam_planes"] + b["vector_halo"] for b in budget)
    assert worst / 153e9 < 0.02
    # one rank plans nothing
    solo = P.comm_budget((1024, 1024, 256), 1)
    assert solo[0]["tiles"] == 625 and solo[0]["block_exchange"] == 0 and solo[0]["vector_halo"] == 0


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


# ---- bench.py --gpus N launch path (no launcher in front of it) ------------------------------------
def _bench(*argv, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True,
                          timeout=600, env=e)


def test_bench_launcher_spawns_ranks_gloo():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts two ranks itself (here: the gloo dry run of the
    launch path) and relays rank 0's single JSON line."""
    import json
    r = _bench("--gpus", "2", "--launcher-dry-run")
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["rank_sum"] == 1 and line["launcher"] == "ok"


def test_bench_refuses_more_gpus_than_devices():
    """A multi-GPU request on a node without that many devices fails loudly: non-zero exit, no JSON line
    (the old behaviour printed n_gpus: 1 for `--gpus 8`)."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("node has 8 devices")
    r = _bench("--gpus", "8", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "refusing" in r.stderr and not r.stdout.strip()


def test_bench_refuses_world_size_mismatch():
    r = _bench("--gpus", "4", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()
