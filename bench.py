#!/usr/bin/env python3
"""bench.py -- Mvoxels/s end-to-end (3-D U-Net forward + instance assignment) on MI355X.

Metric / config: BASELINE.json.  A "step" is one pass of the whole hot path
(skoots/lib/eval.py:126-306: sliding-window U-Net, gate/dilate/scatter, skeleton
labelling, offset following + assignment, renumber) over one synthetic volume that is
already resident in HBM when the timed region starts.

  N = 1 : configs[2], 1024x1024x256 fp16 (the largest single-GPU end-to-end config)
  N > 1 : weak scaling, 2^28 voxels per GPU, Z-sharded: 2048x1024x256 (N=2),
          2048x2048x256 (N=4), 2048x2048x512 (N=8 = configs[3])

``--gpus N`` without a launcher (no WORLD_SIZE in the environment) starts the N ranks itself, one
process per device, and relays rank 0's JSON line; it exits non-zero when the node has fewer than N
devices -- a multi-GPU request never silently turns into a 1-GPU line.

``--precision``: "fp16" (default; the dtype BASELINE's configs name, fp16 MFMA operands -- what the
reference's autocast does), "split" (fp16 hi + lo operand pairs: max-abs <= 1e-3 vs fp32, the
north_star tolerance) or "fp32" (exact-fp32 MFMA).  Every line states its own measured tolerance
(``parity_vs_fp32_mode``: the benched precision against the exact-fp32 mode on one production batch).

``--config train``: BASELINE configs[4], one training step on a synthetic 256^3 crop (see train_measure).

The default N = 1 line (what the driver runs) carries three measurements from ONE process: ``value`` = the fp16 eval
path (BASELINE's dtype), ``also.split`` (and ``also.mix8``, its variant with fp8 correction products in the 3x3x3 convs)
= the same volume at the precision that meets north_star's 1e-3 tolerance
(value, ms_per_step, roofline, parity_vs_fp32_mode measured live) and ``also.train_bf16`` = configs[4] (ms_per_step,
Mvoxels/s trained, roofline); ``--no-also`` prints the first alone.  ``box`` = a bare-MFMA-loop probe of this device:
boxes differ by ~6 % under matrix load, figures from two boxes compare only beside it.

``--streams N`` (default 1): N tile batches in flight on N HIP streams.  ``--streams 2 --tile-batch 32`` measured +1.3 %
over one stream of 64-tile batches (the conv kernels fill the register file, so only launch tails overlap).  Overlapping
launches make per-launch durations meaningless -- rocprofv3's and the bench's own -- so with more than one stream
``roofline`` (and the stage rooflines) are measured in ONE extra single-stream step after the warm-up, outside the timed
region (``roofline.timed_over`` says so); the default stays one stream so that the driver's line and a rocprofv3 run of
the same command time the same launches.

Synthetic data: uint8-range random image, random-init network of the named shape
(DIMS [32,64,128,64,32], DEPTHS [2,2,2,2,2]).  A random-init net never crosses the 0.8
gates, so the post-network stages would see an empty skeleton; as SURVEY.md 8(d)
prescribes, a seeded blob field (tests/workload.py recipe, generated on the device)
replaces each tile's 5-channel output AFTER the network has run on the tile, so the
network is timed in full and the assignment stages do real work.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

SCALE = (60, 60, 12)  # cfg.SKOOTS.VECTOR_SCALING default, skoots/config.py:144
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E ~8 TB/s
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 matrix peak


def workload_shape(n_gpus: int):
    return {1: (1024, 1024, 256), 2: (2048, 1024, 256), 4: (2048, 2048, 256),
            8: (2048, 2048, 512)}.get(n_gpus, (1024, 1024, 256 * n_gpus))


def device_blob_field(shape, z_window, device, seed=0, pitch=(64, 64, 16), dense=False):
    """Blob field of tests/workload.py generated directly on the device for the local
    z-window: one ellipsoid per (64,64,16) lattice cell (jittered centre and radii), prob
    0.95 inside, vectors pointing at the centre / SCALE, skeleton ball at the centre.
    Returns (5, X, Y, zw) fp16 and the number of blobs in the whole volume.

    ``dense``: the worst case of the follow kernel instead of the ~94 %-background field: EVERY voxel of a cell is
    gated in and carries a non-zero vector -- half the pull towards the cell's centre plus a swirl around it, never
    exactly zero -- so every voxel makes all nine dependent hops of N = 10, each to a different voxel."""
    X, Y, Z = shape
    z0, z1 = z_window
    g = torch.Generator(device="cpu").manual_seed(seed)
    nx, ny, nz = X // pitch[0], Y // pitch[1], Z // pitch[2]
    jit = torch.rand((nx, ny, nz, 6), generator=g)
    cx = (torch.arange(nx).view(-1, 1, 1) * pitch[0] + 24 + jit[..., 0] * 16).to(device)
    cy = (torch.arange(ny).view(1, -1, 1) * pitch[1] + 24 + jit[..., 1] * 16).to(device)
    cz = (torch.arange(nz).view(1, 1, -1) * pitch[2] + 6 + jit[..., 2] * 4).to(device)
    rx = (10 + jit[..., 3] * 12).to(device)
    ry = (10 + jit[..., 4] * 12).to(device)
    rz = (2.5 + jit[..., 5] * 2.5).to(device)
    out = torch.zeros((5, X, Y, z1 - z0), dtype=torch.float16, device=device)
    xs = torch.arange(X, device=device, dtype=torch.float32)
    ys = torch.arange(Y, device=device, dtype=torch.float32)
    zs = torch.arange(z0, z1, device=device, dtype=torch.float32)
    ix = (xs / pitch[0]).long().clamp_(max=nx - 1)
    iy = (ys / pitch[1]).long().clamp_(max=ny - 1)
    iz = (zs / pitch[2]).long().clamp_(max=nz - 1)
    slab = 64  # x-slabs to bound temporaries
    for xa in range(0, X, slab):
        xb = min(X, xa + slab)
        I = ix[xa:xb].view(-1, 1, 1), iy.view(1, -1, 1), iz.view(1, 1, -1)
        dx = cx[I] - xs[xa:xb].view(-1, 1, 1)
        dy = cy[I] - ys.view(1, -1, 1)
        dz = cz[I] - zs.view(1, 1, -1)
        inside = (dx / rx[I]) ** 2 + (dy / ry[I]) ** 2 + (dz / rz[I]) ** 2 <= 1.0
        core = (dx * dx + dy * dy + dz * dz <= 2.5) & inside
        o = out[:, xa:xb]
        if dense:
            inside = torch.ones_like(inside)
            eps = 0.26   # a quarter voxel: the vector is never zero, the follow never reaches its fixed-point exit
            vx, vy, vz = 0.5 * dx - 0.5 * dy + eps, 0.5 * dy + 0.5 * dx + eps, 0.5 * dz + eps
            dx, dy, dz = vx, vy, vz.expand_as(vx)
        o[0] = torch.where(inside, (dx / SCALE[0]).clamp(-1, 1), 0).half()
        o[1] = torch.where(inside, (dy / SCALE[1]).clamp(-1, 1), 0).half()
        o[2] = torch.where(inside, (dz / SCALE[2]).clamp(-1, 1), 0).half()
        o[3] = torch.where(core, 0.92, 0.0).half()
        o[4] = torch.where(inside, 0.95, 0.0).half()
    # cells whose blob centre lies inside the written frame produce an instance
    return out, nx * ny * nz


def contiguous_tile_blocks(field, shape, rank, world, zlo):
    """{window-local tile origin: (5, w, h, d) fp16 view} of this rank's tiles of the reference grid, all views of ONE
    contiguous (T, 5, w, h, d) buffer copied from ``field`` (5, X, Y, window planes) -- the layout a network output has."""
    from skoots_amd.parallel import tile_plan
    plan, eff = tile_plan(shape, (300, 300, 20), (50, 50, 5), world)
    mine = plan[rank]
    blocks = torch.empty((len(mine), 5) + tuple(eff), dtype=field.dtype, device=field.device)
    cache = {}
    for i, (x, y, z) in enumerate(mine):
        blocks[i].copy_(field[:, x:x + eff[0], y:y + eff[1], z - zlo:z - zlo + eff[2]])
        cache[(x, y, z - zlo)] = blocks[i]
    return cache, blocks


def cpu_threads() -> int:
    """Host threads for the CPU baseline: the GPU box grants 16 cores per GPU."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(budget_s: float = 8.0):
    """The CPU restatement (oracle/) timed on this host on a bounded sample (~20 s) and reported for
    the SAME workload as the GPU line, BASELINE configs[2] (1024x1024x256), in the units the reference
    does the work in:

      stage 1   seconds per 300x300x20 tile (network + gate/dilate/scatter, eval.py:126-176), measured on
                a few tiles, x 936 tiles -- the reference's generator emits every clamped duplicate
                (cropper.py:97-144)
      stage 2   labelling of one 500x500x50 volume with a blob-field skeleton, x voxels / 12.5 M
      stage 3   one 500x500x50 crop: vector_to_embedding(N=10) + index_skeleton_by_embed
                (eval.py:271-279), x 63 crops
      renumber  one 500x500x50 int volume, x voxels / 12.5 M

    ``config0_value``: the reference's own CPU-runnable case (configs[0], 128x128x32: 100 tiles for 3
    distinct origins -- 62x overcompute, so not comparable with the GPU line) measured the same way."""
    import numpy as np
    from oracle import pipeline as O
    from oracle import unet_spec
    threads = cpu_threads()
    torch.set_num_threads(threads)
    model = unet_spec.build()
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        # ---- stage 1: per-tile cost at the production tile size -----------------------------------
        model(torch.zeros(1, 1, 64, 64, 20))  # warm the allocator / thread pool, untimed
        img = torch.randint(0, 256, (1, 300, 300, 20), generator=g).to(torch.float16)
        prog = {}
        O.stage1(img, model, img.mean(), img.std(), budget_s=budget_s, progress=prog)
        t_tile, n_tiles = prog["seconds"] / prog["done"], prog["done"]
        # ---- configs[0]: 128x128x32, per-tile cost x 100 tiles + stages 2-3 in full ------------
        img0 = torch.randint(0, 256, (1, 128, 128, 32), generator=g).to(torch.float16)
        prog0 = {}
        v0, s0 = O.stage1(img0, model, img0.mean(), img0.std(), budget_s=budget_s / 2, progress=prog0)
        t0 = time.perf_counter()
        O.post_model(v0, s0, SCALE)
        t_c0 = prog0["seconds"] * prog0["total"] / prog0["done"] + time.perf_counter() - t0
        # ---- stages 2-3 + renumber on one 500x500x50 crop of a blob field ------------------------
        from tests.workload import blob_field
        field, _ = blob_field((500, 500, 50), seed=1, n_blobs=60, rmax=(18, 18, 5), noise=0.0)
        skel = (field[3] > 0.8).numpy().astype(np.uint8)[None]
        t0 = time.perf_counter()
        labels = O.stage2(skel)
        t_ccl = time.perf_counter() - t0
        vec = field[0:3].to(torch.float16).unsqueeze(0)
        t0 = time.perf_counter()
        emb = O.vector_to_embedding(torch.as_tensor(SCALE), vec, N=O.FOLLOW_N)      # eval.py:271-273
        inst = O.index_skeleton_by_embed(labels.unsqueeze(0).unsqueeze(0), emb)      # eval.py:277-279
        t_crop = time.perf_counter() - t0
        t0 = time.perf_counter()
        O.renumber(inst[0, 0].numpy())
        t_ren = time.perf_counter() - t0
    vox2 = 1024 * 1024 * 256
    crop_vox = 500 * 500 * 50
    t2 = 936 * t_tile + (vox2 / crop_vox) * (t_ccl + t_ren) + 63 * t_crop
    return {"value": round(vox2 / t2 / 1e6, 4), "unit": "Mvoxels/s", "cores": threads, "kind": "port",
            "config0_value": round(128 * 128 * 32 / t_c0 / 1e6, 5),
            "sample": f"torch CPU {threads} threads, configs[2] 1024x1024x256 extrapolated from measured units: "
                      f"{t_tile:.2f} s per 300x300x20 tile (network + gate/dilate, {n_tiles} tiles timed) x 936 reference "
                      f"tiles + {t_crop:.2f} s per 500x500x50 follow+assign crop x 63 + labelling {t_ccl:.2f} s and "
                      f"renumber {t_ren:.2f} s per 12.5 Mvoxel x {vox2 / crop_vox:.1f} = {t2:.0f} s; config0_value = "
                      f"configs[0] 128x128x32 (100 tiles, 62x overcompute) measured the same way"}


def _newest_profile(stem):
    """profiles/rNN_<stem>[_vK].json with the highest (round, version); None when none is committed."""
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", f"r??_{stem}*.json")):
        m = re.match(rf"r(\d+)_{stem}(?:_v(\d+))?\.json$", os.path.basename(path))
        if m:
            key = (int(m.group(1)), int(m.group(2) or 1))
            if best is None or key > best[0]:
                best = (key, path)
    return best[1] if best else None


def conv_hbm_traffic(tile_batch):
    """HBM bytes per conv3 launch of ``tile_batch`` tiles from the newest committed PMC passes (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs of the same launch shapes, FETCH_SIZE doubled per
    MI355X_MICROARCH.md's gfx950 note; tools/pmc_collect.sh).  A profile taken at another batch size is scaled per
    tile and the source says so.  bench.py cannot collect PMC counters itself; (None, None) when nothing is committed."""
    import re
    path = _newest_profile("conv_hbm_traffic_pmc")
    if path is None:
        return None, None
    prof = json.load(open(path))
    m = re.search(r"--batch (\d+)", prof.get("command", ""))
    pbatch = int(prof.get("batch") or (m.group(1) if m else 8))
    tot, n = 0.0, 0
    for k in prof["kernels"]:
        if any(nm in k["kernel"] for nm in ("conv3_kernel", "conv3_m16_kernel", "conv3_upf_kernel", "conv3_px_kernel")):
            tot += (k["fetch_MB_per_launch_x2_gfx950_correction"] + k["write_MB_per_launch"]) * k["launches"]
            n += k["launches"]
    if not n:
        return None, None
    src = os.path.basename(path) + ("" if pbatch == tile_batch else f", collected at {pbatch} tiles per launch and scaled per tile")
    return round(tot / n * 1024 * 1024 / pbatch * tile_batch), src


def stage_hbm_traffic():
    """{kernel-name substring: bytes per launch} of the stage 2-3 kernels at 1024x1024x256 from the newest committed
    PMC passes over tools/bench_stages.py (tools/pmc_stages.sh); {} when nothing is committed."""
    path = _newest_profile("stage23_hbm_traffic_pmc")
    if path is None:
        return {}, None
    prof = json.load(open(path))
    return prof.get("stages", {}), os.path.basename(path)


def stage_rooflines(sprof):
    """HBM-bound stages.  ``achieved`` = ALGORITHMIC bytes / HIP-event kernel time; ``traffic`` = HBM-side
    bytes per step of the stage's kernels from the committed PMC passes (same volume, same blob field), and when it is
    there ``frac`` is the MEASURED byte rate against the 8 TB/s peak (``frac_algorithmic`` keeps the other one): the
    algorithmic count overstates a sparse field -- a background voxel of the follow kernel moves ~16 B, not 64."""
    out = {}
    measured, src = stage_hbm_traffic()
    names = {"gate_dilate_scatter": "roofline_gate", "ccl": "roofline_ccl", "follow_assign": "roofline_assign"}
    per_voxel = {"gate_dilate_scatter": 17, "ccl": 9, "follow_assign": 64}
    for kname, (kms, kbytes, kn) in sprof.totals().items():
        gbs = kbytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        e = {"bound": "hbm", "kernel": kname, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "launches": kn,
             "avg_launch_ms": round(kms / max(kn, 1), 4), "algorithmic_bytes_per_voxel": per_voxel[kname]}
        m = measured.get(kname)
        nsteps = kbytes / m["algorithmic_bytes_per_step"] if m else 0.0
        if m and kms > 0 and nsteps >= 0.99 and abs(nsteps - round(nsteps)) < 0.01:   # the profiled volume is this volume
            steps = round(nsteps)
            e["traffic"] = int(m["bytes_per_step"])
            mg = m["bytes_per_step"] * steps / (kms * 1e-3) / 1e9
            e["frac_algorithmic"] = e["frac"]
            e["achieved_measured"] = round(mg, 1)
            e["frac"] = round(mg / HBM_PEAK_GBS, 4)
            e["traffic_unit"] = f"HBM-side bytes per step (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/{src}); frac = measured bytes / time / peak"
        out[names[kname]] = e
    return out


def box_probe(dev):
    """Rates of a bare MFMA register loop on this device (sk_mfma_probe; pseudo-random operands, one pair and rotating), printed next to the line: MI355X boxes hold
    different clocks under matrix load (the same code read 548-580 Mvox/s across boxes in round 2), so a figure from
    another box is comparable only beside this one.  ``band``: this device against the rate the loop sustains on most boxes."""
    import ctypes as C
    from skoots_amd import _ffi
    scratch = torch.empty(512 * 256, dtype=torch.float32, device=dev)
    fl = C.c_double(0.0)
    st = _ffi.stream_ptr(dev)
    rates = []
    for vary in (0, 1):
        best = 0.0
        for rep in range(4):   # first launch warms the clocks
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
            _ffi.check(_ffi.lib.sk_mfma_probe(_ffi.ptr(scratch), scratch.numel() * 4, 20000, vary, C.byref(fl), st))
            e1.record(torch.cuda.current_stream(dev))
            e1.synchronize()
            if rep:
                best = max(best, fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        rates.append(best)
    best = rates[0]
    # memory side: a device-to-device copy of 1 GiB (2 GiB of traffic), best of three
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    hbm = 0.0
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
        dst.copy_(src)
        e1.record(torch.cuda.current_stream(dev))
        e1.synchronize()
        if rep:
            hbm = max(hbm, 2.0 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst
    # the same copy over 16 GiB (the split precision's full-resolution tensors are 14.7 GB each): a box whose large
    # allocations are mapped with small page fragments streams them slower than the 1 GiB figure says
    hbm_big = 0.0
    try:
        src = torch.empty((8, 1 << 29), dtype=torch.float32, device=dev)   # 16 GiB
        dst = torch.empty_like(src)
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
            for i in range(8):   # 2 GiB per copy call
                dst[i].copy_(src[i])
            e1.record(torch.cuda.current_stream(dev))
            e1.synchronize()
            if rep:
                hbm_big = max(hbm_big, 2.0 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        del src, dst
    except RuntimeError as e:   # a smaller device
        log(f"16 GiB copy probe skipped: {e}")
        hbm_big = None
    torch.cuda.empty_cache()
    nominal = 1950.0   # the rotating-operand loop on the boxes of round 3 (1 975; round 1's register loop on random data: 1 900)
    r = rates[1] / nominal
    return {"mfma_probe_tflops": round(best, 1), "mfma_probe_varying_operands_tflops": round(rates[1], 1),
            "probe": "bare v_mfma_f32_16x16x32_f16 loop, 2 waves per SIMD, every CU (sk_mfma_probe): one operand pair | four x four "
                     "pseudo-random fragments in rotation (the clock a kernel on real data gets)",
            "vs_nominal_1950": round(r, 3), "band": "slow" if r < 0.97 else ("fast" if r > 1.03 else "typical"),
            "hbm_copy_GBps": round(hbm, 1), "hbm_probe": "torch device-to-device copy of 1 GiB, read + write bytes / time",
            "hbm_copy_16GiB_GBps": None if hbm_big is None else round(hbm_big, 1),
            "device": torch.cuda.get_device_name(dev)}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------------
# launcher: --gpus N without torchrun
# ---------------------------------------------------------------------------------------------------
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv) -> int:
    """One fresh child process per device (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torchrun would);
    rank 0's stdout (the JSON line) is relayed, every rank's stderr passes through.  Nothing here touches
    the GPU: ``device_count()`` does not initialise it."""
    n = args.gpus
    rehearsal = os.environ.get("SKOOTS_DIST_BACKEND") == "gloo"   # ranks share a device, host-staged collectives
    if not args.launcher_dry_run:
        have = torch.cuda.device_count()
        if have < (1 if rehearsal else n):
            print(f"bench.py: --gpus {n} requested but this node exposes {have} device(s); refusing to print a "
                  f"{have}-GPU line for a {n}-GPU request", file=sys.stderr)
            return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    for ln in out0.decode().splitlines():   # stdout carries the JSON line only; library chatter goes to stderr
        print(ln, file=sys.stdout if ln.lstrip().startswith("{") else sys.stderr, flush=True)
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        print(f"bench.py: ranks failed: {bad}", file=sys.stderr)
        return 1
    return 0


def launcher_dry_run(rank: int, world: int) -> None:
    """CPU rehearsal of the launch path (tests/test_parallel_cpu.py): the spawned ranks rendezvous over gloo,
    all-reduce their ranks and rank 0 prints one JSON line."""
    dist.init_process_group("gloo")
    t = torch.tensor([rank], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launcher": "ok", "n_gpus": world, "rank_sum": int(t.item()),
                          "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend()}), flush=True)
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------
def parity_vs_fp32_mode(model, image, origins, eff, mean, std, out_box, precision):
    """The benched precision against the exact-fp32 MFMA mode (pinned to the fp32 oracle at 1e-5 by
    tests/test_hip_unet.py and tests/test_hip_geometry.py) on one production tile batch, over the box
    of every tile that the pipeline consumes.  Measured live, outside the timed region."""
    model.precision = precision
    got = model.forward_tiles(image, origins, eff, mean, std, out_box=out_box).float()
    model.precision = "fp32"
    want = model.forward_tiles(image, origins, eff, mean, std)
    model.precision = precision
    (x0, y0, z0), (x1, y1, z1) = out_box
    e = (got[:, :, x0:x1, y0:y1, z0:z1] - want[:, :, x0:x1, y0:y1, z0:z1]).abs()
    return {"max_abs": float(f"{e.max().item():.3e}"), "rms": float(f"{e.pow(2).mean().sqrt().item():.3e}"),
            "tiles": len(origins), "reference": "precision='fp32' (exact-fp32 MFMA; 1e-5 from the fp32 CPU oracle)",
            "north_star_tolerance": 1e-3}


def conv_c1_measure(dev, iters=5, warmup=2, tile=(512, 512, 128)):
    """BASELINE configs[1]: ONE 512x512x128 fp16 tile through the conv + GroupNorm/SiLU stack (stem, every conv, the
    GroupNorm passes, heads -- no tiling, no assignment; the reference's `model(crop)`, skoots/lib/eval.py:142-143, on
    one crop of that size).  ``roofline``: the 3x3x3 MFMA conv launches (HIP events around each) against the dense fp16
    peak, per layer and for the encoder alone (north_star's >= 40 % target is stated for the conv encoder)."""
    from skoots_amd import unet
    model = unet.smoke_model(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    vol = torch.randint(0, 256, tile, generator=g, device=dev, dtype=torch.uint8).to(torch.float16)
    org = [(0, 0, 0)]
    for _ in range(warmup):
        model.forward_tiles(vol, org, tile, 127.5, 73.9)
    torch.cuda.synchronize(dev)
    prof = unet.ConvProfile()
    model.profile = prof
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream(dev))
    for _ in range(iters):
        model.forward_tiles(vol, org, tile, 127.5, 73.9)
    e1.record(torch.cuda.current_stream(dev))
    e1.synchronize()
    model.profile = None
    fwd_ms = e0.elapsed_time(e1) / iters
    vox = tile[0] * tile[1] * tile[2]
    per = {}
    for a, b, fl, name in prof.named():
        d = per.setdefault(name, [0.0, 0.0, 0])
        d[0] += a.elapsed_time(b)
        d[1] += fl
        d[2] += 1
    layers = {k: {"avg_launch_ms": round(v[0] / v[2], 4), "tflops": round(v[1] / v[0] / 1e9, 1),
                  "frac": round(v[1] / v[0] / 1e9 / MFMA_PEAK_TFLOPS, 4)} for k, v in per.items()}
    conv_ms, conv_fl, n = prof.totals()
    enc = [v for k, v in per.items() if k.startswith(("enc", "mid"))]
    enc_ms, enc_fl = sum(v[0] for v in enc), sum(v[1] for v in enc)
    ach = conv_fl / conv_ms / 1e9
    out = {"metric": "one 512x512x128 fp16 tile, conv + GN/SiLU stack only (BASELINE configs[1])",
           "forward_ms": round(fwd_ms, 3), "value": round(vox / fwd_ms / 1e3, 1), "unit": "Mvoxels/s (tile voxels)",
           "iters": iters, "warmup": warmup, "dtype": "f16",
           "forward_tflops_algorithmic": round(model.flops_per_tile_voxel() * vox / fwd_ms / 1e9, 1),
           "roofline": {"bound": "mfma", "kernel": "all 3x3x3 MFMA conv launches of the tile (conv3_m16_kernel / conv3_kernel)",
                        "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None, "launches": n,
                        "conv3_ms_per_forward": round(conv_ms / iters, 3),
                        "encoder_frac": round(enc_fl / enc_ms / 1e9 / MFMA_PEAK_TFLOPS, 4) if enc_ms > 0 else None,
                        "layers": layers}}
    del model, vol
    torch.cuda.empty_cache()
    return out


def eval_main(args, rank, world, local):
    """The headline line.  At N = 1 the same process then times, on the same resident volume, the precision that
    meets north_star's 1e-3 tolerance (``also.split``) and one training leg (``also.train_bf16``, configs[4]) --
    the driver's command prints one line and that line carries all three (``--no-also`` skips them)."""
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: there is no CPU path")
    torch.cuda.set_device(local % ndev)
    dev = torch.device("cuda", local % ndev)
    backend = "none"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SKOOTS_DIST_BACKEND", "nccl")  # "gloo": single-GPU rehearsal of the N>1 path
        if backend == "nccl":
            if ndev < world:
                raise SystemExit(f"WORLD_SIZE={world} but only {ndev} device(s) visible: RCCL needs one device per rank")
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from skoots_amd import unet
    from skoots_amd.parallel import ShardedVolume, tile_plan
    from skoots_amd.profile import KernelProfile

    shape = tuple(int(v) for v in args.shape.split(",")) if args.shape else workload_shape(world)
    X, Y, Z = shape
    sv = ShardedVolume(shape, rank, world, dev)
    model = unet.smoke_model(dev)
    model.fold_upsample = not args.no_fold

    # ---- synthetic inputs, resident in HBM before the timed region -------------------
    zlo, zhi = sv.window  # local z-window (slab + halo)
    g = torch.Generator(device=dev).manual_seed(1234)
    image = torch.empty((X, Y, zhi - zlo), dtype=torch.float16, device=dev)
    for xa in range(0, X, 128):  # same seed + same call sequence on every rank -> same global volume
        blk = torch.randint(0, 256, (min(128, X - xa), Y, Z), generator=g, dtype=torch.uint8, device=dev)
        image[xa:xa + blk.shape[0]] = blk[:, :, zlo:zhi].to(torch.float16)
        del blk
    mean, std = 127.5, 73.9  # uniform[0,255] statistics ("dataset_mean/std" of the checkpoint, eval.py:87-88)
    inject_vol, n_blobs = (None, 0) if args.no_inject else device_blob_field(shape, (zlo, zhi), dev)

    # The injected field of every tile of this rank as a CONTIGUOUS (5, w, h, d) block -- the layout a network output
    # has -- copied once here, outside every timed region (625 tiles x 18 MB at N = 1).  (Rounds 1-3 handed the scatter
    # kernel strided 300x300x20 windows of the (5, X, Y, Z) field: 40-byte runs of 128-byte lines, 6.8x the bytes.)
    tile_cache, blocks = {}, None
    if inject_vol is not None and not args.strided_inject:
        tile_cache, blocks = contiguous_tile_blocks(inject_vol, shape, rank, world, zlo)

    def inject(out5, origin, eff):
        if inject_vol is None:
            return out5
        t = tile_cache.get(tuple(int(v) for v in origin))
        if t is not None:
            return t
        x, y, z = origin
        return inject_vol[:, x:x + eff[0], y:y + eff[1], z:z + eff[2]]  # strided view: no copy

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    probe = box_probe(dev) if rank == 0 else None

    def measure(precision, steps, warmup, streams):
        """W untimed + K timed steps of the whole path at ``precision``; returns rank 0's line (None elsewhere)."""
        model.precision = precision

        def step(prof=None, sprof=None, nstreams=1):
            return sv.run(image, model, SCALE, mean, std, tile_batch=args.tile_batch, inject=inject,
                          conv_profile=prof, streams=nstreams, stage_profile=sprof)

        log(f"[{precision}] volume {shape}, window {sv.window}, {n_blobs} blobs; warm-up x{warmup}")
        # Two tile batches in flight overlap their launches, which makes a per-launch duration meaningless: with
        # streams > 1 the `roofline` figures come from ONE extra single-stream step with the per-launch HIP events on --
        # every conv launch has the device to itself there -- run after the warm-up and outside the timed region.
        prof, sprof = unet.ConvProfile(), KernelProfile()
        res = None
        for i in range(warmup):
            res = step(None, None, streams)
            log(f"warm-up step done: {sv.timings}")
        roof_steps, roof_src = steps, "timed steps"
        if streams > 1:
            res = step(prof, sprof, 1)
            roof_steps, roof_src = 1, "extra single-stream step after the warm-up, untimed"
        sv.timings.clear()
        sv.comm._acct.clear()
        mstat0 = torch.cuda.memory_stats(dev)
        barrier()
        t0 = time.perf_counter()
        step_marks = []
        for _ in range(steps):
            res = step(prof, sprof, 1) if streams == 1 else step(None, None, streams)
            step_marks.append(time.perf_counter())   # sv.run ends with a device synchronisation (its stage timings)
        barrier()
        dt = time.perf_counter() - t0
        mstat1 = torch.cuda.memory_stats(dev)
        per_step_ms = [round((b_ - a_) * 1e3, 1) for a_, b_ in zip([t0] + step_marks[:-1], step_marks)]
        # device allocations inside the timed region: a hipMalloc / hipFree in the steady state would stall the launch queue
        alloc = {"device_mallocs": int(mstat1.get("num_device_alloc", 0) - mstat0.get("num_device_alloc", 0)),
                 "device_frees": int(mstat1.get("num_device_free", 0) - mstat0.get("num_device_free", 0)),
                 "alloc_retries": int(mstat1.get("num_alloc_retries", 0) - mstat0.get("num_alloc_retries", 0)),
                 "reserved_gib": round(mstat1.get("reserved_bytes.all.current", 0) / 2 ** 30, 1)}
        log(f"[{precision}] timed {steps} steps in {dt:.3f} s; per step {per_step_ms[:8]} ms; allocator {alloc}")
        comm_stats = sv.comm.stats()
        tiles = sv.tiles_this_rank
        if world > 1:
            cdev = dev if backend == "nccl" else torch.device("cpu")   # the gloo rehearsal reduces host tensors
            t = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            # per-rank tile counts and communication time (max over ranks: the slowest rank sets the step time)
            tl = torch.tensor([tiles], dtype=torch.int64, device=cdev)
            gathered = [torch.zeros_like(tl) for _ in range(world)]
            dist.all_gather(gathered, tl)
            tiles = [int(v.item()) for v in gathered]
            keys = sorted(comm_stats)
            cms = torch.tensor([comm_stats[k]["ms"] for k in keys], dtype=torch.float64, device=cdev)
            dist.all_reduce(cms, op=dist.ReduceOp.MAX)
            for k, v in zip(keys, cms.tolist()):
                comm_stats[k]["ms_max_over_ranks"] = round(v, 3)
        line = None
        if rank == 0:
            ms = dt / steps * 1e3
            voxels = X * Y * Z
            conv_ms, conv_flops, conv_launches = prof.totals()
            achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
            executed = prof.executed_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
            traffic, traffic_src = conv_hbm_traffic(args.tile_batch)
            per_layer = {}
            for e0, e1, fl, name in prof.named():
                d = per_layer.setdefault(name, [0.0, 0.0, 0])
                d[0] += e0.elapsed_time(e1)
                d[1] += fl
                d[2] += 1
            layers = {k: {"tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1), "avg_launch_ms": round(v[0] / v[2], 4)}
                      for k, v in per_layer.items() if v[0] > 0}
            line = {
                "metric": "Mvoxels/s end-to-end (3D U-Net fwd + instance assign)",
                "value": round(voxels / (dt / steps) / 1e6, 3), "unit": "Mvoxels/s",
                "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": {"fp16": "f16", "split": "f16 hi+lo pairs (3 MFMA products, f32 accumulate)",
                          "mix8": "f16 hi+lo pairs; 3x3x3 convs: f16 product + block-scaled fp8 (e4m3) correction product, f32 accumulate", "fp32": "f32"}[precision],
                "data": "synthetic",
                "config": {"workload": f"{X}x{Y}x{Z} fp16 volume, 300x300x20 tiles (margin 50,50,5), "
                                       f"U-Net dims [32,64,128,64,32] depths [2,2,2,2,2], N=10 follow, "
                                       f"Z-sharded x{world}", "precision": precision,
                           "tile_batch": args.tile_batch, "streams": streams, "fold_upsample": not args.no_fold,
                           "instances": int(res.get("n_instances", -1)), "blobs_injected": n_blobs,
                           "inject_layout": "none" if inject_vol is None else ("strided windows" if args.strided_inject else "contiguous per-tile blocks"),
                           "stage_ms": {k: round(v / steps * 1e3, 2) for k, v in sv.timings.items()},
                           "step_ms_min_max": [min(per_step_ms), max(per_step_ms)], "allocator_in_timed_steps": alloc},
                "roofline": {"bound": "mfma", "kernel": "conv3_px_kernel (enc0.1, dec0.1: the kernel with the most time) + conv3_upf_kernel (dec0.0, dec1.0) "
                                                          "+ conv3_kernel<64|128> (enc1.x, dec1.1, mid.x): all 3x3x3 MFMA conv launches" if precision == "fp16" else
                                                          "conv3_m16_kernel / conv3_kernel / conv3_upf_kernel, split variants (all 3x3x3 MFMA conv launches)",
                             "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic if precision == "fp16" else None,
                             # the two decoder convs over [skip, upsample(x)] run with the upsample folded into the weights
                             # (8 of 27 taps for the upsampled channels): the pipe executes fewer FLOPs than the algorithm counts
                             "executed": round(executed, 2), "frac_executed": round(executed / MFMA_PEAK_TFLOPS, 4),
                             "traffic_unit": f"bytes per launch of {args.tile_batch} tiles (PMC, profiles/{traffic_src})",
                             "launches": conv_launches, "avg_launch_ms": round(conv_ms / max(conv_launches, 1), 4),
                             "timed_over": f"{roof_steps} {roof_src} (HIP events around every conv launch, one stream)",
                             "flops_counted": "achieved: algorithmic 2*Cin*Cout*27 per output voxel, the conv as the reference states it "
                                              "(split mode issues 3x that on the MFMA pipe); executed: what the launches issue (fp16 mode: "
                                              "dec0.0 / dec1.0 fold the nearest-upsample into the weights, 35 of 54 tap-chunks); "
                                              "dec0.1's launches also contain the GroupNorm + SiLU of their input (fused, fp16 mode)",
                             "layers": layers},
            }
            if precision == "fp32":  # the MFMA conv profile only instruments the fp16 / split kernels
                line["roofline"] = None
            line.update(stage_rooflines(sprof))
            if world > 1:
                line["multi_gpu"] = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                                     "tiles_per_rank": tiles, "slab_planes": [b - a for a, b in sv.slabs],
                                     "comm_rank0_per_step": {k: {"bytes": v["bytes"] // steps,
                                                                 "ms": round(v["ms"] / steps, 3),
                                                                 "ms_max_over_ranks": round(v.get("ms_max_over_ranks", v["ms"]) / steps, 3),
                                                                 "calls": v["calls"] // steps}
                                                             for k, v in comm_stats.items()}}
        if not args.no_parity:
            # every rank runs it (same launches everywhere keeps the ranks in step); rank 0 reports
            plan, eff = tile_plan(shape, (300, 300, 20), (50, 50, 5), world)
            mine = plan[rank][:8]
            reach, ov = (3, 3, 1), (50, 50, 5)
            box = ([max(0, o - r) for o, r in zip(ov, reach)], [min(s, s - o + r) for s, o, r in zip(eff, ov, reach)])
            par = parity_vs_fp32_mode(model, image, [(x, y, z - zlo) for (x, y, z) in mine], eff, mean, std, box, precision)
            if rank == 0:
                line["parity_vs_fp32_mode"] = par
        return line

    line = measure(args.precision, args.steps, args.warmup, args.streams)
    also = {}
    if world == 1 and not args.no_also and args.precision == "fp16":
        # the precision that meets north_star's tolerance, same volume, same process
        sp = measure("split", args.also_steps, 1, args.streams)
        also["split"] = {k: sp[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "parity_vs_fp32_mode")
                         if k in sp}
        also["split"]["roofline"] = {k: sp["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "executed",
                                                                      "frac_executed", "launches", "avg_launch_ms", "timed_over")}
        also["split"]["roofline"]["note"] = ("achieved = ALGORITHMIC FLOPs (2*Cin*Cout*27) / conv time; the split mode issues three fp16 "
                                             "MFMA products per algorithmic product, `executed` counts them")
        also["split"]["stage_ms"] = sp["config"]["stage_ms"]
        also["split"]["step_ms_min_max"] = sp["config"]["step_ms_min_max"]
        also["split"]["allocator_in_timed_steps"] = sp["config"]["allocator_in_timed_steps"]
        # "split" with every 3x3x3 conv's two correction products as one block-scaled fp8 matrix product (sk_conv3d_mix8, sk_conv3d_upfold_mix8)
        mx = measure("mix8", args.also_steps, 1, args.streams)
        also["mix8"] = {k: mx[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "parity_vs_fp32_mode") if k in mx}
        also["mix8"]["layer_launch_ms"] = {k: [sp["roofline"]["layers"][k]["avg_launch_ms"], v["avg_launch_ms"]]
                                           for k, v in mx["roofline"]["layers"].items() if k in sp["roofline"]["layers"]}
        also["mix8"]["layer_launch_ms_note"] = "[split, mix8] per 3x3x3 conv launch of one tile batch, HIP events"
        also["mix8"]["stage_ms"] = mx["config"]["stage_ms"]
        also["mix8"]["roofline"] = {k: mx["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "executed",
                                                                    "frac_executed", "launches", "avg_launch_ms", "timed_over")}
        also["mix8"]["roofline"]["note"] = ("achieved = ALGORITHMIC FLOPs / conv time; executed = fp16-pass equivalents: per chunk one fp16 "
                                            "product + one block-scaled fp8 instruction stream that takes 10/9 (K = 128) or 1 (K = 64, "
                                            "folded taps) fp16-pass times")
        also["mix8"]["step_ms_min_max"] = mx["config"]["step_ms_min_max"]
    if rank == 0:
        line["box"] = probe
        if not args.no_cpu_baseline and world == 1:
            log("timing the CPU restatement (bounded sample)")
            line["cpu_baseline"] = cpu_baseline(args.cpu_budget)
    if world == 1 and not args.no_also and args.precision == "fp16":
        del sv, model, image, inject_vol
        tile_cache.clear()
        blocks = None
        torch.cuda.empty_cache()
        also["conv_c1"] = conv_c1_measure(dev)
        targs = argparse.Namespace(**vars(args))
        targs.precision, targs.steps, targs.warmup, targs.no_cpu_baseline = "bf16", args.also_train_steps, 2, True
        targs.shape = args.also_train_shape
        tr = train_measure(targs, rank, world, dev)
        also["train_bf16"] = {k: tr[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "roofline")}
        also["train_bf16"]["config"] = tr["config"]
    if rank == 0:
        if also:
            line["also"] = also
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------
# BASELINE configs[4]: one training step
# ---------------------------------------------------------------------------------------------------
def train_cpu_baseline(crop=(64, 64, 64), budget_s: float = 8.0):
    """oracle/train_step.py (torch autograd + torch.optim.AdamW on oracle/unet_spec.py, fp32) on a sub-crop."""
    from oracle import train_step as O
    from oracle import unet_spec
    threads = cpu_threads()
    torch.set_num_threads(threads)
    model = unet_spec.build().train()
    opt = O.make_optimizer(model)
    g = torch.Generator().manual_seed(0)
    X, Y, Z = crop
    images = torch.randn((1, 1, X, Y, Z), generator=g)
    masks = (torch.rand((1, 1, X, Y, Z), generator=g) > 0.5).float()
    skele = (torch.rand((1, 1, X, Y, Z), generator=g) > 0.9).float()
    baked = torch.rand((1, 3, X, Y, Z), generator=g) * 64
    sigma, scale = torch.tensor([20.0, 20.0, 20.0]), torch.tensor(SCALE)
    O.train_step(model, opt, images, masks, skele, baked, sigma, scale)   # warm-up, untimed
    t0 = time.perf_counter()
    n = 0
    while n < 2 or time.perf_counter() - t0 < budget_s:
        O.train_step(model, opt, images, masks, skele, baked, sigma, scale)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(X * Y * Z / dt / 1e6, 4), "unit": "Mvoxels/s", "cores": threads, "kind": "port",
            "sample": f"oracle/train_step.py (torch CPU fp32 autograd + AdamW, {threads} threads) on a {X}x{Y}x{Z} crop, "
                      f"batch 1: {dt:.2f} s per step over {n} steps"}


def train_main(args, rank, world, local):
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: there is no CPU path")
    torch.cuda.set_device(local % ndev)
    dev = torch.device("cuda", local % ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if ndev < world:
            raise SystemExit(f"WORLD_SIZE={world} but only {ndev} device(s) visible")
        dist.init_process_group("nccl", device_id=dev)
    line = train_measure(args, rank, world, dev)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def train_measure(args, rank, world, dev):
    """One training step of BASELINE configs[4] (skoots/train/engine.py:456-499: forward, three Tversky terms incl.
    the embedding loss, backward, AdamW) on a synthetic 256^3 crop, batch 1 per GPU, random-init U-Net; N > 1 =
    data-parallel replicas with one all-reduce of the flat gradient buffer per step (engine.py:113-115).
    Returns rank 0's line (None on the other ranks)."""
    from skoots_amd.profile import KernelProfile
    from skoots_amd.train import TrainStep, TrainUNet
    from skoots_amd.unet import random_state_dict
    X, Y, Z = (int(v) for v in args.shape.split(",")) if args.shape else (256, 256, 256)
    B = 1
    model = TrainUNet(random_state_dict(), dev, precision=args.precision)
    step = TrainStep(model, process_group=None if world > 1 else False)
    gen = torch.Generator(device=dev).manual_seed(1 + rank)
    images = torch.randn((B, 1, X, Y, Z), device=dev, generator=gen)
    gx = torch.arange(X, device=dev).view(X, 1, 1)
    gy = torch.arange(Y, device=dev).view(1, Y, 1)
    gz = torch.arange(Z, device=dev).view(1, 1, Z)
    cell = ((gx // 32) * 64 + (gy // 32) * 8 + gz // 32 + 1).float()
    inside = ((gx % 32 - 16) ** 2 + (gy % 32 - 16) ** 2 + (gz % 32 - 16) ** 2) < 12 ** 2
    masks = (cell * inside).expand(B, 1, X, Y, Z).contiguous()
    skele = (((gx % 32 - 16).abs() < 2) & ((gy % 32 - 16).abs() < 2) & ((gz % 32 - 16).abs() < 6)).float() \
        .expand(B, 1, X, Y, Z).contiguous()
    baked = torch.stack([(gx // 32 * 32 + 16).expand(X, Y, Z), (gy // 32 * 32 + 16).expand(X, Y, Z),
                         (gz // 32 * 32 + 16).expand(X, Y, Z)]).float().expand(B, 3, X, Y, Z).contiguous()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        losses = step(images, masks, skele, baked)
    barrier()
    prof = KernelProfile()
    model._L.profile = prof if model.fast16 else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step(images, masks, skele, baked)
    barrier()
    dt = time.perf_counter() - t0
    model._L.profile = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    line = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        line = {"metric": "training Mvoxels/s (U-Net fwd + Tversky/embedding loss + bwd + AdamW)",
                "value": round(world * B * X * Y * Z / (dt / args.steps) / 1e6, 3), "unit": "Mvoxels/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": {"bf16": "bf16", "mixed": "f16 operands (f32 master / accumulate)", "fp32": "f32"}[args.precision],
                "data": "synthetic",
                "config": {"workload": f"BASELINE configs[4]: {X}x{Y}x{Z} crop, batch {B} per GPU, random-init U-Net dims "
                                       f"[32,64,128,64,32], 3 Tversky terms (embedding sigma 20), AdamW; data-parallel x{world}",
                           "precision": args.precision, "steps_per_s": round(args.steps / dt * world, 4),
                           "losses": [round(float(v), 6) for v in losses.cpu()],
                           "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)}}
        tot = prof.totals()
        if tot:
            kms = sum(v[0] for v in tot.values())
            kfl = sum(v[1] for v in tot.values())
            ach = kfl / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
            line["roofline"] = {"bound": "mfma", "kernel": "all MFMA conv launches of the step: forward + data-gradient "
                                                           "(conv3_*/gather_gemm) and weight-gradient (wgrad16*) kernels",
                                "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                                "launches": sum(v[2] for v in tot.values()),
                                "ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in tot.items()},
                                "tflops": {k: round(v[1] / (v[0] * 1e-3) / 1e12, 1) for k, v in tot.items() if v[0] > 0}}
        if not args.no_cpu_baseline and world == 1:
            log("timing the CPU restatement of the training step (bounded sample)")
            line["cpu_baseline"] = train_cpu_baseline(budget_s=args.cpu_budget)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["eval", "train"], default="eval",
                    help="eval: BASELINE configs[2]/[3] (the headline metric); train: configs[4], one training step")
    ap.add_argument("--tile-batch", type=int, default=64, help="tiles per network launch (<= 64) and stream")
    ap.add_argument("--shape", type=str, default="", help="override X,Y,Z (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inject", action="store_true")
    ap.add_argument("--strided-inject", action="store_true",
                    help="rounds 1-3 behaviour: the injected field reaches the scatter kernel as strided windows of one (5, X, Y, Z) "
                         "array instead of per-tile contiguous blocks (what a network output looks like)")
    ap.add_argument("--no-parity", action="store_true", help="skip the live parity_vs_fp32_mode measurement")
    ap.add_argument("--streams", type=int, default=1,
                    help="tile batches in flight (HIP streams).  2 x 32 tiles measured +1.3 %% over 1 x 64 (the convs fill the register "
                         "file, so only launch tails overlap) -- not the default: overlapped launches make rocprofv3's per-kernel "
                         "durations of the same command incomparable with `roofline`, which is then taken in one extra single-stream step")
    ap.add_argument("--no-fold", action="store_true", help="A/B: decoder convs on the direct kernels instead of sk_conv3d_upfold")
    ap.add_argument("--precision", choices=["fp16", "split", "mix8", "fp32", "bf16", "mixed"], default=None,
                    help="eval: fp16 (default) | split (<= 1e-3 vs fp32) | fp32; train: bf16 (default) | mixed (fp16) | fp32")
    ap.add_argument("--no-also", action="store_true",
                    help="N = 1 fp16 eval line only: skip the `also` legs (split precision on the same volume, one bf16 training leg)")
    ap.add_argument("--also-steps", type=int, default=3, help="timed steps of the split-precision leg")
    ap.add_argument("--also-train-steps", type=int, default=5, help="timed steps of the training leg")
    ap.add_argument("--also-train-shape", type=str, default="", help="X,Y,Z of the training leg's crop (default 256,256,256)")
    ap.add_argument("--launcher-dry-run", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=8.0, help="seconds of stage-1 CPU work in the cpu_baseline sample")
    args = ap.parse_args()
    if args.precision is None:
        args.precision = "fp16" if args.config == "eval" else "bf16"
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # launched bare (python bench.py --gpus N): start the N ranks ourselves, before any GPU call
        raise SystemExit(spawn_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line for a different GPU count")
    if args.launcher_dry_run:
        return launcher_dry_run(rank, world)
    if args.config == "train":
        if args.precision not in ("bf16", "mixed", "fp32"):
            raise SystemExit(f"--precision {args.precision} is an eval precision; train takes bf16 | mixed | fp32")
        return train_main(args, rank, world, local)
    if args.precision not in ("fp16", "split", "mix8", "fp32"):
        raise SystemExit(f"--precision {args.precision} is a training precision; eval takes fp16 | split | mix8 | fp32")
    eval_main(args, rank, world, local)


if __name__ == "__main__":
    main()
