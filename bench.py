#!/usr/bin/env python3
"""bench.py -- Mvoxels/s end-to-end (3-D U-Net forward + instance assignment) on MI355X.

Metric / config: BASELINE.json.  A "step" is one pass of the whole hot path
(skoots/lib/eval.py:126-306: sliding-window U-Net, gate/dilate/scatter, skeleton
labelling, offset following + assignment, renumber) over one synthetic volume that is
already resident in HBM when the timed region starts.

  N = 1 : configs[2], 1024x1024x256 fp16 (the largest single-GPU end-to-end config)
  N > 1 : weak scaling, 2^28 voxels per GPU, Z-sharded: 2048x1024x256 (N=2),
          2048x2048x256 (N=4), 2048x2048x512 (N=8 = configs[3])

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
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

SCALE = (60, 60, 12)  # cfg.SKOOTS.VECTOR_SCALING default, skoots/config.py:144


def workload_shape(n_gpus: int):
    return {1: (1024, 1024, 256), 2: (2048, 1024, 256), 4: (2048, 2048, 256),
            8: (2048, 2048, 512)}.get(n_gpus, (1024, 1024, 256 * n_gpus))


def device_blob_field(shape, z_window, device, seed=0, pitch=(64, 64, 16)):
    """Blob field of tests/workload.py generated directly on the device for the local
    z-window: one ellipsoid per (64,64,16) lattice cell (jittered centre and radii), prob
    0.95 inside, vectors pointing at the centre / SCALE, skeleton ball at the centre.
    Returns (5, X, Y, zw) fp16 and the number of blobs in the whole volume."""
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
        o[0] = torch.where(inside, (dx / SCALE[0]).clamp(-1, 1), 0).half()
        o[1] = torch.where(inside, (dy / SCALE[1]).clamp(-1, 1), 0).half()
        o[2] = torch.where(inside, (dz / SCALE[2]).clamp(-1, 1), 0).half()
        o[3] = torch.where(core, 0.92, 0.0).half()
        o[4] = torch.where(inside, 0.95, 0.0).half()
    # cells whose blob centre lies inside the written frame produce an instance
    return out, nx * ny * nz


def cpu_threads() -> int:
    """Host threads for the CPU baseline: the GPU box grants 16 cores per GPU."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(budget_s: float = 20.0):
    """The CPU restatement (oracle/) timed on this host on a bounded sample of BASELINE
    configs[0]: the 128x128x32 volume with the same network shape -- the reference's own
    CPU-runnable case, where it evaluates 100 tiles (every clamped duplicate included).
    Stage 1 runs for at most ``budget_s`` seconds and is scaled to the 100 tiles; stages
    2-3 + renumber run in full."""
    from oracle import pipeline as O
    from oracle import unet_spec
    threads = cpu_threads()
    torch.set_num_threads(threads)
    model = unet_spec.build()
    g = torch.Generator().manual_seed(0)
    image = torch.randint(0, 256, (1, 128, 128, 32), generator=g).to(torch.float16)
    prog = {}
    with torch.no_grad():
        model(torch.zeros(1, 1, 128, 128, 20))  # warm the allocator / thread pool, untimed
        vectors, skeleton = O.stage1(image, model, image.mean(), image.std(), budget_s=budget_s, progress=prog)
        t1 = prog["seconds"] * prog["total"] / prog["done"]
        t0 = time.perf_counter()
        O.post_model(vectors, skeleton, SCALE)
        t23 = time.perf_counter() - t0
    dt = t1 + t23
    return {"value": round(128 * 128 * 32 / dt / 1e6, 5), "unit": "Mvoxels/s", "cores": threads,
            "kind": "port",
            "sample": f"configs[0] 128x128x32 fp32, torch CPU {threads} threads: stage 1 timed on {prog['done']} of "
                      f"{prog['total']} tiles (128x128x20) in {prog['seconds']:.1f} s and scaled to {t1:.1f} s; stages 2-3 + "
                      f"renumber in full {t23:.1f} s"}


def conv_hbm_traffic():
    """HBM bytes per conv3 launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    in separate runs of the same launch shapes, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note).
    bench.py cannot collect PMC counters itself; returns None when the profile is absent."""
    path = os.path.join(ROOT, "profiles", "r01_conv_hbm_traffic_pmc.json")
    try:
        prof = json.load(open(path))
    except OSError:
        return None
    tot, n = 0.0, 0
    for k in prof["kernels"]:
        if "conv3_kernel" in k["kernel"] or "conv3_m16_kernel" in k["kernel"]:
            tot += (k["fetch_MB_per_launch_x2_gfx950_correction"] + k["write_MB_per_launch"]) * k["launches"]
            n += k["launches"]
    return round(tot / n * 1024 * 1024) if n else None


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tile-batch", type=int, default=8)
    ap.add_argument("--shape", type=str, default="", help="override X,Y,Z (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-inject", action="store_true")
    ap.add_argument("--streams", type=int, default=1, help="tile batches in flight (HIP streams)")
    ap.add_argument("--precision", choices=["fp16", "fp32"], default="fp16",
                    help="fp32: every layer on the exact-fp32 matrix instruction (strict 1e-3 parity mode; not the headline)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local % ndev)
    dev = torch.device("cuda", local % ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SKOOTS_DIST_BACKEND", "nccl")  # "gloo": single-GPU rehearsal of the N>1 path
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from skoots_amd import unet
    from skoots_amd.parallel import ShardedVolume

    shape = tuple(int(v) for v in args.shape.split(",")) if args.shape else workload_shape(world)
    X, Y, Z = shape
    sv = ShardedVolume(shape, rank, world, dev)
    model = unet.smoke_model(dev)
    model.precision = args.precision

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

    def inject(out5, origin, eff):
        if inject_vol is None:
            return out5
        x, y, z = origin
        return inject_vol[:, x:x + eff[0], y:y + eff[1], z:z + eff[2]]  # strided view: no copy

    def step(prof=None):
        return sv.run(image, model, SCALE, mean, std, tile_batch=args.tile_batch, inject=inject,
                      conv_profile=prof, streams=args.streams)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    log(f"inputs resident: volume {shape}, window {sv.window}, {n_blobs} blobs; warm-up x{args.warmup}")
    for _ in range(args.warmup):
        res = step()
        log(f"warm-up step done: {sv.timings}")
    barrier()
    prof = unet.ConvProfile()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(prof)
    barrier()
    dt = time.perf_counter() - t0
    log(f"timed {args.steps} steps in {dt:.3f} s")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        voxels = X * Y * Z
        conv_ms, conv_flops, conv_launches = prof.totals()
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        line = {
            "metric": "Mvoxels/s end-to-end (3D U-Net fwd + instance assign)",
            "value": round(voxels / (dt / args.steps) / 1e6, 3), "unit": "Mvoxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if args.precision == "fp16" else "f32",
            "data": "synthetic",
            "config": {"workload": f"{X}x{Y}x{Z} fp16 volume, 300x300x20 tiles (margin 50,50,5), "
                                   f"U-Net dims [32,64,128,64,32] depths [2,2,2,2,2], N=10 follow, "
                                   f"Z-sharded x{world}", "tile_batch": args.tile_batch, "streams": args.streams,
                       "instances": int(res.get("n_instances", -1)), "blobs_injected": n_blobs,
                       "stage_ms": {k: round(v / (args.steps + args.warmup) * 1e3, 2)
                                    for k, v in sv.timings.items()}},
            "roofline": {"bound": "mfma", "kernel": "conv3_m16_kernel / conv3_kernel (all 3x3x3 MFMA conv launches)",
                         "achieved": round(achieved, 2), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(achieved / 2500.0, 4), "traffic": conv_hbm_traffic(),
                         "traffic_unit": "bytes per launch (PMC, profiles/r01_conv_hbm_traffic_pmc.json)",
                         "launches": conv_launches, "avg_launch_ms": round(conv_ms / max(conv_launches, 1), 4)},
        }
        if args.precision != "fp16":  # the MFMA conv profile only instruments the fp16 kernel
            line["roofline"] = None
            line["config"]["precision"] = args.precision
        if not args.no_cpu_baseline and world == 1:
            log("timing the CPU restatement (bounded sample)")
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
