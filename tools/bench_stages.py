#!/usr/bin/env python3
"""Stages 1b-3 alone at BASELINE configs[2] (1024x1024x256): gate / dilate / scatter of an injected blob field, skeleton
labelling, follow + assign, renumber -- no network -- with HIP-event timings per stage.  What tools/pmc_stages.sh runs
its FETCH_SIZE / WRITE_SIZE passes over, and (--dense) the worst-case field of the follow kernel.

    python tools/bench_stages.py --iters 3
    python tools/bench_stages.py --dense
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="1024,1024,256")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dense", action="store_true", help="every voxel gated in with a non-zero vector: nine real hops each")
    ap.add_argument("--strided", action="store_true", help="inject strided windows of the (5, X, Y, Z) field instead of contiguous tiles")
    args = ap.parse_args()
    import bench
    from skoots_amd.parallel import ShardedVolume
    from skoots_amd.profile import KernelProfile
    dev = torch.device("cuda", 0)
    shape = tuple(int(v) for v in args.shape.split(","))
    field, _ = bench.device_blob_field(shape, (0, shape[2]), dev, dense=args.dense)
    image = torch.zeros(shape, dtype=torch.float16, device=dev)

    # per-tile contiguous blocks (what a network output looks like) unless --strided (rounds 1-3: windows of the field)
    cache, _blocks = ({}, None) if args.strided else bench.contiguous_tile_blocks(field, shape, 0, 1, 0)

    def inject(_, origin, eff):
        t = cache.get(tuple(int(v) for v in origin))
        if t is not None:
            return t
        x, y, z = origin
        return field[:, x:x + eff[0], y:y + eff[1], z:z + eff[2]]

    sv = ShardedVolume(shape, 0, 1, dev)
    for _ in range(args.warmup):
        res = sv.run(image, None, bench.SCALE, 0.0, 1.0, inject=inject)
    sv.timings.clear()
    prof = KernelProfile()
    for _ in range(args.iters):
        res = sv.run(image, None, bench.SCALE, 0.0, 1.0, inject=inject, stage_profile=prof)
    vox = shape[0] * shape[1] * shape[2]
    out = {"shape": shape, "dense": args.dense, "inject_layout": "strided windows" if args.strided else "contiguous per-tile blocks",
           "iters": args.iters, "instances": int(res["n_instances"]),
           "foreground_frac": round(float((res["instance_mask"] > 0).float().mean()), 4),
           "skeleton_frac": round(float((res["skeleton"] > 0).float().mean()), 5),
           "stage_ms": {k: round(v / args.iters * 1e3, 3) for k, v in sv.timings.items()}, "kernels": {}}
    for name, (ms, work, n) in prof.totals().items():
        out["kernels"][name] = {"ms_per_step": round(ms / args.iters, 4), "launches_per_step": n // args.iters,
                                "algorithmic_bytes_per_step": int(work / args.iters),
                                "algorithmic_GBps": round(work / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0}
    out["voxels"] = vox
    print(json.dumps(out))


if __name__ == "__main__":
    main()
