#!/usr/bin/env python3
"""Conv-stack-only benchmark (BASELINE.json configs[1]: one 512x512x128 fp16 tile through
the 3-D conv + GroupNorm/SiLU kernels, no tiling, no assignment) and per-layer timing of
the production tile shape.  Prints per-layer TFLOP/s of every 3x3x3 MFMA conv launch and
the whole-forward figure against the 2.5 PFLOP/s dense fp16 peak.

    python tools/bench_conv.py --tile 512,512,128 --batch 1
    python tools/bench_conv.py --tile 300,300,20 --batch 4
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", default="512,512,128")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-box", action="store_true", help="evaluate the heads on the whole tile (the pipeline evaluates the scatter's box: "
                    "interior +- the dilation reach, 28 %% of a 300x300x20 tile)")
    ap.add_argument("--one-pass-stem", action="store_true", help="A/B: sk_conv3d_stem_raw + in-LDS activation in enc0.1 (HipUNet.stem_single_pass)")
    ap.add_argument("--precision", default="fp16", choices=["fp16", "split", "mix8"])
    ap.add_argument("--no-fold", action="store_true", help="decoder convs on the direct kernel (sk_conv3d) instead of sk_conv3d_upfold")
    args = ap.parse_args()
    from skoots_amd import unet
    dev = torch.device("cuda", 0)
    tile = tuple(int(v) for v in args.tile.split(","))
    model = unet.HipUNet(unet.random_state_dict(), dev, precision=args.precision)
    model.fold_upsample = not args.no_fold
    model.stem_single_pass = args.one_pass_stem
    g = torch.Generator(device=dev).manual_seed(0)
    vol = torch.randint(0, 256, (tile[0], tile[1], tile[2] + args.batch - 1), generator=g, device=dev,
                        dtype=torch.uint8).to(torch.float16)
    origins = [(0, 0, b) for b in range(args.batch)]
    box = None
    if not args.no_box and all(t > 2 * o for t, o in zip(tile, (50, 50, 5))):   # as skoots_amd.parallel.ShardedVolume.run
        box = ([o - r for o, r in zip((50, 50, 5), (3, 3, 1))], [t - o + r for t, o, r in zip(tile, (50, 50, 5), (3, 3, 1))])
    for _ in range(args.warmup):
        model.forward_tiles(vol, origins, tile, 127.5, 73.9, out_box=box)
    torch.cuda.synchronize()
    prof = unet.ConvProfile()
    model.profile = prof
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        model.forward_tiles(vol, origins, tile, 127.5, 73.9, out_box=box)
    e1.record()
    torch.cuda.synchronize()
    model.profile = None
    total_ms = e0.elapsed_time(e1) / args.iters
    vox = args.batch * tile[0] * tile[1] * tile[2]
    per = {}
    for a, b, fl, name in prof.named():
        d = per.setdefault(name, [0.0, 0.0, 0])
        d[0] += a.elapsed_time(b)
        d[1] += fl
        d[2] += 1
    layers = {k: {"ms": round(v[0] / v[2], 4), "tflops": round(v[1] / v[0] / 1e9, 1)} for k, v in per.items()}
    conv_ms, conv_fl, n = prof.totals()
    out = {"tile": tile, "batch": args.batch, "out_box": box, "forward_ms": round(total_ms, 3),
           "forward_tflops": round(model.flops_per_tile_voxel() * vox / total_ms / 1e9, 1),
           "mvox_per_s_tile_voxels": round(vox / total_ms / 1e3, 1),
           "conv3_ms": round(conv_ms / args.iters, 3), "conv3_tflops": round(conv_fl / conv_ms / 1e9, 1),
           "conv3_frac_of_2500": round(conv_fl / conv_ms / 1e9 / 2500.0, 4), "layers": layers}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
