#!/usr/bin/env python3
"""Times the training step of BASELINE.json configs[4]: random-init U-Net, synthetic 256^3 crop, batch 1.

    python tools/bench_train.py [--shape 256 256 256] [--steps 3] [--warmup 1]

Prints one JSON line: steps/s plus the split into forward / loss / backward / optimizer (HIP events on
the launch stream).  Not the driver's bench (bench.py measures the inference metric); this is the
secondary number quoted in DESIGN.md."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, nargs=3, default=[256, 256, 256])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--precision", choices=["fp32", "mixed", "bf16"], default="fp32")
    args = ap.parse_args()
    from skoots_amd.train import TrainStep, TrainUNet
    from skoots_amd.unet import random_state_dict
    dev = torch.device("cuda:0")
    X, Y, Z = args.shape
    B = args.batch
    model = TrainUNet(random_state_dict(), dev, precision=args.precision)
    step = TrainStep(model)
    gen = torch.Generator(device=dev).manual_seed(1)
    images = torch.randn((B, 1, X, Y, Z), device=dev, generator=gen)
    gx = torch.arange(X, device=dev).view(X, 1, 1)
    gy = torch.arange(Y, device=dev).view(1, Y, 1)
    gz = torch.arange(Z, device=dev).view(1, 1, Z)
    cell = ((gx // 32) * 64 + (gy // 32) * 8 + gz // 32 + 1).float()
    inside = ((gx % 32 - 16) ** 2 + (gy % 32 - 16) ** 2 + (gz % 32 - 16) ** 2) < 12 ** 2
    masks = (cell * inside).expand(B, 1, X, Y, Z).contiguous()
    skele = (((gx % 32 - 16).abs() < 2) & ((gy % 32 - 16).abs() < 2) & ((gz % 32 - 16).abs() < 6)).float().expand(B, 1, X, Y, Z).contiguous()
    baked = torch.stack([(gx // 32 * 32 + 16).expand(X, Y, Z), (gy // 32 * 32 + 16).expand(X, Y, Z),
                         (gz // 32 * 32 + 16).expand(X, Y, Z)]).float().expand(B, 3, X, Y, Z).contiguous()
    sigma = [20.0, 20.0, 20.0]

    def one(record=None):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        logits = model.forward(images)
        ev[1].record()
        losses, dl = step.fused_loss(logits, masks, skele, baked, sigma)
        ev[2].record()
        model.backward(dl)
        ev[3].record()
        step.optimizer_step()
        ev[4].record()
        if record is not None:
            record.append(ev)
        return losses

    for _ in range(args.warmup):
        one()
    torch.cuda.synchronize()
    rec = []
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(args.steps):
        losses = one(rec)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / args.steps
    phases = [sum(e[i].elapsed_time(e[i + 1]) for e in rec) / len(rec) for i in range(4)]
    print(json.dumps({"metric": "train_steps_per_s", "value": round(1000.0 / ms, 4), "unit": "steps/s",
                      "ms_per_step": round(ms, 2), "dtype": {"fp32": "f32", "mixed": "f16 operands / f32 accumulate + master", "bf16": "bf16 operands / f32 accumulate + master"}[args.precision], "data": "synthetic",
                      "config": {"workload": f"{X}x{Y}x{Z} crop, batch {B}, random-init U-Net, 3 Tversky terms, AdamW"},
                      "phase_ms": {"forward": round(phases[0], 2), "loss": round(phases[1], 2),
                                   "backward": round(phases[2], 2), "optimizer": round(phases[3], 2)},
                      "losses": [round(float(v), 6) for v in losses.cpu()],
                      "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
