#!/usr/bin/env python3
"""Weight-gradient kernels of the training step, layer by layer, at the configs[4] geometry (256^3 crop, batch 1):
HIP-event time and TFLOP/s of sk_train_conv_wgrad_f16 (bf16 twin with --bf16) per k = 3 layer.

    gpurun -- 'python3 tools/bench_wgrad.py'      SK_WGRAD_NOXMARCH=1 SKOOTS_HIP_LIB=.../libskoots_hip_tuning.so ...
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from skoots_amd import _ffi  # noqa: E402

LAYERS = [  # name, out spatial, [(c, up)], cout
    ("enc0.1", 256, [(32, 0)], 32),
    ("enc1.x", 128, [(64, 0)], 64),
    ("mid.x", 64, [(128, 0)], 128),
    ("dec1.0", 128, [(64, 0), (128, 1)], 64),
    ("dec1.1", 128, [(64, 0)], 64),
    ("dec0.0", 256, [(32, 0), (64, 1)], 32),
    ("dec0.1", 256, [(32, 0)], 32),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--scale", type=int, default=1, help="divide the spatial extents (smoke runs)")
    ap.add_argument("--zeros", action="store_true", help="all-zero operands (clock / data-dependence check)")
    ap.add_argument("--phases", action="store_true", help="-DSK_TIMING build (make timing; SKOOTS_HIP_LIB=.../libskoots_hip_timing.so): "
                    "where a wave of wgrad16x_kernel spends its cycles")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if a.bf16 else torch.float16
    fn = getattr(_ffi.lib, "sk_train_conv_wgrad_f16" + ("_bf16" if a.bf16 else ""))
    st = _ffi.stream_ptr(dev)
    zero_page = torch.zeros(4096, dtype=torch.uint8, device=dev)
    out = {}
    mk = torch.zeros if a.zeros else torch.randn
    dbg = None
    if a.phases:
        dbg = torch.zeros((4096, 4, 16), dtype=torch.int64, device=dev)
        _ffi.check(_ffi.lib.sk_debug_set_timing_buffer(_ffi.ptr(dbg), dbg.numel() * 8))
    names = ["groups_0_6", "landing_wait", "barrier", "groups_7_11", "dma_issue", "prologue", "epilogue", "lgkm_after_burst"]
    for name, n, srcdef, cout in LAYERS:
        n //= a.scale
        srcs = [(mk((1, n // 2, n // 2, n // 2, c) if up else (1, n, n, n, c), device=dev).to(dt), up) for c, up in srcdef]
        cin = sum(c for c, _ in srcdef)
        dy = mk((1, n, n, n, cout), device=dev).to(dt)
        scale = torch.tensor([1.0, 1.0, 0.0], device=dev)
        arr = (_ffi.ConvSrc * len(srcs))()
        for i, (t, up) in enumerate(srcs):
            arr[i].data, arr[i].affine, arr[i].c, arr[i].upsample = t.data_ptr(), None, t.shape[-1], up
        dw = torch.empty((cout, cin, 3, 3, 3), device=dev)
        db = torch.empty(cout, device=dev)
        ws = torch.empty(int(_ffi.lib.sk_train_conv_wgrad_workspace_floats(1, n, n, n, cout, cin, 3)), device=dev)

        def run():
            _ffi.check(fn(arr, len(srcs), _ffi.ptr(dy), _ffi.ptr(scale), 1, n, n, n, cout, 3, _ffi.ptr(dw), _ffi.ptr(db),
                          _ffi.ptr(ws), _ffi.ptr(zero_page), st))
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        fl = 2.0 * 27 * cin * cout * n ** 3
        out[name] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}
        if dbg is not None:
            dbg.zero_()
            run()
            torch.cuda.synchronize()
            d = dbg.double()
            m = d[d.sum(dim=(1, 2)) > 0].mean(dim=(0, 1))
            tot = m.sum().item()
            out[name]["phases"] = {k: round(v / tot, 4) for k, v in zip(names, m.tolist())}
            out[name]["slots"] = [round(v / tot, 4) for v in m.tolist()]   # -DSK_WX_TGROUP build: groups 0..11, issue, wait + barrier
            out[name]["cycles_per_wave"] = round(tot)
    if dbg is not None:
        _ffi.check(_ffi.lib.sk_debug_set_timing_buffer(None, 0))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
