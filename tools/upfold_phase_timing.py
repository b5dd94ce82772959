#!/usr/bin/env python3
"""Where do the cycles of a folded decoder conv workgroup go (csrc/conv3d_up.hip)?  -DSK_TIMING build
(make -C skoots_amd/csrc timing): per wave, cycles between the marks of a step's phases, mean over 4096 workgroups.

    SKOOTS_HIP_LIB=skoots_amd/libskoots_hip_timing.so python tools/upfold_phase_timing.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

NAMES = ["skip_mfma", "barrier", "dma_low_issue", "landing_wait", "barrier", "up_mfma", "barrier", "dma_fine_issue",
         "epilogue", "landing_wait2", "closing_barrier"]


def main():
    from skoots_amd import _ffi, unet
    dev = torch.device("cuda", 0)
    B, ext = int(os.environ.get("BATCH", "8")), (300, 300, 20)
    dbg = torch.zeros((4096, 4, 16), dtype=torch.int64, device=dev)
    _ffi.check(_ffi.lib.sk_debug_set_timing_buffer(_ffi.ptr(dbg), dbg.numel() * 8))
    g = torch.Generator(device=dev).manual_seed(1)
    skip = torch.randn((B,) + ext + (32,), generator=g, device=dev).half()
    up = torch.randn((B,) + tuple(e // 2 for e in ext) + (32,), generator=g, device=dev).half()
    w = torch.randn((32, 64, 3, 3, 3)) / (64 * 27) ** 0.5
    wp = unet.pack_conv_weight_upfold(w, 32, dev)
    bias = torch.zeros(32, device=dev)
    sf, uf = skip.float().cpu(), up.float().cpu()
    wps = unet.pack_conv_weight_upfold(w, 32, dev, split=True)
    wpm, wexp = unet.pack_conv_weight_upfold_mix8(w, 32, dev)
    ss, us = unet.split_pair(sf).to(dev), unet.split_pair(uf).to(dev)
    sm, um = unet.mix8_of(sf).to(dev), unet.mix8_of(uf).to(dev)
    runs = (("fp16", lambda: unet.conv3d_upfold(skip, up, wp, bias, 32)),
            ("split", lambda: unet.conv3d_upfold(ss, us, wps, bias, 32, split=True)),
            ("mix8", lambda: unet.conv3d_upfold_mix8(sm, um, wpm, wexp, bias, 32)))
    for name, fn in runs:
        for _ in range(2):
            dbg.zero_()
            fn()
            torch.cuda.synchronize()
        d = dbg.double()
        used = d.sum(dim=(1, 2)) > 0
        m = d[used].mean(dim=(0, 1))
        tot = m.sum().item()
        print(name, json.dumps({"cycles_per_wave": round(tot), **{f"{i}_{n}": round(v / tot, 4) for i, (n, v) in enumerate(zip(NAMES, m.tolist()))}}), flush=True)
    _ffi.check(_ffi.lib.sk_debug_set_timing_buffer(None, 0))   # detach before `dbg` can be freed


if __name__ == "__main__":
    main()
