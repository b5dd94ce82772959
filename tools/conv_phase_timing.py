#!/usr/bin/env python3
"""Where do the cycles of a conv3 workgroup go?  Runs each 3x3x3 layer shape of the production tile batch on the
-DSK_TIMING build (make -C skoots_amd/csrc timing), whose kernels sum, per wave, the cycles between the marks of a phase:
MFMA loop | barrier | LDS-DMA issue (+ landing wait where not deferred) | epilogue | deferred landing wait | in-LDS
activation | closing barrier.  Prints the mean split over the first 4096 workgroups.

    SKOOTS_HIP_LIB=skoots_amd/libskoots_hip_timing.so python tools/conv_phase_timing.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

NAMES = ["mfma", "barrier1", "dma_issue", "epilogue", "landing_wait", "activation", "barrier2", "-"]
# conv3_px_kernel (the single-chunk 32 -> 32 layers at the production tile) uses the same eight slots as:
NAMES_PX = ["mfma", "closing_barrier", "dma_issue", "epilogue", "landing_wait", "activation", "workgroup_prologue", "end_plane_steps"]
LAYERS = [("32->32 (enc0.1 / dec0.1)", (300, 300, 20), [(32, 0)], 32, False),
          ("32->32 + fused activation (dec0.1)", (300, 300, 20), [(32, 0)], 32, True),
          ("32+32up->32 (dec0.0)", (300, 300, 20), [(32, 0), (32, 1)], 32, False),
          ("64->64 (enc1.*, dec1.1)", (150, 150, 10), [(64, 0)], 64, False),
          ("64+64up->64 (dec1.0)", (150, 150, 10), [(64, 0), (64, 1)], 64, False),
          ("128->128 (mid.*)", (75, 75, 5), [(128, 0)], 128, False)]


def main():
    from skoots_amd import _ffi, unet
    dev = torch.device("cuda", 0)
    B = int(os.environ.get("BATCH", "8"))
    dbg = torch.zeros((4096, 4, 16), dtype=torch.int64, device=dev)   # [workgroup][wave][slot]: the library's one record layout
    _ffi.check(_ffi.lib.sk_debug_set_timing_buffer(_ffi.ptr(dbg), dbg.numel() * 8))
    zeros = torch.zeros(4096, dtype=torch.uint8, device=dev)
    out = {}
    for name, ext, srcdef, cout, act in LAYERS:
        g = torch.Generator(device=dev).manual_seed(1)
        srcs = [(torch.randn((B,) + (tuple(s // 2 for s in ext) if up else ext) + (c,), generator=g, device=dev).half(), up)
                for c, up in srcdef]
        cin = sum(c for c, _ in srcdef)
        w = torch.randn((cout, cin, 3, 3, 3)) / (cin * 27) ** 0.5
        wp = unet.pack_conv_weight(w, dev)
        bias = torch.zeros(cout, device=dev)
        aff = torch.stack([torch.ones((B, srcdef[0][0])), torch.zeros((B, srcdef[0][0]))], dim=1).to(dev) if act else None
        res = torch.empty((B,) + ext + (cout,), dtype=torch.float16, device=dev)
        nblk = _ffi.lib.sk_conv3d_num_blocks(B, *ext, cout, 3)
        partial = torch.zeros((B, nblk, cout // 4, 2), device=dev)
        arr = (_ffi.ConvSrc * len(srcs))()
        for i, (t, up) in enumerate(srcs):
            arr[i].data, arr[i].affine, arr[i].c, arr[i].upsample = t.data_ptr(), (aff.data_ptr() if act and i == 0 else None), t.shape[-1], up
        for it in range(2):
            dbg.zero_()
            _ffi.check(_ffi.lib.sk_conv3d(arr, len(srcs), _ffi.ptr(wp), _ffi.ptr(bias), _ffi.ptr(res), B, *ext, cout, 3,
                                          _ffi.ptr(partial), _ffi.ptr(zeros), _ffi.stream_ptr(dev)))
            torch.cuda.synchronize()
        d = dbg.double()
        used = d.sum(dim=(1, 2)) > 0
        m = d[used].mean(dim=(0, 1))
        tot = m.sum().item()
        names = NAMES_PX if (cout == 32 and len(srcdef) == 1 and ext[2] == 20) else NAMES[:7]
        out[name] = {"cycles_per_wave": round(tot), **{n: round(v / tot, 4) for n, v in zip(names, m.tolist())}}
        print(name, json.dumps(out[name]), flush=True)
    # the 32 -> 32 layer of the tolerance-meeting precisions on conv3_m16_kernel: split (three fp16 phases) and mix8 (fp16 + fp8 phase)
    ext = (300, 300, 20)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((B,) + ext + (32,), generator=g, device=dev)
    w = torch.randn((32, 32, 3, 3, 3)) / (32 * 27) ** 0.5
    bias = torch.zeros(32, device=dev)
    xs, xm = unet.split_pair(x), unet.mix8_of(x.cpu()).to(dev)
    wps, (wpm, wexp) = unet.pack_conv_weight(w, dev, split=True), unet.pack_conv_weight_mix8(w, dev)
    for name, fn in (("32->32 split (enc0.1 / dec0.1)", lambda: unet.conv3d([(xs, 0)], wps, bias, 32, 3, ext, zeros, split=True)),
                     ("32->32 mix8 (enc0.1 / dec0.1)", lambda: unet.conv3d_mix8(xm, wpm, wexp, bias, ext, zeros))):
        for it in range(2):
            dbg.zero_()
            fn()
            torch.cuda.synchronize()
        d = dbg.double()
        used = d.sum(dim=(1, 2)) > 0
        m = d[used].mean(dim=(0, 1))
        tot = m.sum().item()
        out[name] = {"cycles_per_wave": round(tot), **{n: round(v / tot, 4) for n, v in zip(NAMES[:7], m.tolist())}}
        print(name, json.dumps(out[name]), flush=True)
    _ffi.check(_ffi.lib.sk_debug_set_timing_buffer(None, 0))   # detach before `dbg` can be freed
    print(json.dumps(out))


if __name__ == "__main__":
    main()
