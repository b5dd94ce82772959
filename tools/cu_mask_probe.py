#!/usr/bin/env python3
"""What a CU-masked HIP stream gives on this device (hipExtStreamCreateWithCUMask through sk_stream_create_cu_mask):

  * which (XCC, SE, CU) the workgroups of a launch on the masked stream land on, for a few mask layouts -- how mask bit i
    maps to the chip;
  * the rate of an HBM stream (torch device-to-device copy, 4 GiB) on N compute units;
  * the time of the conv stack (tools/bench_conv.py's launches) on 256 - N compute units.

    python3 tools/cu_mask_probe.py > gpurun_out/cu_mask_probe.json
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def masked_stream(ffi, bits):
    words = (C.c_uint32 * 8)(*[sum(1 << (b - 32 * w) for b in bits if 32 * w <= b < 32 * w + 32) for w in range(8)])
    h = C.c_void_p()
    ffi.check(ffi.lib.sk_stream_create_cu_mask(words, 8, C.byref(h)))
    return torch.cuda.ExternalStream(h.value), h


def where(ffi, dev, stream, n=2048):
    out = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        ffi.check(ffi.lib.sk_debug_where(ffi.ptr(out), n, 200000, ffi.stream_ptr(dev)))
    stream.synchronize()
    o = out.cpu().view(n, 2).numpy().astype("uint32")
    xcc = o[:, 0] & 0xF
    hw = o[:, 1]
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
    places = sorted({(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(xcc, se, sh, cu)})
    per_xcc = {}
    for p in places:
        per_xcc[p[0]] = per_xcc.get(p[0], 0) + 1
    return {"distinct_cus": len(places), "per_xcc": per_xcc}


def copy_rate(dev, stream, gib=4):
    src = torch.empty(gib << 28, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    best = 0.0
    with torch.cuda.stream(stream):
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            dst.copy_(src)
            e1.record(stream)
            e1.synchronize()
            if rep:
                best = max(best, 2.0 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del src, dst
    return round(best, 1)


def conv_time(dev, stream, model, vol, origins, tile, box, iters=3):
    from skoots_amd import unet
    with torch.cuda.stream(stream):
        model.forward_tiles(vol, origins, tile, 127.5, 73.9, out_box=box)
        prof = unet.ConvProfile()
        model.profile = prof
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            model.forward_tiles(vol, origins, tile, 127.5, 73.9, out_box=box)
        e1.record(stream)
        e1.synchronize()
        model.profile = None
    conv_ms, _, _ = prof.totals()
    return {"forward_ms": round(e0.elapsed_time(e1) / iters, 3), "conv3_ms": round(conv_ms / iters, 3)}


def main():
    from skoots_amd import _ffi as ffi
    from skoots_amd import unet
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = {"layouts": {}, "copy_GBps": {}, "conv": {}}
    allb = list(range(256))
    layouts = {
        "all": allb,
        "bits_0_31": list(range(32)),
        "bits_0_63": list(range(64)),
        "every_8th": list(range(0, 256, 8)),
        "every_4th": list(range(0, 256, 4)),
        "low4_of_each_32": [b for b in allb if b % 32 < 4],
        "low8_of_each_32": [b for b in allb if b % 32 < 8],
    }
    streams = {}
    for name, bits in layouts.items():
        try:
            s, h = masked_stream(ffi, bits)
        except Exception as e:  # noqa: BLE001
            out["layouts"][name] = {"error": str(e)}
            continue
        streams[name] = (s, h, bits)
        out["layouts"][name] = dict(where(ffi, dev, s), bits=len(bits))
        out["copy_GBps"][name] = copy_rate(dev, s)
        print(name, out["layouts"][name], out["copy_GBps"][name], file=sys.stderr, flush=True)
    # conv stack on the complement of the pass masks (and on everything)
    model = unet.smoke_model(dev)
    tile = (300, 300, 20)
    g = torch.Generator(device=dev).manual_seed(0)
    B = 64
    vol = torch.randint(0, 256, (tile[0], tile[1], tile[2] + B - 1), generator=g, device=dev, dtype=torch.uint8).to(torch.float16)
    origins = [(0, 0, b) for b in range(B)]
    box = ([47, 47, 4], [253, 253, 16])
    for name in ("all", "every_8th", "every_4th", "low4_of_each_32", "low8_of_each_32", "bits_0_31", "bits_0_63"):
        if name not in streams:
            continue
        bits = streams[name][2]
        comp = allb if name == "all" else [b for b in allb if b not in set(bits)]
        try:
            s, h = masked_stream(ffi, comp)
        except Exception as e:  # noqa: BLE001
            out["conv"][name] = {"error": str(e)}
            continue
        out["conv"]["complement_of_" + name] = dict(conv_time(dev, s, model, vol, origins, tile, box), cus=len(comp))
        print("conv on complement of", name, out["conv"]["complement_of_" + name], file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
