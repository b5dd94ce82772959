#!/usr/bin/env python3
"""Do kernels on two CU-masked streams with disjoint masks run CONCURRENTLY?  An MFMA loop (sk_mfma_probe) on the conv
share and a device-to-device copy on the hbm share: each alone, then both at once (wall clock around a synchronize)."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from skoots_amd import _ffi as ffi
    from cu_streams import RoleStreams
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = {}
    scratch = torch.empty(512 * 256, dtype=torch.float32, device=dev)
    src = torch.empty(1 << 30, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    fl = C.c_double(0.0)

    def mfma(stream, n):
        with torch.cuda.stream(stream):
            for _ in range(n):
                ffi.check(ffi.lib.sk_mfma_probe(ffi.ptr(scratch), scratch.numel() * 4, 20000, 1, C.byref(fl), ffi.stream_ptr(dev)))

    def copy(stream, n):
        with torch.cuda.stream(stream):
            for _ in range(n):
                dst.copy_(src)

    def wall(fn):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(dev)
        return round((time.perf_counter() - t0) * 1e3, 2)

    for hbm in (4, 8):
        rs = RoleStreams(dev, hbm)
        mfma(rs["conv"], 2)
        copy(rs["hbm"], 1)
        a = wall(lambda: mfma(rs["conv"], 20))
        b = wall(lambda: copy(rs["hbm"], 8))
        both = wall(lambda: (mfma(rs["conv"], 20), copy(rs["hbm"], 8)))
        out[f"masked_{hbm}"] = {"mfma_ms": a, "copy_ms": b, "both_ms": both}
        print(hbm, out[f"masked_{hbm}"], file=sys.stderr, flush=True)
    # two plain streams for comparison
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    a = wall(lambda: mfma(s1, 20))
    b = wall(lambda: copy(s2, 8))
    both = wall(lambda: (mfma(s1, 20), copy(s2, 8)))
    out["plain"] = {"mfma_ms": a, "copy_ms": b, "both_ms": both}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
