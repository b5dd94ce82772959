#!/usr/bin/env python3
"""Are two CU-masked streams with the SAME mask independent hardware queues?  A tiny spinning kernel (8 workgroups, ~20 ms)
on each of several streams at once: wall ~ one spin = concurrent, ~ the sum = serialized."""
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
    out = torch.zeros(64, dtype=torch.int32, device=dev)
    spin = 40_000_000   # ~20 ms at 2 GHz

    def run(streams):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for s in streams:
            with torch.cuda.stream(s):
                ffi.check(ffi.lib.sk_debug_where(ffi.ptr(out), 8, spin, ffi.stream_ptr(dev)))
        torch.cuda.synchronize(dev)
        return round((time.perf_counter() - t0) * 1e3, 1)

    a, b = RoleStreams(dev, 6), RoleStreams(dev, 6)
    p1, p2, p3, p4 = (torch.cuda.Stream(dev) for _ in range(4))
    res = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
           "one": run([a["conv"]]),
           "same_mask_x2": run([a["conv"], b["conv"]]),
           "two_masks": run([a["conv"], a["hbm"]]),
           "four_masked": run([a["conv"], a["hbm"], b["conv"], b["hbm"]]),
           "plain_x2": run([p1, p2]), "plain_x4": run([p1, p2, p3, p4])}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
