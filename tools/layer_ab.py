#!/usr/bin/env python3
"""Per-layer A/B of conv kernels builds on ONE GPU box: runs tools/bench_conv.py once per library (a child process
each, SKOOTS_HIP_LIB), interleaved ROUNDS times so that clock drift hits every variant alike, and prints per library
the conv3 total and the per-layer milliseconds (HIP events around every 3x3x3 launch).

    gpurun -- 'python3 tools/layer_ab.py ab/lib_a.so ab/lib_b.so'
    TILE=512,512,128 BATCH=1 python3 tools/layer_ab.py ...
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    libs = sys.argv[1:]
    rounds = int(os.environ.get("ROUNDS", "2"))
    tile = os.environ.get("TILE", "300,300,20")
    batch = os.environ.get("BATCH", "8")
    for r in range(rounds):
        for lib in libs:
            extra, envx, label = [], {}, lib
            if lib.endswith(":nofold"):   # variant of a library: decoder convs on the direct kernel
                lib, extra = lib[:-len(":nofold")], ["--no-fold"]
            if lib.endswith(":nopx"):     # tuning build only: the single-chunk COUT-32 layers on conv3_m16_kernel
                lib, envx = lib[:-len(":nopx")], {"SK_CONV_NO_PX": "1"}
            path = lib if os.path.isabs(lib) else os.path.join(ROOT, lib)
            env = dict(os.environ, SKOOTS_HIP_LIB=path, **envx)
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_conv.py"), "--tile", tile, "--batch", batch,
                                "--iters", "6", "--warmup", "2"] + extra, env=env, capture_output=True, text=True)
            if p.returncode != 0:
                print(lib, "FAILED", p.stderr[-2000:], flush=True)
                continue
            d = json.loads(p.stdout.strip().splitlines()[-1])
            print(label, d["conv3_ms"], d["conv3_tflops"], "fwd", d["forward_ms"],
                  {k: v["ms"] for k, v in d["layers"].items()}, flush=True)


if __name__ == "__main__":
    main()
