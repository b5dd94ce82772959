#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel of one source file, from hipcc's -Rpass-analysis remarks:

    python tools/kernel_resources.py skoots_amd/csrc/conv3d.hip [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

src = sys.argv[1]
flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage"]
r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + sys.argv[2:] + ["-c", src, "-o", "/tmp/_kres.o"],
                   capture_output=True, text=True)
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
if not blocks:
    sys.stderr.write(r.stderr[-2000:])
    raise SystemExit(1)


def field(b, key):
    m = re.search(re.escape(key) + r": (\d+)", b)
    return m.group(1) if m else "?"


for b in blocks:
    name = b.split("\n")[0].strip()
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn.replace("(anonymous namespace)::", "").replace("void ", ""))
    print("%-60s VGPR %4s AGPR %4s spill %3s scratch %5s LDS %6s occ %s" % (
        dn[:60], field(b, "VGPRs"), field(b, "AGPRs"), field(b, "VGPRs Spill"), field(b, "ScratchSize [bytes/lane]"),
        field(b, "LDS Size [bytes/block]"), field(b, "Occupancy [waves/SIMD]")))
