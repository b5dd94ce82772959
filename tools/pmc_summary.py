#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the PMC slots require)
of tools/bench_conv.py into profiles/r01_conv_hbm_traffic_pmc.json -- the source of bench.py's
`roofline.traffic`.  FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced
reads); both counters are in KiB.

    python tools/pmc_summary.py <fetch_pass_dir> <write_pass_dir> <command string> > profiles/...json
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(directory, counter):
    path = glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True)[0]
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: (tot[k] / len(disp[k]), len(disp[k])) for k in tot}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"command": sys.argv[3],
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); WRITE_SIZE as is",
           "kernels": []}
    for k in sorted(fetch):
        f, n = fetch[k]
        w = write.get(k, (0.0, n))[0]
        out["kernels"].append({"kernel": k[:96], "launches": n, "FETCH_SIZE_KB_per_launch_raw": round(f, 1),
                               "fetch_MB_per_launch_x2_gfx950_correction": round(2 * f / 1024, 1),
                               "write_MB_per_launch": round(w / 1024, 1)})
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
