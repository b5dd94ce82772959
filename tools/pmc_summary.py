#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes over tools/bench_conv.py (see tools/pmc_collect.sh).

    pmc_summary.py traffic <fetch_pass_dir> <write_pass_dir> "<command>"   -> HBM-side bytes per launch and kernel:
        FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads; Infinity-Cache
        hits are counted, not excluded), WRITE_SIZE (KiB) as is -- the source of bench.py's `roofline.traffic`
    pmc_summary.py sq <pass_dir> "<command>"                               -> per-kernel SQ counters + derived fractions
"""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(directory, counters):
    path = glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True)[0]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] in counters:
            tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: ({c: v / len(disp[k]) for c, v in tot[k].items()}, len(disp[k])) for k in tot}


def short(name):
    return re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))[:96]


def traffic(fdir, wdir, cmd):
    fetch = per_kernel(fdir, {"FETCH_SIZE"})
    write = per_kernel(wdir, {"WRITE_SIZE"})
    out = {"command": cmd,
           "note": "per-launch averages; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced "
                   "reads; Infinity-Cache hits are counted); WRITE_SIZE as is",
           "kernels": []}
    for k in sorted(fetch):
        f, n = fetch[k]
        w = write.get(k, ({"WRITE_SIZE": 0.0}, n))[0]["WRITE_SIZE"]
        out["kernels"].append({"kernel": short(k), "launches": n, "FETCH_SIZE_KB_per_launch_raw": round(f["FETCH_SIZE"], 1),
                               "fetch_MB_per_launch_x2_gfx950_correction": round(2 * f["FETCH_SIZE"] / 1024, 1),
                               "write_MB_per_launch": round(w / 1024, 1)})
    return out


def sq(sdir, cmd):
    names = {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
             "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"}
    data = per_kernel(sdir, names)
    out = {"command": cmd,
           "note": "per-launch averages, counters summed over the device (GRBM_GUI_ACTIVE over the 8 XCDs: cycles = / 8).  "
                   "mfma_busy_per_simd = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs); lds_active_per_cu = "
                   "SQ_LDS_IDX_ACTIVE / (cycles * 256 CUs); wave fractions are of SQ_WAVE_CYCLES (round 1's derivation)",
           "kernels": {}}
    for k, (c, n) in sorted(data.items()):
        g = c.get("GRBM_GUI_ACTIVE", 0.0)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        d = {"launches": n, **{kk: round(v) for kk, v in sorted(c.items())}}
        if g and wc:
            d["derived"] = {"mfma_busy_per_simd": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g / 8 * 1024), 4),
                            "waves_parked_frac": round(c.get("SQ_WAIT_ANY", 0) / wc, 4),
                            "issue_stalled_frac": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 4),
                            "issuing_frac": round(c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 4),
                            "lds_active_per_cu": round(c.get("SQ_LDS_IDX_ACTIVE", 0) / (g / 8 * 256), 4),
                            "lds_conflict_share": round(c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c.get("SQ_LDS_IDX_ACTIVE", 1), 1), 4)}
        out["kernels"][short(k)] = d
    return out


def main():
    if sys.argv[1] == "traffic":
        json.dump(traffic(sys.argv[2], sys.argv[3], sys.argv[4]), sys.stdout, indent=1)
    elif sys.argv[1] == "sq":
        json.dump(sq(sys.argv[2], sys.argv[3]), sys.stdout, indent=1)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
