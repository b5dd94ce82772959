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
    m = re.search(r"--batch (\d+)", cmd)
    out = {"command": cmd, "batch": int(m.group(1)) if m else None,
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


STAGE_KERNELS = {"gate_dilate_scatter": ("gate_dilate_scatter_kernel",),
                 "ccl": ("ccl_", "ccl16_"),
                 "follow_assign": ("follow_assign_kernel", "follow_assign2_kernel"),
                 "renumber": ("first_seen_kernel", "mark_kernel", "chunk_popc_kernel", "chunk_scan_kernel", "word_prefix_kernel",
                              "build_lut_kernel", "apply_lut_kernel")}


def stages(fdir, wdir, fdir_dense, wdir_dense, sparse_json, dense_json, cmd):
    """HBM-side bytes per STEP of every stage-2/3 kernel group (a step = one pass of tools/bench_stages.py over the
    1024x1024x256 volume): sum over the group's launches of FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE."""
    out = {"command": cmd,
           "note": "bytes per step = sum over the stage's kernel launches of one pass; FETCH_SIZE (KiB) doubled per "
                   "MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at half; Infinity-Cache hits are counted), WRITE_SIZE "
                   "(KiB) as is.  The 4-byte-per-lane accesses of the labelling kernels are outside the guide's calibration "
                   "(16 B per lane): read their absolute figures with that caveat, ratios between variants are unaffected",
           "stages": {}, "stages_dense_field": {}}
    for key, fd, wd, tj in (("stages", fdir, wdir, sparse_json), ("stages_dense_field", fdir_dense, wdir_dense, dense_json)):
        timing = json.load(open(tj))
        fetch = per_kernel(fd, {"FETCH_SIZE"})
        write = per_kernel(wd, {"WRITE_SIZE"})
        out[key + "_field"] = {k: timing[k] for k in ("shape", "dense", "instances", "foreground_frac", "skeleton_frac")}
        for stage, pats in STAGE_KERNELS.items():
            fb = wb = 0.0
            kern = {}
            for k, (c, n) in fetch.items():
                if any(p_ in k for p_ in pats):
                    # the profiled command runs warm-up + 1 step: launches / 2 per step
                    w = write.get(k, ({"WRITE_SIZE": 0.0}, n))[0]["WRITE_SIZE"]
                    per_step = n / 2.0
                    kern[short(k)] = {"launches_per_step": per_step, "fetch_MB_x2": round(2 * c["FETCH_SIZE"] / 1024 * per_step, 2),
                                      "write_MB": round(w / 1024 * per_step, 2)}
                    fb += 2 * c["FETCH_SIZE"] * 1024 * per_step
                    wb += w * 1024 * per_step
            t = timing["kernels"].get(stage, {})
            e = {"bytes_per_step": int(fb + wb), "fetch_bytes_per_step": int(fb), "write_bytes_per_step": int(wb), "kernels": kern}
            if t:
                e["algorithmic_bytes_per_step"] = t["algorithmic_bytes_per_step"]
                e["ms_per_step_unprofiled"] = t["ms_per_step"]
                e["measured_GBps"] = round((fb + wb) / (t["ms_per_step"] * 1e-3) / 1e9, 1)
                e["algorithmic_GBps"] = t["algorithmic_GBps"]
                e["traffic_over_algorithmic"] = round((fb + wb) / max(t["algorithmic_bytes_per_step"], 1), 3)
            out[key][stage] = e
    return out


def main():
    if sys.argv[1] == "stages":
        json.dump(stages(*sys.argv[2:9]), sys.stdout, indent=1)
    elif sys.argv[1] == "traffic":
        json.dump(traffic(sys.argv[2], sys.argv[3], sys.argv[4]), sys.stdout, indent=1)
    elif sys.argv[1] == "sq":
        json.dump(sq(sys.argv[2], sys.argv[3]), sys.stdout, indent=1)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
