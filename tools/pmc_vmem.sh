#!/bin/bash
# Vector-memory path counters of the conv stack (8 production tiles): is the texture-address / L1 path what the conv3
# kernels wait for?  One rocprofv3 --pmc pass per counter pair (block slot limits), kernel trace only.
#   gpurun -- 'bash tools/pmc_vmem.sh r02'   ->  gpurun_out/<tag>_conv_vmem_counters.json
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CMD="python3 $ROOT/tools/bench_conv.py --tile 300,300,20 --batch 8 --iters 1 --warmup 1"
PASSES=("TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
        "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum GRBM_GUI_ACTIVE"
        "TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE"
        "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
        "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
        "TD_TD_BUSY_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE"
        "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE")
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_v
i=0
for P in "${PASSES[@]}"; do
    rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/pmc_v/$i -o p -- $CMD > /dev/null 2>&1 || echo "pass $i failed: $P"
    i=$((i + 1))
done
mkdir -p "$ROOT/gpurun_out"
python3 - "$ROOT/gpurun_out/${TAG}_conv_vmem_counters.json" <<'PY'
import collections, csv, glob, json, re, sys
out = collections.defaultdict(dict)
for d in sorted(glob.glob("/tmp/pmc_v/*")):
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        tot = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(path)):
            tot[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
        for k in tot:
            name = re.sub(r"\(.*", "", k.replace("(anonymous namespace)::", "").replace("void ", ""))[:80]
            if "conv3" not in name and "down2" not in name and "gn_silu" not in name:
                continue
            for c, v in tot[k].items():
                if c == "GRBM_GUI_ACTIVE" and c in out[name]:
                    continue
                out[name][c] = round(v / len(disp[k]))
            out[name]["launches"] = len(disp[k])
res = {"note": "per-launch averages, summed over the device; GRBM_GUI_ACTIVE / 8 = kernel cycles; *_BUSY / (cycles * 256 CUs) = busy fraction per CU",
       "kernels": {}}
for k, c in sorted(out.items()):
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    d = dict(c)
    if cyc:
        for key in ("TA_TA_BUSY_sum", "TD_TD_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_ADDR_STALLED_BY_TD_CYCLES_sum",
                    "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_GATE_EN1_sum", "TA_BUFFER_TOTAL_CYCLES_sum"):
            if key in c:
                d[key.replace("_sum", "") + "_frac_per_cu"] = round(c[key] / (cyc * 256), 4)
        if c.get("TA_BUFFER_WAVEFRONTS_sum"):
            d["cycles_per_vmem_instruction_per_cu"] = round(cyc * 256 / c["TA_BUFFER_WAVEFRONTS_sum"], 1)
    res["kernels"][k] = d
json.dump(res, open(sys.argv[1], "w"), indent=1)
print("wrote", sys.argv[1])
PY
