#!/bin/bash
# PMC passes over the conv stack at the production launch geometry (BATCH tiles of 300x300x20, default 64 = what
# bench.py launches), on the GPU box:
#   pass 1  FETCH_SIZE      pass 2  WRITE_SIZE      (separate passes, as MI355X_MICROARCH.md prescribes)
#   pass 3  SQ counters     (MFMA busy, wave wait / issue split, LDS activity)
# Each pass is `rocprofv3 --kernel-trace --pmc ...` only (no other trace domains).  Writes
#   gpurun_out/<tag>_conv_hbm_traffic_pmc.json   and   gpurun_out/<tag>_conv_sq_counters.json
# which are then copied into profiles/ (bench.py's roofline.traffic reads the newest committed traffic file).
#
#   gpurun -- 'bash tools/pmc_collect.sh r03'          BATCH=8 bash tools/pmc_collect.sh r03_b8
set -e
TAG=${1:-r03}
BATCH=${BATCH:-64}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CMD="python3 $ROOT/tools/bench_conv.py --tile 300,300,20 --batch $BATCH --iters 1 --warmup 1"
# PRECISION=split | mix8: the same passes over that precision's kernels, written under names bench.py does not pick up
TRAFFIC_NAME=conv_hbm_traffic_pmc
SQ_NAME=conv_sq_counters
if [ -n "$PRECISION" ]; then
    CMD="$CMD --precision $PRECISION"
    TRAFFIC_NAME=${PRECISION}_hbm_traffic
    SQ_NAME=${PRECISION}_sq_counters
fi
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_s
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d /tmp/pmc_s -o s -- $CMD > /dev/null 2>&1
mkdir -p "$ROOT/gpurun_out"
python3 "$ROOT/tools/pmc_summary.py" traffic /tmp/pmc_f /tmp/pmc_w "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- $CMD" \
    > "$ROOT/gpurun_out/${TAG}_${TRAFFIC_NAME}.json"
python3 "$ROOT/tools/pmc_summary.py" sq /tmp/pmc_s "rocprofv3 --kernel-trace --pmc $SQ -- $CMD" \
    > "$ROOT/gpurun_out/${TAG}_${SQ_NAME}.json"
echo "wrote gpurun_out/${TAG}_${TRAFFIC_NAME}.json gpurun_out/${TAG}_${SQ_NAME}.json"
