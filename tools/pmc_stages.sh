#!/bin/bash
# PMC passes over stages 1b-3 at 1024x1024x256 (tools/bench_stages.py: gate/dilate/scatter, the ccl_* kernels, follow +
# assign, the renumber kernels) -- FETCH_SIZE and WRITE_SIZE in SEPARATE passes, kernel trace only, as
# MI355X_MICROARCH.md prescribes -- and a timing run of the sparse and the dense field.  Writes
#   gpurun_out/<tag>_stage23_hbm_traffic_pmc.json     (bench.py's roofline_gate / _ccl / _assign read the committed copy)
#
#   gpurun -- 'bash tools/pmc_stages.sh r03'
set -e
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CMD="python3 $ROOT/tools/bench_stages.py --iters 1 --warmup 1"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcs_f /tmp/pmcs_w /tmp/pmcs_fd /tmp/pmcs_wd
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcs_f -o f -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcs_w -o w -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcs_fd -o f -- $CMD --dense > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcs_wd -o w -- $CMD --dense > /dev/null 2>&1
mkdir -p "$ROOT/gpurun_out"
python3 "$ROOT/tools/bench_stages.py" --iters 3 > "$ROOT/gpurun_out/${TAG}_stages_sparse.json"
python3 "$ROOT/tools/bench_stages.py" --iters 3 --dense > "$ROOT/gpurun_out/${TAG}_stages_dense.json"
python3 "$ROOT/tools/pmc_summary.py" stages /tmp/pmcs_f /tmp/pmcs_w /tmp/pmcs_fd /tmp/pmcs_wd \
    "$ROOT/gpurun_out/${TAG}_stages_sparse.json" "$ROOT/gpurun_out/${TAG}_stages_dense.json" \
    "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- $CMD [--dense]" \
    > "$ROOT/gpurun_out/${TAG}_stage23_hbm_traffic_pmc.json"
echo "wrote gpurun_out/${TAG}_stage23_hbm_traffic_pmc.json"
