#!/bin/bash
# Per-kernel A/B on the GPU box, precise to ~0.3 %: average kernel durations from rocprofv3's rocpd database
# (view top_kernels) over tools/bench_conv.py, one run per library / environment variant, all inside ONE gpurun call
# (boxes differ by +-5 %; end-to-end bench.py pairs are too noisy for changes below 2 %).
#
#   gpurun -- 'bash tools/kernel_ab.sh ab/libold.so skoots_amd/libskoots_hip.so'
#   gpurun -- 'SK_CONV_XC=48 bash tools/kernel_ab.sh skoots_amd/libskoots_hip.so'
#
# Arguments: shared libraries to compare (absolute or repo-relative).  Prints one line per library with the average
# microseconds of every kernel whose name matches $KERNELS (default: conv3_|stem_|gather_|gn_silu|heads).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
KERNELS=${KERNELS:-'conv3_|stem_|gather_|gn_silu|heads'}
TILE=${TILE:-300,300,20}
BATCH=${BATCH:-8}
cd /tmp && export TMPDIR=/tmp
for LIB in "$@"; do
    case "$LIB" in /*) ;; *) LIB="$ROOT/$LIB" ;; esac
    TAG=$(basename "$LIB" .so)_$RANDOM
    SKOOTS_HIP_LIB="$LIB" rocprofv3 --kernel-trace --stats -d /tmp/kernel_ab -o "$TAG" -- \
        python3 "$ROOT/tools/bench_conv.py" --tile "$TILE" --batch $BATCH --iters 6 --warmup 2 > /dev/null 2>&1
    python3 - "$TAG" "$KERNELS" <<'PY'
import glob, re, sqlite3, sys
tag, pat = sys.argv[1], re.compile(sys.argv[2])
db = sqlite3.connect(glob.glob(f"/tmp/kernel_ab/{tag}_results.db")[0])
rows = [(n, c, a) for n, c, a in db.execute("select name, total_calls, average from top_kernels order by name") if pat.search(n)]
short = lambda n: re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
print(tag, "  ".join(f"{short(n)} x{c} = {a:.1f} us" for n, c, a in rows), flush=True)
PY
done
