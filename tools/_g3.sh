set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_unet.py tests/test_hip_geometry.py tests/test_hip_upfold.py -x -q -m gpu > gpurun_out/t_r4b.log 2>&1 || (tail -40 gpurun_out/t_r4b.log; exit 1)
tail -2 gpurun_out/t_r4b.log
BATCH=64 bash tools/kernel_ab.sh skoots_amd/libskoots_hip_zw0.so skoots_amd/libskoots_hip.so skoots_amd/libskoots_hip_zw0.so skoots_amd/libskoots_hip.so > gpurun_out/r04_ab_zero_window.txt 2>&1
cat gpurun_out/r04_ab_zero_window.txt
SKOOTS_HIP_LIB=$GRAFT_REPO_ROOT/skoots_amd/libskoots_hip_zw0.so bash tools/pmc_collect.sh r04zw0 > /dev/null
bash tools/pmc_collect.sh r04 > /dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --precision split --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-also > $GRAFT_REPO_ROOT/gpurun_out/r04_split_tl2.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04_split_tl2.err
python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py /tmp/tl > $GRAFT_REPO_ROOT/gpurun_out/r04_split_timeline2.txt
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 5 --warmup 2 > gpurun_out/r04_bench_b.json 2> gpurun_out/r04_bench_b.err
tail -3 gpurun_out/r04_bench_b.err
