set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_hip_unet.py tests/test_hip_geometry.py tests/test_hip_eval_file.py tests/test_hip_sharded.py -x -q -m gpu > gpurun_out/t_r4d.log 2>&1 || (tail -60 gpurun_out/t_r4d.log; exit 1)
tail -2 gpurun_out/t_r4d.log
python3 bench.py --steps 5 --warmup 2 > gpurun_out/r04_bench_d.json 2> gpurun_out/r04_bench_d.err
tail -3 gpurun_out/r04_bench_d.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st1 -o st -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also > $GRAFT_REPO_ROOT/gpurun_out/r04_bench_default_under_rocprof.json 2>/dev/null
cp $(find /tmp/st1 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04_bench_default_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st2 -o st -- python3 $GRAFT_REPO_ROOT/bench.py --precision split --steps 3 --warmup 1 --no-cpu-baseline --no-also > $GRAFT_REPO_ROOT/gpurun_out/r04_bench_split_under_rocprof.json 2>/dev/null
cp $(find /tmp/st2 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04_split_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st3 -o st -- python3 $GRAFT_REPO_ROOT/bench.py --config train --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r04_train_bf16_under_rocprof.json 2>/dev/null
cp $(find /tmp/st3 -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/r04_train_kernel_stats.csv
echo profiles done
