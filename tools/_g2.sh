set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1500 python -m pytest tests/test_hip_unet.py tests/test_hip_postmodel.py tests/test_hip_geometry.py tests/test_hip_upfold.py -x -q -m gpu > gpurun_out/t_r4a.log 2>&1 || (tail -40 gpurun_out/t_r4a.log; exit 1)
tail -3 gpurun_out/t_r4a.log
python3 bench.py --steps 5 --warmup 2 > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err
tail -2 gpurun_out/r04_bench_a.err
