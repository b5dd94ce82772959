set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t_r4c.log 2>&1 || (tail -60 gpurun_out/t_r4c.log; exit 1)
tail -2 gpurun_out/t_r4c.log
python3 bench.py --steps 5 --warmup 2 > gpurun_out/r04_bench_c.json 2> gpurun_out/r04_bench_c.err
tail -3 gpurun_out/r04_bench_c.err
bash tools/pmc_stages.sh r04 > /dev/null 2>&1
