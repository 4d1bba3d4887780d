cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r03_gputests.log
timeout 1500 python bench.py --others-scale 0.25 > gpurun_out/r03_bench_debug.log 2>&1
tail -c 6000 gpurun_out/r03_bench_debug.log; cat gpurun_out/r03_gputests.log
