set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_lz4.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/v7_t.log
for v in 0 20; do
 timeout 200 python bench.py --variant $v --no-cpu --no-e2e --no-others --steps 20 > gpurun_out/v7_b$v.log 2>&1
 for k in runs rand mix; do timeout 200 python bench.py --variant $v --kind $k --no-cpu --no-e2e --no-others --steps 10 > gpurun_out/v7_b${v}_$k.log 2>&1; done
done
tail -n 3 gpurun_out/v7_t.log; for f in gpurun_out/v7_b*.log; do echo $f; tail -c 600 $f; echo; done
