# A/B of LZ4 decode kernel variants on the GPU box: bash benchmarks/ab_lz4.sh "0 23" [kinds]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_lz4.py -x -q -m gpu 2>&1 | tail -3
for v in ${1:-0}; do
 for k in ${2:-text}; do
  RCX_AB=1 RCX_BENCH_EXPERIMENT_NOCHECK=1 timeout 200 python bench.py --variant $v --kind $k --no-cpu --no-e2e --no-others --steps 20 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('variant %s %-5s %.4f ms  %.1f GiB/s  frac %.4f' % (d['config']['kernel_variant'], d['config']['distribution'], d['roofline']['kernel_ms_avg'], d['value'], d['roofline']['frac']))
    elif 'rror' in l or 'ssert' in l: print(l.rstrip())
"
 done
done
