# A/B of the DEFLATE kernel variants on BASELINE config 3: bash benchmarks/ab_inflate.sh "0 11 12" [scale]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_codecs.py tests/test_gpu_gzip.py -x -q -m gpu -k "inflate or zlib or gzip" 2>&1 | tail -3
for v in ${1:-0}; do
  RCX_INFLATE_VARIANT=$v timeout 600 python benchmarks/bench_configs.py --configs 3 --scale ${2:-1.0} 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('inflate variant $v: %.3f ms  %.1f GiB/s  frac %.4f' % (d['ms'], d['GiB/s'], d['roofline']['frac']))
    elif 'rror' in l or 'ssert' in l: print(l.rstrip())
"
done
