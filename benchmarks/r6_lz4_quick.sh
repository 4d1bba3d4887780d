#!/bin/bash
# Round 6: the LZ4 decoder's quick loop -- parity of the LZ4 GPU tests, the headline time and the other distributions (no side legs).
# bash benchmarks/r6_lz4_quick.sh [tag]
T=${1:-quick}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_lz4.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r6_${T}_tests.log
for i in 1 2; do
timeout 300 python bench.py --no-cpu --no-e2e --no-others --steps 40 --warmup 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms', d['ms_per_step'], 'kernel', d['roofline'].get('kernel_ms_avg'), 'dists', {k: v.get('ms_per_step') for k, v in d.get('per_distribution', {}).items()} if isinstance(d.get('per_distribution'), dict) else d.get('per_distribution'))
" >> gpurun_out/r6_${T}_times.log
done
cat gpurun_out/r6_${T}_tests.log gpurun_out/r6_${T}_times.log | grep -v amdgpu.ids
