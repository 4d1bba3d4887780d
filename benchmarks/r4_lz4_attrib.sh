#!/bin/bash
# Round 4: where the LZ4 decoder's instructions go.  Correctness of the default first, then the timing and the instruction
# counters of the default (0) and of the A/B builds with phases cut out (41..45, results wrong on purpose).
REPO=$(pwd)
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_lz4.py -x -q 2>&1 | tail -3 > gpurun_out/r4_attrib_tests.log
for V in 0 41 42 43 44 45; do
  RCX_AB=1 RCX_BENCH_EXPERIMENT_NOCHECK=1 timeout 200 python bench.py --variant $V --no-cpu --no-e2e --no-others --no-dists --steps 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('variant $V ms', d['ms_per_step'], 'kernel', d['roofline'].get('kernel_ms_avg'))
" >> gpurun_out/r4_attrib_times.log
done
bash benchmarks/pmc_insts.sh "0 41 42 43 44 45"
cat gpurun_out/r4_attrib_tests.log gpurun_out/r4_attrib_times.log
