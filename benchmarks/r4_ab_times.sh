#!/bin/bash
# timing of LZ4 decode variants from the A/B library: bash benchmarks/r4_ab_times.sh "0 46 47" [kind]
for V in $1; do
  RCX_AB=1 RCX_BENCH_EXPERIMENT_NOCHECK=1 timeout 200 python bench.py --variant $V --kind ${2:-text} --no-cpu --no-e2e --no-others --no-dists --steps 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('variant $V ${2:-text} ms', d['ms_per_step'], 'kernel', d['roofline'].get('kernel_ms_avg'))
"
done
