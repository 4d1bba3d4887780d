#!/bin/bash
# the headline under build flags: bash benchmarks/r6_modes.sh "<flags 1>" "<flags 2>" ...   (time a step, G-runs, G-rand)
for f in "$@"; do
  RCX_EXTRA_FLAGS="$f" python -c "from rust_compress_amd.csrc import build; build.build()" 2>&1 | grep -i " error" | head -3
  for i in 1 2; do
  RCX_EXTRA_FLAGS="$f" timeout 300 python bench.py --no-cpu --no-e2e --no-others --steps 40 --warmup 20 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$f', 'ms', d['ms_per_step'], 'kernel', d['roofline'].get('kernel_ms_avg'), {k: v.get('ms_per_step') for k, v in d.get('per_distribution', {}).items()})
"
  done
done
