#!/bin/bash
# builds the shipped library with each flag set in turn and times the LZ4 decode default on G-text:
#   bash benchmarks/r4_flag_sweep.sh "-DA=1" "-DB=2 -DC=3" ...
for F in "$@"; do
  RCX_EXTRA_FLAGS="$F" python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build(force=False)" > /dev/null 2>&1
  RCX_EXTRA_FLAGS="$F" timeout 200 python bench.py --no-cpu --no-e2e --no-others --no-dists --steps 30 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('flags [$F] ms', d['ms_per_step'], 'kernel', d['roofline'].get('kernel_ms_avg'))
"
done
