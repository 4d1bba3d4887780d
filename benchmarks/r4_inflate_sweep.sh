#!/bin/bash
# config 3 under build flags (INF3_* geometry): bash benchmarks/r4_inflate_sweep.sh "-DINF3_OCC=7" ...
for F in "$@"; do
  RCX_EXTRA_FLAGS="$F" python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
  RCX_EXTRA_FLAGS="$F" RCX_INFLATE_VARIANT=12 RCX_CFG_NOCHECK=1 timeout 300 python benchmarks/bench_configs.py --configs 3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('statuses'): print('   ', l.strip()[:80])
    if l.startswith('{'): print('flags [$F] ms', json.loads(l)['ms'])
"
done
