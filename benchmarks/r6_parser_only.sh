#!/bin/bash
# the parser wave alone (A/B variant 45: the executor only empties the ring) under build flags: time a launch and instructions a batch
# bash benchmarks/r6_parser_only.sh "<flags>" ...
for f in "$@"; do
  RCX_EXTRA_FLAGS="$f" RCX_AB=1 python -c "from rust_compress_amd.csrc import build; build.build(ab=True)" 2>&1 | grep -i " error" | head -3
  RCX_EXTRA_FLAGS="$f" bash benchmarks/pmc_insts.sh "45" >/dev/null 2>&1
  python -c "
import json
d=json.load(open('gpurun_out/pmc_insts_v45.json'))
print('$f', {k[9:]: round(d[k]['per_CU']/1568) for k in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_INSTS_BRANCH')})"
  for i in 1 2; do RCX_EXTRA_FLAGS="$f" RCX_AB=1 RCX_BENCH_EXPERIMENT_NOCHECK=1 python bench.py --no-cpu --no-e2e --no-others --no-dists --steps 40 --warmup 20 --variant 45 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('   parser-only ms', json.loads(l)['ms_per_step'])"; done
done
