#!/bin/bash
# config 3 with the executor cut out of k_inflate3 (INF3_CUT_EMIT: wrong output on purpose, results not checked), first pass only
# (variant 12: the segment pass without the second pass): time and instruction counters -> the front end's share
for F in "" "-DINF3_CUT_EMIT=1"; do
  RCX_EXTRA_FLAGS="$F" python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
  RCX_EXTRA_FLAGS="$F" RCX_CFG_NOCHECK=1 RCX_INFLATE_VARIANT=12 timeout 300 python benchmarks/bench_configs.py --configs 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'): print('flags [$F] ms', json.loads(l)['ms'])
"
  RCX_EXTRA_FLAGS="$F" RCX_CFG_NOCHECK=1 bash benchmarks/pmc_inflate_insts.sh 12 > /dev/null 2>&1
  python -c "
import json; d = json.load(open('gpurun_out/pmc_insts_inflate_v12.json'))
print('flags [$F] M instructions per CU', {k[9:]: round(d[k]['per_CU'] / 1e6, 2) for k in d if k.startswith('SQ_INSTS')})"
done
