#!/bin/bash
# instruction counters of k_inflate3 with and without its executor (INF3_CUT_EMIT), first pass only (variant 12: the segment pass, no second pass)
for F in "" "-DINF3_CUT_EMIT=1"; do
  RCX_EXTRA_FLAGS="$F" python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
  echo "flags [$F]"
  RCX_EXTRA_FLAGS="$F" RCX_CFG_NOCHECK=1 bash benchmarks/pmc_inflate_insts.sh 12 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        try: d = json.loads(l)
        except Exception: print(l[:300]); continue
        print({k[9:]: round(d[k]['per_CU'] / 1e6, 2) for k in d if k.startswith('SQ_INSTS')})
"
done
