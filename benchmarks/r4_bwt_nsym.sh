#!/bin/bash
# forward BWT time against the number of symbols in the first key (RCX_BWT_NSYM, an experiment knob of the host driver)
for k in text dna4; do
  for ns in $( [ $k = text ] && echo "10 9 8 7 6" || echo "16 14 13 12 11 10" ); do
    RCX_BWT_NSYM=$ns timeout 300 python benchmarks/bench_configs.py --configs 4 --kinds $k 2>&1 | grep forward_ms | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('$k nsym $ns forward', j['forward_ms'])"
  done
done
