#!/bin/bash
# RCX_AGE_PRIO experiment: the headline's kernel time with the age-aware issue priorities (0 = the shipped kernel, 1, 2, 3)
for v in ${AGE_MODES:-0 2 4 8 10 12}; do
    RCX_EXTRA_FLAGS="-DRCX_AGE_PRIO=$v $AGE_EXTRA" python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
    for i in 1 2; do
        RCX_EXTRA_FLAGS="-DRCX_AGE_PRIO=$v $AGE_EXTRA" python bench.py --no-cpu --no-e2e --no-others --no-dists 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('RCX_AGE_PRIO=$v $AGE_EXTRA', j['ms_per_step'], j['roofline']['kernel_ms_avg'], j['value'])"
    done
done
python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
