#!/bin/bash
# Which issue port binds the LZ4 decoder?  N extra instructions per batch on the scalar port / on the vector ALU (the executor
# wave, results untouched): the headline's time with each.  ~1570 batches a CU and launch: 100 a batch = +157 K instructions.
for flags in ${PORT_FLAGS:-"-DRCX_NONE=1" "-DRCX_DUMMY_SALU=100" "-DRCX_DUMMY_SALU=200" "-DRCX_DUMMY_VALU=100" "-DRCX_DUMMY_VALU=200"}; do
    RCX_EXTRA_FLAGS="$flags" python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
    for i in 1; do
        RCX_EXTRA_FLAGS="$flags" python bench.py --no-cpu --no-e2e --no-others --no-dists --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$flags', j['ms_per_step'], j['roofline']['kernel_ms_avg'])"
    done
done
python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
