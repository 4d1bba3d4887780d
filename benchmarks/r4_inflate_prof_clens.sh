#!/bin/bash
# as r4_inflate_prof.sh, the code-length decode (P_CLENS) counted apart from the other header work (it lands in the "wide copies" slot)
RCX_EXTRA_FLAGS="-DINF3_PROF=1 -DINF3_PROF_CLENS=1" python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
RCX_EXTRA_FLAGS="-DINF3_PROF=1 -DINF3_PROF_CLENS=1" RCX_INF3_PROF=1 RCX_CFG_NOCHECK=1 RCX_INFLATE_VARIANT=12 timeout 300 python benchmarks/bench_configs.py --configs 3 2>&1 | grep "per member\|\"ms\"" | cut -c1-400
bash benchmarks/r4_inflate_prof.sh
python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
