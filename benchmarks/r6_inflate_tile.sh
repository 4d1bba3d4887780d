#!/bin/bash
# the inflate front end's tile walks (k_inflate3, -DINF3_PROF -DINF3_PROF_TILE): rounds of [ISA walk, portable step] per tile, cycles in each
F="-DINF3_PROF=1 ${2:--DINF3_PROF_TILE=1} $1"
RCX_EXTRA_FLAGS="$F" python -c "from rust_compress_amd.csrc import build; build.build()" 2>&1 | grep -i " error" | head -3
RCX_EXTRA_FLAGS="$F" RCX_INF3_PROF=1 RCX_CFG_NOCHECK=1 RCX_INFLATE_VARIANT=12 timeout 300 python benchmarks/bench_configs.py --configs 3 2>&1 | grep "per member\|\"ms\"" | cut -c1-500
