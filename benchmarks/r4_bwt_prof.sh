#!/bin/bash
# phase tick totals of k_bws_local_wg (the largest kernel of the forward BWT) on config 4 (-DBWS_PROF build);
# $1: further flags (e.g. -DBWS_CUT_RANK=1: no rank stores -- wrong results, round 0's line is what counts)
FL="-DBWS_PROF=1 $1"
RCX_EXTRA_FLAGS="$FL" python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
echo "== $FL"
RCX_EXTRA_FLAGS="$FL" RCX_BWT_TRACE=1 RCX_CFG_NOCHECK=1 timeout 600 python benchmarks/bench_configs.py --configs 4 --once --kinds text 2>&1 | grep -A1 "bwt forward round 0\|forward_ms" | cut -c1-400 | head -${2:-8}
