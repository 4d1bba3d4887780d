#!/bin/bash
# the range coder's rotating issue priorities (RCX_ARI_ROTATE: the period in symbols, 0 = off): per-stage times of config 5
for v in ${ROT_MODES:-0 256 64 1024}; do
    RCX_EXTRA_FLAGS="-DRCX_ARI_ROTATE=$v $ROT_EXTRA" python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
    RCX_EXTRA_FLAGS="-DRCX_ARI_ROTATE=$v $ROT_EXTRA" python benchmarks/pipeline_stages.py 1.0 2>/dev/null | grep "ari_\|^bytes" | tr '\n' ' '; echo " RCX_ARI_ROTATE=$v $ROT_EXTRA"
done
python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
