#!/bin/bash
for flags in "-DINF3_OWNER_SCAN=1" "-DINF3_OWNER_SCAN=0" "-DINF3_OWNER_SCAN=1 -DRCX_Y=1" "-DINF3_OWNER_SCAN=0 -DRCX_Y=1"; do
  RCX_EXTRA_FLAGS="$flags" python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
  RCX_EXTRA_FLAGS="$flags" RCX_CFG_NOHOST=1 timeout 300 python benchmarks/bench_configs.py --configs 3 2>/dev/null | grep '^{' | head -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$flags', d['ms'])"
done
python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
timeout 600 python -m pytest tests/test_gpu_codecs.py tests/test_gpu_gzip.py tests/test_gpu_fullsize.py -x -q -k "inflate or zlib or gzip or config3" 2>&1 | tail -2
