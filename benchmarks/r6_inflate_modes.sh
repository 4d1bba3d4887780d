#!/bin/bash
# config 3 under build flags: bash benchmarks/r6_inflate_modes.sh "<flags 1>" "<flags 2>" ...
for f in "$@"; do
  RCX_EXTRA_FLAGS="$f" python -c "from rust_compress_amd.csrc import build; build.build()" 2>&1 | grep -i " error" | head -3
  for i in 1 2 3; do echo "$f $(RCX_EXTRA_FLAGS="$f" python benchmarks/bench_configs.py --configs 3 2>&1 | grep -o '"ms": [0-9.]*' | head -1)"; done
done
python -c "from rust_compress_amd.csrc import build; build.build()"
