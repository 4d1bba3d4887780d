#!/bin/bash
# phase timers of the LZ4 decoder (A/B library, variant 24) built with the given flags: bash benchmarks/r6_prof.sh "<flags>" ...
for f in "$@"; do
  echo "=== $f"
  RCX_EXTRA_FLAGS="$f" RCX_AB=1 python -c "from rust_compress_amd.csrc import build; build.build(ab=True)" 2>&1 | grep -i " error" | head -3
  RCX_EXTRA_FLAGS="$f" RCX_AB=1 timeout 300 python benchmarks/lz4_v8_profile.py text 4096 2>&1 | grep -v amdgpu.ids
  RCX_EXTRA_FLAGS="$f" RCX_AB=1 timeout 300 python benchmarks/lz4_v8_profile.py text 256 2>&1 | grep -v amdgpu.ids
done
