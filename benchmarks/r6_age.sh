#!/bin/bash
# block end times by age rank under build flags (A/B variant 24): bash benchmarks/r6_age.sh "<flags>" ...
for f in "$@"; do
  RCX_EXTRA_FLAGS="$f" RCX_AB=1 python -c "from rust_compress_amd.csrc import build; build.build(ab=True)" 2>&1 | grep -i " error" | head -3
  echo "=== $f"; RCX_EXTRA_FLAGS="$f" RCX_AB=1 python benchmarks/lz4_v8_profile.py text 2>&1 | grep "block end\|LAST\|FIRST"
done
python -c "from rust_compress_amd.csrc import build; build.build()"
