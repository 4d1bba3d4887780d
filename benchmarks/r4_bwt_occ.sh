#!/bin/bash
# k_bws_local_wg against its residency: LDS padded so that 3 or 2 workgroups share a CU instead of 4 (is it bound by what a CU
# shares, or by each workgroup's own chain?)
for pad in 0 14000 40000; do
  FL=""; [ $pad != 0 ] && FL="-DBWS_LDS_PAD=$pad"
  RCX_EXTRA_FLAGS="$FL" python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
  echo "== pad $pad"
  cd /tmp; rm -rf /tmp/kto; RCX_EXTRA_FLAGS="$FL" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kto -- python $OLDPWD/benchmarks/bench_configs.py --configs 4 --kinds text > /tmp/kto.log 2>&1; cd $OLDPWD
  f=$(find /tmp/kto -name "*kernel_stats.csv" | head -1); grep "k_bws_local_wg<unsigned long>" $f | cut -d, -f1-4
done
