#!/bin/bash
# LDS / VALU / instruction counters of the headline kernel under build flags: bash benchmarks/r6_pmc.sh "<flags>" ...
REPO=$(pwd)
for f in "$@"; do
  echo "=== $f"
  RCX_EXTRA_FLAGS="$f" python -c "from rust_compress_amd.csrc import build; build.build()" 2>&1 | grep -i " error" | head -3
  cd /tmp && export TMPDIR=/tmp
  i=0
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
             "SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    rm -rf /tmp/pl_$i
    RCX_EXTRA_FLAGS="$f" timeout 300 rocprofv3 --pmc $set -d /tmp/pl_$i -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --no-others --no-dists > /tmp/pl_$i.log 2>&1
    db=$(find /tmp/pl_$i -name "*.db" | head -1)
    python $REPO/benchmarks/pmcq.py $db lz4_decode 2>&1 | cut -c1-20,40- || tail -5 /tmp/pl_$i.log
  done
  cd $REPO
done
