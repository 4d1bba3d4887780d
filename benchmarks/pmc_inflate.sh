#!/bin/bash
# SQ counters for the zlib decode kernel (run on the GPU box via gpurun): bash benchmarks/pmc_inflate.sh <variant>
VAR=${1:-0}
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/infl1.py <<PY
import sys, os, zlib, numpy as np, torch
sys.path.insert(0, "$REPO")
from multiprocessing import Pool
import rust_compress_amd as R
from rust_compress_amd import _native as N, synth, batch as B
def _z(a): return zlib.compress(a[1], (1,6,9)[a[0]%3])
if __name__ == "__main__":
    dev = torch.device("cuda", 0); ctx = R.Context(0)
    nb, BLOCK = 65536, 16384
    raw_np = synth.gen_blocks("text", nb, BLOCK, 0x5A11)
    with Pool(32) as pool: members = pool.map(_z, [(i, raw_np[i*BLOCK:(i+1)*BLOCK].tobytes()) for i in range(nb)], chunksize=512)
    base, off, lens = B.pack(members); ar = np.arange(nb, dtype=np.int64)
    db = R.DeviceBatch.from_host(base, off, lens, nb*BLOCK, (ar*BLOCK).astype(np.uint64), np.full(nb, BLOCK, dtype=np.uint64), dev)
    ctx.set_variant(N.ZLIB_DECODE, $VAR)
    for _ in range(3): ctx.launch_dev(N.ZLIB_DECODE, db)
    torch.cuda.synchronize()
PY
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); rm -rf /tmp/pi_$i
    rocprofv3 --pmc $set -d /tmp/pi_$i -- python /tmp/infl1.py > /tmp/pi_$i.log 2>&1
    db=$(find /tmp/pi_$i -name "*.db" | head -1)
    python $REPO/benchmarks/pmcq.py $db inflate 2>&1 | awk "{print \$(NF-2), \$(NF-1), \$NF}" || tail -5 /tmp/pi_$i.log
done
