#!/bin/bash
# Round profile of the headline benchmark (run on the GPU box via gpurun): kernel-trace stats + HBM traffic PMC passes.
# (--no-dists: only the headline distribution's dispatches, so that the per-kernel averages are G-text's)
# Outputs under gpurun_out/: lz4_decode_kernel_stats.csv, pmc_lz4_decode.json, bench_line.json
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $REPO/bench.py --steps 20 --warmup 3 --no-cpu --no-e2e --no-others --no-dists > /tmp/kt.log 2>&1
grep "^{\"metric\"" /tmp/kt.log | tail -1 > $REPO/gpurun_out/bench_line_profiled.json
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/lz4_decode_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu --no-e2e --no-others --no-dists > /tmp/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu --no-e2e --no-others --no-dists > /tmp/pw.log 2>&1
ALG=$(python -c "import json,sys; print(json.loads(open('$REPO/gpurun_out/bench_line_profiled.json').read())['roofline']['algorithmic_bytes_per_launch'])")
python $REPO/benchmarks/pmc_traffic.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) lz4_decode_v8 $ALG $REPO/gpurun_out/pmc_lz4_decode.json
head -4 $REPO/gpurun_out/lz4_decode_kernel_stats.csv | cut -c1-200
cd $REPO && python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_line.json
