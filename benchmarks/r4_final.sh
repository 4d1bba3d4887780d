#!/bin/bash
# Round 4, the evidence run (after the last kernel change): kernel-trace stats and HBM-traffic counters of the headline, the
# instruction mix of the two decoders, kernel stats + traffic of configs 3 / 4 / 5, the SQ busy / wait counters.  Everything
# lands under gpurun_out/ and is copied to profiles/r04_* by hand once looked at.
set -x
bash benchmarks/profile_round.sh > gpurun_out/r4_final_profile_round.log 2>&1
cp gpurun_out/lz4_decode_kernel_stats.csv gpurun_out/r04_lz4_decode_kernel_stats.csv
cp gpurun_out/bench_line.json gpurun_out/r04_bench_line.json
bash benchmarks/pmc_insts.sh "0" > /dev/null 2>&1
cp gpurun_out/pmc_insts_v0.json gpurun_out/r04_pmc_insts_lz4_decode_v0.json
bash benchmarks/pmc_inflate_insts.sh "0" > /dev/null 2>&1
cp gpurun_out/pmc_insts_inflate_v0.json gpurun_out/r04_pmc_insts_inflate_v0.json
bash benchmarks/profile_configs.sh r04 > gpurun_out/r4_final_profile_configs.log 2>&1
bash benchmarks/pmc_configs.sh "3 4 5" > gpurun_out/r4_final_pmc_configs.log 2>&1
bash benchmarks/pmc_lz4.sh text > gpurun_out/r04_lz4_sq_counters.txt 2>&1
ls -la gpurun_out | tail -30
