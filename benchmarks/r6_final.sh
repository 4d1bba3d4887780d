#!/bin/bash
# Round 6, the evidence run (after the last source change): kernel-trace stats and HBM-traffic counters of the headline, the
# instruction mix of the two decoders, kernel stats + traffic of configs 3 / 4 / 5, the host path's timeline.  Everything lands under
# gpurun_out/r06f_* and is copied to profiles/ once looked at.
set -x
mkdir -p gpurun_out
bash benchmarks/profile_round.sh > gpurun_out/r06f_profile_round.log 2>&1
cp gpurun_out/lz4_decode_kernel_stats.csv gpurun_out/r06f_lz4_decode_kernel_stats.csv
cp gpurun_out/bench_line.json gpurun_out/r06f_bench_line.json
cp gpurun_out/bench_line_profiled.json gpurun_out/r06f_bench_line_profiled.json
cp gpurun_out/pmc_lz4_decode.json gpurun_out/r06f_pmc_lz4_decode.json
bash benchmarks/pmc_insts.sh "0" > /dev/null 2>&1
cp gpurun_out/pmc_insts_v0.json gpurun_out/r06f_pmc_insts_lz4_decode_v0.json
bash benchmarks/pmc_inflate_insts.sh "0" > /dev/null 2>&1
cp gpurun_out/pmc_insts_inflate_v0.json gpurun_out/r06f_pmc_insts_inflate_v0.json
bash benchmarks/profile_configs.sh r06f > gpurun_out/r06f_profile_configs.log 2>&1
bash benchmarks/pmc_configs.sh "3 4 5" > gpurun_out/r06f_pmc_configs.log 2>&1
timeout 600 python benchmarks/single_stream.py 2>/dev/null | tail -1 > gpurun_out/r06f_single_stream.json
timeout 300 python benchmarks/host_path_rate.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06f_host_path_rate.txt
ls -la gpurun_out | tail -40
