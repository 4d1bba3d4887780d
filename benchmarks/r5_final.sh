#!/bin/bash
# Round 5, the evidence run (after the last source change): kernel-trace stats and HBM-traffic counters of the headline, the
# instruction mix of the two decoders, kernel stats + traffic of configs 3 / 4 / 5, the host path's timeline.  Everything lands under
# gpurun_out/r05f_* and is copied to profiles/ once looked at.
set -x
mkdir -p gpurun_out
bash benchmarks/profile_round.sh > gpurun_out/r05f_profile_round.log 2>&1
cp gpurun_out/lz4_decode_kernel_stats.csv gpurun_out/r05f_lz4_decode_kernel_stats.csv
cp gpurun_out/bench_line.json gpurun_out/r05f_bench_line.json
cp gpurun_out/bench_line_profiled.json gpurun_out/r05f_bench_line_profiled.json
cp gpurun_out/pmc_lz4_decode.json gpurun_out/r05f_pmc_lz4_decode.json
bash benchmarks/pmc_insts.sh "0" > /dev/null 2>&1
cp gpurun_out/pmc_insts_v0.json gpurun_out/r05f_pmc_insts_lz4_decode_v0.json
bash benchmarks/pmc_inflate_insts.sh "0" > /dev/null 2>&1
cp gpurun_out/pmc_insts_inflate_v0.json gpurun_out/r05f_pmc_insts_inflate_v0.json
bash benchmarks/profile_configs.sh r05f > gpurun_out/r05f_profile_configs.log 2>&1
bash benchmarks/pmc_configs.sh "3 4 5" > gpurun_out/r05f_pmc_configs.log 2>&1
bash benchmarks/r5_hostpath_trace.sh > gpurun_out/r05f_hostpath_trace.txt 2>&1
timeout 300 python benchmarks/host_path_rate.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05f_host_path_rate.txt
ls -la gpurun_out | tail -40
