#!/bin/bash
# copy the evidence of benchmarks/r6_final.sh (gpurun_out/r06f_*) and of r6_verify_final.sh into profiles/ under the names DESIGN.md cites
set -e
G=gpurun_out; P=profiles
cp $G/r06f_bench_line.json $P/r06_bench_line.json
cp $G/r06f_bench_line_profiled.json $P/r06_bench_line_profiled.json
cp $G/r06f_lz4_decode_kernel_stats.csv $P/r06_lz4_decode_kernel_stats.csv
cp $G/r06f_pmc_lz4_decode.json $P/pmc_lz4_decode.json
cp $G/r06f_pmc_insts_lz4_decode_v0.json $P/pmc_insts_lz4_decode_v0.json
cp $G/r06f_pmc_insts_inflate_v0.json $P/pmc_insts_inflate_v0.json
for c in 3 4 5; do cp $G/r06f_cfg${c}_kernel_stats.csv $P/r06_cfg${c}_kernel_stats.csv; cp $G/pmc_cfg$c.json $P/pmc_cfg$c.json; done
cat $G/r06f_cfg3_lines.jsonl $G/r06f_cfg4_lines.jsonl $G/r06f_cfg5_lines.jsonl > $P/r06_configs_3_4_5.jsonl
cp $G/r06f_single_stream.json $P/r06_single_stream.json
cp $G/r06f_host_path_rate.txt $P/r06_host_path_rate.txt
[ -f $G/r06_final_verify.log ] && cp $G/r06_final_verify.log $P/r06_final_verify.log
[ -f $G/pmc_insts_v45.json ] && cp $G/pmc_insts_v45.json $P/r06_pmc_insts_lz4_parser_only_v45.json
ls -la $P | grep -E "r06|pmc_" | wc -l
