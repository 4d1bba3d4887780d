#!/bin/bash
for seed in 11 12 13 14; do t0=$(date +%s); timeout 1200 python benchmarks/fuzz_gpu_enc.py 3000 $seed 2>&1 | grep -v " 0 mismatches" | tail -4; echo "seed $seed: $(( $(date +%s) - t0 )) s"; done
