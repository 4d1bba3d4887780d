#!/bin/bash
# after the sorter changes: the whole GPU suite, the encoder / transform fuzz (BWT forward included, every input against the oracle),
# the inverse-BWT fuzz, configs 4 and 5
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for seed in 11 12 13; do t0=$(date +%s); timeout 900 python benchmarks/fuzz_gpu_enc.py 3000 $seed 2>&1 | tail -2; echo "seed $seed: $(( $(date +%s) - t0 )) s"; done
timeout 600 python benchmarks/fuzz_gpu_bwti.py 2>&1 | tail -2
timeout 900 python benchmarks/bench_configs.py --configs 4,5 2>&1 | grep '^{"config"' | cut -c1-420
