#!/bin/bash
# the last check of a tree: GPU suite, smoke(), decoder fuzz, encoder / transform fuzz, inverse-BWT fuzz, DC fuzz
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for seed in 31 32; do timeout 900 python benchmarks/fuzz_gpu.py 400000 $seed 2>&1 | tail -1; done
for seed in 21 22 23; do timeout 900 python benchmarks/fuzz_gpu_enc.py 3000 $seed 2>&1 | grep -v " 0 mismatches" | tail -2; done
timeout 600 python benchmarks/fuzz_gpu_bwti.py 2>&1 | tail -2
timeout 600 python benchmarks/fuzz_gpu_dc.py 2>&1 | tail -2
