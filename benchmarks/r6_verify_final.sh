#!/bin/bash
# the last check of the round's final tree, every line kept: GPU suite, smoke(), decoder fuzz (4 seeds), encoder / transform fuzz (3 seeds),
# inverse-BWT fuzz, DC fuzz
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for seed in 111 112 113 114; do timeout 900 python benchmarks/fuzz_gpu.py 400000 $seed 2>&1 | grep -v "amdgpu.ids"; done
for seed in 115 116 117; do timeout 900 python benchmarks/fuzz_gpu_enc.py 3000 $seed 2>&1 | grep -v "amdgpu.ids" | tail -12; done
timeout 600 python benchmarks/fuzz_gpu_bwti.py 2>&1 | tail -2
timeout 600 python benchmarks/fuzz_gpu_dc.py 2>&1 | tail -2
