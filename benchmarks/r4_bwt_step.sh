#!/bin/bash
# config 4 (both directions, both distributions) with the shipped build + the BWT GPU tests: after a change to the sorter
timeout 900 python -m pytest tests/test_gpu_codecs.py tests/test_gpu_fullsize.py -x -q -m gpu -k "bwt or BWT or suffix or pipeline" 2>&1 | tail -3
for i in 1 2; do timeout 600 python benchmarks/bench_configs.py --configs 4 2>&1 | grep forward_ms | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['workload'], 'forward', j['forward_ms'], 'inverse', j['inverse_ms'])"; done
