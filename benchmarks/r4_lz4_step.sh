#!/bin/bash
# Round 4 iteration step for the LZ4 decoder: parity tests, the in-suite fuzz, timing per distribution, instruction counters.
#   bash benchmarks/r4_lz4_step.sh <tag> ["variants for pmc"]
TAG=${1:-step}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_lz4.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -k "lz4 or fuzz" 2>&1 | tail -4 > gpurun_out/r4_${TAG}_tests.log
timeout 250 python bench.py --no-cpu --no-others --steps 20 2>/dev/null | grep '^{' > gpurun_out/r4_${TAG}_bench.json
python - <<PY
import json
d = json.loads(open("gpurun_out/r4_${TAG}_bench.json").read().strip().splitlines()[-1])
print("${TAG}: ms", d["ms_per_step"], "kernel", d["roofline"].get("kernel_ms_avg"), "e2e", d.get("end_to_end"))
print({k: v["ms_per_step"] for k, v in d.get("per_distribution", {}).items()})
PY
bash benchmarks/pmc_insts.sh "${2:-0}" > /dev/null 2>&1
for V in ${2:-0}; do python -c "
import json; d = json.load(open('gpurun_out/pmc_insts_v$V.json'))
print('variant $V', {k[9:]: round(d[k]['per_CU'] / 1e3, 1) for k in d if k.startswith('SQ_INSTS')})"; done
cat gpurun_out/r4_${TAG}_tests.log
