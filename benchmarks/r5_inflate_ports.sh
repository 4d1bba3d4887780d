#!/bin/bash
# Which port binds the inflate kernel?  N extra instructions per chunk of 64 symbols (69 chunks a 16 KiB member) on the scalar
# port / the vector ALU / the LDS: config 3's time with each.
for flags in ${PORT_FLAGS:-"-DINF3_NONE=1" "-DINF3_DUMMY_SALU=100" "-DINF3_DUMMY_VALU=100" "-DINF3_DUMMY_VALU=400" "-DINF3_DUMMY_LDS=50"}; do
    RCX_EXTRA_FLAGS="$flags" python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
    RCX_EXTRA_FLAGS="$flags" python benchmarks/bench_configs.py --configs 3 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('$flags', j['ms'])"
done
python -c "
import sys; sys.path.insert(0, '.')
from rust_compress_amd.csrc import build
build.build()" > /dev/null 2>&1
