#!/bin/bash
# Every dispatch of one BWT forward pass in order (rocprofv3 --kernel-trace): bash benchmarks/bwt_forward_ktrace.sh [kind] [nblocks]
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt -- python $REPO/benchmarks/bwt_forward_profile.py ${1:-text} ${2:-1024} > /tmp/kt.log 2>&1
python3 $REPO/benchmarks/ktrace.py /tmp/kt
