#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per rocprofv3 pass) of BASELINE configs 3, 4, 5 -> gpurun_out/pmc_cfg{3,4,5}.json
# (copy them to profiles/ once checked).  bash benchmarks/pmc_configs.sh "3 4 5"
REPO=$(pwd)
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
pass() {   # pass <tag> <counter> <bench_configs args...>
    local tag=$1 c=$2; shift 2
    rm -rf /tmp/pc_$tag
    timeout 900 rocprofv3 --pmc $c -d /tmp/pc_$tag -- python $REPO/benchmarks/bench_configs.py "$@" --once > /tmp/pc_$tag.log 2>&1
    find /tmp/pc_$tag -name "*.db" | head -1
}
for cfg in ${1:-3 4 5}; do
    if [ "$cfg" = "4" ]; then
        FT=$(pass 4ft FETCH_SIZE --configs 4 --kinds text); WT=$(pass 4wt WRITE_SIZE --configs 4 --kinds text)
        FD=$(pass 4fd FETCH_SIZE --configs 4 --kinds dna4); WD=$(pass 4wd WRITE_SIZE --configs 4 --kinds dna4)
        python $REPO/benchmarks/pmc_configs.py 4 $FT $WT $REPO/gpurun_out/pmc_cfg4.json dna4 $FD $WD || tail -5 /tmp/pc_4ft.log
    else
        F=$(pass ${cfg}f FETCH_SIZE --configs $cfg); W=$(pass ${cfg}w WRITE_SIZE --configs $cfg)
        python $REPO/benchmarks/pmc_configs.py $cfg $F $W $REPO/gpurun_out/pmc_cfg$cfg.json || tail -5 /tmp/pc_${cfg}f.log
    fi
done
