#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for cfg in "host_trap time 1000" "host_trap time 100" "host_trap time 10" "stochastic cycles 1048576" "stochastic cycles 65536" "stochastic instructions 65536"; do
  set -- $cfg
  rm -rf /tmp/pcsp
  timeout 120 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 --output-format csv -d /tmp/pcsp -- python -c "import torch; a=torch.ones(1<<24,device='cuda'); [a.mul_(1.0001) for _ in range(200)]; torch.cuda.synchronize()" > /tmp/pcsp.log 2>&1
  echo "== $cfg: rc=$? $(grep -c . /tmp/pcsp.log) lines; $(grep -i 'not supported\|error' /tmp/pcsp.log | head -2 | cut -c1-160)"
  find /tmp/pcsp -name "*pc_sampl*" 2>/dev/null | head -3
done
rocprofv3 --list-avail 2>/dev/null | grep -i -A12 "pc sampl" | head -40
