#!/bin/bash
# A/B an environment knob in the full bench inside ONE gpurun call (same device): tools/ab_bench.sh VAR v0 v1 [rounds]
VAR=$1; A=$2; B=$3; R=${4:-2}
for r in $(seq 1 $R); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ab_${VAR}_$v.log 2>&1
    tail -1 gpurun_out/ab_${VAR}_$v.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['value'], d['ms_per_step'], {k: v['ms'] for k, v in d['kernel_classes'].items()})"
  done
done
