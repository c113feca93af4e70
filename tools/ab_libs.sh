#!/bin/bash
# A/B two builds of the library in the full bench inside ONE gpurun call (same device): tools/ab_libs.sh "tagA:flags" "tagB:flags" [rounds]
# (experiments builds, tools/exp_variants.sh naming; bench.py loads each through AGD_LIB)
A=$1; B=$2; R=${3:-3}
KB_CFGS=0 KB_TOOL=/dev/null bash tools/exp_variants.sh "$A" "$B" > gpurun_out/ab_libs_build.log 2>&1
for r in $(seq 1 $R); do
  for v in "$A" "$B"; do
    tag=${v%%:*}
    AGD_LIB=/tmp/exp_$tag/libagenda_hip_exp.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/ab_lib_$tag.log 2>/dev/null
    tail -1 gpurun_out/ab_lib_$tag.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag', d['value'], d['ms_per_step'])"
  done
done
