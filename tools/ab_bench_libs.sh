#!/bin/bash
# A/B several PREBUILT libraries in the full bench inside ONE gpurun call (same device), alternating: R=3 tools/ab_bench_libs.sh <libA> <libB> [...]
R=${R:-3}
for r in $(seq 1 $R); do
  for l in "$@"; do
    AGD_LIB=$GRAFT_REPO_ROOT/$l python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/ab_bl.log 2>/dev/null || exit 1
    tail -1 gpurun_out/ab_bl.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l', d['value'], d['ms_per_step'])"
  done
done
