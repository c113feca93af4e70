#!/bin/bash
# A/B two experiments libraries with one micro-benchmark tool inside ONE gpurun call, alternating: tools/ab_kb.sh <tool.py> <libA> <libB> [rounds]
T=$1; A=$2; B=$3; R=${4:-2}
for r in $(seq 1 $R); do
  for l in $A $B; do echo "== $l (round $r)"; AGD_LIB=$GRAFT_REPO_ROOT/$l timeout -k 10 300 python3 $T || exit 1; done
done
