#!/bin/bash
# A/B of built libraries in the full bench inside ONE gpurun call (same device), interleaved: tools/ab_sofiles.sh ROUNDS lib1.so lib2.so ...   (each loaded through AGD_LIB)
R=$1; shift
for r in $(seq 1 $R); do
  for lib in "$@"; do
    AGD_LIB=$PWD/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['ms_per_step'])"
  done
done
