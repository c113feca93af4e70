#!/bin/bash
# usage: _tmp_smap.sh  (run on the GPU box from the repo root)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in 0 240 256 512 768 752; do for k in 1280 2560; do
  d=$R/gpurun_out/smapprof/c${c}_k${k}
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/_tmp_smap.py $c $k > $d.out 2>&1 || exit 1
  echo "== cfg $c Cin $k: $(grep -v rocprof $d.out | tail -1)"
  find $d -name "*kernel_stats.csv" | xargs grep -h "smap\|splitk\|igemm" | cut -d, -f1-4
done; done
