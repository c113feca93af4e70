#!/bin/bash
# per-layer report only (tools/profile_all.sh's last step): one short bench run of the experiments library with the shape log, joined with its kernel trace
set -e
R=${1:-r04}; OUT=gpurun_out/prof_$R; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
AGD_LIB=$GRAFT_REPO_ROOT/agenda_amd/libagenda_hip_exp.so AGD_IGEMM_LOG=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/layers -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $OUT/bench_layers.log 2> $OUT/layers.err
LT=$(find $OUT/layers -name "*kernel_trace.csv" | head -1)
python3 tools/layer_report.py $OUT/layers.err $LT 24 > $OUT/${R}_layer_report.txt 2>&1
rm -rf $OUT/layers $OUT/layers.err
tail -25 $OUT/${R}_layer_report.txt
