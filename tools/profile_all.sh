#!/bin/bash
# Round profile set on the GPU box (one gpurun call): kernel stats of the bench command, FETCH/WRITE PMC passes (separate
# runs, no trace domains mixed with --pmc), MFMA-busy PMC pass, per-layer report.  Outputs under gpurun_out/prof_$1/.
set -e
R=${1:-r05}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
# PMC passes: one batch of 10 DDIM steps (the per-launch counters do not depend on the step count; rocprofv3 --pmc crashed
# in its dispatch interception on the full 3-batch x 50-step run, ~45k dispatches)
PMCB="python3 bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-profile"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_stats.out 2> $OUT/bench_stats.log
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $PMCB > $OUT/bench_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $PMCB > $OUT/bench_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- $PMCB > $OUT/bench_mfma.log 2>&1
echo "mfma done"
ST=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); KT=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
FE=$(find $OUT/fetch -name "*counter_collection.csv" | head -1); WR=$(find $OUT/write -name "*counter_collection.csv" | head -1)
MF=$(find $OUT/mfma -name "*counter_collection.csv" | head -1)
cp $ST $OUT/${R}_bench_kernel_stats.csv
python3 tools/pmc_traffic.py $FE $WR $KT > $OUT/${R}_pmc_traffic_summary.csv
python3 tools/pmc_mfma.py $MF > $OUT/${R}_pmc_mfma_util.csv
grep '"metric"' $OUT/bench_stats.out | tail -1 > $OUT/${R}_bench_line_profiled.json
# per-layer report: one short run with the shape log
# (the shape log exists only in the experiments library: agenda_amd/libagenda_hip_exp.so, `make -C agenda_amd/csrc exp`)
AGD_LIB=$GRAFT_REPO_ROOT/agenda_amd/libagenda_hip_exp.so AGD_IGEMM_LOG=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/layers -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $OUT/bench_layers.log 2> $OUT/layers.err || true
LT=$(find $OUT/layers -name "*kernel_trace.csv" | head -1)
python3 tools/layer_report.py $OUT/layers.err $LT 24 > $OUT/${R}_layer_report.txt 2>&1 || true
# keep only the summaries (the raw traces are hundreds of MB)
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/mfma $OUT/layers $OUT/layers.err
ls -la $OUT
