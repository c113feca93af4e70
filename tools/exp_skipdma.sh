# timing experiment (GPU box): how much of a launch is LDS-DMA issue / ingest?  Builds the experiments library three more times with
# the main loop's A pieces, B pieces or all pieces compiled out (results are garbage) and times the SD-1.5 shapes with each.
set -e
bash tools/build_exp.sh
for v in 1 2 3; do
  rm -rf /tmp/exp$v && cp -r /tmp/exp /tmp/exp$v && cd /tmp/exp$v/csrc && rm -f igemm.o && make EXTRA="-DAGD_EXPERIMENTS -DEXP_SKIP_DMA=$v" OUT=/tmp/exp$v/libagenda_hip.so > /tmp/exp$v/build.log 2>&1; tail -1 /tmp/exp$v/build.log; cd $GRAFT_REPO_ROOT
done
for v in "" 1 2 3; do echo "== EXP_SKIP_DMA=${v:-0}"; AGD_LIB=/tmp/exp$v/libagenda_hip.so KB_CFGS=0 timeout -k 10 300 python3 tools/kb_lin.py; done
