# timing experiments (GPU box): build the experiments library once per variant ("tag:extra compiler flags") into /tmp/exp_<tag>/ and
# run a kb tool with each.  Example: bash tools/exp_variants.sh "base:" "s1:-DEXP_SCHED=1" "s3:-DEXP_SCHED=3"   (KB_TOOL=tools/kb_ff.py ...)
set -e
CS=$GRAFT_REPO_ROOT/agenda_amd/csrc
for v in "$@"; do
  tag=${v%%:*}; fl=${v#*:}
  rm -rf /tmp/exp_$tag && mkdir -p /tmp/exp_$tag && cp -r $CS /tmp/exp_$tag/csrc && mkdir -p /tmp/exp_$tag/include && cp $GRAFT_REPO_ROOT/include/agenda_hip.h /tmp/exp_$tag/include/
  (cd /tmp/exp_$tag/csrc && sed -i "s#../../include/agenda_hip.h#../include/agenda_hip.h#" model.hip Makefile && rm -rf exp *.o && make -j16 exp EXTRA="$fl" EXP_OUT=/tmp/exp_$tag/libagenda_hip_exp.so > /tmp/exp_$tag/build.log 2>&1; tail -1 /tmp/exp_$tag/build.log)
done
for v in "$@"; do tag=${v%%:*}; echo "== $v"; AGD_LIB=/tmp/exp_$tag/libagenda_hip_exp.so KB_CFGS=${KB_CFGS:-0} timeout -k 10 300 python3 ${KB_TOOL:-tools/kb_lin.py}; done
