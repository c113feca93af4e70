# timing experiments (GPU box): build the experiments library once per variant ("tag:extra compiler flags") and run tools/kb_lin.py
# with each.  Example: bash tools/exp_variants.sh "base:" "s1:-DEXP_SCHED=1" "s3:-DEXP_SCHED=3"
set -e
bash tools/build_exp.sh
for v in "$@"; do
  tag=${v%%:*}; fl=${v#*:}
  if [ -n "$fl" ]; then
    rm -rf /tmp/exp_$tag && cp -r /tmp/exp /tmp/exp_$tag && cd /tmp/exp_$tag/csrc && rm -f igemm.o attention.o && make EXTRA="-DAGD_EXPERIMENTS $fl" OUT=/tmp/exp_$tag/libagenda_hip.so > /tmp/exp_$tag/build.log 2>&1; tail -1 /tmp/exp_$tag/build.log; cd $GRAFT_REPO_ROOT
  else
    rm -rf /tmp/exp_$tag && ln -s /tmp/exp /tmp/exp_$tag
  fi
done
for v in "$@"; do tag=${v%%:*}; echo "== $v"; AGD_LIB=/tmp/exp_$tag/libagenda_hip.so KB_CFGS=${KB_CFGS:-0} timeout -k 10 300 python3 ${KB_TOOL:-tools/kb_lin.py}; done
