#!/bin/bash
# FETCH_SIZE (bytes the L2s fetch from beyond them) per launch of the row-halo 3x3 kernels, with / without XCD tile blocks and on the producer / consumer kernel
# (tools/kb_fetch.py runs 10 launches per (shape, mode); rocprofv3 --pmc alone, no trace domains).  On the GPU box: tools/kb_fetch.sh > gpurun_out/fetch.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/fetchprof
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d -- python3 $R/tools/kb_fetch.py > $d.out 2>&1 || exit 1
grep -v rocprof $d.out | tail -14
f=$(find $d -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE"]
ker = [r for r in rows if "igemm" in r["Kernel_Name"]]
# 10 launches (2 warm + 8) per (shape, mode), in order
for i in range(0, len(ker), 10):
    g = ker[i:i + 10]
    print(i // 10, g[0]["Kernel_Name"][:60], "fetch MB per launch: %.1f" % (sum(float(r["Counter_Value"]) for r in g) / len(g) * 1024 * 2 / 1e6))
PY
