#!/bin/bash
# Counter table for the 6.2x fetch of the split-K 16 x 16 conv: L2 requests, fabric reads, L2 hit / miss per launch for
# the row-halo kernel (A-major), the general kernel A-major and the general kernel W-major.  Separate --pmc passes, no trace domains.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_conv16; mkdir -p $O
for form in halo gen:0 gen:1; do
  f=${form%%:*}; w=${form##*:}; [ "$f" = "$w" ] && w=""
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum TCC_TAG_STALL_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
    i=$((i+1))
    if [ -n "$w" ]; then export AGD_IGEMM_WMAJOR=$w; else unset AGD_IGEMM_WMAJOR; fi
    KB_FORM=$f rocprofv3 --pmc $set --output-format csv -d $O/p -- python3 tools/kb_conv16.py > $O/${form}_$i.log 2>&1 || echo "pass $form $i failed"
    c=$(find $O/p -name "*counter_collection.csv" | head -1)
    if [ -n "$c" ]; then python3 tools/pmc_generic.py $c | grep -v "fill_random\|rocclr" | sed "s/^/$form,/" >> $O/table.csv; fi
    rm -rf $O/p
  done
done
cat $O/table.csv
