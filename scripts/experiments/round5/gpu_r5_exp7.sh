#!/bin/bash
# round 5: pass caps on the sweep plan (gates per pass limited so that the VALU-bound first pass hands work to the
# memory-bound ones), measured
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_exp7
rm -rf $OUT; mkdir -p $OUT
export TCMI_SPECIALIZE=1
for F in "3,0" "3,0,70" "3,0,56" "3,0,52" "3,0,64" "3,1,56" "3,1,64"; do
  tag=$(echo "f$F" | tr ',' '_')
  TCMI_ADJ_FORCE=$F timeout 900 python3 scripts/gpu_live_passes.py 28 12 8 > $OUT/$tag.log 2>&1
  echo "== FORCE '$F'"; grep -A 8 "reverse sweep" $OUT/$tag.log
done
