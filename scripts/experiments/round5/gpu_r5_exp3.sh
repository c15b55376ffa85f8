#!/bin/bash
# round 5: the six candidate sweep plans (pinned low bits x tie-break) measured instead of modelled
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_exp3
rm -rf $OUT; mkdir -p $OUT
export TCMI_SPECIALIZE=1
for F in "3,0" "3,1" "4,0" "4,1" "5,0" "5,1"; do
  tag=$(echo "f$F" | tr ',' '_')
  TCMI_ADJ_FORCE=$F timeout 900 python3 scripts/gpu_live_passes.py 28 12 8 > $OUT/$tag.log 2>&1
  echo "== FORCE '$F'"; grep -A 10 "reverse sweep" $OUT/$tag.log
done
