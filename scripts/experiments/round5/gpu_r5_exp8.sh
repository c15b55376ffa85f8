#!/bin/bash
# round 5: nontemporal hints on tiles of sub-line runs
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_exp8
rm -rf $OUT; mkdir -p $OUT
export TCMI_SPECIALIZE=1
for E in "" "ntl=0" "nts=0" "ntl=0,nts=0"; do
  tag=$(echo "x$E" | tr ',=' '__')
  TCMI_SPEC_EXP=$E timeout 600 python3 scripts/gpu_live_passes.py 28 12 8 > $OUT/$tag.log 2>&1
  echo "== EXP '$E'"; grep -A 7 "reverse sweep" $OUT/$tag.log; grep -A 10 "^forward" $OUT/$tag.log | tail -4
done
