#!/bin/bash
# round 5: XCD-aware tile order in the generated kernels
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_exp6
rm -rf $OUT; mkdir -p $OUT
export TCMI_SPECIALIZE=1
for E in "" "xcd=0" "prio=1" "xcd=0,prio=1"; do
  tag=$(echo "x$E" | tr ',=' '__')
  TCMI_SPEC_EXP=$E timeout 600 python3 scripts/gpu_live_passes.py 28 12 8 > $OUT/$tag.log 2>&1
  echo "== EXP '$E'"; grep -A 7 "reverse sweep" $OUT/$tag.log; grep -A 10 "^forward" $OUT/$tag.log | tail -5
done
