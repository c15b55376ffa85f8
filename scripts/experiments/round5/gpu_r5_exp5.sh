#!/bin/bash
# round 5: wave priorities, and register-heavier tiles with fewer waves behind a barrier
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_exp5
rm -rf $OUT; mkdir -p $OUT
export TCMI_SPECIALIZE=1
for E in "prio=1" "prio=2" "prio=3"; do
  tag=$(echo "x$E" | tr ',=' '__')
  TCMI_SPEC_EXP=$E timeout 600 python3 scripts/gpu_live_passes.py 28 12 8 > $OUT/$tag.log 2>&1
  echo "== EXP '$E'"; grep -A 4 "reverse sweep" $OUT/$tag.log; grep -A 10 "^forward" $OUT/$tag.log | tail -5
done
for T in "5,6" "5,7"; do
  tag=$(echo "t$T" | tr ',' '_')
  TCMI_ADJ_TILE=$T timeout 900 python3 scripts/gpu_live_passes.py 28 12 8 > $OUT/$tag.log 2>&1
  echo "== ADJ TILE '$T'"; grep -A 12 "reverse sweep" $OUT/$tag.log; tail -2 $OUT/$tag.log
done
