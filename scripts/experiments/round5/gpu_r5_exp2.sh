#!/bin/bash
# round 5: what bounds the dense reverse-sweep passes?  (per-pass times under the emitter's experiment switches)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_exp2
rm -rf $OUT; mkdir -p $OUT
export TCMI_SPECIALIZE=1
for E in "" "nomem" "nomem,noexch" "noexch" "waves=5" "noevents" "nomem,waves=3" "nomem,waves=2"; do
  tag=$(echo "x$E" | tr ',=' '__')
  TCMI_SPEC_EXP=$E timeout 600 python3 scripts/gpu_live_passes.py 28 12 8 > $OUT/$tag.log 2>&1
  echo "== EXP '$E'"; grep -A 4 "reverse sweep" $OUT/$tag.log; grep -A 10 "^forward" $OUT/$tag.log | tail -4
done
