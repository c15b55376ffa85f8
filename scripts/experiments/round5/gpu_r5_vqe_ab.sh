#!/bin/bash
# round 5: the VQE step with this round's kernel changes switched off one at a time (interleaved repeats)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_vqe_ab
rm -rf $OUT; mkdir -p $OUT
export TCMI_SPECIALIZE=1
for rep in 1 2; do
for E in "" "xcd=0" "single8=0" "prio=0" "xcd=0,single8=0,prio=0"; do
  tag=$(echo "x${E}_r$rep" | tr ',=' '__')
  TCMI_SPEC_EXP=$E timeout 600 python3 scripts/gpu_vqe_only.py 28 12 8 3 > $OUT/$tag.log 2>&1
  echo "== EXP '$E' rep $rep"; grep -v "Warn\|amdgpu.ids" $OUT/$tag.log | tail -4
done
done
