#!/bin/bash
# round 5: smaller workgroups (fewer waves behind one barrier, more workgroups per CU)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_exp4
rm -rf $OUT; mkdir -p $OUT
export TCMI_SPECIALIZE=1
echo "== adjoint LT=7"; TCMI_ADJ_LT=7 timeout 900 python3 scripts/gpu_live_passes.py 28 12 8 > $OUT/adj_lt7.log 2>&1; grep -A 11 "reverse sweep" $OUT/adj_lt7.log; tail -3 $OUT/adj_lt7.log
echo "== forward R5 LT7"; TCMI_FWD_TILE=5,7 timeout 900 python3 scripts/gpu_live_passes.py 28 12 8 > $OUT/fwd_lt7.log 2>&1; grep -A 14 "^forward" $OUT/fwd_lt7.log; tail -3 $OUT/fwd_lt7.log
echo "== forward R4 LT7"; TCMI_FWD_TILE=4,7 timeout 900 python3 scripts/gpu_live_passes.py 28 12 8 > $OUT/fwd_r4lt7.log 2>&1; grep -A 14 "^forward" $OUT/fwd_r4lt7.log; tail -3 $OUT/fwd_r4lt7.log
echo "== adjoint LT=6"; TCMI_ADJ_LT=6 timeout 900 python3 scripts/gpu_live_passes.py 28 12 8 > $OUT/adj_lt6.log 2>&1; grep -A 22 "reverse sweep" $OUT/adj_lt6.log; tail -3 $OUT/adj_lt6.log
