#!/bin/bash
# round 5, first contact: A/B of the 8-byte LDS access form in the reverse sweep + the new tests
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_ab1
rm -rf $OUT; mkdir -p $OUT
python3 scripts/gpu_live_passes.py 28 12 8 > $OUT/live_single8.log 2>&1
TCMI_SPECIALIZE=1 TCMI_SPEC_EXP=single8=0 python3 scripts/gpu_live_passes.py 28 12 8 > $OUT/live_paired.log 2>&1
python3 scripts/gpu_vqe_only.py 28 12 8 3 > $OUT/vqe_only.log 2>&1
TCMI_SPARSE_START=0 python3 scripts/gpu_vqe_only.py 28 12 8 2 > $OUT/vqe_only_dense.log 2>&1
TCMI_SPARSE_START=0 TCMI_SPECIALIZE=1 TCMI_SPEC_EXP=single8=0 python3 scripts/gpu_vqe_only.py 28 12 8 2 > $OUT/vqe_only_dense_paired.log 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_multirank.py tests/test_gpu_api_gaps.py tests/test_gpu_specialize.py tests/test_gpu_grad.py -x -q -m gpu > $OUT/pytest_subset.log 2>&1
tail -3 $OUT/pytest_subset.log
tail -15 $OUT/live_single8.log | head -14
grep "reverse sweep" $OUT/live_paired.log
cat $OUT/vqe_only.log $OUT/vqe_only_dense.log $OUT/vqe_only_dense_paired.log | grep -v Warn
