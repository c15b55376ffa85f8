#!/bin/bash
# timing experiments on the specialised kernels: scripts/gpu_spec_exp.sh "<exp1>" "<exp2>" ...   (exp = TCMI_SPEC_EXP string, "-" = none)
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  echo "== TCMI_SPEC_EXP='$e'"
  TCMI_SPEC_EXP="$e" TCMI_SPECIALIZE=1 python scripts/gpu_spec_fwd.py 28 12 8 2>&1 | grep -E "specialised  ms|mode 1|Error|error" | head -4
  TCMI_SPEC_EXP="$e" TCMI_SPECIALIZE=1 python scripts/gpu_spec_adj.py 28 12 8 2>&1 | grep -E "mode 1|specialised  sweep|Error|error" | tail -3
done
