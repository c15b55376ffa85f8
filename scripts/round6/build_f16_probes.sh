#!/bin/bash
# probe libraries for scripts/round6/gpu_gemm_f16_modes.py: libtcmi_probe.so (TCMI_SPLIT_MODE instantiations) and variants of
# it with a different number of vector instructions scheduled behind each MFMA of the f16 kernel (TCMI_S2_VPM2).  Built on
# the host (hipcc cross-compiles), never loaded by the product.
set -e
cd "$(dirname "$0")/../../tensorcircuit-ng_amd/csrc"
make libtcmi.so libtcmi_probe.so >/dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -simplifycfg-sink-common=false -fno-slp-vectorize -munsafe-fp-atomics"
OBJS="tcmi_vm.o tcmi_vm2.o tcmi_measure2.o tcmi_adjoint.o tcmi_adjoint2.o tcmi_hsum.o tcmi_tensordot.o tcmi_mps.o tcmi_host.o tcmi_spec.o tcmi_comm.o"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DTCMI_SPLIT_PROBE -DTCMI_S2_VPM2=$v -c tcmi_gemm_split.hip -o /tmp/split_vpm2_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $OBJS /tmp/split_vpm2_$v.o -ldl -o libtcmi_probe_vpm2_$v.so
done
ls -la libtcmi_probe*.so
