"""Per-pass time of the tiled Pauli-sum (TFIM cotangent) against the flat kernel: gpu_pauli_tiled.py n [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from tcmi import _lib
from tcmi.executor import ATOMIC_COPIES, plan_pauli_passes
n = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = [(1 << (n - 1 - q), 0, 0, q) for q in range(n)] + [(0, (1 << (n - 1 - q)) | (1 << (n - 2 - q)), 0, n + q) for q in range(n - 1)]
code = _lib.TCMI_C64; T = 12
passes = plan_pauli_passes(n, rows, T)
psi = torch.randn(B, 2**n, dtype=torch.complex64, device="cuda"); out = torch.empty_like(psi)
w = torch.ones(B, len(rows), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
dots = torch.zeros(B, ATOMIC_COPIES, dtype=torch.float64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0
for i, ps in enumerate(passes):
    tp = torch.tensor(ps["tilepos"], dtype=torch.int32, device="cuda")
    tr = torch.from_numpy(np.asarray(ps["rows"], dtype=np.int64).astype(np.uint32).view(np.int32).reshape(-1, 4).copy()).cuda()
    wp = w[:, ps["order"]].contiguous()
    def run(): _lib.check(_lib.lib().tcmi_apply_pauli_sum_tiled(psi.data_ptr(), out.data_ptr(), 2**n, B, n, tp.data_ptr(), tr.data_ptr(), len(ps["order"]), ps["ndiag"], wp.data_ptr(), wp.stride(0), int(i > 0), dots.data_ptr(), dots.stride(0), ATOMIC_COPIES, code, st), "t")
    run(); torch.cuda.synchronize(); e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 5; tot += ms
    nb = (2 if i == 0 else 3) * B * 2**n * 8
    print(f"pass {i}: {len(ps['order'])} terms, tile {ps['tilepos']}: {ms*1e3:.0f} us = {nb/ms/1e6:.0f} GB/s")
print(f"tiled total {tot:.2f} ms")
order = sorted(range(len(rows)), key=lambda k: rows[k][0])
arr = np.array([[rows[k][0], rows[k][1], rows[k][2]] for k in order], dtype=np.int64).astype(np.uint32).view(np.int32)
tdev = torch.from_numpy(arr.reshape(-1, 3).copy()).cuda(); wf = w[:, order].contiguous()
def flat(): _lib.check(_lib.lib().tcmi_apply_pauli_sum(psi.data_ptr(), out.data_ptr(), 2**n, B, n, tdev.data_ptr(), len(rows), wf.data_ptr(), wf.stride(0), code, st), "f")
flat(); torch.cuda.synchronize(); e0.record()
for _ in range(5): flat()
e1.record(); torch.cuda.synchronize(); print(f"flat {e0.elapsed_time(e1)/5:.2f} ms")
