"""A few launches of the join kernels on the headline's shape (M = N = 4096, K = 128, batch 32, with the 4 x 4 epilogue) for
rocprofv3 counter passes: tcmi_cgemm_split_f16 and tcmi_cgemm_split_epi."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tensorcircuit-ng_amd"))
import torch
from tcmi import _lib
L = _lib.lib()
M = N = 4096
K, B = int(os.environ.get("GEMM_K", "128")), int(os.environ.get("GEMM_B", "32"))
st = torch.cuda.current_stream().cuda_stream
a = torch.randn(B, K, M, 2, device="cuda"); b = torch.randn(B, K, N, 2, device="cuda")
a = a / torch.linalg.vector_norm(a, dim=(2, 3), keepdim=True); b = b / torch.linalg.vector_norm(b, dim=(2, 3), keepdim=True)
A, Bm = torch.view_as_complex(a), torch.view_as_complex(b)
c = torch.empty(B, M, N, dtype=torch.complex64, device="cuda")
X = torch.eye(4, dtype=torch.complex64, device="cuda").reshape(1, 16).repeat(B, 1).contiguous()
for _ in range(4):
    _lib.check(L.tcmi_cgemm_split_f16(A.data_ptr(), Bm.data_ptr(), c.data_ptr(), M, N, K, B, K * M, K * N, M * N, X.data_ptr(),
                                      2.0**14, 2.0**14, st), "f16")
    _lib.check(L.tcmi_cgemm_split_epi(A.data_ptr(), Bm.data_ptr(), c.data_ptr(), M, N, K, B, K * M, K * N, M * N, X.data_ptr(), st), "bf16")
torch.cuda.synchronize()
