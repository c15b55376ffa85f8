"""Device linear algebra of the MPS path: thin wrappers (torch tensors in, torch tensors out) over the
C-ABI entry points ``tcmi_cgemm``, ``tcmi_svd_trunc_batched``, ``tcmi_qr_batched`` and
``tcmi_mps_gate_mix``.  These are what ``backend.svd / qr / rq / matmul`` resolve to on the hip backend
(reference call sites: ``mps_base.py:123-175``, ``mpscircuit.py:35-64``; truncation rule
``backends/jax_backend.py:62-112``).  torch supplies memory and views only; there is no fallback to
``torch.linalg`` — without ``libtcmi.so`` every function raises ``TcmiError``.
"""

from typing import Any, Dict, Optional, Tuple

from . import _lib

_WORK: Dict[Any, Any] = {}
_CHECK_SVD = __import__("os").environ.get("TCMI_CHECK_SVD", "0") == "1"


def _code(t):
    import torch

    if t.dtype == torch.complex64:
        return _lib.TCMI_C64
    if t.dtype == torch.complex128:
        return _lib.TCMI_C128
    raise TypeError(f"tcmi linalg needs complex64/complex128 tensors, got {t.dtype}")


def _stream(t):
    import torch

    return torch.cuda.current_stream(t.device).cuda_stream


def _workspace(kind, nbytes, device):
    import torch

    key = (kind, device)
    w = _WORK.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _WORK[key] = w
    return w


def matmul(a, b):
    """[M,K] @ [K,N] (or batched [B,M,K] @ [B,K,N]) through ``tcmi_cgemm``."""
    import torch

    a, b = a.contiguous(), b.contiguous()
    if a.dim() == 2:
        M, K = a.shape
        K2, N = b.shape
        assert K == K2, (a.shape, b.shape)
        c = torch.empty((M, N), dtype=a.dtype, device=a.device)
        _lib.check(_lib.lib().tcmi_cgemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, 1, 0, 0, 0, 0, _code(a),
                                         _stream(a)), "tcmi_cgemm")
        return c
    B, M, K = a.shape
    _, K2, N = b.shape
    assert K == K2 and b.shape[0] == B
    c = torch.empty((B, M, N), dtype=a.dtype, device=a.device)
    _lib.check(_lib.lib().tcmi_cgemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, B, M * K, K * N, M * N, 0,
                                     _code(a), _stream(a)), "tcmi_cgemm")
    return c


def site_gate(gate, tensor):
    """``ncon([gate, A], [[-2, 1], [-1, 1, -3]])``: out[l,a,r] = sum_b gate[a,b] A[l,b,r] — one batched
    (d x d)(d x r) GEMM with the gate shared by every l (stride 0)."""
    import torch

    tensor = tensor.contiguous()
    gate = gate.contiguous()
    l, d, r = tensor.shape
    out = torch.empty_like(tensor)
    _lib.check(_lib.lib().tcmi_cgemm(gate.data_ptr(), tensor.data_ptr(), out.data_ptr(), d, r, d, l, 0, d * r, d * r,
                                     0, _code(tensor), _stream(tensor)), "tcmi_cgemm")
    return out


def gate_mix(t, gate, L, R):
    """theta[l,a',b',r] = sum_ab gate[a',b',a,b] t[l,a,b,r]; t flat [L*4*R]."""
    import torch

    t = t.contiguous()
    gate = gate.contiguous()
    out = torch.empty_like(t)
    _lib.check(_lib.lib().tcmi_mps_gate_mix(t.data_ptr(), gate.data_ptr(), out.data_ptr(), L, R, 1, 0, _code(t),
                                            _stream(t)), "tcmi_mps_gate_mix")
    return out


def _svd_rows(mat, kmax, max_sv, max_err, relative, absorb):
    """SVD of a [m, n] matrix with m <= n.  Returns u [m,kmax], s [m], vh [kmax,n], keep (device int32
    [1]), tw2 (device real [1])."""
    import torch

    mat = mat.contiguous()
    m, n = mat.shape
    rdt = torch.float32 if mat.dtype == torch.complex64 else torch.float64
    code = _code(mat)
    nbytes = _lib.lib().tcmi_svd_work_bytes(m, n, 1, code)
    if nbytes < 0:
        raise _lib.TcmiError("tcmi_svd_work_bytes: bad arguments")
    work = _workspace("svd", nbytes, mat.device)
    u = torch.empty((m, kmax), dtype=mat.dtype, device=mat.device)
    s = torch.empty((m,), dtype=rdt, device=mat.device)
    vh = torch.empty((kmax, n), dtype=mat.dtype, device=mat.device)
    keep = torch.empty((1,), dtype=torch.int32, device=mat.device)
    tw2 = torch.empty((1,), dtype=rdt, device=mat.device)
    _lib.check(_lib.lib().tcmi_svd_trunc_batched(
        mat.data_ptr(), u.data_ptr(), s.data_ptr(), vh.data_ptr(), keep.data_ptr(), tw2.data_ptr(), m, n, kmax, 1,
        int(max_sv or 0), float(-1.0 if max_err is None else max_err), int(bool(relative)), absorb, 0,
        work.data_ptr(), work.numel(), code, _stream(mat)), "tcmi_svd_trunc_batched")
    return u, s, vh, keep, tw2


def last_svd_status(device=None) -> int:
    """0 if the most recent SVD launch on ``device`` completed normally, 1 if its inter-workgroup barrier
    timed out (results invalid).  Synchronises; meant for tests and debugging (``TCMI_CHECK_SVD=1`` makes
    every ``svd_trunc`` call check it and raise)."""
    import torch

    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    w = _WORK.get(("svd", device))
    if w is None:
        return 0
    return int(w[4:8].view(torch.int32).item())


def svd_trunc(mat, max_singular_values: Optional[int] = None, max_truncation_err: Optional[float] = None,
              relative: bool = False, absorb: int = 0) -> Tuple[Any, Any, Any, Any]:
    """``backend.svd(mat, pivot_axis=1, ...)`` with the reference truncation rule.  Returns
    ``(u, s, vh, s_rest)``; ``absorb`` 1 / 2 folds ``s`` into ``u`` / ``vh`` inside the kernel (the
    TEBD update's ``U * S`` / ``S * V``, mps_base.py:136-146).  When only ``max_singular_values`` is
    given the kept rank is known on the host and nothing synchronises; ``max_truncation_err`` needs
    the device's count (one 4-byte read-back)."""
    m, n = mat.shape
    k = min(m, n)
    static_keep = k if max_singular_values is None else min(int(max_singular_values), k)
    if m <= n:
        u, s, vh, keep, _ = _svd_rows(mat, static_keep, max_singular_values, max_truncation_err, relative, absorb)
    else:
        # mat^T = U' S V'h  ->  mat = V'h^T S U'^T
        sw = {0: 0, 1: 2, 2: 1}[absorb]
        u2, s, vh2, keep, _ = _svd_rows(mat.t(), static_keep, max_singular_values, max_truncation_err, relative, sw)
        u, vh = vh2.t().contiguous(), u2.t().contiguous()
    if _CHECK_SVD and last_svd_status(mat.device) != 0:
        raise _lib.TcmiError("tcmi_svd_trunc_batched: inter-workgroup barrier timed out")
    kk = static_keep
    if max_truncation_err is not None:
        kk = int(keep.item())
        if kk < static_keep:
            u, vh = u[:, :kk].contiguous(), vh[:kk, :].contiguous()
    return u, s[:kk].to(mat.dtype), vh, s[kk:].to(mat.dtype)


def qr(mat):
    """Householder QR: [m,n] -> q [m,K], r [K,n] (complete isometry also for rank-deficient input)."""
    import torch

    mat = mat.contiguous()
    m, n = mat.shape
    K = min(m, n)
    code = _code(mat)
    nbytes = _lib.lib().tcmi_qr_work_bytes(m, n, 1, code)
    work = _workspace("qr", nbytes, mat.device)
    q = torch.empty((m, K), dtype=mat.dtype, device=mat.device)
    r = torch.empty((K, n), dtype=mat.dtype, device=mat.device)
    _lib.check(_lib.lib().tcmi_qr_batched(mat.data_ptr(), q.data_ptr(), r.data_ptr(), m, n, 1, work.data_ptr(),
                                          work.numel(), code, _stream(mat)), "tcmi_qr_batched")
    return q, r


def rq(mat):
    """mat = r q with q q^H = 1 (tensornetwork ``rq``: QR of the conjugate transpose)."""
    q, r = qr(mat.conj().t().resolve_conj())
    return r.conj().t().resolve_conj().contiguous(), q.conj().t().resolve_conj().contiguous()


def einsum2(expr: str, a, b):
    """Two-operand einsum without repeated / batch labels: permute (views), one ``tcmi_cgemm``, permute."""
    lhs, out = expr.split("->")
    la, lb = lhs.split(",")
    con = [c for c in la if c in lb and c not in out]
    fa = [c for c in la if c not in con]
    fb = [c for c in lb if c not in con]
    assert all(c in out for c in fa + fb) and len(out) == len(fa) + len(fb), expr
    a2 = a.permute([la.index(c) for c in fa + con]).contiguous()
    b2 = b.permute([lb.index(c) for c in con + fb]).contiguous()
    sa = [a.shape[la.index(c)] for c in fa]
    sb = [b.shape[lb.index(c)] for c in fb]
    K = 1
    for c in con:
        K *= a.shape[la.index(c)]
    M = 1
    for v in sa:
        M *= v
    N = 1
    for v in sb:
        N *= v
    c2 = matmul(a2.reshape(M, K), b2.reshape(K, N)).reshape(sa + sb)
    cur = fa + fb
    return c2.permute([cur.index(c) for c in out])
