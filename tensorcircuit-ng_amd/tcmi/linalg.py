"""Device linear algebra of the MPS path: thin wrappers (torch tensors in, torch tensors out) over the
C-ABI entry points ``tcmi_cgemm``, ``tcmi_svd_trunc_batched``, ``tcmi_qr_batched`` and
``tcmi_mps_gate_mix``.  These are what ``backend.svd / qr / rq / matmul`` resolve to on the hip backend
(reference call sites: ``mps_base.py:123-175``, ``mpscircuit.py:35-64``; truncation rule
``backends/jax_backend.py:62-112``).  torch supplies memory and views only; there is no fallback to
``torch.linalg`` — without ``libtcmi.so`` every function raises ``TcmiError``.
"""

import os
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


def _devkey(device):
    """Canonical dictionary key of a device: ``cuda`` without an index means the current device."""
    import torch

    if device is None:
        return ("cuda", torch.cuda.current_device())
    device = torch.device(device)
    return (device.type, device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else 0))


def _workspace(kind, nbytes, device):
    import torch

    key = (kind, _devkey(device))
    w = _WORK.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _WORK[key] = w
    return w


def _tracked(*ts) -> bool:
    """True when a result must stay on torch's autograd tape: an operand requires grad (and grad mode is
    on) or is a ``torch.func`` wrapper, whose storage the C ABI cannot be handed directly."""
    import torch

    for t in ts:
        if torch.is_tensor(t):
            if torch._C._functorch.is_functorch_wrapped_tensor(t):
                return True
            if t.requires_grad and torch.is_grad_enabled():
                return True
    return False


def _h(x):
    return x.mH.resolve_conj()


def _timed(tag, launches, work):
    """bench.py's HIP-event log around a launch (no-op unless ``executor.EVENT_LOG`` is a list)."""
    from .executor import _timed as T

    return T(tag, launches, work)


def matmul(a, b):
    """[M,K] @ [K,N] (or batched [B,M,K] @ [B,K,N]) through ``tcmi_cgemm``; differentiable."""
    if _tracked(a, b):
        return _ad()["matmul"](a, b)
    return _matmul_raw(a, b)


def _matmul_raw(a, b):
    import torch

    a, b = a.resolve_conj().contiguous(), b.resolve_conj().contiguous()
    if a.dim() == 2:
        M, K = a.shape
        K2, N = b.shape
        assert K == K2, (a.shape, b.shape)
        c = torch.empty((M, N), dtype=a.dtype, device=a.device)
        with _timed("mps_gemm", 1, 8.0 * M * N * K):
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

    if _tracked(gate, tensor):
        l, d, r = tensor.shape
        out = matmul(gate, tensor.permute(1, 0, 2).reshape(d, l * r))
        return out.reshape(d, l, r).permute(1, 0, 2).contiguous()
    return _site_gate_raw(gate, tensor)


def _site_gate_raw(gate, tensor):
    import torch

    tensor = tensor.resolve_conj().contiguous()
    gate = gate.resolve_conj().contiguous()
    l, d, r = tensor.shape
    out = torch.empty_like(tensor)
    with _timed("mps_gemm", 1, 8.0 * d * d * r * l):
        _lib.check(_lib.lib().tcmi_cgemm(gate.data_ptr(), tensor.data_ptr(), out.data_ptr(), d, r, d, l, 0, d * r, d * r,
                                         0, _code(tensor), _stream(tensor)), "tcmi_cgemm")
    return out


def gate_mix(t, gate, L, R):
    """theta[l,a',b',r] = sum_ab gate[a',b',a,b] t[l,a,b,r]; t flat [L*4*R]."""
    import torch

    if _tracked(t, gate):
        out = matmul(gate.reshape(4, 4), t.reshape(L, 4, R).permute(1, 0, 2).reshape(4, L * R))
        return out.reshape(4, L, R).permute(1, 0, 2).reshape(-1)
    return _gate_mix_raw(t, gate, L, R)


def _gate_mix_raw(t, gate, L, R):
    import torch

    t = t.resolve_conj().contiguous()
    gate = gate.resolve_conj().contiguous()
    out = torch.empty_like(t)
    with _timed("mps_mix", 1, 2.0 * t.numel() * t.element_size()):
        _lib.check(_lib.lib().tcmi_mps_gate_mix(t.data_ptr(), gate.data_ptr(), out.data_ptr(), L, R, 1, 0, _code(t),
                                                _stream(t)), "tcmi_mps_gate_mix")
    return out


def _svd_rows(mat, kmax, max_sv, max_err, relative, absorb):
    """SVD of a [m, n] matrix with m <= n, or of a stack [B, m, n] of them in ONE call (the ``batch`` argument of the
    ABI: the matrices of a launch run side by side, 16 workgroups each at 256 x 256, so up to 16 chains cost the time
    of one).  Returns u [.., m,kmax], s [.., m], vh [.., kmax,n], keep (device int32 [B]), tw2 (device real [B])."""
    import torch

    mat = mat.resolve_conj().contiguous()
    lead = tuple(mat.shape[:-2])
    B = int(lead[0]) if lead else 1
    m, n = mat.shape[-2:]
    rdt = torch.float32 if mat.dtype == torch.complex64 else torch.float64
    code = _code(mat)
    nbytes = _lib.lib().tcmi_svd_work_bytes(m, n, B, code)
    if nbytes < 0:
        raise _lib.TcmiError("tcmi_svd_work_bytes: bad arguments")
    work = _workspace("svd", nbytes, mat.device)
    u = torch.empty(lead + (m, kmax), dtype=mat.dtype, device=mat.device)
    s = torch.empty(lead + (m,), dtype=rdt, device=mat.device)
    vh = torch.empty(lead + (kmax, n), dtype=mat.dtype, device=mat.device)
    keep = torch.empty((B,), dtype=torch.int32, device=mat.device)
    tw2 = torch.empty((B,), dtype=rdt, device=mat.device)
    with _timed("mps_svd", 1, 0.0):
        _lib.check(_lib.lib().tcmi_svd_trunc_batched(
            mat.data_ptr(), u.data_ptr(), s.data_ptr(), vh.data_ptr(), keep.data_ptr(), tw2.data_ptr(), m, n, kmax, B,
            int(max_sv or 0), float(-1.0 if max_err is None else max_err), int(bool(relative)), absorb, 0,
            work.data_ptr(), work.numel(), code, _stream(mat)), "tcmi_svd_trunc_batched")
    return u, s, vh, keep, tw2


def last_svd_status(device=None) -> int:
    """0 if the most recent SVD launch on ``device`` completed normally, 1 if its inter-workgroup barrier
    timed out (results invalid).  Synchronises; meant for tests and debugging (``TCMI_CHECK_SVD=1`` makes
    every ``svd_trunc`` call check it and raise)."""
    import torch

    w = _WORK.get(("svd", _devkey(device)))
    if w is None:
        return 0
    return int(w[4:8].view(torch.int32).item())


_SVD_PENDING: Dict[Any, bool] = {}    # device -> SVD launches since the last status read-back


def svd_health_check(device=None) -> None:
    """Raise if an SVD launched since the last check ended on a timed-out inter-workgroup barrier (its factors are
    poisoned: s = NaN, kept rank -1, discarded weight NaN).  ``svd_trunc`` with only ``max_singular_values`` never
    reads anything back, so the MPS front end calls this where it hands results to the user (one 4-byte read,
    only when SVDs ran since the previous check)."""
    import torch

    if not _SVD_PENDING.get(_devkey(device)):
        return
    _SVD_PENDING[_devkey(device)] = False
    if last_svd_status(device) != 0:
        raise _lib.TcmiError("tcmi_svd_trunc_batched: inter-workgroup barrier timed out (workgroups not co-resident); "
                             "the factors of that decomposition are invalid")


# QR-preconditioned Jacobi (Drmac / Veselic): ``SVD_PRECONDITION = True`` (or TCMI_SVD_PRECOND=1) sends every
# ``svd_trunc`` of an untaped complex matrix through ``A^H = Q R`` first.  Plain one-sided Jacobi needs more sweeps the
# more the spectrum is graded (256 x 256 complex64: 11 / 13 / 19 / 27 sweeps of 0.275 ms for spectra graded over 1 / 2 /
# 4 / 6 decades); on the triangular factor it needs 10 whatever the grading.  The QR (two register-resident panels +
# GEMMs) costs ~2 ms, so the switch pays from about 3.5 decades on and is off by default: the bond matrices of random
# circuits have flat spectra (scripts/gpu_svd_precond.py, bench ``mps_tebd.graded``).
SVD_PRECONDITION = os.environ.get("TCMI_SVD_PRECOND", "0") == "1"


def last_svd_sweeps(device=None) -> int:
    """Sweeps the most recent SVD launch on ``device`` needed (synchronises; tests and the bench's graded leg)."""
    import torch

    w = _WORK.get(("svd", _devkey(device)))
    if w is None:
        return 0
    ctl = w[:256].view(torch.int32).cpu().numpy()
    return int((ctl[2:62] > 0).sum()) + 1


def qr_two_panels(mat):
    """QR of a tall or square [m, n] matrix with 128 < n <= 256 <= ... as two panels the register-resident kernel takes
    (n <= 128 each): block Gram-Schmidt with one re-orthogonalisation of the second panel against the first ("twice is
    enough").  Returns (q [m, n], r [n, n]) with q^H q = 1 to working precision."""
    import torch

    m, n = mat.shape
    h = n // 2
    a1, a2 = mat[:, :h].contiguous(), mat[:, h:].contiguous()
    q1, r11 = _qr_raw(a1)
    q1h = q1.conj().t().contiguous()
    r12 = _matmul_raw(q1h, a2)
    w = a2 - _matmul_raw(q1, r12)
    c = _matmul_raw(q1h, w)
    w = w - _matmul_raw(q1, c)
    r12 = r12 + c
    q2, r22 = _qr_raw(w.contiguous())
    # columns of w below the working precision are completed to an isometry arbitrarily -- not orthogonally to q1
    # (|q1^H q2| = 4e-5 for a spectrum graded over six decades): project once more and move the component into r12
    # (q2 stays orthonormal to the square of that number)
    c2 = _matmul_raw(q1h, q2)
    q2 = q2 - _matmul_raw(q1, c2)
    r12 = r12 + _matmul_raw(c2, r22)
    q = torch.cat([q1, q2], dim=1)
    r = torch.zeros(n, n, dtype=mat.dtype, device=mat.device)
    r[:h, :h] = r11
    r[:h, h:] = r12
    r[h:, h:] = r22
    return q, r


def _svd_preconditioned(mat, static_keep, max_singular_values, max_truncation_err, relative, absorb):
    """m <= n: ``mat^H = Q R`` (Q [n, m], R [m, m]), Jacobi on R = U_R S V_R^h, then mat = R^H Q^H = V_R S (Q U_R)^H."""
    m, n = mat.shape
    ah = mat.conj().t().contiguous()
    q, r = qr_two_panels(ah) if m > 128 else _qr_raw(ah)
    sw = {0: 0, 1: 2, 2: 1}[absorb]          # U_A S = (S V_R^h)^H, S Vh_A = (Q U_R S)^H
    ur, s, vhr, keep, tw2 = _svd_rows(r, static_keep, max_singular_values, max_truncation_err, relative, sw)
    u = vhr.conj().t().contiguous()
    vh = _matmul_raw(q, ur).conj().t().contiguous()
    return u, s, vh, keep, tw2


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
    if _tracked(mat):
        return _svd_trunc_ad(mat, static_keep, max_truncation_err, relative, absorb)
    _SVD_PENDING[_devkey(mat.device)] = True
    if SVD_PRECONDITION and mat.dim() == 2 and str(mat.dtype) == "torch.complex64" and 16 <= min(m, n) and max(m, n) <= 256:
        if m <= n:
            u, s, vh, keep, tw2 = _svd_preconditioned(mat, static_keep, max_singular_values, max_truncation_err, relative, absorb)
        else:
            sw = {0: 0, 1: 2, 2: 1}[absorb]
            u2, s, vh2, keep, tw2 = _svd_preconditioned(mat.t(), static_keep, max_singular_values, max_truncation_err, relative, sw)
            u, vh = vh2.t().contiguous(), u2.t().contiguous()
    elif m <= n:
        u, s, vh, keep, tw2 = _svd_rows(mat, static_keep, max_singular_values, max_truncation_err, relative, absorb)
    else:
        # mat^T = U' S V'h  ->  mat = V'h^T S U'^T
        sw = {0: 0, 1: 2, 2: 1}[absorb]
        u2, s, vh2, keep, tw2 = _svd_rows(mat.t(), static_keep, max_singular_values, max_truncation_err, relative, sw)
        u, vh = vh2.t().contiguous(), u2.t().contiguous()
    if _CHECK_SVD and last_svd_status(mat.device) != 0:
        raise _lib.TcmiError("tcmi_svd_trunc_batched: inter-workgroup barrier timed out")
    kk = static_keep
    if max_truncation_err is not None:
        kk = int(keep.item())
        if kk < 0:   # the kernel poisons its outputs when its inter-workgroup barrier times out
            raise _lib.TcmiError("tcmi_svd_trunc_batched: inter-workgroup barrier timed out (workgroups not co-resident)")
        if kk < static_keep:
            u, vh = u[:, :kk].contiguous(), vh[:kk, :].contiguous()
    rest = s[kk:].to(mat.dtype)
    rest._tcmi_tw2 = tw2          # sum of the squared discarded values, already reduced inside the kernel
    return u, s[:kk].to(mat.dtype), vh, rest


def qr(mat):
    """Householder QR: [m,n] -> q [m,K], r [K,n] (complete isometry also for rank-deficient input).
    On the autograd tape the factors are gauge-fixed to a real non-negative diagonal of ``r`` (the form
    the backward rule assumes) and the rule of ``jax_ops.py:84-150`` applies."""
    if _tracked(mat):
        return _ad()["qr"](mat)
    return _qr_raw(mat)


def _qr_raw(mat):
    import torch

    mat = mat.resolve_conj().contiguous()
    lead = tuple(mat.shape[:-2])                       # [B] for a stack of matrices: one batched launch
    B = int(lead[0]) if lead else 1
    m, n = mat.shape[-2:]
    K = min(m, n)
    code = _code(mat)
    nbytes = _lib.lib().tcmi_qr_work_bytes(m, n, B, code)
    work = _workspace("qr", nbytes, mat.device)
    q = torch.empty(lead + (m, K), dtype=mat.dtype, device=mat.device)
    r = torch.empty(lead + (K, n), dtype=mat.dtype, device=mat.device)
    with _timed("mps_qr", 1, 0.0):
        _lib.check(_lib.lib().tcmi_qr_batched(mat.data_ptr(), q.data_ptr(), r.data_ptr(), m, n, B, work.data_ptr(),
                                              work.numel(), code, _stream(mat)), "tcmi_qr_batched")
    return q, r


def rq(mat):
    """mat = r q with q q^H = 1 (tensornetwork ``rq``: QR of the conjugate transpose)."""
    q, r = qr(_h(mat))
    return _h(r).contiguous(), _h(q).contiguous()


# ----------------------------------------------------------------------------- reverse mode
# The MPS path under ``backend.value_and_grad`` (reference tests/test_mpscircuit.py:437-497).  Forward
# passes are the same kernels; the backward rules below are written in terms of ``matmul`` (so every
# product is again ``tcmi_cgemm``) and elementwise torch ops.  Regularisation follows the reference's
# AD-aware rules (``backends/jax_ops.py:24-25, 80, 107-113``): x / (x^2 + 1e-15) for the reciprocal
# gaps and singular values, |r_ii| clamped from below at 1e-8 in the QR rule.
_AD: Dict[str, Any] = {}
_SVD_EPS = 1e-15
_QR_EPS = 1e-8


def _svd_full_raw(mat):
    """thin SVD of [m, n] or of a stack [B, m, n] (one batched launch)"""
    m, n = mat.shape[-2:]
    if m <= n:
        u, s, vh, _, _ = _svd_rows(mat, m, None, None, False, 0)
        return u, s, vh
    u2, s, vh2, _, _ = _svd_rows(mat.transpose(-1, -2), n, None, None, False, 0)
    return vh2.transpose(-1, -2).contiguous(), s, u2.transpose(-1, -2).contiguous()


def _svd_backward(u, s, vh, gu, gs, gvh):
    """dL/dA of the thin SVD A = u diag(s) vh (u [m,k], vh [k,n]); arXiv:1909.02659 with the gauge term on
    the diagonal, as in ``jax_ops.py:33-75``."""
    import torch

    m, k = u.shape[-2:]                                # leading batch dimension allowed (stacked chains)
    n = vh.shape[-1]
    cdt = u.dtype
    v, gv, uh = _h(vh), _h(gvh), _h(u)
    s2 = s * s
    E = s2[..., None, :] - s2[..., :, None]            # E_ij = s_j^2 - s_i^2, zero on the diagonal
    F = (E / (E * E + _SVD_EPS)).to(cdt)
    sinv = (s / (s * s + _SVD_EPS)).to(cdt)
    sc = s.to(cdt)
    GU = matmul(uh, gu)
    GV = matmul(vh, gv)
    core = ((GU - _h(GU)) * F) * sc[..., None, :] + sc[..., :, None] * ((GV - _h(GV)) * F)
    dg = gs.to(cdt) + 1j * (torch.diagonal(GU, dim1=-2, dim2=-1).imag.to(cdt) * sinv)
    core = core + torch.diag_embed(dg)
    ga = matmul(matmul(u, core), vh)
    if m > k:
        ga = ga + matmul((gu - matmul(u, GU)) * sinv[..., None, :], vh)
    if n > k:
        ga = ga + matmul(u * sinv[..., None, :], _h(gv - matmul(v, GV)))
    return ga


def _triu_inv(r):
    """Inverse of an upper-triangular [n,n] matrix by recursive doubling over its diagonal blocks:
    inv([[A,B],[0,C]]) = [[A^-1, -A^-1 B C^-1],[0, C^-1]], every level one pair of batched GEMMs."""
    import torch

    n = r.shape[0]
    N = 1
    while N < n:
        N *= 2
    if N != n:
        z = torch.zeros((n, N - n), dtype=r.dtype, device=r.device)
        e = torch.eye(N - n, dtype=r.dtype, device=r.device)
        r = torch.cat([torch.cat([r, z], 1), torch.cat([z.t(), e], 1)], 0)
    X = (1.0 / torch.diagonal(r)).reshape(N, 1, 1)
    b = 1
    while b < N:
        nb = N // (2 * b)
        idx = torch.arange(nb, device=r.device)
        B = r.reshape(nb, 2 * b, nb, 2 * b)[idx, :, idx, :][:, :b, b:]
        A, C = X[0::2], X[1::2]
        T = -(A * B * C) if b == 1 else -matmul(matmul(A, B), C)
        X = torch.cat([torch.cat([A, T], 2), torch.cat([torch.zeros_like(A), C], 2)], 1)
        b *= 2
    return X[0][:n, :n]


def _qr_backward_tall(q, r, gq, gr):
    """m >= n, r [n,n] upper triangular with a real diagonal."""
    import torch

    d = torch.diagonal(r)
    small = d.abs() < _QR_EPS
    r = r + torch.diag_embed(torch.where(small, _QR_EPS - d, torch.zeros_like(d)))
    M = matmul(r, _h(gr)) - matmul(_h(gq), q)
    L = torch.tril(M, -1)
    sym = L + _h(L) + torch.diag_embed(torch.diagonal(M).real.to(M.dtype))
    B = gq + matmul(q, sym)
    return matmul(B, _h(_triu_inv(r)))


def _qr_backward(q, r, gq, gr):
    import torch

    m = q.shape[0]
    n = r.shape[1]
    if m >= n:
        return _qr_backward_tall(q, r, gq, gr)
    # wide: A = [X | Y] = q [U | V]
    U, V = r[:, :m], r[:, m:]
    gU, gV = gr[:, :m], gr[:, m:]
    gx = _qr_backward_tall(q, U, gq + matmul(matmul(q, V), _h(gV)), gU)
    return torch.cat([gx, matmul(q, gV)], 1)


def _svd_trunc_ad(mat, static_keep, max_truncation_err, relative, absorb):
    import torch

    u, s, vh = _ad()["svd"](mat)
    kk = static_keep
    if max_truncation_err is not None:
        sd = s.detach()
        errs = torch.sqrt(torch.cumsum(torch.flip(sd, [0]) ** 2, 0))
        bound = max_truncation_err * sd[0] if relative else max_truncation_err
        kk = min(kk, int((errs > bound).sum().item()))
    sc = s.to(mat.dtype)
    uk, sk, vk = u[:, :kk], sc[:kk], vh[:kk, :]
    if absorb == 1:
        uk = uk * sk[None, :]
    elif absorb == 2:
        vk = sk[:, None] * vk
    return uk.contiguous(), sk, vk.contiguous(), sc[kk:]


def _ad():
    if _AD:
        return _AD
    import torch

    def vmap_loop(fn, info, in_dims, *args):
        """fallback vmap rule of the three primitives: one call per batch element (nested vmap; 2-D GEMMs and the
        decompositions of 2-D matrices use their batched launches instead)."""
        outs = []
        for i in range(info.batch_size):
            outs.append(fn(*[a.select(d, i) if d is not None else a for a, d in zip(args, in_dims)]))
        if isinstance(outs[0], tuple):
            res = tuple(torch.stack([o[k] for o in outs]) for k in range(len(outs[0])))
            return res, tuple(0 for _ in res)
        return torch.stack(outs), 0

    class Matmul(torch.autograd.Function):
        generate_vmap_rule = False

        @staticmethod
        def forward(a, b):
            return _matmul_raw(a, b)

        @staticmethod
        def vmap(info, in_dims, a, b):
            if a.dim() - (in_dims[0] is not None) == 2 and b.dim() - (in_dims[1] is not None) == 2:
                B = info.batch_size
                a3 = a.movedim(in_dims[0], 0) if in_dims[0] is not None else a.unsqueeze(0).expand(B, *a.shape)
                b3 = b.movedim(in_dims[1], 0) if in_dims[1] is not None else b.unsqueeze(0).expand(B, *b.shape)
                return Matmul.apply(a3, b3), 0
            return vmap_loop(Matmul.apply, info, in_dims, a, b)

        @staticmethod
        def setup_context(ctx, inputs, output):
            ctx.save_for_backward(*inputs)

        @staticmethod
        def backward(ctx, g):
            a, b = ctx.saved_tensors
            ga = matmul(g, _h(b)) if ctx.needs_input_grad[0] else None
            gb = matmul(_h(a), g) if ctx.needs_input_grad[1] else None
            return ga, gb

    class Svd(torch.autograd.Function):
        generate_vmap_rule = False

        @staticmethod
        def forward(mat):
            return _svd_full_raw(mat.contiguous())

        @staticmethod
        def vmap(info, in_dims, mat):
            if mat.dim() == 3:   # a batch of 2-D matrices (vmapped chains): ONE launch, the matrices side by side
                return Svd.apply(mat.movedim(in_dims[0], 0)), (0, 0, 0)
            return vmap_loop(Svd.apply, info, in_dims, mat)

        @staticmethod
        def setup_context(ctx, inputs, output):
            ctx.save_for_backward(*output)

        @staticmethod
        def backward(ctx, gu, gs, gvh):
            u, s, vh = ctx.saved_tensors
            return _svd_backward(u, s, vh, gu, gs, gvh)

    class Qr(torch.autograd.Function):
        generate_vmap_rule = False

        @staticmethod
        def forward(mat):
            q, r = _qr_raw(mat)
            d = torch.diagonal(r, dim1=-2, dim2=-1)
            a = d.abs()
            ph = torch.where(a > 0, d / a.clamp_min(1e-300 if d.dtype == torch.complex128 else 1e-30),
                             torch.ones_like(d))
            return q * ph[..., None, :], ph.conj()[..., :, None] * r

        @staticmethod
        def vmap(info, in_dims, mat):
            if mat.dim() == 3:   # batched launch, as Svd
                return Qr.apply(mat.movedim(in_dims[0], 0)), (0, 0)
            return vmap_loop(Qr.apply, info, in_dims, mat)

        @staticmethod
        def setup_context(ctx, inputs, output):
            ctx.save_for_backward(*output)

        @staticmethod
        def backward(ctx, gq, gr):
            q, r = ctx.saved_tensors
            if q.dim() == 3:     # stacked chains: the rule is written for one matrix
                return torch.stack([_qr_backward(q[i], r[i], gq[i], gr[i]) for i in range(q.shape[0])])
            return _qr_backward(q, r, gq, gr)

    # the decompositions' saved tensors are their own outputs: under torch.func they arrive wrapped, and
    # the rules above only use torch ops and ``matmul`` on them
    _AD.update(matmul=Matmul.apply, svd=Svd.apply, qr=Qr.apply)
    return _AD


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
