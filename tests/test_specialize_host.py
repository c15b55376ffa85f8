"""Plan-specialised pass kernels, host side (no GPU): the emitter turns pass descriptors into HIP source that hipcc
compiles for gfx950, declines what it has no straight-line form for, and keys its cache on the generated text."""

import os
import shutil

import numpy as np
import pytest

import tcmi as tc
from tcmi import executor as X, plan as P, specialize as S
from oracle import gates as G


def _records(c):
    return c._gate_records(), len(c._params)


def _mixed_circuit(n, seed=3):
    """One-qubit gates of every structure class, diagonals of every emitted form, dense / CNOT / SWAP two-qubit gates."""
    rng = np.random.default_rng(seed)
    c = tc.Circuit(n)
    for i in range(n):
        c.h(i)
    for layer in range(3):
        for i in range(n - 1):
            c.rzz(i, i + 1, theta=float(rng.uniform(0, 6)))
        for i in range(n):
            c.rx(i, theta=float(rng.uniform(0, 6)))
        for i in range(0, n, 3):
            c.ry(i, theta=float(rng.uniform(0, 6)))
        for i in range(1, n, 4):
            c.u(i, theta=0.3 + layer, phi=0.5, lbd=0.7)
        c.cnot(layer, n - 1 - layer)
        c.cz(2 + layer, 5)
        c.swap(1, n - 2)
        c.any(3, 4 + layer, unitary=G.random_two_qubit_gate(5 + layer))
        c.cphase(0, n - 1, theta=0.4)
    return c


def _emits(d):
    try:
        S.forward_source(d)
        return True
    except S.Unsupported:
        return False


@pytest.fixture(autouse=True)
def _c64():
    old = tc.dtypestr
    tc.set_dtype("complex64")
    yield
    tc.set_dtype(old)


def test_forward_and_reverse_sources_are_emitted_for_hea_and_mixed_circuits():
    n = 16
    c = _mixed_circuit(n)
    gates, nparams = _records(c)
    n_exec, cfg, plan, eg = X.choose_plan(n, gates, nparams, "complex64", None)
    assert cfg.gen == 2
    nsrc, text = 0, ""
    for d in plan.descs:
        try:
            src, meta = S.forward_source(d)
        except S.Unsupported:       # a pass with the generic per-thread phase op keeps the interpreter
            continue
        assert 'extern "C" __global__' in src and meta["T"] == cfg.T and meta["lds"] == 4 << cfg.T
        # straight-line: no descriptor argument
        assert "desc" not in src.split("{", 1)[0]
        nsrc += 1
        text += src
    assert nsrc >= 1
    # dense two-qubit gates, CNOT / SWAP (register renames: no code), general and real one-qubit classes
    c2 = tc.Circuit(14)
    for i in range(14):
        c2.h(i)
    c2.any(3, 4, unitary=G.random_two_qubit_gate(5))
    c2.cnot(0, 1)
    c2.swap(5, 6)
    for i in range(14):
        c2.rx(i, theta=0.1 + i)
    c2.cz(2, 3)
    c2.u(7, theta=0.3, phi=0.5, lbd=0.7)
    c2.ry(8, theta=0.4)
    c2.any(7, 8, unitary=G.random_two_qubit_gate(6))
    g2, np2 = _records(c2)
    _, _, plan2, _ = X.choose_plan(14, g2, np2, "complex64", None)
    for d in plan2.descs:
        text += S.forward_source(d)[0]
    for body in ("vm2_g2x2", "vm2_shear8_real", "vm2_cmul8s", "vm2_cmul8v", "vm2_shear23_8_rx"):
        assert body in text, body
    # same descriptor -> same text -> same cache key; another descriptor -> another key
    ok = [d for d in plan.descs if S.prepare("forward", [d], compile_missing=False) is not None and _emits(d)]
    a, _ = S.forward_source(ok[0])
    b, _ = S.forward_source(ok[0])
    assert a == b and S.pass_digest(a) == S.pass_digest(b)
    if len(ok) > 1:
        assert S.pass_digest(S.forward_source(ok[1])[0]) != S.pass_digest(a)


def test_reverse_sweep_events_are_static_and_cover_every_gradient_slot():
    n, d = 14, 3
    c = tc.Circuit(n)
    pr = np.random.default_rng(0).uniform(0, 6, [2 * d, n])
    import torch

    pt = torch.from_numpy(pr)
    for i in range(n):
        c.h(i)
    for j in range(d):
        for i in range(n - 1):
            c.exp1(i, i + 1, unitary=tc.gates._zz_matrix, theta=pt[2 * j, i])
        for i in range(n):
            c.rx(i, theta=pt[2 * j + 1, i])
    gates, nparams = _records(c)
    n_exec, cfg, plan, eg = X.choose_plan(n, gates, nparams, "complex64", None)
    acfg, ap = X.choose_adjoint_plan(eg, n_exec, "complex64", True)
    slots = []
    for dsc in ap.descs:
        e = S._Adjoint(dsc, S.adjoint_opts(acfg))
        src = e.source("k")
        assert "tcmi_spec_slots" in src
        slots += [s for s in e.events if s >= 0]
        assert len(e.events) % e.EVB == 0
    # every gradient slot of the plan is the target of exactly one event of the sweep
    assert sorted(slots) == list(range(len(ap.gslot_param)))


def test_plans_outside_the_emitters_reach_keep_the_interpreter():
    """Tiles with fewer than four register bits (small circuits, complex128-style tiles) have no straight-line form: the
    emitter declines (Unsupported), `prepare` reports None for the pass and the executor launches the interpreting
    kernel.  The generic per-thread phase op (the final flush of a plan that ends on diagonal gates) IS emitted."""
    n = 12
    c = tc.Circuit(n)
    for i in range(n):
        c.h(i)
    rng = np.random.default_rng(0)
    for i in range(n):
        for j in range(i + 1, n):
            c.rzz(i, j, theta=float(rng.uniform(0, 2 * np.pi)))
    for i in range(n):
        c.rx(i, theta=0.1 * (i + 1))
    c.cz(0, 5)
    gates, nparams = _records(c)
    plan = P.compile_plan(gates, n, P.PlanConfig(R=4, LT=8, lowbits=5, vec=2, gen=2), nparams=nparams)
    assert any("sincos_turns" in S.forward_source(d)[0] for d in plan.descs)
    small = P.compile_plan(gates, n, P.PlanConfig(R=3, LT=8, lowbits=5, vec=2, gen=1), nparams=nparams)
    with pytest.raises(S.Unsupported):
        S.forward_source(small.descs[0])
    assert S.prepare("forward", small.descs, compile_missing=False) == [None] * len(small.descs)


@pytest.mark.skipif(not S.have_compiler(), reason="hipcc not available")
def test_generated_kernels_compile_for_gfx950(tmp_path, monkeypatch):
    """Cross-compile one forward and one reverse-sweep pass (mixed gate set: dense two-qubit gate, register renames,
    every diagonal form) into a private cache directory; a second `prepare` finds them there."""
    monkeypatch.setattr(S, "CACHE_DIR", str(tmp_path / "plancache"))
    n = 14
    c = _mixed_circuit(n)
    gates, nparams = _records(c)
    n_exec, cfg, plan, eg = X.choose_plan(n, gates, nparams, "complex64", None)
    before = dict(S.STATS)
    first = [d for d in plan.descs if _emits(d)][:1]
    res = S.prepare("forward", first)
    assert res[0] is not None and os.path.exists(res[0][0]) and res[0][0].startswith(str(tmp_path))
    assert S.STATS["compiled"] == before["compiled"] + 1
    res2 = S.prepare("forward", first)
    assert res2[0][0] == res[0][0] and S.STATS["cache_hits"] == before["cache_hits"] + 1
    # reverse sweep of an HEA-like circuit (the packed adjoint op set: one-qubit gates + diagonals)
    c2 = tc.Circuit(n)
    for i in range(n):
        c2.h(i)
    for i in range(n - 1):
        c2.rzz(i, i + 1, theta=0.3 + i)
    for i in range(n):
        c2.rx(i, theta=0.2 * i + 0.1)
    g2, np2 = _records(c2)
    n2, cfg2, plan2, eg2 = X.choose_plan(n, g2, np2, "complex64", None)
    acfg, ap = X.choose_adjoint_plan(eg2, n2, "complex64", True)
    r3 = S.prepare("adjoint", [np.asarray(ap.descs[0])], S.adjoint_opts(acfg))
    assert r3[0] is not None and r3[0][1]["kind"] == "adjoint"
    shutil.rmtree(str(tmp_path / "plancache"), ignore_errors=True)


def test_switch_off(monkeypatch):
    monkeypatch.setenv("TCMI_SPECIALIZE", "0")
    ps = S.PassSet("forward", [np.zeros(64, dtype=np.int32)], 28)
    assert ps.get() == [None]


def test_a_failing_compiler_degrades_to_the_interpreter(monkeypatch, tmp_path):
    """hipcc missing / failing, or a cache nobody can write to: `PassSet.get` warns once and hands back no kernels (the
    executor then launches the interpreting kernel); it never raises into the user's computation."""
    monkeypatch.setenv("TCMI_SPECIALIZE", "1")
    monkeypatch.setattr(S, "CACHE_DIR", str(tmp_path / "plancache"))
    monkeypatch.setattr(S, "HIPCC", "/bin/false")
    monkeypatch.setattr(S.shutil, "which", lambda *_a, **_k: None)
    n = 14
    c = tc.Circuit(n)
    for i in range(n):
        c.h(i)
        c.rx(i, theta=0.1 * i)
    gates, nparams = _records(c)
    _, _, plan, _ = X.choose_plan(n, gates, nparams, "complex64", None)
    ps = S.PassSet("forward", plan.descs, n)
    with pytest.warns(RuntimeWarning):
        ks = ps.get()
    assert ks == [None] * len(plan.descs) and ps.state == "done"
    assert ps.get() == [None] * len(plan.descs)          # and stays quiet afterwards


def test_one_compiler_run_per_code_object_when_processes_share_a_cache(tmp_path):
    """Ranks of one node ask for the same generated kernel at the same moment: the lock file of specialize._compile lets
    ONE of them run hipcc, the others wait for the code object."""
    import stat
    import subprocess
    import sys
    import textwrap

    log = tmp_path / "calls.log"
    fake = tmp_path / "fake_hipcc.sh"
    fake.write_text(textwrap.dedent(f"""\
        #!/bin/bash
        echo run >> {log}
        sleep 0.5
        while [ "$1" != "-o" ]; do shift; done
        echo code > "$2"
        """))
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    out = tmp_path / "cache" / "k.hsaco"
    child = textwrap.dedent(f"""\
        import sys
        sys.path[:0] = {[p for p in sys.path if p]!r}
        from tcmi import specialize as S
        S._compile("// source", {str(out)!r}, False)
        assert open({str(out)!r}).read() == "code\\n"
        """)
    env = dict(os.environ, HIPCC=str(fake))
    procs = [subprocess.Popen([sys.executable, "-c", child], env=env) for _ in range(4)]
    assert [p.wait(timeout=120) for p in procs] == [0, 0, 0, 0]
    assert log.read_text().count("run") == 1
    assert not os.path.exists(str(out) + ".lock")
    # a lock left behind by a process that no longer exists is taken over
    os.remove(out)
    with open(str(out) + ".lock", "w") as fh:
        fh.write("999999999")
    assert subprocess.run([sys.executable, "-c", child], env=env, timeout=120).returncode == 0
    assert S._compile_workers(64) >= 1
