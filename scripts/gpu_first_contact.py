"""First GPU contact: parity of the tile-VM against the dense oracle + raw pass timings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import tcmi as tc
from oracle import dense, workloads as W

def check(n, d, dtype, kind="b", opts=None):
    tc.set_dtype(dtype)
    if opts: tc.set_contractor("greedy", **opts)
    rng = np.random.default_rng(n * 100 + d)
    if kind == "b":
        params = rng.uniform(0, 2*np.pi, [2*d, n]); ops = W.hea_b_ops(n, d, params)
    else:
        params = rng.uniform(0, 2*np.pi, [d, n]); ops = W.hea_a_ops(n, d, params)
    pt = tc.backend.convert_to_tensor(params, dtype=tc.rdtypestr)
    c = tc.Circuit(n)
    (W.hea_b if kind == "b" else W.hea_a)(c, n, d, pt, **({"zz": tc.gates._zz_matrix} if kind=="b" else {}))
    psi = tc.backend.numpy(c.wavefunction())
    ref = dense.run(n, ops)
    err = np.abs(psi - ref).max()
    cc = c._compiled()
    print(f"n={n} d={d} {dtype} hea-{kind} cfg=(R{cc.cfg.R},LT{cc.cfg.LT},low{cc.cfg.lowbits}) passes={len(cc.descs)} err={err:.3e} norm-1={abs(np.vdot(psi,psi)-1):.2e}", flush=True)
    tc.set_contractor("greedy")
    return err

def check_mixed(n, dtype):
    from oracle import gates as G
    tc.set_dtype(dtype)
    d = 2
    params = np.random.default_rng(n).uniform(0,2*np.pi,[2*d,n]); pa = np.random.default_rng(n+1).uniform(0,2*np.pi,[d,n])
    pt = tc.backend.convert_to_tensor(params, dtype=tc.rdtypestr)
    c = tc.Circuit(n); W.hea_b(c,n,d,pt, zz=tc.gates._zz_matrix); c.ry(0, theta=0.3); c.u(1, theta=0.2, phi=0.5, lbd=0.7); c.cnot(2,3); c.s(4); c.cz(5,1); c.swap(0,6); c.cnot(7,2)
    W.hea_a(c, n, d, pa); c.cphase(3,7,theta=0.4); c.rzz(0,5,theta=0.9); c.iswap(1,2,theta=0.3); c.phase(4, theta=1.1)
    c.any(1, 6, unitary=G.random_two_qubit_gate(5)); c.any(6, 2, unitary=G.random_two_qubit_gate(6)); c.r(3, theta=0.3, alpha=0.4, phi=0.5)
    ops = W.hea_b_ops(n,d,params) + [(G.ry(0.3),[0]),(G.u(0.2,0.5,0.7),[1]),(G.CNOT,[2,3]),(G.S,[4]),(G.CZ,[5,1]),(G.SWAP,[0,6]),(G.CNOT,[7,2])] + W.hea_a_ops(n,d,pa) + [(G.controlled(G.phase(0.4)),[3,7]),(G.rzz(0.9),[0,5]),(G.iswap(0.3),[1,2]),(G.phase(1.1),[4]),(G.random_two_qubit_gate(5),[1,6]),(G.random_two_qubit_gate(6),[6,2]),(G.r(0.3,0.4,0.5),[3])]
    psi = tc.backend.numpy(c.wavefunction()); ref = dense.run(n, ops)
    print(f"mixed n={n} {dtype} err={np.abs(psi-ref).max():.3e}", flush=True)

for dtype in ("complex64", "complex128"):
    for n in (8, 11, 13, 15, 18):
        check_mixed(n, dtype)
for dtype in ("complex64", "complex128"):
    for (n, d, kind) in [(3,2,"b"),(8,2,"b"),(10,4,"b"),(12,3,"a"),(13,3,"b"),(14,3,"a"),(16,4,"b"),(18,3,"a"),(20,4,"b")]:
        check(n, d, dtype, kind)
check(16, 4, "complex64", "b", {"lowbits": 7})
check(16, 4, "complex64", "b", {"lowbits": 3})
check(18, 3, "complex64", "a", {"R": 4, "LT": 9})
check(18, 3, "complex64", "b", {"R": 5, "LT": 9})

# timing at n=24, d=8
tc.set_dtype("complex64")
for n, d in [(24, 8), (26, 8)]:
  for opts in [{"lowbits":5}, {"lowbits":4}, {"lowbits":6}, {"lowbits":7}, {"R":5,"LT":9,"lowbits":5}, {"R":4,"LT":8,"lowbits":5}]:
    tc.set_contractor("greedy", **opts)
    params = np.random.default_rng(24).uniform(0, 2*np.pi, [2*d, n]).astype(np.float32)
    pt = tc.backend.convert_to_tensor(params)
    c = tc.Circuit(n); W.hea_b(c, n, d, pt, zz=tc.gates._zz_matrix)
    t0 = time.time(); cc = c._compiled(); t_compile = time.time() - t0
    p = c._param_tensor().reshape(1, -1)
    out = torch.empty(1, 2**n, dtype=torch.complex64, device="cuda")
    for _ in range(2): cc.state(p, out=out)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 10
    evs[0].record()
    for _ in range(reps): cc.state(p, out=out)
    evs[1].record(); torch.cuda.synchronize()
    ms = evs[0].elapsed_time(evs[1]) / reps
    st = cc.stats()
    print(f"TIMING n={n} d={d} opts={opts} passes={st['passes']} rounds={st['rounds']} compile={t_compile*1e3:.1f}ms  {ms:.3f} ms/state  amps/s={2**n/ms*1e3:.3e}  plan GB/s={st['bytes']/ms/1e6:.0f}", flush=True)
    # per-pass timing
    stream = torch.cuda.current_stream().cuda_stream
    ptab = torch.empty(1, cc.ptab_size, dtype=torch.float32, device="cuda")
    times = []
    for i in range(len(cc.descs)):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): cc.run_passes(out, ptab, 1, stream, i, i+1)
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1)/5)
    print("   per-pass us:", [int(t*1e3) for t in times], " gates/pass:", [len(pp.gate_ids) for pp in cc.plan.passes], "rounds:", [len(pp.rounds) for pp in cc.plan.passes], flush=True)
