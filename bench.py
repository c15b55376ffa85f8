#!/usr/bin/env python3
"""bench.py -- headline benchmark of the tcmi hot path (BASELINE.json metric: amplitudes/sec).

A "step" = the full ``Circuit.wavefunction`` contraction of a batch of HEA-B circuits (reference
``templates/blocks.py:146-185`` ansatz, seeded random parameters, SURVEY.md section 8(d) config 2:
24 qubits, depth 8, complex64; 64 circuits per step and GPU, 32 per vmap call) through the compiled plan, with
parameters and states resident in HBM.  One process per GPU; the circuits are independent units sharded over the ranks in
contiguous blocks (vmap-batch sharding, no data-path collective).  By default every GPU gets 64 circuits per step
("scaling": "weak": the per-GPU work is fixed, the global batch is 64 x the number of GPUs); ``--global-batch G`` fixes
the total instead ("scaling": "strong").  Timing is max-over-ranks between barriers.  Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 --steps 20 --warmup 3        (starts its own 8 rank processes, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md chip table (spec; 6.29 TB/s measured copy)
MFMA_F32_PEAK_TFS = 157.3  # dense FP32 (f32-input) MFMA peak, same table
MFMA_BF16_PEAK_TFS = 2516.6  # dense bf16 MFMA peak (256 CUs x 4 SIMDs x 32768 flop / 32 cycles x 2.4 GHz), same table


def build_circuit(tc, n, d, params_row):
    """HEA-B: the reference's own ansatz template (templates/blocks.py:146-185), through the product's mirror of it."""
    return tc.templates.blocks.example_block(tc.Circuit(n), params_row, nlayers=d)


def cpu_baseline(n_sample, d, seed, budget_s=25.0):
    """The oracle's TN contraction (numpy tensordot chain, what the reference's numpy backend
    executes through tensornetwork) timed on the host cores on a bounded sample of the workload:
    the same HEA-B ansatz at ``n_sample`` qubits (full n=24 takes minutes on 8 cores)."""
    import numpy as np
    from oracle import tn, workloads as W

    try:
        import threadpoolctl

        cores = max(x["num_threads"] for x in threadpoolctl.threadpool_info()) if threadpoolctl.threadpool_info() else os.cpu_count()
    except Exception:
        cores = os.cpu_count()
    params = np.random.default_rng(seed).uniform(0, 2 * np.pi, [2 * d, n_sample]).astype(np.float32)
    reps, t_total = 0, 0.0
    while reps < 3 or (t_total < budget_s / 2 and reps < 64):   # about 12 s of CPU work, at least 3 runs
        c = tn.Circuit(n_sample, dtype=np.complex64)
        W.hea_b(c, n_sample, d, params)
        t0 = time.perf_counter()
        c.wavefunction()
        t_total += time.perf_counter() - t0
        reps += 1
    t = t_total / reps
    return {
        "value": (2**n_sample) / t,
        "unit": "amplitudes/s",
        "cores": int(cores or 1),
        "kind": "port",
        "sample": f"oracle.tn greedy TN contraction of HEA-B n={n_sample} d={d} complex64, "
                  f"mean of {reps} run(s), {t:.2f} s each",
    }


def _host_threads():
    try:
        import threadpoolctl

        info = threadpoolctl.threadpool_info()
        return int(max(x["num_threads"] for x in info)) if info else int(os.cpu_count() or 1)
    except Exception:  # noqa: BLE001
        return int(os.cpu_count() or 1)


def cpu_baseline_vqe(n_full, d, budget_s=20.0):
    """CPU column of the VQE step (config 3): value_and_grad of the 2n-1 term TFIM energy of ONE HEA-B sample on the host
    with oracle/adjoint.py (numpy adjoint state-vector method, the cheapest CPU formulation of the step -- the reference's
    tape through the tensordot chain does more work), complex64, at the largest n whose estimated time fits the budget
    (time doubles per qubit; n = 16 is timed first).  The n_full figure is an EXTRAPOLATION by 2^(n_full - n) and says so."""
    import numpy as np
    from oracle import adjoint as A

    def run(n):
        p = np.random.default_rng(28).normal(0, 0.1, [2 * d, n]).astype(np.float32)
        t0 = time.perf_counter()
        A.hea_b_tfim_value_and_grad(n, d, p, dtype=np.complex64)
        return time.perf_counter() - t0

    n = min(16, n_full)
    t = run(n)
    spent = t
    # two qubits at a time while the next size (x 4, and the caches stop helping: x 5) still fits what is left of the budget
    while n + 2 <= min(n_full, 24) and spent + 5.0 * t <= budget_s:
        n += 2
        t = run(n)
        spent += t
    return {"value": 1.0 / t, "unit": "samples/s (one value_and_grad of one circuit)", "cores": 1, "kind": "port",
            "sample": f"oracle.adjoint (numpy, strided in-place updates: one thread) HEA-B n={n} d={d} TFIM value_and_grad, "
                      f"complex64, one run of {t:.2f} s",
            "qubits_timed": n, "seconds_per_sample": t,
            "extrapolated_seconds_per_sample_at_full_size": t * 2.0 ** (n_full - n),
            "extrapolation": f"x 2^({n_full} - {n}): the work per gate is linear in the state size; caches make the real "
                             f"n = {n_full} run slower than this"}


def cpu_baseline_mps(n, chi, tensors, gate_mats, budget_s=25.0):
    """CPU column of config 5: the same TEBD sweep through oracle.mps (numpy GEMMs + LAPACK SVD of the (2 chi) x (2 chi)
    bond matrix, complex64 like the GPU leg) on the host cores, on as many bonds from the left end as fit the budget (the
    bonds in the saturated middle all cost the same; the time per bond is over the bonds timed)."""
    import numpy as np
    from oracle import mps as OM

    m = OM.MPSCircuit(n, tensors=[np.asarray(t) for t in tensors], split=OM.split_rules(max_singular_values=chi))
    m.position(0)
    t0 = time.perf_counter()
    done, sat, t_sat = 0, 0, 0.0
    for i in range(n - 1):
        t1 = time.perf_counter()
        m.apply(gate_mats[i], i, i + 1)
        dt = time.perf_counter() - t1
        done += 1
        if min(2 ** (i + 1), 2 ** (n - i - 1)) >= chi:       # a bond of full dimension on both sides
            sat += 1
            t_sat += dt
        if time.perf_counter() - t0 > budget_s:
            break
    el = time.perf_counter() - t0
    per_bond = (t_sat / sat) if sat else el / done
    return {"value": 1.0 / (per_bond * (n - 1)), "unit": "sweeps/s", "cores": _host_threads(), "kind": "port",
            "sample": f"oracle.mps (numpy + LAPACK) on the first {done} of {n - 1} bonds of the same sweep, complex64, "
                      f"{el:.1f} s; us_per_bond over the {sat} saturated bonds among them",
            "us_per_bond": per_bond * 1e6, "bonds_timed": done}


def cpu_baseline_rqc(tree, arrays, budget_s=30.0):
    """CPU column of config 4: ONE slice of the executed tree as a numpy tensordot chain (oracle/sliced.py: what the
    reference's numpy backend does per slice, cons.py:845-961) on the host cores; the amplitude costs nslices of them.
    The slice is abandoned when the budget is exceeded (then the figure is a lower bound of the time and says so)."""
    import numpy as np
    from oracle import sliced as OS

    host = [a.detach().cpu().numpy() for a in arrays]
    sliced = list(tree.sliced_inds)
    t0 = time.perf_counter()
    base = {"unit": "amplitudes/s (one amplitude = all slices)", "cores": _host_threads(), "kind": "port"}
    try:
        val = OS.contract_path(host, [list(x) for x in tree.inputs], [tuple(p_) for p_ in tree.path], sliced,
                               OS.slice_values(0, len(sliced)), budget_s=budget_s)
    except TimeoutError as e:
        done, total, t = e.args[0]
        return {**base, "value": 1.0 / (t * tree.nslices), "value_is": "an UPPER bound of the CPU rate: the slice was abandoned",
                "sample": f"oracle.sliced numpy tensordot chain of slice 0 of {tree.nslices} of the executed tree, complex64: "
                          f"{done} of {total} steps in {t:.1f} s (budget {budget_s:.0f} s), abandoned",
                "seconds_per_slice_at_least": t}
    t = time.perf_counter() - t0
    return {**base, "value": 1.0 / (t * tree.nslices),
            "sample": f"oracle.sliced numpy tensordot chain of slice 0 of {tree.nslices} of the executed tree, "
                      f"complex64, {t:.1f} s; x {tree.nslices} slices",
            "seconds_per_slice": t, "seconds_per_amplitude": t * tree.nslices, "slice0_value": [val.real, val.imag]}


def rank_times(torch, dist, dev, seconds, steps):
    """[ms per step of every rank] (all_gather of this rank's own wall time between the barriers): the max is what
    ``value`` is computed from; the spread tells a straggling rank from a slow job.  None on one rank."""
    if dist is None:
        return None
    mine = torch.tensor([seconds / max(1, steps) * 1e3], device=dev, dtype=torch.float64)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [round(float(x.item()), 4) for x in out]


def allreduce_latency(torch, dist, dev, numel, reps=50):
    """Mean wall time in microseconds of the step's one collective: a float64 all-reduce(SUM) of ``numel`` values
    ([value || gradient], a few KB -- latency-bound over xGMI), each one followed by a device synchronisation (the
    step reads the sum on the host side right away).  None on one rank."""
    if dist is None:
        return None
    buf = torch.zeros(max(1, int(numel)), device=dev, dtype=torch.float64)
    for _ in range(5):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(buf)
        torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps * 1e6
    tt = torch.tensor([t], device=dev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return round(float(tt.item()), 1)


def summarize_events(log):
    """Per-tag totals of the executor's HIP-event log: {tag: {ms, launches, work, calls}} (call after a sync)."""
    out = {}
    for tag, e0, e1, launches, work in log:
        d = out.setdefault(tag, {"ms": 0.0, "launches": 0, "work": 0.0, "calls": 0})
        d["ms"] += e0.elapsed_time(e1)
        d["launches"] += launches
        d["work"] += work
        d["calls"] += 1
    return out


def hbm_entry(kernel, ev, steps):
    """Roofline entry of an HBM-bound kernel family from its event-log totals."""
    if not ev or ev["launches"] == 0 or ev["ms"] <= 0:
        return None
    per_launch = ev["work"] / ev["launches"]
    avg_us = ev["ms"] * 1e3 / ev["launches"]
    gbs = per_launch / (avg_us * 1e-6) / 1e9
    return {"kernel": kernel, "bound": "hbm", "launches_per_step": ev["launches"] / steps, "avg_launch_us": avg_us,
            "algorithmic_bytes_per_launch": per_launch, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS}


def traffic_probe(args, kernels, timeout_s=150):
    """HBM traffic of the headline's dominant kernel from the PMC counters of THIS command: two child runs of a
    short headline-only bench under rocprofv3 (--pmc FETCH_SIZE, then WRITE_SIZE: the two do not fit one pass;
    the program itself after `--`), corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950
    (counter unit KiB; FETCH_SIZE reports half of a wide coalesced streaming read): bytes = (2 FETCH + WRITE) * 1024.
    Returns {kernel substring: bytes per launch} or {} when the profiler is unavailable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return {}
    res = {}
    tmp = tempfile.mkdtemp(prefix="tcmi_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--probe-child", "--steps", "2", "--warmup", "1",
             "--qubits", str(args.qubits), "--depth", str(args.depth), "--batch", str(args.batch),
             "--contractor", args.contractor]
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            subprocess.run(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child,
                           cwd="/tmp", env=env, timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                acc = {}
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] != ctr:
                        continue
                    for k in kernels:
                        if k in r["Kernel_Name"]:
                            a = acc.setdefault(k, [0.0, set()])
                            a[0] += float(r["Counter_Value"])
                            a[1].add(r["Dispatch_Id"])
                for k, (v, ids) in acc.items():
                    res.setdefault(k, {})[ctr] = v / max(1, len(ids))
    except Exception:  # noqa: BLE001  (no profiler, timeout ...): traffic stays null
        res = {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for k, v in res.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out[k] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
    return out


def vqe_leg(tc, torch, dist, args, rank, world, dev):
    """BASELINE config 3: one VQE step = vectorized value_and_grad of the 2n-1 term TFIM energy
    (reference benchmarks/scripts/vqe_tc.py:75-81,107-141) over a vmap batch of HEA-B circuits,
    batch sharded over the ranks, followed by ONE packed all-reduce of [sum of energies || summed
    gradient] (the reference's jnp.sum over devices, tensorcircuit/experimental.py:1145-1152)."""
    import numpy as np
    from tcmi import distributed as D

    n, d, Bg = args.vqe_qubits, args.vqe_depth, args.vqe_batch
    lo, hi = D.shard_range(Bg, rank, world)
    params_np = np.random.default_rng(28).normal(0, 0.1, [Bg, 2 * d, n]).astype(np.float32)
    params = torch.from_numpy(params_np[lo:hi]).to(dev)

    def energy(p):
        c = tc.templates.blocks.example_block(tc.Circuit(n), p, nlayers=d)
        e = 0.0
        for i in range(n):
            e += -1.0 * c.expectation((tc.gates.x(), [i]))
        for i in range(n - 1):
            e += 1.0 * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
        return tc.backend.real(e)

    # the reference harness jits the step (benchmarks/scripts/vqe_tc.py:136-141); here jit traces the host side
    vvag = tc.backend.jit(tc.backend.vvag(energy, argnums=0, vectorized_argnums=0))
    mb = max(1, args.vqe_microbatch)

    nstreams = max(1, int(os.environ.get("TCMI_BENCH_VQE_STREAMS", str(args.vqe_streams))))
    side = [torch.cuda.Stream(device=dev) for _ in range(nstreams)] if nstreams > 1 else []

    def step():
        vals, grads = [], []
        if side:
            # micro-batches are independent: alternate them over HIP streams, so that the VALU-bound passes of one overlap
            # the HBM-bound passes of another (every allocation of the traced pipeline is per call and stream-ordered)
            cur = torch.cuda.current_stream(dev)
            for k, b0 in enumerate(range(0, hi - lo, mb)):
                st = side[k % len(side)]
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    v, g = vvag(params[b0: b0 + mb])
                vals.append(v)
                grads.append(g)
            for st in side:
                cur.wait_stream(st)
            for t_ in vals + grads:
                t_.record_stream(cur)
        else:
            for b0 in range(0, hi - lo, mb):
                v, g = vvag(params[b0: b0 + mb])
                vals.append(v)
                grads.append(g)
        v = torch.cat(vals) if vals else torch.zeros(0, device=dev)
        g = torch.cat(grads) if grads else torch.zeros(0, 2 * d, n, device=dev)
        esum, gsum = D.allreduce_sum_packed([v.sum().reshape(1), g.sum(0)])
        return v, g, esum, gsum

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t0 = time.perf_counter()
    step()  # staging: plan + adjoint plan + measurement plan compile, first launch
    sync()
    staging = time.perf_counter() - t0
    from tcmi import executor as X

    step()  # second call: the traced pipeline is validated against the plain path on the first two
    sync()
    X.EVENT_LOG = []
    X.VALU_LOG = {}
    t0 = time.perf_counter()
    for _ in range(args.vqe_steps):
        v, g, esum, gsum = step()
    sync()
    el = time.perf_counter() - t0
    ev = summarize_events(X.EVENT_LOG)
    valu_log = X.VALU_LOG
    X.EVENT_LOG = None
    X.VALU_LOG = None
    # the same kernels with every tile live (TCMI_SPARSE_START=0: the dense plan, every pass moves the whole state): what
    # the pass kernels reach on full passes, next to the step that skips the tiles that are still zero
    dense = None
    if X.SPARSE_START and dist is None:      # one rank only: step() holds a collective
        X.SPARSE_START = False
        try:
            vd, gd, _, _ = step()
            sync()
            X.EVENT_LOG = []
            td0 = time.perf_counter()
            vd, gd, _, _ = step()
            torch.cuda.synchronize()
            td = time.perf_counter() - td0
            evd = summarize_events(X.EVENT_LOG)
            X.EVENT_LOG = None
            dense = {"ms_per_step": td * 1e3, "steps": 1,
                     "forward_pass": hbm_entry("the forward pass kernels, every tile live", evd.get("pass"), 1),
                     "adjoint_pass": hbm_entry("the reverse-sweep pass kernels, every tile live", evd.get("adjoint"), 1),
                     "max_abs_energy_difference": float((vd - v).abs().max().item()),
                     "max_abs_gradient_difference": float((gd - g).abs().max().item())}
        finally:
            X.EVENT_LOG = None
            X.SPARSE_START = True
    per_rank_ms = rank_times(torch, dist, dev, el, args.vqe_steps)
    ar_us = allreduce_latency(torch, dist, dev, 1 + 2 * d * n)
    if dist is not None:
        sync()
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    # roofline of the step on the EXECUTED plan: every launch's algorithmic bytes / its HIP-event time
    S = float(hi - lo) * (2 ** n) * 8          # one pass over this rank's batch of states, bytes
    nf = ev.get("pass", {}).get("launches", 0) / max(1, args.vqe_steps) / max(1, -(-(hi - lo) // mb))
    nb_ = ev.get("adjoint", {}).get("launches", 0) / max(1, args.vqe_steps) / max(1, -(-(hi - lo) // mb))
    exec_bytes = sum(d["work"] for d in ev.values()) / max(1, args.vqe_steps)
    step_s = el / args.vqe_steps
    b_fwd = nf * 2.0 * S
    fcfg = X.pick_variant(n, "complex64")[1]
    acfg = X.pick_adjoint_variant(n, "complex64", [])
    mcfg = X.pick_measure_variant(n, "complex64")
    from tcmi import specialize as SP

    # the passes run as plan-specialised straight-line kernels when the plan's code objects are there (tcmi/specialize.py;
    # kernel names tcmi_spec_forward / tcmi_spec_adjoint in a rocprofv3 trace), else on the interpreting kernels
    nspec = {k: sum(1 for v in SP._LOADED.values() if v.meta.get("kind") == k) for k in ("forward", "adjoint")}
    fname = (f"tcmi_spec_forward (plan-specialised gate passes, {nspec['forward']} code objects; programs of pass2_kernel<{fcfg.R},{fcfg.LT}>)"
             if nspec["forward"] else f"tcmi::pass2_kernel<{fcfg.R},{fcfg.LT}> (gate passes)")
    aname = (f"tcmi_spec_adjoint (plan-specialised reverse sweep on psi and lambda, {nspec['adjoint']} code objects; programs of "
             f"adjoint2_kernel<{acfg.R},{acfg.LT}>)" if nspec["adjoint"]
             else f"tcmi::adjoint2_kernel<{acfg.R},{acfg.LT}> (reverse sweep on psi and lambda)")
    roof = {
        "forward_pass": hbm_entry(fname, ev.get("pass"), args.vqe_steps),
        "adjoint_pass": hbm_entry(aname, ev.get("adjoint"), args.vqe_steps),
        "measure_pass": hbm_entry(f"tcmi::measure2_kernel<{mcfg.R},{mcfg.LT}> (fused Pauli-sum measurement)"
                                  if mcfg.gen >= 2 else f"tcmi::pass_kernel<float,{mcfg.R},{mcfg.LT},1> (fused Pauli-sum measurement)",
                                  ev.get("measure"), args.vqe_steps),
        # tile passes (3 launches per micro-batch for the TFIM: 2 + 3 + 3 state transfers) that also return the energy:
        # the traced step runs no measurement pass (measure_pass is null then)
        "pauli_sum": hbm_entry("tcmi::pauli_tile_kernel (cotangent of the energy as tile passes, returns the energy too)",
                               ev.get("pauli_sum"), args.vqe_steps),
        "step": {
            "bound": "hbm", "executed_bytes_per_step": exec_bytes, "achieved": exec_bytes / step_s / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": exec_bytes / step_s / 1e9 / HBM_PEAK_GBS,
            "kernel_ms_per_step": sum(d["ms"] for d in ev.values()) / max(1, args.vqe_steps),
            "forward_passes": nf, "adjoint_passes": nb_,
            # the survey's cross-plan convention B_vg = 3 B_fwd + B_exp (reference-equivalent term loop) on this plan
            "B_vg_convention_bytes": 3.0 * b_fwd + (2 * n - 1) * 2.0 * S,
        },
    }
    # The gate passes are bound by VALU issue, not by HBM (PMC: profiles/r03*_vqe_n28_d12_pmc.txt): next to the byte rates,
    # the FP32 rate of the GATE ARITHMETIC ALONE (one-qubit gates in three-shear form: 3 packed FMAs = 12 flops per
    # amplitude pair; the reverse sweep applies U^dagger to psi and lambda and forms the generator's expectation: 32 flops
    # per pair) against the 157.3 TFLOP/s vector peak -- phase tables, Walsh transforms and exchanges come on top
    ngates = n * (d + 1)
    def _valu(ev_key, flops_per_pair):
        e = ev.get(ev_key)
        if not e or e["ms"] <= 0:
            return None
        per_step_s = e["ms"] * 1e-3 / max(1, args.vqe_steps)
        fl = float(hi - lo) * ngates * (2.0 ** (n - 1)) * flops_per_pair
        return {"bound": "valu", "gate_arithmetic_flops_per_step": fl, "achieved": fl / per_step_s / 1e12,
                "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s", "frac": fl / per_step_s / 1e12 / MFMA_F32_PEAK_TFS}
    # live-tile passes (executor.live_masks): a state that starts from |0...0> has few tiles that can be non-zero during its
    # first passes (and psi is back to that shape in the last passes of the reverse sweep); `frac` above is on the bytes of
    # the live tiles -- what the sparse-start algorithm has to move; `dense_plan_bytes_per_launch` = what the same passes
    # move with every tile live (measured in the `dense_plan` block below)
    for key, evk, dense_units in (("forward_pass", "pass", 2.0 * nf), ("adjoint_pass", "adjoint", 4.0 * nb_ - 2.0)):
        ent, e_ = roof.get(key), ev.get(evk)
        if ent and e_ and e_["launches"]:
            dbytes = dense_units * S * max(1, args.vqe_steps) / e_["launches"]
            ent["dense_plan_bytes_per_launch"] = dbytes
            ent["live_tile_passes"] = bool(X.SPARSE_START and abs(dbytes - ent["algorithmic_bytes_per_launch"]) > 1e-6 * dbytes)
    if dense is not None:
        roof["dense_plan"] = dense
    # VALU roofline on the EXECUTED plan (the fraction of whichever roof bounds the kernel): flops and packed
    # instructions counted from the generated source of every specialised pass (specialize.pass_arithmetic: the asm bodies
    # it calls in their two-shear form, the pruned Walsh butterflies, the wave folds), x the waves of the LIVE tiles x batch,
    # against the FP32 vector peak (a packed FMA = 256 flops holds a SIMD's issue port for 4 clocks: 1024 SIMDs x 64 flops
    # x 2.4 GHz = 157.3 TFLOP/s).  ``issue_frac`` = the time the VALU needs to issue those instructions / the measured time.
    def _valu_exec(tag):
        e, vl = ev.get(tag), valu_log.get(tag)
        if not e or e["ms"] <= 0 or not vl or vl[0] <= 0:
            return None
        per_step_s = e["ms"] * 1e-3 / max(1, args.vqe_steps)
        fl, ins = vl[0] / max(1, args.vqe_steps), vl[1] / max(1, args.vqe_steps)
        issue_s = ins * 4.0 / (1024 * 2.4e9)
        return {"bound": "valu", "flops_per_step": fl, "achieved": fl / per_step_s / 1e12, "peak": MFMA_F32_PEAK_TFS,
                "unit": "TFLOP/s", "frac": fl / per_step_s / 1e12 / MFMA_F32_PEAK_TFS,
                "valu_instructions_x_waves_per_step": ins, "issue_time_ms_per_step": issue_s * 1e3,
                "issue_frac": issue_s / per_step_s, "interpreted_passes_not_counted": int(vl[2] / max(1, args.vqe_steps)),
                "counted_from": "generated source of the executed passes (tcmi/specialize.py::pass_arithmetic), live tiles only"}
    roof["forward_pass_valu"] = _valu_exec("pass")
    roof["adjoint_pass_valu"] = _valu_exec("adjoint")
    # the older convention (gate arithmetic alone on full passes), kept for the dense plan's comparison with round 3
    live_on = bool(roof.get("forward_pass") and roof["forward_pass"].get("live_tile_passes"))
    roof["forward_pass_gate_flops_only"] = None if live_on else _valu("pass", 12.0)
    roof["adjoint_pass_gate_flops_only"] = None if live_on else _valu("adjoint", 32.0)
    interp = sum(int(v_[2]) for v_ in valu_log.values()) // max(1, args.vqe_steps)
    cpu = None
    if dist is None and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline_vqe(n, d)
        cpu["gpu_seconds_per_sample"] = el / args.vqe_steps / Bg
    return {
        "roofline": roof,
        **({"cpu_baseline": cpu} if cpu is not None else {}),
        "workload": f"HEA-B n={n} depth={d} TFIM value_and_grad (55-term style energy), vmap batch {Bg} "
                    f"(SURVEY 8d config 3), complex64",
        "ms_per_step": el / args.vqe_steps * 1e3,
        **({"per_rank_ms_per_step": per_rank_ms, "allreduce_us": ar_us,
            "allreduce": f"one packed float64 all-reduce of [sum of energies || summed gradient] = {1 + 2 * d * n} values per step"}
           if dist is not None else {}),
        "steps": args.vqe_steps,
        "samples_per_s": Bg * args.vqe_steps / el,
        "batch_per_gpu": hi - lo,
        "staging_s": round(staging, 3),
        "mean_energy": float(esum.item()) / Bg,
        "grad_norm": float(gsum.norm().item()),
        "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
        "specialised_kernels": {**nspec, "mode": SP.mode(), "interpreted_passes_per_step": interp,
                                "compiled_in_this_process": SP.STATS["compiled"],
                                "compile_s": round(SP.STATS["compile_s"], 2), "cache_hits": SP.STATS["cache_hits"]},
    }


def mps_leg(tc, torch, args):
    """BASELINE config 5: MPSCircuit n=64, chi=128 (max_singular_values), one TEBD sweep = 63 adjacent random
    SU(4) gates left to right on a chi-saturated random MPS, complex64.  Latency-bound: per bond
    one QR (centre move), one bond GEMM, one gate mix, one Jacobi-SVD launch; no host synchronisation."""
    import numpy as np
    from scipy.stats import unitary_group

    n, chi = args.mps_qubits, args.mps_chi
    rng = np.random.default_rng(64)
    dims = [min(2 ** i, 2 ** (n - i), chi) for i in range(n + 1)]
    tensors = [((rng.normal(size=(dims[i], 2, dims[i + 1])) + 1j * rng.normal(size=(dims[i], 2, dims[i + 1])))
                / np.sqrt(2 * dims[i])).astype(np.complex64) for i in range(n)]
    gates = [tc.gates.Gate(unitary_group.rvs(4, random_state=5000 + i).reshape(2, 2, 2, 2).astype(np.complex64))
             for i in range(n - 1)]
    t0 = time.perf_counter()
    m = tc.MPSCircuit(n, tensors=tensors, split=tc.cons.split_rules(max_singular_values=chi))

    def sweep():
        for i in range(n - 1):
            m.apply(gates[i], i, i + 1)

    sweep()
    m.position(0)
    torch.cuda.synchronize()
    staging = time.perf_counter() - t0
    from tcmi import executor as X

    times = []
    X.EVENT_LOG = []
    for _ in range(max(1, args.mps_sweeps)):
        t0 = time.perf_counter()
        sweep()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        log, X.EVENT_LOG = X.EVENT_LOG, None
        m.position(0)                    # back to the sweep's starting gauge: not part of the sweep, so it must
        torch.cuda.synchronize()         # not run into the next timed sweep either
        X.EVENT_LOG = log
    ev = summarize_events(X.EVENT_LOG)
    X.EVENT_LOG = None
    t = sum(times) / len(times)
    nbonds = (n - 1) * len(times)
    split = {k.replace("mps_", "") + "_us_per_bond": v["ms"] * 1e3 / nbonds for k, v in ev.items()}
    split["launches_per_bond"] = sum(v["launches"] for v in ev.values()) / nbonds
    if not all(bool(torch.isfinite(x.abs()).all()) for x in m.get_tensors()):
        raise FloatingPointError("non-finite MPS tensor after the TEBD sweeps")
    # Independent chains from the Python layer: backend.vmap over C chains that start from the same saturated MPS and
    # apply their own random SU(4) gates.  GEMMs, QRs and SVDs of the C chains go out as one batched launch each (the
    # batch argument of the ABI: 16 workgroups per 256 x 256 SVD, so 16 chains fill the chip).
    chains = None
    C = int(args.mps_chains)
    if C > 1:
        start = [x.clone() for x in m.get_tensors()]
        gs = torch.stack([torch.stack([torch.as_tensor(
            unitary_group.rvs(4, random_state=9000 + 100 * c + i).reshape(2, 2, 2, 2).astype(np.complex64))
            for i in range(n - 1)]) for c in range(C)]).cuda()

        def chain(g):
            mc = tc.MPSCircuit(n, tensors=start, center_position=0, split=tc.cons.split_rules(max_singular_values=chi))
            for i in range(n - 1):
                mc.apply(tc.gates.Gate(g[i]), i, i + 1)
            last = mc.get_tensors()[-1]
            return tc.backend.real(tc.backend.sum(last * tc.backend.conj(last)))

        vchain = tc.backend.vmap(chain)
        nrm = vchain(gs)                      # staging run
        torch.cuda.synchronize()
        tt = []
        for _ in range(max(1, args.mps_sweeps)):
            t0 = time.perf_counter()
            nrm = vchain(gs)
            torch.cuda.synchronize()
            tt.append(time.perf_counter() - t0)
        tb = sum(tt) / len(tt)
        if not bool(torch.isfinite(nrm).all()):
            raise FloatingPointError("non-finite norm after the batched TEBD sweeps")
        chains = {"chains": C, "api": "backend.vmap over chains (one batched GEMM / QR / SVD launch per bond)",
                  "chain_sweeps_per_s": C / tb, "us_per_bond_per_chain": tb / (n - 1) / C * 1e6,
                  "us_per_bond_all_chains": tb / (n - 1) * 1e6, "last_tensor_norm2": [float(x) for x in nrm[:4]]}
    # Graded spectra: the bond matrices of random circuits are flat, those of ground-state searches are not.  One
    # (2 chi) x (2 chi) matrix with singular values spread over six decades through the plain one-sided Jacobi and
    # through the QR-preconditioned path (linalg.SVD_PRECONDITION): sweeps and milliseconds by HIP events.
    graded = None
    try:
        from tcmi import linalg as LA

        def haar(k, seed):
            z = np.random.default_rng(seed).normal(size=(k, k, 2)) @ np.array([1.0, 1j])
            qh, rh = np.linalg.qr(z)
            return qh * (np.diag(rh) / np.abs(np.diag(rh)))

        md = 2 * chi
        ag = torch.from_numpy(((haar(md, 1) * np.logspace(0, -6, md)) @ haar(md, 2)).astype(np.complex64)).cuda()
        graded = {"matrix": f"{md} x {md} complex64, singular values 1 .. 1e-6 (log-spaced)"}
        keep_flag = LA.SVD_PRECONDITION
        for name, pre in (("plain", False), ("qr_preconditioned", True)):
            LA.SVD_PRECONDITION = pre
            LA.svd_trunc(ag, max_singular_values=chi, absorb=1)
            torch.cuda.synchronize()
            ev = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ug, sg, vg, _rest = LA.svd_trunc(ag, max_singular_values=chi, absorb=1)
                e1.record()
                torch.cuda.synchronize()
                ev.append(e0.elapsed_time(e1))
            graded[name] = {"ms": float(np.median(ev)), "sweeps": int(LA.last_svd_sweeps(ag.device)),
                            "sigma_max": float(sg.real.max())}
        LA.SVD_PRECONDITION = keep_flag
    except Exception as e:  # noqa: BLE001 - an auxiliary figure must not take the leg down
        graded = {"error": repr(e)}
    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline_mps(n, chi, tensors, [tc.backend.numpy(g.tensor) for g in gates])
    return {
        **({"cpu_baseline": cpu} if cpu is not None else {}),
        "workload": f"MPSCircuit n={n} chi={chi} TEBD sweep of {n - 1} adjacent random SU(4) gates, complex64 "
                    f"(SURVEY 8d config 5)",
        "sweeps_per_s": 1.0 / t, "us_per_bond": t / (n - 1) * 1e6, "sweeps": len(times),
        "max_bond": int(max(m.get_bond_dimensions())), "staging_s": round(staging, 3),
        "fidelity_estimate": float(m._fidelity),
        # neither HBM- nor MFMA-bound: latency of dependent launches; kernel time per bond by kind
        "roofline": {"bound": "latency", "kernel_us_per_bond": split},
        "batched_chains": chains,
        "graded": graded,
    }


def rqc_network(tc, depth, rows=4, cols=8):
    """nodes_fn of config 4: <0^32|C|0^32> of a brickwork of Haar-random two-qubit gates on a rows x cols grid."""
    gates = [tc.gates.random_two_qubit_gate(7000 + i).tensor for i in range(depth * rows * cols)]
    q = lambda r, c: r * cols + c  # noqa: E731

    def nodes_fn(_):
        c = tc.Circuit(rows * cols)
        k = 0
        for d in range(depth):
            pat = d % 4
            if pat in (0, 1):
                pairs = [(q(r, cc), q(r, cc + 1)) for r in range(rows) for cc in range(pat, cols - 1, 2)]
            else:
                pairs = [(q(r, cc), q(r + 1, cc)) for r in range(pat - 2, rows - 1, 2) for cc in range(cols)]
            for a, b in pairs:
                c.any(a, b, unitary=gates[k])
                k += 1
        return c.amplitude_before("0" * (rows * cols))

    return nodes_fn


def rqc_search_options(log2_target, seeds):
    return {"slicing_opts": {"target_size": 2 ** log2_target}, "max_repeats": 128, "seed": list(range(seeds))}


def svqa_network(tc, n, d):
    """nodes_fn of the sliced-VQA leg: rzz / rx ladder, <Z> of the middle qubit (reuse=False: ket and bra networks)."""
    def nodes_fn(params):
        c = tc.Circuit(n)
        for i in range(d):
            for j in range(n - 1):
                c.rzz(j, j + 1, theta=params[j, i, 0])
            for j in range(n):
                c.rx(j, theta=params[j, i, 1])
        return c.expectation_before([tc.gates.z(), [n // 2]], reuse=False)

    return nodes_fn


def svqa_search_options(slices, seeds=1, seed0=0, minimize="combo"):
    return {"slicing_opts": {"target_slices": slices}, "max_repeats": 32, **({"minimize": minimize} if minimize else {}),
            **({"seed": list(range(seed0, seed0 + seeds))} if (seeds > 1 or seed0) else {})}


def presearch_trees(tc, args=None):
    """The contraction trees of the two sliced legs at the bench's default sizes, searched on the HOST (no GPU: the search
    needs the networks' index structure only) and left in the tree cache next to the generated kernels --
    __graft_entry__.build() calls this like it pre-compiles the bench's plan kernels; the timed run then loads them
    (``path_search_cached``).  The reference persists searched trees the same way (``find_path(filepath)`` /
    ``from_path``, experimental.py:923-991; cotengra's ReusableHyperOptimizer)."""
    import numpy as np
    from tcmi.experimental import DistributedContractor as DC

    ap = {"rqc_depth": 16, "rqc_log2_target": 27, "rqc_seeds": 8, "svqa_qubits": 30, "svqa_depth": 8, "svqa_slices": 8,
          "svqa_seeds": 8}
    if args is not None:
        ap.update({k: getattr(args, k) for k in ap if hasattr(args, k)})
    out = {}
    t0 = time.perf_counter()
    # (build hosts have no GPU context: the seeds of a hyper-search run on a process pool, cotengra's ``parallel``)
    DC._get_tree_data(rqc_network(tc, ap["rqc_depth"]), None,
                      dict(rqc_search_options(ap["rqc_log2_target"], ap["rqc_seeds"]), parallel=True))
    out["rqc_s"] = round(time.perf_counter() - t0, 2)
    out["rqc_cached"] = bool(DC.last_search and DC.last_search[0].get("cached"))
    n, d = ap["svqa_qubits"], ap["svqa_depth"]
    pt = tc.backend.convert_to_tensor(np.random.default_rng(5).uniform(0.2, 1.2, [n, d, 2]).astype(np.float32))
    t0 = time.perf_counter()
    DC._get_tree_data(svqa_network(tc, n, d), pt, dict(svqa_search_options(ap["svqa_slices"], ap["svqa_seeds"]), parallel=True))
    out["svqa_s"] = round(time.perf_counter() - t0, 2)
    out["svqa_cached"] = bool(DC.last_search and DC.last_search[0].get("cached"))
    return out


def rqc_leg(tc, torch, dist, args, rank, world):
    """BASELINE config 4: single amplitude <0^32|C|0^32> of a 32-qubit random circuit on a 4x8 grid (brickwork
    of Haar-random two-qubit gates, reference gates.py:852-863), complex64, through DistributedContractor:
    random-greedy path search + slicing to 2^27 elements, slices sharded over the ranks as
    reference experimental.py:881-890 and summed with one packed all-reduce."""
    import numpy as np
    from tcmi.experimental import DistributedContractor

    depth = args.rqc_depth
    nodes_fn = rqc_network(tc, depth)

    t0 = time.perf_counter()
    dc = DistributedContractor(nodes_fn, None, cotengra_options=rqc_search_options(args.rqc_log2_target, args.rqc_seeds))
    search_s = time.perf_counter() - t0
    seeds = list(DistributedContractor.last_search)
    v = dc.value(None, op=lambda x: x)          # staging run
    v = dc.value(None, op=lambda x: x)          # (the second call validates the traced node function)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    from tcmi import tn as TN

    TN.COUNTERS = TN.new_counters()
    t0 = time.perf_counter()
    v = dc.value(None, op=lambda x: x)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t = time.perf_counter() - t0
    per_rank_ms = rank_times(torch, dist, v.device, t, 1)
    if dist is not None:
        tt = torch.tensor([t], device=v.device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())
    cnt, TN.COUNTERS = TN.COUNTERS, None
    tree = dc.tree
    # slice-invariant work is repeated on every rank: time one slice and all local slices to split t = t_inv + S t_slice
    # (what an 8-GPU run can gain: experimental.py:881-890 gives each rank ceil(S / G) slices)
    split = None
    if tree.nslices > 1 and len(dc.my_slices) > 1:
        arrays = dc._arrays(None)

        def run(ids):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for r_ in tree.contract_slices(arrays, ids):
                pass
            torch.cuda.synchronize()
            return time.perf_counter() - t1

        run(dc.my_slices[:1])
        t_one = min(run(dc.my_slices[:1]) for _ in range(3))
        t_all = min(run(dc.my_slices) for _ in range(3))
        S = len(dc.my_slices)
        t_slice = max(0.0, (t_all - t_one) / (S - 1))
        t_inv = max(0.0, t_one - t_slice)
        per_rank = min(S, -(-tree.nslices // 8))      # slices a rank of an 8-GPU run holds
        t_rank = t_one if per_rank == 1 else min(run(dc.my_slices[:per_rank]) for _ in range(3))

        # the same with the slice-invariant subtrees split over 8 ranks (ContractionTree.invariant_shards; what
        # DistributedContractor does when it runs on more than one rank): this rank computes the subtrees of rank 0 --
        # the most loaded one -- and the other ranks' roots are already in place (the all-gather itself, a few MB over
        # xGMI, is not in this one-rank figure)
        def run_sharded(ids):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for r_ in tree.contract_slices(arrays, ids, shard=(0, 8, "emulate")):
                pass
            torch.cuda.synchronize()
            return time.perf_counter() - t1

        run_sharded(dc.my_slices[:per_rank])
        t_rank_sh = min(run_sharded(dc.my_slices[:per_rank]) for _ in range(3))
        loads = tree.invariant_shards(8)[2]
        for r_ in tree.contract_slices(arrays, dc.my_slices):     # back to the unsharded graphs for later callers
            pass
        split = {"slice_invariant_s": t_inv, "per_slice_s": t_slice, "local_slices": S,
                 "one_slice_run_s": t_one, "invariant_fraction_of_one_slice_run": t_inv / max(t_one, 1e-12),
                 # slices go out in pairs on two streams, so per_slice_s is the paired rate; the 8-rank figures are
                 # measured times of the slices ONE rank would hold, not a model
                 "one_rank_of_8_replicated_invariants_s": t_rank, "one_rank_of_8_sharded_invariants_s": t_rank_sh,
                 "projected_speedup_8_ranks_replicated_invariants": t_all / t_rank,
                 "projected_speedup_8_ranks": t_all / t_rank_sh,
                 "invariant_shard_model_us": [round(x * 1e6) for x in loads],
                 "projection_basis": "ONE-rank measurement: time of all local slices / time of the slices one rank of "
                                     "an 8-rank run would hold (with the invariant work that rank executes); no 8-GPU run"}
    cpu = None
    if dist is None and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline_rqc(tree, dc._arrays(None))
        cpu["gpu_seconds_per_amplitude"] = t
    flops = float(tree.total_flops())          # all slices (ContractionTree.total_flops includes nslices)
    steps, dep, _, _ = tree._symbolic_steps()
    n_inv = sum(1 for st in steps if not dep[st[4]])   # slice-invariant steps: computed once per rank
    return {
        "workload": f"32-qubit 4x8 random circuit depth {depth}, amplitude <0|C|0>, complex64, sliced to "
                    f"2^{args.rqc_log2_target} elements (SURVEY 8d config 4)",
        **({"cpu_baseline": cpu} if cpu is not None else {}),
        "nslices": int(tree.nslices), "slices_per_gpu": int(-(-tree.nslices // world)),
        "contraction_width": float(tree.contraction_width()), "log2_flops_total": float(np.log2(flops)),
        # primary rate = the flops the engine EXECUTED (slice-invariant steps once per call); the sliced tree's own
        # count (every step x nslices) is kept as algorithmic_tflops_sliced_tree
        **({"per_rank_contract_ms": per_rank_ms} if dist is not None else {}),
        "contract_s": t, "tflops": (cnt["gemm_flops"] + cnt["scattered_flops"]) / t / 1e12,
        "algorithmic_tflops_sliced_tree": flops / t / 1e12, "path_search_s": round(search_s, 2),
        "steps_per_slice": len(steps), "slice_invariant_steps": n_inv, "time_split": split,
        # the search is a small hyper-search over seeds (whole pipeline per seed, best tree by the engine's time model)
        # a tree found earlier (build(), a previous run) is loaded from the tree cache: path_search_s is then the load;
        # per_seed_search_s are the times of the search that produced it, on the host that ran it
        "path_search_cached": bool(seeds and seeds[0].get("cached")),
        "path_search": {"seeds": len(seeds), "best_model_ms": min(x["model_time_s"] for x in seeds) * 1e3,
                        "median_model_ms": float(np.median([x["model_time_s"] for x in seeds])) * 1e3,
                        "per_seed_model_ms": [round(x["model_time_s"] * 1e3, 1) for x in seeds],
                        "per_seed_search_s": [x["search_s"] for x in seeds],
                        # with W ranks the seeds are dealt to them (seed k to rank k mod W) and the best tree is broadcast
                        "per_seed": [{"seed": x["seed"], "rank": x.get("rank", 0)} for x in seeds]} if seeds else None,
        "amplitude": [float(v.real), float(v.imag)],
        # F_alg of the executed (sliced, slice-invariant parts once) steps over the wall time against the f32 MFMA
        # peak; stand-alone permutes are traffic outside B_alg ("wasted")
        "roofline": {
            "bound": "mfma", "achieved": (cnt["gemm_flops"] + cnt["scattered_flops"]) / t / 1e12,
            "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s",
            "frac": (cnt["gemm_flops"] + cnt["scattered_flops"]) / t / 1e12 / MFMA_F32_PEAK_TFS,
            "executed_flops_this_rank": cnt["gemm_flops"] + cnt["scattered_flops"],
            "algorithmic_bytes": cnt["gemm_bytes"] + cnt["scattered_bytes"],
            "wasted_traffic": cnt["permute_bytes"],
            "launches": {k: int(cnt[k + "_launches"]) for k in ("gemm", "scattered", "permute")},
        },
    }


def sliced_vqa_leg(tc, torch, dist, args, rank, world):
    """The reference's sliced-VQA workload scaled up (examples/slicing_auto_pmap_vqa.py:20-41,86-94): rzz / rx ladder,
    <Z> of the middle qubit, contraction tree sliced with ``slicing_opts={"target_slices": S}``, one
    ``DistributedContractor.value_and_grad`` per step (reference experimental.py:1182-1211): forward and backward sweep of
    every slice on the untaped kernels, replayed from HIP graphs (tn.contract_slices_vjp), slices sharded over the ranks,
    one packed all-reduce of [value || gradient]."""
    import numpy as np
    from tcmi.experimental import DistributedContractor

    n, d, S = args.svqa_qubits, args.svqa_depth, args.svqa_slices
    pv = np.random.default_rng(5).uniform(0.2, 1.2, [n, d, 2]).astype(np.float32)
    pt = tc.backend.convert_to_tensor(pv)

    nodes_fn = svqa_network(tc, n, d)
    t0 = time.perf_counter()
    dc = DistributedContractor(nodes_fn, pt, cotengra_options=svqa_search_options(S, args.svqa_seeds, args.svqa_seed0, args.svqa_minimize or None))
    search_s = time.perf_counter() - t0

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t0 = time.perf_counter()
    v, g = dc.value_and_grad(pt)          # staging: kernels, bit tables, the four graph captures
    sync()
    staging = time.perf_counter() - t0
    v, g = dc.value_and_grad(pt)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.svqa_steps):
        v, g = dc.value_and_grad(pt)
    sync()
    el = (time.perf_counter() - t0) / args.svqa_steps
    per_rank_ms = rank_times(torch, dist, pt.device, el, 1)
    ar_us = allreduce_latency(torch, dist, pt.device, 1 + pt.numel())
    if dist is not None:
        tt = torch.tensor([el], device=pt.device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    tree = dc.tree
    steps, dep, _, _ = tree._symbolic_steps()
    out = {
        "workload": f"rzz/rx ladder n={n} depth={d}, <Z_{n // 2}>, value_and_grad of {2 * n * d} parameters through "
                    f"DistributedContractor, tree sliced to {tree.nslices} slices (target_slices={S}), complex64",
        "ms_per_value_and_grad": el * 1e3, "nslices": int(tree.nslices), "slices_per_gpu": len(dc.my_slices),
        "steps_per_slice": len(steps), "slice_invariant_steps": sum(1 for st in steps if not dep[st[4]]),
        "contraction_width": float(tree.contraction_width()), "log10_flops_forward": dc.tree_info["log10_flops"],
        **({"per_rank_ms_per_value_and_grad": per_rank_ms, "allreduce_us": ar_us} if dist is not None else {}),
        "path_search_s": round(search_s, 2), "staging_s": round(staging, 2),
        "path_search_cached": bool(DistributedContractor.last_search and DistributedContractor.last_search[0].get("cached")),
        "path_search": [{k_: x.get(k_) for k_ in ("seed", "objective", "model_time_s", "search_s")} for x in DistributedContractor.last_search],
        "path_search_uncached_s": round(sum(x.get("search_s", 0.0) for x in DistributedContractor.last_search), 2),
        "value": float(v), "grad_norm": float(g.norm()),
    }
    # device time of the four graphs (HIP events around one replay each): what is left is host time of the user's
    # node function, the op on the result and torch's backward through the gate matrices
    c = getattr(tree, "_vjp_graph_cache", None)
    if c is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gt = {}
        for name, key in (("invariant_forward", "g_a"), ("slice_forward", "g_b"), ("slice_backward", "g_c"),
                          ("invariant_backward", "g_d")):
            if c[key] is None:
                continue
            torch.cuda.synchronize()
            e0.record()
            c[key].replay()
            e1.record()
            torch.cuda.synchronize()
            gt[name + "_ms"] = e0.elapsed_time(e1)
        ns = len(dc.my_slices)
        dev_ms = gt.get("invariant_forward_ms", 0) + gt.get("invariant_backward_ms", 0) + \
            ns * (gt.get("slice_forward_ms", 0) + gt.get("slice_backward_ms", 0))
        per8 = -(-tree.nslices // 8)
        dev8 = gt.get("invariant_forward_ms", 0) + gt.get("invariant_backward_ms", 0) + \
            per8 * (gt.get("slice_forward_ms", 0) + gt.get("slice_backward_ms", 0))
        # F_alg of forward + backward (the VJP of a tensordot is two tensordots: 3 x the forward flops) over device time
        fl = 3.0 * 10.0 ** dc.tree_info["log10_flops"]
        out["graphs"] = gt
        # achieved = F_alg over the measured step (the per-slice graphs run in pairs on two streams, so the step can be
        # shorter than the sum of its graphs' stand-alone times, which `device_ms_per_step` keeps for comparison)
        out["roofline"] = {"bound": "latency", "device_ms_per_step": dev_ms, "host_bound": bool(el * 1e3 > 1.2 * dev_ms),
                           "achieved": fl / el / 1e12, "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s",
                           "frac": fl / el / 1e12 / MFMA_F32_PEAK_TFS,
                           "note": "thousands of rank <= 21 steps of a few microseconds each: launch-latency bound; "
                                   "device_ms_per_step = sum of the graphs' stand-alone times (one stream)"}
        out["one_rank_of_8_device_ms"] = dev8
        out["projected_speedup_8_ranks_device_time"] = dev_ms / dev8 if dev8 > 0 else None
    # What ONE rank of an 8-rank run executes, measured in this process: its contiguous block of the slice table
    # (reference experimental.py:881-890) and ITS share of the slice-invariant subtrees in both directions (forward
    # roots / backward root cotangents of the other ranks arrive by all-gather / all-reduce in a real run and are not part
    # of a one-rank figure).  The most loaded rank is taken.
    if world == 1 and tree.nslices >= 8 and c is not None:
        try:
            from tcmi import distributed as D

            tab = D.slice_table(int(tree.nslices), 8)
            loads = tree.invariant_shards(8)[2]
            r8 = int(np.argmax(loads))
            keep = (dc.my_slices, getattr(dc, "_emulate_rank", None))
            dc.my_slices = [int(x) for x in tab[r8] if x >= 0]
            dc._emulate_rank = (r8, 8)
            for _ in range(2):
                dc.value_and_grad(pt)
            sync()
            t0 = time.perf_counter()
            for _ in range(args.svqa_steps):
                dc.value_and_grad(pt)
            sync()
            el8 = (time.perf_counter() - t0) / args.svqa_steps
            c8 = tree._vjp_graph_cache
            g8 = {}
            for name, key in (("invariant_forward", "g_a"), ("slice_forward", "g_b"), ("slice_backward", "g_c"),
                              ("invariant_backward", "g_d")):
                if c8.get(key) is None:
                    continue
                torch.cuda.synchronize()
                e0.record()
                c8[key].replay()
                e1.record()
                torch.cuda.synchronize()
                g8[name + "_ms"] = e0.elapsed_time(e1)
            dc.my_slices, dc._emulate_rank = keep
            out["one_rank_of_8_sharded"] = {
                "rank": r8, "slices": len([x for x in tab[r8] if x >= 0]), "ms_per_value_and_grad": el8 * 1e3,
                "graphs": g8, "invariant_shard_model_us": [round(x * 1e6) for x in loads],
                "projected_speedup_8_ranks": el / el8,
                "basis": "ONE-rank measurement of the work of the most loaded rank of 8 (its slice block, its share of the "
                         "invariant subtrees forward and backward); wall time of value_and_grad, host side included; no "
                         "8-GPU run"}
        except Exception as e:  # noqa: BLE001
            out["one_rank_of_8_sharded"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def statevector_leg(tc, torch, dist, args, rank, world, dev):
    """north_star's own size: ``Circuit.wavefunction`` (reference circuit.py:701-721) of HEA-B n = 28, depth 12, complex64
    through backend.jit(backend.vmap(f)) -- the state-vector plan (a cut of 12 ZZ layers would need a bond of 4096).
    Global batch --sv-batch states, sharded over the ranks in contiguous blocks (strong scaling, no collective),
    --sv-microbatch states per vmap call.  Reported: amplitudes/s; the HBM fraction of the EXECUTED plan (live-tile passes:
    algorithmic bytes = the bytes of the tiles that can be non-zero); the same kernels with every tile live
    (``dense_plan``: every pass moves the whole state); SURVEY 8(d)'s cross-plan anchor B_sv / t (the bytes the canonical
    gate-by-gate state-vector plan would move, over this time); batch-1 latency."""
    import numpy as np
    from tcmi import distributed as D
    from tcmi import executor as X

    n, d, Bg, mb = args.sv_qubits, args.sv_depth, args.sv_batch, max(1, args.sv_microbatch)
    lo, hi = D.shard_range(Bg, rank, world)
    params_all = np.random.default_rng(n).uniform(0, 2 * np.pi, [Bg, 2 * d, n]).astype(np.float32)
    params = torch.from_numpy(params_all[lo:hi]).to(dev)
    chunks = [params[b0: b0 + mb] for b0 in range(0, hi - lo, mb)]

    def wavefunction(p):
        return build_circuit(tc, n, d, p).wavefunction()

    fwd = tc.backend.jit(tc.backend.vmap(wavefunction))

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        out = None
        for ch in chunks:
            out = fwd(ch)
        return out

    t0 = time.perf_counter()
    for _ in range(3):                   # staging; the first two calls validate the traced pipeline
        st = step()
    sync()
    staging = time.perf_counter() - t0
    X.EVENT_LOG = []
    t0 = time.perf_counter()
    for _ in range(args.sv_steps):
        st = step()
    sync()
    el = time.perf_counter() - t0
    ev = summarize_events(X.EVENT_LOG)
    X.EVENT_LOG = None
    per_rank_ms = rank_times(torch, dist, dev, el, args.sv_steps)
    if dist is not None:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    nrm = float((st[0].abs() ** 2).sum().item()) if st is not None else None
    chk = torch.zeros(1, device=dev, dtype=torch.float64)      # sum over the global batch of <psi|Z_0|psi>
    for ch in chunks:
        pr = (fwd(ch).abs() ** 2).to(torch.float64)
        chk += (pr[:, : 2 ** (n - 1)].sum() - pr[:, 2 ** (n - 1):].sum())
    if dist is not None:
        dist.all_reduce(chk)
    step_s = el / args.sv_steps
    executed = hbm_entry("tcmi_spec_fwd_p* (plan-specialised forward passes, live tiles)", ev.get("pass"), args.sv_steps)
    cc = build_circuit(tc, n, d, params[0] if hi > lo else torch.from_numpy(params_all[0]).to(dev))._compiled()
    is_cut = isinstance(cc, X.CutCircuit)
    npass = None if is_cut else len(cc.descs)
    dense = None
    if X.SPARSE_START and dist is None and not is_cut:
        X.SPARSE_START = False
        try:
            step()
            sync()
            X.EVENT_LOG = []
            td0 = time.perf_counter()
            std = step()
            torch.cuda.synchronize()
            td = time.perf_counter() - td0
            evd = summarize_events(X.EVENT_LOG)
            dense = hbm_entry("the forward pass kernels, every tile live", evd.get("pass"), 1)
            if dense is not None:
                dense["ms_per_step"] = td * 1e3
                dense["amplitudes_per_s"] = float(hi - lo) * (2 ** n) / td
                dense["max_abs_state_difference"] = float((std - st).abs().max().item())
            del std
        finally:
            X.EVENT_LOG = None
            X.SPARSE_START = True
    del st
    f1 = tc.backend.jit(wavefunction)
    lat1 = None
    if hi > lo:
        for _ in range(3):
            f1(params[0])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            f1(params[0])
        torch.cuda.synchronize()
        lat1 = (time.perf_counter() - t1) / 5
    b_sv = ((d - 1) * (n - 1) + 2) * 2.0 * (2 ** n) * 8      # bytes of the gate-by-gate plan per state
    kernel_ms = sum(v["ms"] for v in ev.values()) / max(1, args.sv_steps)
    from tcmi import specialize as SP

    return {
        "workload": f"HEA-B statevector contraction n={n} depth={d} complex64 (north_star size): one step = {Bg} circuits, "
                    f"{mb} per vmap call, timed through backend.jit(backend.vmap(wavefunction))",
        "amplitudes_per_s": float(Bg) * (2 ** n) * args.sv_steps / el, "ms_per_step": step_s * 1e3, "steps": args.sv_steps,
        "ms_per_state": step_s * 1e3 / max(1, hi - lo), "global_batch": Bg, "batch_per_gpu": hi - lo,
        **({"per_rank_ms_per_step": per_rank_ms} if per_rank_ms is not None else {}),
        "contraction": "cut" if is_cut else "state-vector", "passes": npass, "staging_s": round(staging, 3),
        "state_norm": nrm, "z0_checksum": float(chk.item()),
        "latency_batch1_ms": None if lat1 is None else lat1 * 1e3,
        "roofline": {
            "bound": "hbm",
            # the fraction north_star's ">= 50 % of the HBM roofline" is read against: bytes the executed plan has to move
            # (live tiles: read as far as they can be non-zero, written whole) over the HIP-event time of its launches
            "executed_plan": executed,
            "step": {"bound": "hbm", "executed_bytes_per_step": sum(v["work"] for v in ev.values()) / max(1, args.sv_steps),
                     "achieved": sum(v["work"] for v in ev.values()) / max(1, args.sv_steps) / step_s / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": sum(v["work"] for v in ev.values()) / max(1, args.sv_steps) / step_s / 1e9 / HBM_PEAK_GBS,
                     "kernel_ms_per_step": kernel_ms, "wall_ms_per_step": step_s * 1e3},
            "dense_plan": dense,
            "B_sv_bytes_per_state": b_sv, "B_sv_GBps": b_sv * (hi - lo) / step_s / 1e9,
            "B_sv_note": "cross-plan anchor of SURVEY 8(d): [(d-1)(n-1)+2] x 2 x 2^n x 8 B per state over the measured time; "
                         "above the 8 TB/s peak = the plan moves fewer bytes than gate-by-gate execution, not a bandwidth",
        },
        "specialised_kernels": {"forward": sum(1 for v in SP._LOADED.values() if v.meta.get("kind") == "forward"),
                                "compiled_in_this_process": SP.STATS["compiled"]},
    }


def heisenberg_leg(tc, torch, args, dev):
    """The VQE step with a Heisenberg chain as the energy (reference tensorcircuit/quantum.py:2131-2219
    heisenberg_hamiltonian; templates/measurements.py:156-191): sum_i (XX + YY + ZZ)_{i, i+1} on the HEA-B state of config
    3's size, one micro-batch through jit(vvag).  Two-factor strings are born in the reverse sweep where their qubits meet
    untouched in a tile (OP_XFOLD2, executor.fold_setup); the pairs that straddle two tiles keep going through the Pauli-sum
    tile passes.  Reported next to the same step with the fold off (every string through the tile passes).  One rank."""
    import numpy as np
    from tcmi import executor as X

    n, d, B = args.vqe_qubits, args.vqe_depth, max(1, args.vqe_microbatch)
    params = torch.from_numpy(np.random.default_rng(29).normal(0, 0.1, [B, 2 * d, n]).astype(np.float32)).to(dev)

    def energy(p):
        c = tc.templates.blocks.example_block(tc.Circuit(n), p, nlayers=d)
        e = 0.0
        for i in range(n - 1):
            e += c.expectation_ps(x=[i, i + 1]) + c.expectation_ps(y=[i, i + 1]) + c.expectation_ps(z=[i, i + 1])
        return tc.backend.real(e)

    out = {}
    old = os.environ.get("TCMI_PAULI_FOLD")
    try:
        for flag in ("1", "0"):
            os.environ["TCMI_PAULI_FOLD"] = flag
            vvag = tc.backend.jit(tc.backend.vvag(energy, argnums=0, vectorized_argnums=0))
            for _ in range(3):            # staging + validation of the traced pipeline (+ kernels of a hot plan compiled)
                v, g = vvag(params)
            torch.cuda.synchronize()
            X.EVENT_LOG = []
            t0 = time.perf_counter()
            v, g = vvag(params)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            ev = summarize_events(X.EVENT_LOG)
            X.EVENT_LOG = None
            out[flag] = {"ms_per_sample": el / B * 1e3, "mean_energy": float(v.mean().item()), "grad_norm": float(g.norm().item()),
                         "kernel_ms": {k: round(x["ms"], 3) for k, x in ev.items()},
                         "pauli_sum_launches": ev.get("pauli_sum", {}).get("launches", 0),
                         "pauli_sum_bytes": ev.get("pauli_sum", {}).get("work", 0.0)}
            del vvag, v, g
            torch.cuda.empty_cache()
    finally:
        X.EVENT_LOG = None
        if old is None:
            os.environ.pop("TCMI_PAULI_FOLD", None)
        else:
            os.environ["TCMI_PAULI_FOLD"] = old
    on, off = out["1"], out["0"]
    return {"workload": f"HEA-B n={n} depth={d}, energy = sum_i (XX + YY + ZZ)_(i,i+1) ({3 * (n - 1)} strings), value_and_grad, "
                        f"vmap batch {B}, complex64",
            "ms_per_sample": on["ms_per_sample"], "strings_born_in_the_sweep": on, "every_string_through_tile_passes": off,
            "speedup_from_the_fold": off["ms_per_sample"] / on["ms_per_sample"],
            "energy_difference": abs(on["mean_energy"] - off["mean_energy"])}


def hea_a_leg(tc, torch, args, dev):
    """SURVEY 8(d) config 2, secondary workload: HEA-A (reference benchmarks/scripts_v2/benchmark_core.py:6-14: H layer, then
    per layer rx on every qubit and a CNOT ladder), same size and call as the headline -- backend.jit(backend.vmap(
    wavefunction)), batch --batch, this rank only (it is a per-GPU figure next to the headline, not a second headline)."""
    import numpy as np
    from tcmi import executor as X

    n, d, B = args.qubits, args.depth, args.batch
    params = torch.from_numpy(np.random.default_rng(n + 7).uniform(0, 2 * np.pi, [B, d, n]).astype(np.float32)).to(dev)

    def wavefunction(p):
        c = tc.Circuit(n)
        for i in range(n):
            c.h(i)
        for j in range(d):
            for i in range(n):
                c.rx(i, theta=p[j, i])
            for i in range(n - 1):
                c.cx(i, i + 1)
        return c.wavefunction()

    fwd = tc.backend.jit(tc.backend.vmap(wavefunction))
    for _ in range(3):
        st = fwd(params)
    torch.cuda.synchronize()
    X.EVENT_LOG = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = fwd(params)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / args.steps
    ev = summarize_events(X.EVENT_LOG)
    X.EVENT_LOG = None
    nrm = float((st[0].abs() ** 2).sum().item())
    from tcmi.executor import CutCircuit

    c0 = tc.Circuit(n)
    for i in range(n):
        c0.h(i)
    for j in range(d):
        for i in range(n):
            c0.rx(i, theta=params[0, j, i])
        for i in range(n - 1):
            c0.cx(i, i + 1)
    cc = c0._compiled()
    return {"workload": f"HEA-A statevector contraction n={n} depth={d} complex64, vmap batch {B}, through "
                        f"backend.jit(backend.vmap(wavefunction))",
            "amplitudes_per_s_per_gpu": B * (2**n) / el, "ms_per_call": el * 1e3, "state_norm": nrm,
            "contraction": "cut" if isinstance(cc, CutCircuit) else "state-vector",
            "passes": hbm_entry("tile-VM gate passes", ev.get("pass"), args.steps),
            "kernel_ms_per_call": {k: v["ms"] / args.steps for k, v in ev.items()}}


_T0 = time.perf_counter()


def _trace(msg):
    """One stderr line per leg boundary and rank when there is more than one rank (or TCMI_BENCH_TRACE=1): which leg a
    rank was in when a job died or hung is otherwise unknowable -- the JSON line is printed last."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("TCMI_BENCH_TRACE") == "1":
        try:
            import torch

            free = torch.cuda.mem_get_info()[0] / 2**30 if torch.cuda.is_initialized() else float("nan")
        except Exception:  # noqa: BLE001
            free = float("nan")
        print(f"[bench rank {os.environ.get('RANK', '0')} +{time.perf_counter() - _T0:7.1f}s free {free:6.1f} GiB] {msg}",
              file=sys.stderr, flush=True)


def _guard(name, fn, *a, dist=None, need_bytes=0):
    """Secondary legs must never take the headline line down with them.  Before a leg starts, the objects the earlier legs
    left alive (compiled plans, captured graphs, traced pipelines: millions of Python objects by the fourth leg) are moved
    out of the garbage collector's young generations: a full collection in the middle of a host-bound timed loop (the
    sliced-VQA leg issues thousands of small tensor ops per step) otherwise shows up as a 100 ms step.

    With more than one rank a leg holds collectives, so the ranks must agree on running it and none may leave it alone:
    * BEFORE the leg every rank compares the device memory the leg needs (``need_bytes``: working set + 2 GiB for the
      runtime's own allocations -- kernel scratch, code objects, graph pools; a device that is full makes THOSE fail, and
      that surfaces as a queue abort, HSA_STATUS_ERROR_EXCEPTION, not as a Python error) with what is free on its device
      (divided by the ranks that share the device in the oversubscribed launch-path test), and ONE all-reduce(MIN) decides:
      the leg runs on every rank or is skipped on every rank, with the reason in the JSON line;
    * an exception INSIDE the leg on one rank would leave the others in a collective forever: the rank prints the
      traceback and exits with code 17, the launcher (self_launch, torchrun) takes the job down -- non-zero, never a hang.
    One rank: the exception becomes an ``error`` entry and the line is still printed."""
    import gc

    import torch

    gc.collect()
    gc.freeze()
    if need_bytes:
        torch.cuda.empty_cache()
        if dist is not None:          # every rank has let go of the previous leg's memory before anybody looks
            torch.cuda.synchronize()
            dist.barrier()
        free, _total = torch.cuda.mem_get_info()
        sharers = max(1, int(os.environ.get("TCMI_BENCH_SHARERS", "1")))
        need = need_bytes + (2 << 30)
        ok = free / sharers >= need
        if dist is not None:
            flag = torch.tensor([1.0 if ok else 0.0, free / sharers], device="cuda", dtype=torch.float64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok, free_min = bool(flag[0].item() > 0.5), float(flag[1].item())
        else:
            free_min = free / sharers
        if not ok:
            _trace(f"leg {name}: skipped (memory)")
            return {"skipped": f"{name}: needs {need / 2**30:.1f} GiB of device memory per rank, "
                               f"{free_min / 2**30:.1f} GiB free on the fullest device ({sharers} rank(s) per device)"}
    try:
        _trace(f"leg {name}: start")
        if os.environ.get("TCMI_BENCH_RAISE_IN") == f"{name}:{os.environ.get('RANK', '0')}":     # tests: a rank failing mid-leg
            raise MemoryError("injected failure (TCMI_BENCH_RAISE_IN)")
        out = fn(*a)
        _trace(f"leg {name}: done")
        return out
    except Exception as e:  # noqa: BLE001
        if dist is not None:
            import traceback

            traceback.print_exc()
            print(f"bench.py: rank {os.environ.get('RANK', '0')} failed in leg {name!r} ({type(e).__name__}); the leg holds "
                  f"collectives, leaving the job (exit 17)", file=sys.stderr, flush=True)
            os._exit(17)
        return {"error": f"{name}: {type(e).__name__}: {e}"[:300]}


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes of this very command (the
    reference drives all devices from one command, examples/slicing_auto_pmap_vqa.py:8-10,60-66), one per GPU, with
    the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT) and the nccl
    (= RCCL) backend.  The parent makes no GPU call before or after the spawn -- a process that has initialised the
    GPU must not start other programs on this pool -- it only waits; rank 0 prints the JSON line on the inherited
    stdout.  Returns the worst exit code; if a rank dies the others are terminated (by PID)."""
    import socket
    import subprocess

    import torch   # device_count() does not initialise the GPU

    have = torch.cuda.device_count()
    share = have < n and have >= 1 and os.environ.get("TCMI_BENCH_OVERSUBSCRIBE") == "1"
    if have < n and not share:
        print(f"bench.py: --gpus {n} but only {have} device(s) visible "
              f"(TCMI_BENCH_OVERSUBSCRIBE=1 runs the ranks round-robin on the visible devices over gloo: a test of the "
              f"launch path, not a measurement)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if share:
            env.update(LOCAL_RANK=str(r % have), TCMI_BENCH_BACKEND="gloo",
                       TCMI_BENCH_SHARERS=os.environ.get("TCMI_BENCH_SHARERS") or str(-(-n // have)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while procs:
            for p_ in list(procs):
                code = p_.poll()
                if code is None:
                    continue
                procs.remove(p_)
                if code != 0:
                    rc = rc or code
                    for q_ in procs:      # a dead rank would leave the others in a collective forever
                        q_.terminate()
            time.sleep(0.2)
    finally:
        for p_ in procs:
            p_.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--qubits", type=int, default=24)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--batch", type=int, default=32, help="circuits per vmap call (micro-batch of the headline step; 8: 1.66e11, 16: 1.74e11, 32: 1.81e11, 64: 1.82e11 amplitudes/s)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="headline step = this many circuits contracted, whatever the number of GPUs (strong scaling): "
                         "sharded over the ranks in contiguous blocks, each rank works through its block --batch at a time; "
                         "0 (default) = --batch-per-gpu circuits on every GPU (weak scaling)")
    ap.add_argument("--batch-per-gpu", type=int, default=64, help="circuits per step and GPU when --global-batch is not given")
    ap.add_argument("--cpu-qubits", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vqe-qubits", type=int, default=28, help="VQE leg (config 3): qubits; 0 disables the leg")
    ap.add_argument("--vqe-depth", type=int, default=12)
    ap.add_argument("--vqe-batch", type=int, default=32, help="VQE leg: global vmap batch (sharded over ranks)")
    ap.add_argument("--vqe-steps", type=int, default=2)
    ap.add_argument("--vqe-microbatch", type=int, default=8, help="samples per vvag call (bounds HBM use)")
    ap.add_argument("--vqe-streams", type=int, default=1, help="micro-batches of the VQE step alternate over this many HIP streams")
    ap.add_argument("--sv-qubits", type=int, default=28, help="n = 28 statevector leg (north_star size): qubits; 0 disables")
    ap.add_argument("--sv-depth", type=int, default=12)
    ap.add_argument("--sv-batch", type=int, default=16, help="statevector leg: global batch of circuits per step (sharded over ranks)")
    ap.add_argument("--sv-microbatch", type=int, default=8, help="statevector leg: circuits per vmap call")
    ap.add_argument("--sv-steps", type=int, default=3)
    ap.add_argument("--mps-qubits", type=int, default=64, help="MPS TEBD leg (config 5): qubits; 0 disables the leg")
    ap.add_argument("--mps-chi", type=int, default=128)
    ap.add_argument("--mps-sweeps", type=int, default=2)
    ap.add_argument("--mps-chains", type=int, default=16, help="config 5: independent chains through backend.vmap (0/1 disables)")
    ap.add_argument("--rqc-depth", type=int, default=16, help="config 4 leg (32-qubit RQC amplitude): depth; 0 disables")
    ap.add_argument("--rqc-log2-target", type=int, default=27)
    ap.add_argument("--rqc-seeds", type=int, default=8, help="config 4: seeds of the path hyper-search")
    ap.add_argument("--svqa-qubits", type=int, default=30, help="sliced-VQA leg (value_and_grad of a sliced network): qubits; 0 disables")
    ap.add_argument("--svqa-depth", type=int, default=8)
    ap.add_argument("--svqa-slices", type=int, default=8)
    ap.add_argument("--svqa-steps", type=int, default=6)
    ap.add_argument("--svqa-seeds", type=int, default=8,
                    help="sliced-VQA leg: seeds of the path hyper-search (seed 0 alone finds one of the worst trees of 0..7: "
                         "14.9 ms per value_and_grad against 9.7 for the best of eight, gpurun_out/r6i)")
    ap.add_argument("--svqa-seed0", type=int, default=0, help="sliced-VQA leg: first seed")
    ap.add_argument("--svqa-minimize", default="combo", help="sliced-VQA leg: cotengra `minimize` ('' = the engine's time model)")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph replay measurement")
    ap.add_argument("--no-heisenberg", action="store_true", help="skip the Heisenberg-chain variant of the VQE step")
    ap.add_argument("--no-hea-a", action="store_true", help="skip the HEA-A secondary workload of config 2")
    ap.add_argument("--no-traffic-probe", action="store_true", help="skip the rocprofv3 PMC child runs (traffic = null)")
    ap.add_argument("--probe-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--contractor", default="greedy",
                    help="greedy/auto: cost model picks the contraction order; plain: state-vector plan; cut: cut contraction")
    ap.add_argument("--lowbits", type=int, default=None)
    ap.add_argument("--R", type=int, default=None)
    ap.add_argument("--LT", type=int, default=None)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.probe_child:
        sys.exit(self_launch(args.gpus))   # one fresh process per GPU; this parent never touches the GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.probe_child:
        args.no_graph = args.no_cpu_baseline = args.no_traffic_probe = True
        args.vqe_qubits = args.mps_qubits = args.rqc_depth = args.svqa_qubits = args.sv_qubits = 0

    # HBM traffic of the dominant kernel, measured by the PMC counters on this very command (child processes,
    # started before this process touches the GPU)
    KERNELS = ("cgemm_split_kernel", "cgemm_dma128_kernel", "cgemm_dma_kernel", "cgemm_mfma_kernel", "pass2_kernel")
    traffic = {}
    if rank == 0 and world == 1 and not args.no_traffic_probe:
        traffic = traffic_probe(args, KERNELS)

    import numpy as np
    import torch

    dist = None
    # TCMI_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL init, barriers, all-reduces) with a
    # world of one rank too (used to validate that path on a one-GPU box under torchrun)
    if world > 1 or os.environ.get("TCMI_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if os.environ.get("TCMI_BENCH_BACKEND", "nccl") == "gloo":   # ranks sharing a device (launch-path test only)
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import tcmi as tc
    from tcmi import executor as X
    from tcmi.executor import CutCircuit

    tc.set_backend("hip")
    tc.set_dtype("complex64")
    opts = {k: getattr(args, k) for k in ("lowbits", "R", "LT") if getattr(args, k) is not None}
    tc.set_contractor(args.contractor, **opts)
    n, d, B = args.qubits, args.depth, args.batch

    if os.environ.get("TCMI_BENCH_KILL_RANK") == str(rank) and world > 1:
        os._exit(3)        # tests/test_gpu_bench_multirank.py: a rank that dies must take the whole job down

    # synthetic parameters: SURVEY 8(d) config-2 generator, one independent row per circuit of the GLOBAL batch (the
    # same rows whatever the number of ranks); this rank contracts its contiguous block of them, B per vmap call
    from tcmi import distributed as D_

    weak = args.global_batch <= 0
    Bg = max(args.batch_per_gpu * world if weak else args.global_batch, B)
    lo, hi = D_.shard_range(Bg, rank, world)
    rng = np.random.default_rng(n)
    params_all = rng.uniform(0, 2 * np.pi, [Bg, 2 * d, n]).astype(np.float32)
    params_loc = torch.from_numpy(params_all[lo:hi]).to(dev)
    chunks = [params_loc[b0: b0 + B] for b0 in range(0, hi - lo, B)]
    params = chunks[0] if chunks else torch.from_numpy(params_all[:B]).to(dev)

    # the product call, through the reference's public API: K.jit(K.vmap(f)) with f = Circuit(...).wavefunction()
    def wavefunction(p):
        return build_circuit(tc, n, d, p).wavefunction()

    fwd = tc.backend.jit(tc.backend.vmap(wavefunction))
    _trace("headline: start")
    t0 = time.perf_counter()
    state = fwd(params)               # staging: records the structure, compiles the plan (cached by structure)
    torch.cuda.synchronize()
    staging_s = time.perf_counter() - t0
    cc = build_circuit(tc, n, d, params[0])._compiled()
    is_cut = isinstance(cc, CutCircuit)
    st = cc.stats()

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 2)):   # the first two calls validate the traced pipeline against the plain path
        state = fwd(params)
    sync()
    for ch in chunks[-1:]:                 # a ragged last chunk has its own batch size: stage it outside the timed region
        if ch.shape[0] != B:
            for _ in range(3):
                fwd(ch)
    sync()
    X.EVENT_LOG = []
    t0 = time.perf_counter()
    for k in range(args.steps):
        for ch in chunks:                  # one step = the whole global batch: this rank's block, B circuits per call
            state = fwd(ch)
    sync()
    elapsed = time.perf_counter() - t0
    ev = summarize_events(X.EVENT_LOG)
    X.EVENT_LOG = None
    traced = bool(getattr(fwd, "stats", {}).get("fast", 0) >= args.steps)
    # the same step with every join on the exact-f32 MFMA kernel (tcmi_cgemm), and how far the two results differ
    join_f32 = None
    if is_cut and X.JOIN_GEMM != "f32" and not args.probe_child:
        split_state = fwd(chunks[0]).clone() if chunks else None
        X.JOIN_GEMM = "f32"
        try:
            for _ in range(2):
                state = fwd(params)
            sync()
            tf0 = time.perf_counter()
            for k in range(args.steps):
                for ch in chunks:
                    state = fwd(ch)
            sync()
            tf = time.perf_counter() - tf0
            diff = float((fwd(chunks[0]) - split_state).abs().max().item()) if chunks else None
            join_f32 = {"ms_per_step": tf / args.steps * 1e3, "amplitudes_per_s": float(Bg) * (2**n) * args.steps / tf,
                        "max_abs_difference_of_the_two_states": diff,
                        "largest_amplitude": float(split_state.abs().max().item()) if chunks else None}
        finally:
            X.JOIN_GEMM = "split"
        del split_state
    # a checksum of the step's result that every world size must reproduce: sum over the global batch of <psi|Z_0|psi>
    chk = torch.zeros(1, device=dev, dtype=torch.float64)
    for ch in chunks:
        pr = (fwd(ch).abs() ** 2).to(torch.float64)
        chk += (pr[:, : 2 ** (n - 1)].sum() - pr[:, 2 ** (n - 1):].sum())
    if dist is not None:
        dist.all_reduce(chk)
    checksum = float(chk.item())
    if args.probe_child:
        return
    # sanity: the state is normalised (cheap property check at full size)
    nrm = float((state[0].abs() ** 2).sum().item())

    # batch-1 latency through the same API (the metric is per batch element)
    f1 = tc.backend.jit(wavefunction)
    for _ in range(3):
        f1(params[0])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        f1(params[0])
    torch.cuda.synchronize()
    lat1 = (time.perf_counter() - t1) / args.steps

    # the same step replayed from a hipGraph (launch overhead removed); reported next to the eager figure
    graph_info = None
    if not args.no_graph:
        try:
            from tcmi.executor import GraphedState

            order = []
            for j in range(d):
                order += [(2 * j) * n + i for i in range(n - 1)]
                order += [(2 * j + 1) * n + i for i in range(n)]
            # (this rank's chunk: with more than --global-batch / --batch ranks a chunk holds fewer than B circuits -- sizing
            # the replay by B instead was the out-of-range gather that killed every 4-rank run of round 5 with
            # HSA_STATUS_ERROR_EXCEPTION, a device-side index assertion; profiles/r06_over4_fault.txt)
            Bl = int(params.shape[0])
            pmat = params.reshape(Bl, -1)[:, torch.tensor(order, device=dev)].contiguous()
            gs = GraphedState(cc, Bl)
            for _ in range(args.warmup):
                gs(pmat)
            sync()
            tg0 = time.perf_counter()
            for _ in range(args.steps):
                gs(pmat)
            sync()
            tg = time.perf_counter() - tg0
            same = bool(torch.allclose(gs.out, fwd(params), atol=1e-6, rtol=0))     # the eager result of the same chunk
            graph_info = {"ms_per_step": tg / args.steps * 1e3, "amplitudes_per_s_per_gpu": Bl * (2**n) * args.steps / tg,
                          "circuits_per_replay": Bl, "matches_eager": same}
            del gs
        except Exception as e:  # noqa: BLE001
            graph_info = {"error": f"{type(e).__name__}: {e}"[:200]}
    per_rank_ms = rank_times(torch, dist, dev, elapsed, args.steps)
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    del state
    torch.cuda.empty_cache()
    _trace("headline: done")

    hea_a = None
    if rank == 0 and not args.probe_child and not args.no_hea_a:
        hea_a = _guard("hea_a", hea_a_leg, tc, torch, args, dev)
        torch.cuda.empty_cache()
    sv28 = None
    if args.sv_qubits:
        lo_, hi_ = D_.shard_range(args.sv_batch, rank, world)
        # per micro-batch: the output, the previous step's output still alive, the plain path's result on the validating
        # call and one working copy (measured: 34 GiB allocated at 4 states of 2 GiB)
        sv28 = _guard("statevector_n%d" % args.sv_qubits, statevector_leg, tc, torch, dist, args, rank, world, dev, dist=dist,
                      need_bytes=int(4.5 * max(1, min(args.sv_microbatch, hi_ - lo_)) * (2 ** args.sv_qubits) * 8))
        torch.cuda.empty_cache()
    vqe = None
    if args.vqe_qubits:
        # working set of a micro-batch of the traced step: psi, lambda and the per-call copies = 4.25 states per sample
        # (68 GiB measured at 8 samples of 2 GiB)
        lo_, hi_ = D_.shard_range(args.vqe_batch, rank, world)
        vqe = _guard("vqe_step", vqe_leg, tc, torch, dist, args, rank, world, dev, dist=dist,
                     need_bytes=int(4.5 * max(1, min(args.vqe_microbatch, hi_ - lo_)) * (2 ** args.vqe_qubits) * 8))
    heis = None
    if args.vqe_qubits and rank == 0 and dist is None and not args.no_heisenberg:
        torch.cuda.empty_cache()
        heis = _guard("vqe_heisenberg", heisenberg_leg, tc, torch, args, dev,
                      need_bytes=int(5.5 * args.vqe_microbatch * (2 ** args.vqe_qubits) * 8))
    # the host-bound leg first: it is the one that feels what earlier legs leave behind (graph memory pools, cached plans)
    svqa = None
    if args.svqa_qubits:
        _burn = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get("TCMI_BENCH_BURN_STREAMS", "0")))]  # experiment
        for st_ in _burn:          # ... and USED: a stream gets its hardware queue with its first work
            with torch.cuda.stream(st_):
                torch.zeros(8, device=dev).add_(1)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        svqa = _guard("sliced_vqa", sliced_vqa_leg, tc, torch, dist, args, rank, world, dist=dist, need_bytes=1 << 30)

    rqc = None
    if args.rqc_depth:
        torch.cuda.empty_cache()
        # two slices in flight (two streams), each a chain of intermediates of up to the target size
        rqc = _guard("rqc_amplitude", rqc_leg, tc, torch, dist, args, rank, world, dist=dist,
                     need_bytes=12 * (2 ** args.rqc_log2_target) * 8)

    if rank == 0:
        amps = float(Bg) * (2**n) * args.steps
        value = amps / elapsed
        pcfg = cc.cfg
        if is_cut:                       # the half-circuit batches have their own (small-tile) plan configuration
            half = cc.left.single if cc.left.s == 0 else cc.left.prefix
            pcfg = half.cfg
        from tcmi import specialize as SP_

        nspec_fwd = sum(1 for v in SP_._LOADED.values() if v.meta.get("kind") == "forward")
        pass_entry = hbm_entry(("tcmi_spec_forward (plan-specialised gate passes; programs of pass2_kernel<%d,%d>)" if nspec_fwd
                                else "tcmi::pass2_kernel<%d,%d> (tile-VM gate passes)") % (pcfg.R, pcfg.LT), ev.get("pass"),
                               args.steps)
        if is_cut:
            g = ev["gemm"]
            M, N, K = 2**cc.spec.n_left, 2 ** (n - cc.spec.n_left), cc.K
            avg_us = g["ms"] * 1e3 / g["launches"]
            gl_call = g["launches"] / float(args.steps * max(1, len(chunks)))
            alg = g["work"] / g["launches"]           # 8 real flops per complex MAC
            exe = 0.75 * alg                          # the kernel issues Gauss's 3-product form: 6 flops per MAC
            # the join runs on cgemm_dma128_kernel when M and N are multiples of 128 (every n >= 14 cut), else on the
            # 64-tile DMA kernel / the register-staged kernel
            gk = "cgemm_dma128_kernel" if (M % 128 == 0 and N % 128 == 0 and K % 16 == 0) else \
                ("cgemm_dma_kernel" if (M % 64 == 0 and N % 64 == 0 and K % 16 == 0) else "cgemm_mfma_kernel<true>")
            peak = MFMA_F32_PEAK_TFS
            split = X.JOIN_GEMM != "f32" and M % 128 == 0 and N % 128 == 0 and K % 32 == 0
            f16 = split and X.JOIN_GEMM == "split" and getattr(cc, "_f16", None) is not None
            if f16:
                # tcmi_cgemm_split_f16: 3 real products x 3 f16 piece products per complex MAC = 18 flops on the f16 pipe
                # (same dense peak as bf16): HALF the executed flops of the three-piece kernel for the same product, so this
                # fraction is not comparable with the three-piece kernel's -- compare `avg_launch_us` and `algorithmic_achieved`
                gk, exe, peak = "cgemm_split_kernel<0, %d, 2>" % int(cc.spec.epilogue is not None), 2.25 * alg, MFMA_BF16_PEAK_TFS
            elif split:
                # tcmi_cgemm_split: 3 real products x 6 bf16 piece products per complex MAC = 36 flops on the bf16 pipe
                gk, exe, peak = "cgemm_split_kernel<0, %d, 3>" % int(cc.spec.epilogue is not None), 4.5 * alg, MFMA_BF16_PEAK_TFS
            tr = traffic.get(gk.split("<")[0])
            roof = {
                "bound": "mfma", "kernel": f"tcmi::{gk} (cut-contraction join GEMM)",
                # frac = EXECUTED flops (what the MFMA pipe does) against the dense MFMA peak of the pipe it runs on
                "achieved": exe / (avg_us * 1e-6) / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": exe / (avg_us * 1e-6) / 1e12 / peak,
                **({"pipe": "f16 MFMA, f32 operands (bounded by the cut: scales %g, %g) cut into two f16 pieces, three piece "
                            "products per real product: 18 executed flops per complex multiply-add where the three-piece bf16 "
                            "kernel of rounds 4-5 executed 36 (f32 accuracy: error against float64 equal to the f32 MFMA "
                            "kernel's, tests/test_gpu_gemm_split.py)" % cc._f16} if f16 else
                   {"pipe": "bf16 MFMA, f32 operands cut into three bf16 pieces, six piece products per real product "
                            "(f32 accuracy: error against float64 equal to the f32 MFMA kernel's, "
                            "tests/test_gpu_gemm_split.py)"} if split else {}),
                # the kernel's job is to WRITE the joined states once: the same launch against the HBM roof
                "hbm_frac_on_algorithmic_bytes": 8.0 * (B / gl_call) * (K * (M + N) + M * N) / (avg_us * 1e-6) / (HBM_PEAK_GBS * 1e9),
                # the 8-flops-per-complex-MAC count over the time, and its ratio to the peak of the exact-f32 MFMA pipe: a
                # comparison with the kernel this one replaced (> 1 on the bf16 pipe), NOT a fraction of anything achievable
                "algorithmic_achieved": alg / (avg_us * 1e-6) / 1e12,
                "vs_f32_mfma_peak": alg / (avg_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFS,
                "traffic": tr, "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command, "
                                                 "(2 FETCH + WRITE) KiB per launch" if tr else None,
                "launches_per_step": g["launches"] / args.steps, "avg_launch_us": avg_us,
                "timing": "HIP events on the launch stream around every launch of the timed steps",
                "executed_flops_per_launch": exe, "algorithmic_flops_per_launch": alg,
                # a vmap call of B circuits is joined in sub-batches (CutCircuit._state_pipelined): batch per launch
                "algorithmic_bytes_per_launch": 8.0 * (B / gl_call) * (K * (M + N) + M * N),
                "gemm_shape": {"M": M, "N": N, "K": K, "batch": B / gl_call, "launches_per_vmap_call": gl_call},
                "half_circuit_passes": pass_entry,
            }
            if cc.spec.epilogue is not None:
                roof["deferred_gate"] = ("the last crossing gate and the one-qubit gates after it on its two qubits are not a bond: "
                                         "their 4 x 4 product per circuit is applied to the GEMM result in the accumulators "
                                         "(tcmi_cgemm_split_epi; bond %d instead of %d; VALU work, not counted in `achieved`)"
                                         % (K, cc.spec.plain.bond_dim))
            plan_info = {"contraction": "cut", "bond": K, "n_left": cc.spec.n_left,
                         "bond_with_every_crossing_gate_a_bond": getattr(getattr(cc.spec, "plain", None), "bond_dim", K),
                         "half_circuit_passes": len(cc.left.descs) + len(cc.right.descs),
                         "staging_s": round(staging_s, 4)}
        else:
            roof = dict(pass_entry)
            tr = traffic.get("pass2_kernel")
            roof["traffic"] = tr
            roof["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of this command, "
                                      "(2 FETCH + WRITE) KiB per launch") if tr else None
            roof["timing"] = "HIP events on the launch stream around the pass launches of every timed step"
            plan_info = {"contraction": "state-vector", "passes": len(cc.descs), "rounds": st["rounds"], "R": cc.cfg.R,
                         "LT": cc.cfg.LT, "lowbits": cc.cfg.lowbits, "staging_s": round(staging_s, 4)}
            # cross-plan anchor of SURVEY 8d: the canonical gate-by-gate state-vector plan's bytes over this time
            roof["B_sv_GBps"] = ((d - 1) * (n - 1) + 2) * 2.0 * (2**n) * 8 * B / (elapsed / args.steps) / 1e9
        out = {
            "metric": "amplitudes/sec",
            "value": value,
            "unit": "amplitudes/s",
            "n_gpus": world,
            **({"oversubscribed": "ranks share the visible device(s), gloo: launch-path test, not a measurement"}
               if os.environ.get("TCMI_BENCH_BACKEND") == "gloo" else {}),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            **({"per_rank_ms_per_step": per_rank_ms} if per_rank_ms is not None else {}),
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "c64 (f32 arithmetic)" if join_f32 is None else
                     "c64 (f32 arithmetic; join GEMM at f32 accuracy on the f16 / bf16 MFMA pipe with split operands, see "
                     "roofline.pipe and join_on_f32_mfma)",
            "data": "synthetic",
            "config": {
                "workload": f"HEA-B statevector contraction n={n} depth={d} complex64 (SURVEY 8d config 2): one step = "
                            f"{Bg} circuits (" + (f"{args.batch_per_gpu} per GPU" if weak else "fixed for every number of GPUs") +
                            f", sharded in contiguous blocks), {B} per vmap call, "
                            f"timed through backend.jit(backend.vmap(wavefunction))",
                "qubits": n, "depth": d, "global_batch": Bg, "batch_per_call": B, "calls_per_step_per_gpu": len(chunks),
                "parallelism": f"batch-shard x{world}", "z0_checksum": checksum,
                "contractor": args.contractor, "plan": plan_info, "state_norm": nrm, "host_side_traced": traced,
            },
            "roofline": roof,
            "latency_batch1": {"ms_per_state": lat1 * 1e3, "amplitudes_per_s": (2**n) / lat1},
        }
        if join_f32 is not None:
            if is_cut and cc.spec.epilogue is not None:
                join_f32["contraction"] = ("the plain cut (every crossing gate a bond: %d) joined by tcmi_cgemm -- the exact-f32 "
                                           "kernel has no epilogue" % cc.spec.plain.bond_dim)
            out["join_on_f32_mfma"] = join_f32
        if graph_info is not None:
            out["hipgraph_replay"] = graph_info
        if hea_a is not None:
            out["hea_a"] = hea_a
        if sv28 is not None:
            out["statevector_n%d" % args.sv_qubits] = sv28
        if vqe is not None:
            out["vqe_step"] = vqe
        if heis is not None:
            out["vqe_heisenberg"] = heis
        if rqc is not None:
            out["rqc_amplitude"] = rqc
        if svqa is not None:
            out["sliced_vqa"] = svqa
        if args.mps_qubits > 0 and world == 1:
            out["mps_tebd"] = _guard("mps_tebd", mps_leg, tc, torch, args)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_qubits, d, seed=n)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
