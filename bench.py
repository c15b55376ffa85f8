#!/usr/bin/env python3
"""bench.py -- headline benchmark of the tcmi hot path (BASELINE.json metric: amplitudes/sec).

A "step" = one full ``Circuit.wavefunction`` contraction of a batch of HEA-B circuits (reference
``templates/blocks.py:146-185`` ansatz, seeded random parameters, SURVEY.md section 8(d) config 2:
24 qubits, depth 8, complex64) through the compiled tile-VM plan, with parameters and states
resident in HBM.  One process per GPU; ranks run independent batches (vmap-batch sharding, no
data-path collective), timing is max-over-ranks between barriers.  Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 20 --warmup 3
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


def build_circuit(tc, n, d, params_row):
    c = tc.Circuit(n)
    for i in range(n):
        c.h(i)
    for j in range(d):
        for i in range(n - 1):
            c.exp1(i, i + 1, unitary=tc.gates._zz_matrix, theta=params_row[2 * j, i])
        for i in range(n):
            c.rx(i, theta=params_row[2 * j + 1, i])
    return c


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
        c = tc.Circuit(n)
        for i in range(n):
            c.h(i)
        for j in range(d):
            for i in range(n - 1):
                c.exp1(i, i + 1, unitary=tc.gates._zz_matrix, theta=p[2 * j, i])
            for i in range(n):
                c.rx(i, theta=p[2 * j + 1, i])
        e = 0.0
        for i in range(n):
            e += -1.0 * c.expectation((tc.gates.x(), [i]))
        for i in range(n - 1):
            e += 1.0 * c.expectation((tc.gates.z(), [i]), (tc.gates.z(), [i + 1]))
        return tc.backend.real(e)

    # the reference harness jits the step (benchmarks/scripts/vqe_tc.py:136-141); here jit traces the host side
    vvag = tc.backend.jit(tc.backend.vvag(energy, argnums=0, vectorized_argnums=0))
    mb = max(1, args.vqe_microbatch)

    def step():
        vals, grads = [], []
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
    t0 = time.perf_counter()
    for _ in range(args.vqe_steps):
        v, g, esum, gsum = step()
    sync()
    el = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    return {
        "workload": f"HEA-B n={n} depth={d} TFIM value_and_grad (55-term style energy), vmap batch {Bg} "
                    f"(SURVEY 8d config 3), complex64",
        "ms_per_step": el / args.vqe_steps * 1e3,
        "steps": args.vqe_steps,
        "samples_per_s": Bg * args.vqe_steps / el,
        "batch_per_gpu": hi - lo,
        "staging_s": round(staging, 3),
        "mean_energy": float(esum.item()) / Bg,
        "grad_norm": float(gsum.norm().item()),
        "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
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
    times = []
    for _ in range(max(1, args.mps_sweeps)):
        t0 = time.perf_counter()
        sweep()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        m.position(0)                    # back to the sweep's starting gauge: not part of the sweep, so it must
        torch.cuda.synchronize()         # not run into the next timed sweep either
    t = sum(times) / len(times)
    if not all(bool(torch.isfinite(x.abs()).all()) for x in m.get_tensors()):
        raise FloatingPointError("non-finite MPS tensor after the TEBD sweeps")
    return {
        "workload": f"MPSCircuit n={n} chi={chi} TEBD sweep of {n - 1} adjacent random SU(4) gates, complex64 "
                    f"(SURVEY 8d config 5)",
        "sweeps_per_s": 1.0 / t, "us_per_bond": t / (n - 1) * 1e6, "sweeps": len(times),
        "max_bond": int(max(m.get_bond_dimensions())), "staging_s": round(staging, 3),
        "fidelity_estimate": float(m._fidelity),
    }


def rqc_leg(tc, torch, dist, args, rank, world):
    """BASELINE config 4: single amplitude <0^32|C|0^32> of a 32-qubit random circuit on a 4x8 grid (brickwork
    of Haar-random two-qubit gates, reference gates.py:852-863), complex64, through DistributedContractor:
    random-greedy path search + slicing to 2^27 elements, slices sharded over the ranks as
    reference experimental.py:881-890 and summed with one packed all-reduce."""
    import numpy as np
    from tcmi.experimental import DistributedContractor

    rows, cols, depth = 4, 8, args.rqc_depth
    gates = [tc.gates.random_two_qubit_gate(7000 + i).tensor for i in range(depth * rows * cols)]
    q = lambda r, c: r * cols + c

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

    t0 = time.perf_counter()
    dc = DistributedContractor(nodes_fn, None, cotengra_options={
        "slicing_opts": {"target_size": 2 ** args.rqc_log2_target}, "max_repeats": 128})
    search_s = time.perf_counter() - t0
    v = dc.value(None, op=lambda x: x)          # staging run
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    v = dc.value(None, op=lambda x: x)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t = time.perf_counter() - t0
    tree = dc.tree
    flops = float(tree.total_flops())          # all slices (ContractionTree.total_flops includes nslices)
    steps, dep, _, _ = tree._symbolic_steps()
    n_inv = sum(1 for st in steps if not dep[st[4]])   # slice-invariant steps: computed once per rank
    return {
        "workload": f"32-qubit 4x8 random circuit depth {depth}, amplitude <0|C|0>, complex64, sliced to "
                    f"2^{args.rqc_log2_target} elements (SURVEY 8d config 4)",
        "nslices": int(tree.nslices), "slices_per_gpu": int(-(-tree.nslices // world)),
        "contraction_width": float(tree.contraction_width()), "log2_flops_total": float(np.log2(flops)),
        "contract_s": t, "tflops": flops / t / 1e12, "path_search_s": round(search_s, 2),
        "steps_per_slice": len(steps), "slice_invariant_steps": n_inv,
        "amplitude": [float(v.real), float(v.imag)],
    }


def _guard(name, fn, *a):
    """Secondary legs must never take the headline line down with them."""
    try:
        return fn(*a)
    except Exception as e:  # noqa: BLE001
        return {"error": f"{name}: {type(e).__name__}: {e}"[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--qubits", type=int, default=24)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--batch", type=int, default=8, help="circuits per GPU per step (vmap batch)")
    ap.add_argument("--cpu-qubits", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vqe-qubits", type=int, default=28, help="VQE leg (config 3): qubits; 0 disables the leg")
    ap.add_argument("--vqe-depth", type=int, default=12)
    ap.add_argument("--vqe-batch", type=int, default=32, help="VQE leg: global vmap batch (sharded over ranks)")
    ap.add_argument("--vqe-steps", type=int, default=2)
    ap.add_argument("--vqe-microbatch", type=int, default=8, help="samples per vvag call (bounds HBM use)")
    ap.add_argument("--mps-qubits", type=int, default=64, help="MPS TEBD leg (config 5): qubits; 0 disables the leg")
    ap.add_argument("--mps-chi", type=int, default=128)
    ap.add_argument("--mps-sweeps", type=int, default=2)
    ap.add_argument("--rqc-depth", type=int, default=16, help="config 4 leg (32-qubit RQC amplitude): depth; 0 disables")
    ap.add_argument("--rqc-log2-target", type=int, default=27)
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph replay measurement")
    ap.add_argument("--contractor", default="greedy",
                    help="greedy/auto: cost model picks the contraction order; plain: state-vector plan; cut: cut contraction")
    ap.add_argument("--lowbits", type=int, default=None)
    ap.add_argument("--R", type=int, default=None)
    ap.add_argument("--LT", type=int, default=None)
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # TCMI_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL init, barriers, all-reduces) with a
    # world of one rank too (used to validate that path on a one-GPU box under torchrun)
    if world > 1 or os.environ.get("TCMI_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype("complex64")
    opts = {k: getattr(args, k) for k in ("lowbits", "R", "LT") if getattr(args, k) is not None}
    tc.set_contractor(args.contractor, **opts)
    n, d, B = args.qubits, args.depth, args.batch

    # synthetic parameters: SURVEY 8(d) config-2 generator, one independent row per batch element
    rng = np.random.default_rng(n + 1000 * rank)
    params_np = rng.uniform(0, 2 * np.pi, [B, 2 * d, n]).astype(np.float32)
    params = torch.from_numpy(params_np).to(dev)

    # staging: record the structure once, compile the plan (cached by structure)
    t0 = time.perf_counter()
    c = build_circuit(tc, n, d, params[0])
    cc = c._compiled()
    staging_s = time.perf_counter() - t0
    # parameter vector layout = recording order; build the [B, P] matrix with the same gather
    idx = torch.stack(c._params)  # values of batch row 0 in recording order
    flat0 = params[0].reshape(-1)
    # map each recorded parameter to its position in the flat [2d*n] row (exact float match is
    # ambiguous, so recompute the gather order structurally)
    order = []
    for j in range(d):
        order += [(2 * j) * n + i for i in range(n - 1)]
        order += [(2 * j + 1) * n + i for i in range(n)]
    gather = torch.tensor(order, device=dev)
    assert torch.equal(flat0[gather], idx)
    pmat = params.reshape(B, -1)[:, gather].contiguous()

    from tcmi.executor import CutCircuit

    is_cut = isinstance(cc, CutCircuit)
    state = torch.empty(B, 2**cc.n_exec, dtype=torch.complex64, device=dev)
    st = cc.stats()

    def step(ev=None):
        # the product call: CompiledCircuit.state / CutCircuit.state on resident parameters, with HIP
        # events around the dominant kernel's launches (pass kernels, or the join GEMM of a cut plan)
        if is_cut:
            cc.gemm_events = ev
        else:
            cc.pass_events = ev
        cc.state(pmat, out=state)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    sync()
    elapsed = time.perf_counter() - t0
    cc.gemm_events = cc.pass_events = None
    # the same step replayed from a hipGraph (launch overhead removed); reported next to the eager figure
    graph_info = None
    if not args.no_graph:
        try:
            from tcmi.executor import GraphedState

            gs = GraphedState(cc, B)
            for _ in range(args.warmup):
                gs(pmat)
            sync()
            tg0 = time.perf_counter()
            for _ in range(args.steps):
                gs(pmat)
            sync()
            tg = time.perf_counter() - tg0
            same = bool(torch.allclose(gs.out, state, atol=1e-6, rtol=0))
            graph_info = {"ms_per_step": tg / args.steps * 1e3, "amplitudes_per_s_per_gpu": B * (2**n) * args.steps / tg,
                          "matches_eager": same}
            del gs
        except Exception as e:  # noqa: BLE001
            graph_info = {"error": f"{type(e).__name__}: {e}"[:200]}
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = sum(a.elapsed_time(b_) for a, b_ in events) / args.steps  # dominant-kernel time per step

    # sanity: the state is normalised (cheap property check at full size)
    nrm = float((state[0].abs() ** 2).sum().item())
    del state
    torch.cuda.empty_cache()

    vqe = None
    if args.vqe_qubits:
        vqe = _guard("vqe_step", vqe_leg, tc, torch, dist, args, rank, world, dev)
    rqc = None
    if args.rqc_depth:
        torch.cuda.empty_cache()
        rqc = _guard("rqc_amplitude", rqc_leg, tc, torch, dist, args, rank, world)

    if rank == 0:
        amps = float(world) * B * (2**n) * args.steps
        value = amps / elapsed
        if is_cut:
            M, N, K = 2**cc.spec.n_left, 2 ** (n - cc.spec.n_left), cc.K
            flops_per_launch = 8.0 * M * N * K * B          # complex MAC = 8 real flops
            achieved = flops_per_launch / (kern_ms * 1e-3) / 1e12
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01l_traffic.json")
            if os.path.exists(tpath):
                traffic = json.load(open(tpath))["hbm_bytes_per_output_element"] * B * M * N
            roof = {
                "bound": "mfma", "kernel": "tcmi::cgemm_mfma_kernel<true> (cut-contraction join GEMM)",
                "achieved": achieved, "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_F32_PEAK_TFS, "traffic": traffic,
                "traffic_source": "profiles/r01l_traffic.json (rocprofv3 PMC, scaled by output elements)" if traffic else None,
                "launches_per_step": 1, "avg_launch_us": kern_ms * 1e3,
                "algorithmic_flops_per_launch": flops_per_launch,
                # the kernel uses Gauss's 3-product form: 6 real flops per complex MAC are issued to the MFMA
                # pipe, so the algorithmic rate (8 per MAC) can touch or pass the f32 peak
                "executed_flops_per_launch": 0.75 * flops_per_launch,
                "executed_frac": 0.75 * achieved / MFMA_F32_PEAK_TFS,
                "algorithmic_bytes_per_launch": 8.0 * B * (K * (M + N) + M * N),
                "gemm_shape": {"M": M, "N": N, "K": K, "batch": B},
            }
            plan_info = {"contraction": "cut", "bond": K, "n_left": cc.spec.n_left,
                         "half_circuit_passes": len(cc.left.descs) + len(cc.right.descs),
                         "staging_s": round(staging_s, 4)}
        else:
            npass = len(cc.descs)
            bytes_per_launch = 2.0 * B * (2**cc.n_exec) * 8  # read + write the batched state once
            avg_launch_s = kern_ms * 1e-3 / npass
            achieved = bytes_per_launch / avg_launch_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01b_traffic.json")
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                traffic = tj["hbm_bytes_per_amplitude_per_launch"] * B * (2**cc.n_exec)
            roof = {
                "bound": "hbm", "kernel": "tcmi::pass_kernel<float,%d,%d,0>" % (cc.cfg.R, cc.cfg.LT),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": "profiles/r01b_traffic.json (rocprofv3 PMC, scaled by batch)" if traffic else None,
                "launches_per_step": npass, "avg_launch_us": avg_launch_s * 1e6,
                "algorithmic_bytes_per_launch": bytes_per_launch,
            }
            plan_info = {"contraction": "state-vector", "passes": npass, "rounds": st["rounds"], "R": cc.cfg.R,
                         "LT": cc.cfg.LT, "lowbits": cc.cfg.lowbits, "staging_s": round(staging_s, 4)}
        out = {
            "metric": "amplitudes/sec",
            "value": value,
            "unit": "amplitudes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "c64 (f32 arithmetic)",
            "data": "synthetic",
            "config": {
                "workload": f"HEA-B statevector contraction n={n} depth={d} complex64 (SURVEY 8d config 2), "
                            f"vmap batch {B} per GPU",
                "qubits": n, "depth": d, "batch_per_gpu": B, "parallelism": f"batch-shard x{world}",
                "contractor": args.contractor, "plan": plan_info, "state_norm": nrm,
            },
            "roofline": roof,
        }
        if graph_info is not None:
            out["hipgraph_replay"] = graph_info
        if vqe is not None:
            out["vqe_step"] = vqe
        if rqc is not None:
            out["rqc_amplitude"] = rqc
        if args.mps_qubits > 0 and world == 1:
            out["mps_tebd"] = _guard("mps_tebd", mps_leg, tc, torch, args)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_qubits, d, seed=n)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
