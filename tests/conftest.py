import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tensorcircuit-ng_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


BENCH_SMALL = ["--steps", "2", "--warmup", "1", "--qubits", "16", "--depth", "4", "--batch", "4", "--global-batch", "4",
               "--vqe-qubits", "16", "--vqe-depth", "3", "--vqe-batch", "4", "--vqe-microbatch", "2", "--vqe-steps", "1",
               "--rqc-depth", "8", "--rqc-log2-target", "20", "--rqc-seeds", "2", "--svqa-qubits", "12", "--svqa-depth", "2",
               "--svqa-slices", "4", "--svqa-steps", "1", "--svqa-seeds", "1", "--mps-qubits", "0", "--sv-qubits", "16", "--sv-depth", "3", "--sv-batch", "4",
               "--sv-microbatch", "2", "--sv-steps", "1", "--no-cpu-baseline", "--no-traffic-probe"]
BENCH_OUT = {k: os.path.join(ROOT, ".pytest_cache", f"bench_{k}.json") for k in ("w1", "w2", "dead", "nccl1", "full", "raise")}


def _run_bench(tag, gpus, extra_env, timeout):
    """`python bench.py --gpus N` exactly as the driver starts it, at toy sizes; {rc, seconds, line} -> BENCH_OUT[tag]."""
    import json
    import subprocess
    import time

    env = dict(os.environ, **extra_env)
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + BENCH_SMALL,
                           env=env, capture_output=True, text=True, timeout=timeout)
        rc, out, err = r.returncode, r.stdout, r.stderr
    except subprocess.TimeoutExpired as e:
        rc, out, err = -999, (e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or ""), "timeout"
    line = None
    for ln in out.splitlines():
        if ln.startswith("{"):
            try:
                line = json.loads(ln)
            except ValueError:
                pass
    with open(BENCH_OUT[tag], "w") as fh:
        json.dump({"rc": rc, "seconds": time.time() - t0, "line": line, "stderr": err[-6000:]}, fh)


MULTIRANK_OUT = os.path.join(ROOT, ".pytest_cache", "multirank_slices.json")
MULTIRANK4_OUT = os.path.join(ROOT, ".pytest_cache", "multirank_slices_w4.json")


def pytest_sessionstart(session):
    """The two-rank and four-rank DistributedContractor runs of tests/test_gpu_multirank.py are started here, BEFORE this process
    makes any HIP call: its ranks are fresh child processes with their own GPU contexts (a process that has
    initialised the GPU must not be the one that starts them on this pool).  Counting devices does not initialise."""
    expr = session.config.getoption("-m", default="") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return
    try:
        import subprocess

        import torch

        if torch.cuda.device_count() < 1:
            return
        os.makedirs(os.path.dirname(MULTIRANK_OUT), exist_ok=True)
        for world, out in ((2, MULTIRANK_OUT), (4, MULTIRANK4_OUT)):
            if os.path.exists(out):
                os.remove(out)
            subprocess.run([sys.executable, os.path.join(ROOT, "tests", "multirank_slices.py"), str(world), out],
                           timeout=900, check=False)
        # the driver's multi-GPU command (tests/test_gpu_bench_multirank.py): one rank, two ranks sharing this box's
        # device over gloo (TCMI_BENCH_OVERSUBSCRIBE: the launch path, the sharding and the collectives, not a
        # measurement), and two ranks one of which dies
        for f in BENCH_OUT.values():
            if os.path.exists(f):
                os.remove(f)
        # (TCMI_TREE_CACHE=0: both worlds SEARCH their contraction trees -- the seeds dealt to the ranks are part of what is
        # compared; the RCCL run below goes through the tree cache)
        _run_bench("w1", 1, {"TCMI_TREE_CACHE": "0"}, 900)
        _run_bench("w2", 2, {"TCMI_BENCH_OVERSUBSCRIBE": "1", "TCMI_TREE_CACHE": "0"}, 900)
        _run_bench("dead", 2, {"TCMI_BENCH_OVERSUBSCRIBE": "1", "TCMI_BENCH_KILL_RANK": "1"}, 300)
        # a device too full for the legs (here: pretended, every rank is told it shares its device with 10^6 others): the
        # ranks agree to skip them; and an exception inside a leg on ONE rank: the job ends, non-zero, instead of hanging
        _run_bench("full", 2, {"TCMI_BENCH_OVERSUBSCRIBE": "1", "TCMI_BENCH_SHARERS": "1000000"}, 600)
        _run_bench("raise", 2, {"TCMI_BENCH_OVERSUBSCRIBE": "1", "TCMI_BENCH_RAISE_IN": "vqe_step:1"}, 300)
        # RCCL itself on this box's one GPU: the multi-rank code path of bench.py (init_process_group("nccl"), barriers,
        # the packed all-reduces, the sharded legs' collectives) with a world of ONE rank, torchrun-style environment
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        _run_bench("nccl1", 1, {"TCMI_BENCH_FORCE_DIST": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                                "LOCAL_WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                                "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, 900)
    except Exception as e:  # noqa: BLE001  (the test reports the missing file)
        print("multirank launcher failed:", e)


@pytest.fixture
def hipb():
    """Mirror of the reference's backend fixtures (tests/conftest.py:16-71)."""
    import tcmi as tc

    tc.set_backend("hip")
    tc.set_dtype("complex64")
    yield tc
    tc.set_dtype("complex64")


@pytest.fixture
def highp():
    import tcmi as tc

    tc.set_dtype("complex128")
    yield
    tc.set_dtype("complex64")
