"""Where the time of the f16 join kernel goes (probe libraries, scripts/round6/build_f16_probes.sh): TCMI_SPLIT_MODE 0 whole,
1 no conversion, 2 no MFMA, 3 no result stores, 5 conversion without plane writes, 4 per-tile stamps of workgroup 100;
variants with 3 / 5 / 10 vector instructions behind each MFMA.  M = N = 4096, K = 128, batch 32, no epilogue (the probe
instantiations are plain-product kernels) -- one child process per (library, mode): the mode is read once."""
import sys, os, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "..", "tensorcircuit-ng_amd", "csrc")
if len(sys.argv) > 1:
    lib, mode, K, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    os.environ["TCMI_SPLIT_MODE"] = str(mode)
    sys.path.insert(0, os.path.join(HERE, "..", "..", "tensorcircuit-ng_amd"))
    import torch
    from tcmi import _lib
    _lib.LIB_PATH = os.path.join(CSRC, lib)
    L = _lib.lib()
    M = N = 4096
    st = torch.cuda.current_stream().cuda_stream
    A = torch.view_as_complex(torch.randn(B, K, M, 2, device="cuda") * 0.01)
    Bm = torch.view_as_complex(torch.randn(B, K, N, 2, device="cuda") * 0.01)
    c = torch.empty(B, M, N, dtype=torch.complex64, device="cuda")
    X = torch.eye(4, dtype=torch.complex64, device="cuda").reshape(1, 16).repeat(B, 1).contiguous()
    epi = mode in (7, 8, 9) or os.environ.get("F16_PROBE_EPI") == "1"
    if mode >= 100:      # 1xx: mode xx with the epilogue
        mode, epi = mode - 100, True
        os.environ["TCMI_SPLIT_MODE"] = str(mode)
    SC = float(os.environ.get("F16_PROBE_SCALE", str(2.0**14)))
    entry = L.tcmi_cgemm_split_f16
    fn = lambda: _lib.check(entry(A.data_ptr(), Bm.data_ptr(), c.data_ptr(), M, N, K, B, K * M, K * N, M * N,
                                                   X.data_ptr() if epi else None, SC, SC, st), "x")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    ntile = 1024 * B / 256
    line = f"{lib:28s} mode {mode} epi {int(epi)} K={K} B={B}: {ms:.3f} ms = {ms * 1e3 / ntile:.2f} us per tile"
    if mode == 4:
        nt = 4 * B
        v = torch.view_as_real(c[0, 0, : nt + 1]).cpu().numpy()
        real = v[1:, 1] / 100.0
        d = [real[0]] + [real[i] - real[i - 1] for i in range(1, nt)]
        line += f"  life {v[0, 1] / 100:.1f} us at {v[0, 0] / v[0, 1] * 100:.0f} MHz; tiles (us): " + " ".join(f"{x:.1f}" for x in d[:16])
    if mode == 10:
        v = torch.view_as_real(c[0, 0, :2]).cpu().numpy().reshape(-1)
        nt = 1024 * B // 256
        line += "  cycles per tile: step0 %.0f step1 %.0f rest %.0f (%.0f per step) epilogue %.0f" % (
            v[0] / nt, v[1] / nt, v[2] / nt, v[2] / nt / (K // 16 - 2), v[3] / nt)
    print(line, flush=True)
    sys.exit(0)
libs = [l for l in os.environ.get("F16_PROBE_LIBS", "libtcmi_probe.so").split(",") if l]
for lib in libs:
    for mode in [int(m) for m in os.environ.get("F16_PROBE_MODES", "0,1,2,3,5,4").split(",")]:
        for K, B in ((128, 32), (512, 8)):      # tile time t(K) = steps(K) * step + transition: two K give both
            subprocess.run([sys.executable, os.path.abspath(__file__), lib, str(mode), str(K), str(B)], timeout=300)
    os.environ["F16_PROBE_EPI"] = "1"           # mode 0 with the 4 x 4 epilogue
    for K, B in ((128, 32), (512, 8)):
        subprocess.run([sys.executable, os.path.abspath(__file__), lib, "0", str(K), str(B)], timeout=300)
    del os.environ["F16_PROBE_EPI"]
