"""profiles/<tag>_summary.txt from the captures of scripts/gpu_final_profiles.sh: every fraction of the bench line recomputed
from the committed files.  usage: python scripts/summarize_final.py <tag>"""
import csv, json, sys

tag = sys.argv[1]
d = json.loads(open(f"profiles/{tag}_bench_default.json").read().strip().splitlines()[-1])
ks = {r["Name"]: r for r in csv.DictReader(open(f"profiles/{tag}_bench_default_kernel_stats.csv"))}
hs = {r["Name"]: r for r in csv.DictReader(open(f"profiles/{tag}_headline_kernel_stats.csv"))}


def find(tab, sub):
    for k, v in tab.items():
        if sub in k:
            return v


L = [f"Final captures {tag} (MI355X, one GPU), all from the binary of the last commit that touched csrc/.", "",
     f"{tag}_bench_default.json                   python3 bench.py   (bench line: HIP-event launch times, PMC traffic probe)",
     f"{tag}_bench_default_under_rocprof.json     the same command under rocprofv3 --kernel-trace --stats (program directly after --)",
     f"{tag}_bench_default_kernel_stats.csv       rocprofv3 kernel stats of that run (all legs: launches of different batch sizes share a row)",
     f"{tag}_headline_kernel_stats.csv            rocprofv3 kernel stats of the timed region of config 2 only (bench.py --probe-child --steps 20",
     "                                          --warmup 3): every join-GEMM launch (cgemm_split_kernel / cgemm_dma128_kernel) is a batch-8 join",
     f"{tag}_vqe_n28_d12_pmc.txt                  rocprofv3 --pmc passes (4 separate runs) over scripts/gpu_vqe_timing.py 28,12,1, per-dispatch averages", ""]
r = d["roofline"]
L.append(f"Config 2 headline ({d['config']['plan'].get('contraction')} order), dominant kernel {r.get('kernel')}:")
if r.get("bound") == "mfma":
    g = find(hs, "cgemm_split_kernel") or find(hs, "cgemm_dma128_kernel") or find(hs, "cgemm_dma_kernel") or find(hs, "cgemm_mfma_kernel<true>")
    pk = r["peak"]
    pipe = "bf16 MFMA peak, 36 executed flops per complex MAC" if pk > 1000 else "f32 MFMA peak, 6 executed flops per complex MAC"
    L.append(f"  HIP events (bench line):   {r['avg_launch_us']:.1f} us per launch -> executed flops {r['executed_flops_per_launch']:.4g} / t = {r['achieved']:.1f} TF = {r['frac']:.3f} of {pk:.1f} TF ({pipe}); "
             f"on 8 flops per MAC: {r['algorithmic_achieved']:.1f} TF = {r.get('vs_f32_mfma_peak', r.get('algorithmic_frac', 0.0)):.3f} of the 157.3 TF f32 MFMA peak")
    if g:
        tf = r["executed_flops_per_launch"] / float(g["AverageNs"]) * 1e9 / 1e12
        L.append(f"  rocprofv3 (headline csv):  {float(g['AverageNs'])/1e3:.1f} us average over {g['Calls']} calls (min {float(g['MinNs'])/1e3:.1f}) -> {tf:.1f} TF = {tf/pk:.3f}")
    j = d.get("join_on_f32_mfma")
    if j:
        L.append(f"  the same step with the join on the exact-f32 MFMA kernel: {j['amplitudes_per_s']:.4g} amplitudes/s, {j['ms_per_step']:.3f} ms per step; "
                 f"largest difference of the two states {j['max_abs_difference_of_the_two_states']:.2e} at amplitudes up to {j['largest_amplitude']:.2e}")
    if r.get("traffic"):
        L.append(f"  PMC traffic (bench line):  {r['traffic']/1e9:.3f} GB per launch ((2 FETCH + WRITE) KiB) vs {r['algorithmic_bytes_per_launch']/1e9:.3f} GB algorithmic")
L.append(f"  value {d['value']:.4g} amplitudes/s, {d['ms_per_step']:.3f} ms per step; batch 1: {d['latency_batch1']['ms_per_state']:.3f} ms per state")
v = d["vqe_step"]
vr = v["roofline"]
L += ["", f"Config 3 VQE step n=28 d=12 batch 32: {v['ms_per_step']:.1f} ms per step = {v['ms_per_step']/32:.1f} ms per sample"]
for k in ("forward_pass", "adjoint_pass", "measure_pass", "pauli_sum"):
    x = vr.get(k)
    if not x:
        L.append(f"  {k:13s} (no launch in the timed steps: the traced step takes the energy from the cotangent passes)")
        continue
    L.append(f"  {k:13s} {x['kernel']:66s} {x['launches_per_step']:.0f} launches/step, {x['avg_launch_us']/1e3:.2f} ms each, {x['algorithmic_bytes_per_launch']/1e9:.1f} GB algorithmic -> {x['achieved']:.0f} GB/s = {x['frac']:.3f} of 8 TB/s")
for k in ("forward_pass_valu", "adjoint_pass_valu"):
    x = vr.get(k)
    if x and not vr.get("dense_plan"):     # the count assumes every gate acts on the whole state: meaningless with live-tile passes
        L.append(f"  {k:18s} gate arithmetic alone {x['gate_arithmetic_flops_per_step']/1e12:.1f} Tflop per step -> {x['achieved']:.1f} TF = {x['frac']:.3f} of the 157.3 TF FP32 vector peak")
dp = vr.get("dense_plan")
if dp:
    L.append(f"  live-tile passes are on (bytes above = bytes of the live tiles).  The same kernels with every tile live (one step, TCMI_SPARSE_START=0 semantics): "
             f"{dp['ms_per_step']:.0f} ms per step; forward {dp['forward_pass']['avg_launch_us']/1e3:.2f} ms per launch = {dp['forward_pass']['frac']:.3f} of 8 TB/s, "
             f"sweep {dp['adjoint_pass']['avg_launch_us']/1e3:.2f} ms per launch = {dp['adjoint_pass']['frac']:.3f}; "
             f"energy difference {dp['max_abs_energy_difference']:.1e}, largest gradient difference {dp['max_abs_gradient_difference']:.1e}")
x = vr["step"]
L.append(f"  step: executed bytes {x['executed_bytes_per_step']/1e12:.2f} TB / wall -> {x['achieved']:.0f} GB/s = {x['frac']:.3f}; kernel time {x['kernel_ms_per_step']:.0f} ms of {v['ms_per_step']:.0f}; {x['forward_passes']:.0f} forward / {x['adjoint_passes']:.0f} adjoint passes")
for sub, name in (("tcmi_spec_forward", "tcmi_spec_forward"), ("tcmi_spec_adjoint", "tcmi_spec_adjoint"),
                  ("pass2_kernel<5, 8", "pass2_kernel<5,8>"), ("adjoint2_kernel<4, 8", "adjoint2_kernel<4,8>"),
                  ("measure2_kernel<5, 8", "measure2_kernel<5,8>"), ("pauli_tile_kernel<float", "pauli_tile_kernel"),
                  ("pauli_sum_kernel<float, 4096", "pauli_sum_kernel")):
    g = find(ks, sub)
    if g:
        L.append(f"  rocprofv3 {name:22s} {g['Calls']:>5s} calls, average {float(g['AverageNs'])/1e6:.3f} ms (all batch sizes of the run share the row)")
q = d["rqc_amplitude"]
qr = q["roofline"]
L += ["", f"Config 4 RQC amplitude: {q['contract_s']*1e3:.1f} ms, executed flops {qr['executed_flops_this_rank']:.4g} -> {qr['achieved']:.1f} TF = {qr['frac']:.3f} of 157.3; "
          f"algorithmic bytes {qr['algorithmic_bytes']/1e9:.1f} GB, stand-alone permute bytes {qr['wasted_traffic']/1e6:.1f} MB; launches {qr['launches']}",
      f"  amplitude {q['amplitude']}", f"  time split {q['time_split']}", f"  path search {q.get('path_search')}"]
sv = d.get("sliced_vqa")
ha = d.get("hea_a")
if ha and "error" not in ha:
    L += ["", f"Config 2 secondary workload HEA-A: {ha['amplitudes_per_s_per_gpu']:.4g} amplitudes/s per GPU, {ha['ms_per_call']:.3f} ms per call "
              f"({ha['contraction']} order), kernel ms per call {ha['kernel_ms_per_call']}"]
if sv and "error" not in sv:
    L += ["", f"Sliced value_and_grad (n={sv['workload'].split('n=')[1].split(' ')[0]}): {sv['ms_per_value_and_grad']:.1f} ms per call, graphs {sv.get('graphs')}",
          f"  roofline {sv.get('roofline')}", f"  one rank of 8, invariants sharded forward and backward: {sv.get('one_rank_of_8_sharded')}"]
m = d["mps_tebd"]
L += ["", f"Config 5 MPS sweep: {m['us_per_bond']:.0f} us per bond, kernel us per bond {m['roofline']['kernel_us_per_bond']}", "",
      f"cpu_baseline: {d['cpu_baseline']}"]
open(f"profiles/{tag}_summary.txt", "w").write("\n".join(L) + "\n")
print("\n".join(L))
