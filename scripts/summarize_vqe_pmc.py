"""Per-pass table of the VQE step from a scripts/gpu_vqe_profiles.sh capture (kernel stats CSV + PMC text): time, HBM traffic
against the algorithmic bytes of the pass's live tiles, VALU occupancy, LDS conflicts and the shader clock the pass ran at.
Host only (the algorithmic bytes come from the plans, recomputed here).

    python scripts/summarize_vqe_pmc.py profiles/r05c [n] [depth] [batch] > profiles/r05c_summary.txt
"""
import ast, csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd"))
import numpy as np, torch
import tcmi as tc
from tcmi import cons, executor as X, plan as P

stem = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 28
d = int(sys.argv[3]) if len(sys.argv) > 3 else 12
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
tc.set_dtype("complex64")
c = tc.templates.blocks.example_block(tc.Circuit(n), torch.zeros(2 * d * n), nlayers=d)
gates, nparams = c._gate_records(), len(c._params)
n_exec, cfg, plan, eg = X.choose_plan(n, gates, nparams, cons.dtypestr, cons._plan_options)
fm, ff = X.live_masks(plan.descs, n_exec)
plans = {}
def get_plan(full):
    if full not in plans:
        r_ = X.choose_adjoint_plan(eg, n_exec, cons.dtypestr, full, full) or X.choose_adjoint_plan(eg, n_exec, cons.dtypestr, True, False)
        plans[full] = {"plan": r_[1], "cfg": r_[0]}
    return plans[full]
adj0, am, af = X.pick_adjoint_from_zero(eg, n_exec, get_plan)
S1 = float(B) * (2 ** n) * 8 / 1e9          # one transfer of the batch, GB
zb = []
touched = 0
for dsc in plan.descs:
    w = np.asarray(dsc).view(np.uint32).astype(np.int64)
    tb = sum(1 << int(w[8 + i]) for i in range(int(w[2])))
    zb.append(2.0 ** -bin(tb & ~touched).count("1")); touched |= tb
alg = {}
for i, f in enumerate(ff):
    alg[f"fwd_p{i}"] = f * (1.0 + zb[i]) * S1
na = len(af)
for i, f in enumerate(af):
    alg[f"adj_p{i}"] = f * (4.0 if i < na - 1 else 2.0) * S1
rows = {r["Name"]: r for r in csv.DictReader(open(stem + "_vqe_kernel_stats.csv"))}
pm = {}
for line in open(stem + "_vqe_pmc.txt"):
    if line.startswith("tcmi_spec"):
        name, rest = line.split(" ", 1)
        pm[name] = ast.literal_eval(rest.strip())
print(f"# {stem}: VQE step n={n} d={d}, {B} samples per call -- one row per generated kernel = per pass of the executed plan")
print("# time = rocprofv3 kernel-trace average; traffic = (2 FETCH_SIZE + WRITE_SIZE) KiB from the PMC passes; algorithmic = bytes of")
print("# the pass's live tiles (a sweep pass with every term of the cotangent born in it reads psi only: 3 transfers);")
print("# VALU occupancy = 4 SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES (both quad-cycles; 4 waves per SIMD: 1.0 = a VALU instruction in")
print("# flight on every SIMD whenever its waves are resident); clock = SQ_BUSY_CYCLES / 32 shader engines / time")
print(f"{'kernel':28s} {'calls':>5s} {'ms':>8s} {'traffic GB':>10s} {'algorithmic':>11s} {'ratio':>6s} {'TB/s':>5s} {'VALU occ':>8s} {'LDS confl':>9s} {'GHz':>5s}")
def key(nm):
    return (nm[10:13], int(nm.split("_p")[1].split("_")[0]), nm)
for name in sorted(pm, key=key):
    dct = pm[name]
    g = lambda k: dct.get(k, (0, 0))[0]
    k = [r for nm, r in rows.items() if nm.startswith(name)]
    if not k:
        continue
    ms, calls = float(k[0]["AverageNs"]) / 1e6, int(k[0]["Calls"])
    tr = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / 1e9
    short = name.split("tcmi_spec_")[1].rsplit("_", 1)[0]
    a = alg.get(short)
    if a is not None and short == "adj_p0" and tr < 0.85 * a:
        a = a * 0.75                      # the folded first pass: lambda is not read
    print(f"{name:28s} {calls:5d} {ms:8.3f} {tr:10.2f} {a if a is not None else float('nan'):11.2f} {tr / a if a else float('nan'):6.2f} "
          f"{tr / ms if ms else 0:5.2f} {4 * g('SQ_ACTIVE_INST_VALU') / max(1.0, g('SQ_WAVE_CYCLES')):8.2f} "
          f"{g('SQ_LDS_BANK_CONFLICT') / max(1.0, g('SQ_ACTIVE_INST_LDS')):9.3f} {g('SQ_BUSY_CYCLES') / 32 / (ms * 1e-3) / 1e9 if ms else 0:5.2f}")
