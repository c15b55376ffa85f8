"""LDS instruction mix of the generated kernels of the n-qubit HEA-B plans: which DS forms did hipcc choose?
Compiles the sources to assembly (hipcc -S, device only); host only.   python scripts/isa_lds_mix.py [adjoint|forward] [n] [d] [EXP]"""
import collections, os, re, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tensorcircuit-ng_amd"))
import torch
import tcmi as tc
import tcmi.specialize as _SPX; _SPX.ALLOW_PROBE = True   # this script times kernels, also the wrong-result variants of TCMI_SPEC_EXP
from tcmi import cons, executor as X, specialize as S

kind = sys.argv[1] if len(sys.argv) > 1 else "adjoint"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 28
d = int(sys.argv[3]) if len(sys.argv) > 3 else 12
if len(sys.argv) > 4:
    os.environ["TCMI_SPEC_EXP"] = sys.argv[4]
tc.set_dtype("complex64")
c = tc.templates.blocks.example_block(tc.Circuit(n), torch.zeros(2 * d * n), nlayers=d)
gates, nparams = c._gate_records(), len(c._params)
n_exec, cfg, plan, eg = X.choose_plan(n, gates, nparams, cons.dtypestr, cons._plan_options)
if kind == "forward":
    descs, opts = plan.descs, None
else:
    acfg, ap = X.choose_adjoint_plan(eg, n_exec, cons.dtypestr, True, True)
    descs, opts = [np.asarray(x) for x in ap.descs], S.adjoint_opts(acfg)
keep = os.environ.get("KEEP_S")
for i, dsc in enumerate(descs):
    src, meta = S._source(kind, dsc, opts, i)
    with tempfile.TemporaryDirectory() as td:
        hip = os.path.join(td, "k.hip")
        open(hip, "w").write(src)
        flags = [f for f in S.HIPCC_FLAGS if f != "--genco"]
        out = os.path.join(td, "k.s")
        r = subprocess.run([S.HIPCC] + flags + ["--cuda-device-only", "-S", hip, "-o", out], capture_output=True, text=True)
        if r.returncode:
            print(r.stderr[-2000:]); sys.exit(1)
        asm = open(out).read()
        if keep:
            open(f"{keep}_{kind}_p{i}.s", "w").write(asm); open(f"{keep}_{kind}_p{i}.hip", "w").write(src)
    body = asm[asm.index(meta["kernel"] + ":"):]
    body = body[:body.index("s_endpgm")]
    ins = [l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    cn = collections.Counter(ins)
    ds = {k: v for k, v in cn.items() if k.startswith("ds_")}
    vg = re.search(r"\.vgpr_count:\s+(\d+)", asm)
    sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", asm)
    print(f"{meta['kernel']}: {len(ins)} instr, VALU {sum(v for k, v in cn.items() if k.startswith('v_'))}, "
          f"SALU {sum(v for k, v in cn.items() if k.startswith('s_'))}, s_waitcnt {cn['s_waitcnt']}, s_nop {cn['s_nop']}, "
          f"vgprs {vg.group(1) if vg else '?'} spill {sp.group(1) if sp else '?'}  {ds}")
