"""Plan-specialised pass kernels: a pass descriptor lowered to straight-line gfx950 code.

The tile-VM kernels (csrc/tcmi_vm2.hip, csrc/tcmi_adjoint2.hip) INTERPRET a pass descriptor: per op they
fetch descriptor words, build one-hot dispatch flags, branch over the bodies they do not need, walk LDS
addresses from mask words held in SGPRs.  The PMC counters of the n = 28 passes (profiles/r03d_*_pmc.txt)
show what that costs: 41 % of the issued VALU instructions and 0.84e9 scalar instructions per dispatch of
the reverse sweep are neither gate arithmetic nor data movement.  This module removes the interpreter for
plans that are worth it: the SAME descriptor (written by tcmi/plan.py, same tables, same arithmetic
bodies from csrc/tcmi_vm2_asm.inc) is translated once into HIP source in which

  * every op is emitted in program order with its register indices, table offsets and masks as literals
    (no dispatch, no descriptor loads, no taken branches around unused bodies);
  * the LDS exchange addresses are `thread part + literal offset` wherever the register bits own their slot
    bits (the literal goes into the DS instruction's offset field: no address arithmetic at all);
  * all scalar table loads of the pass sit in one basic block, so the compiler hoists them far ahead of use.

The source is compiled with hipcc to a code object (one kernel per pass), cached on disk by a digest of the
descriptor, and launched through `tcmi_spec_launch_*` (hipModuleLaunchKernel; include/tcmi.h).  This is the
role `jax.jit` plays for the reference (tensorcircuit/backends/jax_backend.py `jit`; the reference harness
jits the VQE step, benchmarks/scripts/vqe_tc.py:136-141): compile once per circuit structure, reuse for every
parameter value.  Passes with ops the emitter does not know keep the interpreter; so does everything when
hipcc is unavailable (TCMI_SPECIALIZE=0 switches the whole mechanism off).
"""

import hashlib
import os
import shutil
import subprocess
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import plan as P

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
CACHE_DIR = os.environ.get("TCMI_SPEC_CACHE") or os.path.join(CSRC, "plancache")
HIPCC = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-munsafe-fp-atomics",
               "-mllvm", "-simplifycfg-sink-common=false", "-mllvm", "-disable-promote-alloca-to-vector",
               "--genco", "-I", CSRC]


NT_MIN_N = 26


class Unsupported(Exception):
    """The pass contains an op (or a shape) the emitter has no straight-line form for: it stays interpreted."""


def _ins0(k: int, j: int) -> int:
    return ((k >> j) << (j + 1)) | (k & ((1 << j) - 1))


def _u32(x) -> int:
    return int(x) & 0xFFFFFFFF


def _i32(x) -> int:
    x = int(x) & 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def _runs(pairs: Sequence[Tuple[int, int]]):
    """[(source bit, destination bit)] sorted by source -> maximal runs (src0, dst0, length) of consecutive bits."""
    out = []
    for s, d in pairs:
        if out and out[-1][0] + out[-1][2] == s and out[-1][1] + out[-1][2] == d:
            out[-1][2] += 1
        else:
            out.append([s, d, 1])
    return [tuple(x) for x in out]


def _deposit_expr(var: str, pairs: Sequence[Tuple[int, int]], ctype: str = "uint32_t") -> str:
    """C expression moving bit s of ``var`` to bit d for every (s, d)."""
    terms = []
    for s, d, ln in _runs(sorted(pairs)):
        m = ((1 << ln) - 1) << s
        t = f"(({ctype}){var} & {m:#x}u)"
        if d > s:
            t = f"({t} << {d - s})"
        elif d < s:
            t = f"({t} >> {s - d})"
        terms.append(t)
    return " | ".join(terms) if terms else "0u"


def _xor_const(masks: Sequence[int], r: int) -> int:
    out = 0
    for j, m in enumerate(masks):
        if (r >> j) & 1:
            out ^= m
    return out


class _Round:
    def __init__(self, w, pc, R, LT):
        self.pc = pc
        self.nops = int(w[pc])
        self.nwords = int(w[pc + 1])
        self.reg_phys = [_u32(w[pc + 2 + j]) for j in range(R)]
        self.thr_phys = [_u32(w[pc + 8 + i]) for i in range(LT)]
        self.reg_rd = [_u32(w[pc + 18 + j]) for j in range(R)]
        self.thr_rd = [_u32(w[pc + 24 + i]) for i in range(LT)]
        self.reg_wr = [_u32(w[pc + 34 + j]) for j in range(R)]
        self.thr_wr = [_u32(w[pc + 40 + i]) for i in range(LT)]
        self.ops_at = pc + P.RR_WORDS


class Seg:
    """One schedulable unit of a pass: wave-uniform operand loads + body pieces.  The linearisation (``_Emitter.linear``)
    issues the loads of segment i + 1 right after the FIRST piece of segment i, i.e. after the wait that piece needs for
    its own operands and a whole body ahead of their use; scheduling barriers pin that order.  (All scalar loads of a
    wave share one out-of-order counter, so a wait for any of them waits for all that were issued before it.)"""

    def __init__(self, tag: str = ""):
        self.tag = tag
        self.loads: List[str] = []
        self.parts: List[List[str]] = [[]]

    def new_part(self) -> List[str]:
        if self.parts[-1]:
            self.parts.append([])
        return self.parts[-1]


class _Emitter:
    """Common parts of the forward and the reverse-sweep emitters: header decode, index arithmetic, tile load /
    store, LDS exchange, linearisation."""

    SB = "  __builtin_amdgcn_sched_barrier(0);"

    def __init__(self, words, vectors: Sequence[str], opts: Optional[dict] = None):
        w = np.asarray(words).view(np.int32).astype(np.int64)
        if int(w[0]) != P.MAGIC:
            raise Unsupported("not a pass descriptor")
        self.w = w
        self.opts = dict(opts or {})
        self.n, self.T, self.R, self.LT = int(w[1]), int(w[2]), int(w[3]), int(w[4])
        # a young workgroup issues its tile loads at raised priority (s_setprio 3 until they are out): its index arithmetic
        # no longer queues behind the older workgroups' gate bodies (forward dense passes -3 %, sweep -2 %; raising the
        # exchanges as well (prio=2) gained nothing)
        self.opts.setdefault("prio", 1)
        self.nrounds, self.flags = int(w[5]), int(w[6])
        self.NR = 1 << self.R
        if self.n > 32:
            raise Unsupported("n > 32")
        self.tile_bits = [int(w[8 + i]) for i in range(self.T)]
        # contiguous run of a tile in memory: 8 bytes x 2^(tile bits 0, 1, 2 ... without a gap).  Runs shorter than an L2
        # line (128 bytes) get the XCD-aware tile order (index_lines); measured on the n = 28 VQE step: the sweep (64-byte
        # runs) 111 -> 98 ms per 8 samples, the forward passes (128-byte runs) 2.5 % SLOWER with it, so not there
        run = 0
        while run < self.T and self.tile_bits[run] == run:
            run += 1
        self.opts.setdefault("xcd", int((8 << run) < 128))
        # ... and no nontemporal hint: the other half of every 128-byte L2 line belongs to the neighbouring tile, which the
        # XCD-aware order runs on the same L2 a moment later -- a line marked streaming is gone by then (PMC, sweep pass 1 of
        # n = 28: 88 GB moved for 69 GB algorithmic with the hint; passes on 64-byte runs 5-27 % faster without it).  Tiles of
        # whole lines keep it for states that no cache holds until the next pass (2^26 amplitudes = 512 MiB and up: forward
        # 8.96 -> 8.85 ms per state at n = 28; specialize.EXTRA_OPTS = {'ntl': 0, 'nts': 0} to compare); smaller states are re-read from L2 /
        # MALL by the next pass or the join.
        nt = int(self.n >= NT_MIN_N and (8 << run) >= 128)
        self.opts.setdefault("ntl", nt)
        self.opts.setdefault("nts", nt)
        self.vectors = list(vectors)
        self.rounds: List[_Round] = []
        pc = P.HDR_WORDS
        for _ in range(self.nrounds):
            rd = _Round(w, pc, self.R, self.LT)
            self.rounds.append(rd)
            pc = rd.ops_at + rd.nwords
        self.segs: List[Seg] = []
        self.uid = 0
        # logical register index -> variable index (CNOT / SWAP between register bits are renamings here)
        self.reg = list(range(self.NR))

    # ---- small helpers -------------------------------------------------------------------------------
    def seg(self, tag: str = "") -> Seg:
        s = Seg(tag)
        self.segs.append(s)
        return s

    def fresh(self, stem: str) -> str:
        self.uid += 1
        return f"{stem}{self.uid}"

    def A(self, r: int, vec: str = "a") -> str:
        return f"{vec}{self.reg[r]}"

    def toff_type(self) -> str:
        return "uint32_t" if self.n <= 29 else "unsigned long long"

    def linear(self) -> List[str]:
        out: List[str] = []
        segs = [s for s in self.segs if s.loads or any(s.parts)]
        pf = 1     # prefetch distance in segments (operand loads may name values defined by the segment right before)
        sb = [] if self.opts.get("nosb") else [self.SB]
        for s in segs[:pf]:
            out += s.loads
        for i, s in enumerate(segs):
            parts = [p for p in s.parts if p] or [[]]
            out.append(f"  // -- {s.tag}")
            out += parts[0]
            out += sb
            if i + pf < len(segs) and segs[i + pf].loads:
                out += segs[i + pf].loads
                out += sb
            for p in parts[1:]:
                out += p
                out += sb
        return out

    # ---- prologue ------------------------------------------------------------------------------------
    def index_lines(self) -> Tuple[List[str], List[str]]:
        """(lines before the tile loop, lines that open it)."""
        free = [p for p in range(self.n) if p not in self.tile_bits]
        pairs = [(i, p) for i, p in enumerate(free)]
        pre = ["  const uint32_t tid = threadIdx.x;"]
        # live_mask (kernel argument, over the bits of the tile index): 0xffffffff = every tile has its workgroup; else
        # the grid holds one workgroup per tile whose index is zero outside the mask -- the tiles that can be non-zero
        # while some qubits of a circuit started from |0...0> have not been touched yet (executor.live_masks)
        loop = ["  { uint32_t bx = blockIdx.x;"]
        if self.opts.get("xcd", 1):
            # XCD-aware tile order.  Workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md, workgroup dispatch), every XCD
            # has its own L2 (128-byte lines), and tile index t and t + 1 differ in the LOWEST physical bit outside the
            # tile: with tiles of 64-byte runs (pinned low bits 0..2) they are the two halves of the same L2 lines.
            # Dealt round-robin, the two halves go to two XCDs and each L2 fetches (and writes back) whole lines for
            # half their bytes; here XCD x walks the contiguous range [x N/8, (x + 1) N/8) of tile indices instead, so
            # neighbours in the tile order are neighbours in time on ONE L2.
            loop += ["  if (gridDim.x >= 16u) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);"]
        loop += ["  if (live_mask != 0xffffffffu) {",
                 "    uint32_t b_ = 0, s_ = bx;",
                 "    for (uint32_t m_ = live_mask; m_; m_ &= m_ - 1u) { if (s_ & 1u) b_ |= m_ & (0u - m_); s_ >>= 1; }",
                 "    bx = b_;",
                 "  }"]
        loop.append(f"  const uint32_t wg_base = {_deposit_expr('bx', pairs)};")
        return pre, loop

    def local_tid(self, out: List[str]) -> str:
        """A copy of the thread index the optimiser cannot see through: index arithmetic derived from it is computed
        where it is written instead of being merged with the other exchanges' and kept alive from the top of the kernel
        (20 spilled VGPRs in a 9-round pass)."""
        t = self.fresh("tid")
        out.append(f'  uint32_t {t} = tid; asm volatile("" : "+v"({t}));')
        return t

    def thread_xor(self, out: List[str], name: str, masks: Sequence[int], shift: int = 0, tid: str = "tid"):
        """uint32_t name = XOR of masks[i] << shift over the set bits of tid."""
        pairs = [(i, m.bit_length() - 1 + shift) for i, m in enumerate(masks) if m and (m & (m - 1)) == 0]
        multi = [(i, m) for i, m in enumerate(masks) if m and (m & (m - 1)) != 0]
        if not multi:
            out.append(f"  const uint32_t {name} = {_deposit_expr(tid, pairs)};")
            return
        out.append(f"  uint32_t {name} = {_deposit_expr(tid, pairs)};")
        for i, m in multi:
            out.append(f"  {name} ^= (0u - (({tid} >> {i}) & 1u)) & {(m << shift):#x}u;")

    def tile_io(self, out: List[str], rd: _Round, store: bool, ptrs: Dict[str, str], tphys: str, vecs=None):
        """16-byte accesses, two amplitudes each (register bit 0 = the lowest physical bit of the tile)."""
        tt = self.toff_type()
        if rd.reg_phys[0] != 1 or self.tile_bits[0] != 0:
            raise Unsupported("register bit 0 is not physical bit 0 in the load / store layout")
        toff = self.fresh("toff")
        if store:
            tp2 = self.fresh("tps")
            out.append(f'  uint32_t {tp2} = {tphys}; asm volatile("" : "+v"({tp2}));')
            tphys = tp2
        out.append(f"  const {tt} {toff} = ({tt}){tphys} * 8u;")
        for vec in (self.vectors if vecs is None else vecs):
            base = self.fresh("gb")
            cq = "" if store else "const "
            out.append(f"  {cq}char* __restrict__ {base} = reinterpret_cast<{cq}char*>({ptrs[vec]} + wg_base);")
            for r in range(0, self.NR, 2):
                c = _xor_const(rd.reg_phys, r) * 8
                addr = f"{base} + {c:#x}ull + {toff}"
                if store:
                    v = self.fresh("sv")
                    st = (f"__builtin_nontemporal_store({v}, reinterpret_cast<v4f*>({addr}))" if self.opts.get("nts")
                          else f"*reinterpret_cast<v4f*>({addr}) = {v}")
                    out.append(f"  {{ v4f {v}; {v}.xy = {self.A(r, vec)}; {v}.zw = {self.A(r + 1, vec)}; {st}; }}")
                else:
                    v = self.fresh("lv")
                    ld = (f"__builtin_nontemporal_load(reinterpret_cast<const v4f*>({addr}))" if self.opts.get("ntl")
                          else f"*reinterpret_cast<const v4f*>({addr})")
                    if self.opts.get("umask_arg"):
                        # umask (kernel argument, physical bits inside the tile): amplitudes whose index has one of these
                        # bits set are known to be zero (no pass has touched the qubit yet, executor.zero_start) and may
                        # never have been written: they are not read
                        out.append(f"  v4f {v} = {{0.f, 0.f, 0.f, 0.f}};")
                        out.append(f"  if (((({c >> 3:#x}u | {tphys}) & umask) == 0u)) {{ {v} = {ld}; if (umask & 1u) {{ {v}.z = 0.f; {v}.w = 0.f; }} }}")
                    else:
                        out.append(f"  const v4f {v} = {ld};")
                    out.append(f"  {self.A(r, vec)} = {v}.xy; {self.A(r + 1, vec)} = {v}.zw;")

    # ---- LDS exchange --------------------------------------------------------------------------------
    def exchange_plan(self, reg_masks: Sequence[int], thr_masks: Sequence[int]):
        """Split the slot of register index r into (dirty, clean): clean bits are touched by no thread mask, so
        `thread part ^ dirty` + clean is the address and `clean` rides in the instruction's offset field."""
        U = 0
        for m in thr_masks:
            U |= m
        out = []
        for r in range(self.NR):
            c = _xor_const(reg_masks, r)
            out.append((c & U, c & ~U))
        return out

    def exchange(self, k: int, planes: Sequence[Tuple[str, str]], elem_bytes: int) -> Seg:
        """Exchange from the layout of round k into round k + 1.  ``planes``: (vector, component suffix) moved one
        after the other through the one LDS buffer ('' = the whole 8-byte value).  Returns the segment; the name of
        the new thread-part of the physical index is left in ``self.tphys``."""
        wr, rdn = self.rounds[k], self.rounds[k + 1]
        sh = {4: 2, 8: 3}[elem_bytes]
        reg_wr, thr_wr, reg_rd, thr_rd = wr.reg_wr, wr.thr_wr, rdn.reg_rd, rdn.thr_rd
        if elem_bytes == 8 and self.opts.get("own_slots", True):
            # The slot map is private to the kernel.  The descriptor's masks are laid out for 4-byte planes (32-lane
            # groups on 32 banks); an 8-byte exchange through them conflicts (PMC, reverse sweep: 0.57 conflict cycles per
            # active LDS cycle).  The planner's rule for 8-byte elements -- 16-lane write groups, 32-lane read groups --
            # is applied here instead (plan.exchange_masks(planar=False)).
            tb = {1 << p: i for i, p in enumerate(self.tile_bits)}

            class _R:          # the two rounds in the planner's terms: tile-bit index of every register / thread bit
                pass

            a_, b_ = _R(), _R()
            a_.reg_tb, a_.thr_tb = [tb[m] for m in wr.reg_phys], [tb[m] for m in wr.thr_phys]
            b_.reg_tb, b_.thr_tb = [tb[m] for m in rdn.reg_phys], [tb[m] for m in rdn.thr_phys]
            A = P.exchange_masks(self.T, a_, b_, planar=False)
            reg_wr, thr_wr = [A[x] for x in a_.reg_tb], [A[x] for x in a_.thr_tb]
            reg_rd, thr_rd = [A[x] for x in b_.reg_tb], [A[x] for x in b_.thr_tb]
        sg = self.seg(f"exchange {k} -> {k + 1}")
        out = sg.parts[0]
        prio = int(self.opts.get("prio", 0))
        if prio & 2:       # the barrier / LDS chain of an exchange is latency-bound: let it overtake the other workgroups' arithmetic
            out.append("  __builtin_amdgcn_s_setprio(2);")
        ws, rs, tpn = self.fresh("ws"), self.fresh("rs"), self.fresh("tph")
        lt = self.local_tid(out)
        self.thread_xor(out, ws, thr_wr, sh, lt)
        self.thread_xor(out, rs, thr_rd, sh, lt)
        self.thread_xor(out, tpn, rdn.thr_phys, 0, lt)
        wplan = self.exchange_plan(reg_wr, thr_wr)
        rplan = self.exchange_plan(reg_rd, thr_rd)
        ctype = "float" if elem_bytes == 4 else "v2f"
        # volatile reads: the load / store optimiser would pair two 4-byte reads into one ds_read2, whose two results
        # land in ONE register pair -- but the two values are the same component of two different amplitudes, so every
        # such read costs two v_mov afterwards (428 of them in a 9-round pass)
        vq = "volatile " if (elem_bytes == 4 and self.opts.get("single_reads", True)) else ""
        # 8-byte elements (reverse sweep): left alone, the load / store optimiser pairs EVERY access into ds_read2_b64 /
        # ds_read2st64_b64 / ds_write2_b64 (scripts/isa_lds_mix.py).  A ds_read2_b64 is served as two accesses of four
        # contiguous 16-lane groups on 32 banks -- 8 LDS cycles per 1 KiB where two ds_read_b64 (32-lane groups on 64
        # banks, the groups the slot map is laid out for) take 4 -- so the paired form runs at half the read rate AND
        # conflicts on a conflict-free map (PMC round 4: 0.45 conflict cycles per active LDS cycle in the sweep, 0 in the
        # forward kernel).  Volatile accesses keep the single-element forms.
        vq8 = "volatile " if (elem_bytes == 8 and self.opts.get("single8", True)) else ""
        vqw = vq8
        vq = vq or vq8

        def walk(out, stem, var, plan, stmt):
            """One address variant (thread part ^ dirty bits) at a time, defined right before its accesses (short live
            ranges: the variants never sit next to each other in registers); clean bits ride in the offset field."""
            order = sorted(range(self.NR), key=lambda r: (plan[r][0], plan[r][1]))
            cur, nm = None, var
            for r in order:
                d, c = plan[r]
                if d != cur:
                    cur = d
                    if d == 0:
                        nm = var
                    else:
                        nm = self.fresh(stem)
                        out.append(f"  const uint32_t {nm} = {var} ^ {(d << sh):#x}u;")
                out.append("  " + stmt(r, f"lb + {nm} + {c << sh}"))

        first = True
        for vec, comp in planes:
            sfx = f".{comp}" if comp else ""
            walk(out, "wsd", ws, wplan, lambda r, ad: f"*({vqw}{ctype} LDS_AS*)({ad}) = {self.A(r, vec)}{sfx};")
            out.append("  __syncthreads();")
            if first:
                out = sg.new_part()
                first = False
            walk(out, "rsd", rs, rplan,
                 lambda r, ad: f"{self.A(r, vec)}{sfx} = *(const {vq}{ctype} LDS_AS*)({ad});")
            out.append("  __syncthreads();")
        if prio & 2:
            out.append("  __builtin_amdgcn_s_setprio(0);")
        self.tphys = tpn
        return sg

    # ---- shared op pieces ----------------------------------------------------------------------------
    def sign_update(self, sg: "Seg", expr: str):
        """sgn ^= sign bit of a shear record.  The asm fence keeps the update where it is written: left alone, the compiler
        defers the whole xor chain to the end of the pass and keeps every record alive until then (250 spilled SGPRs)."""
        if getattr(self, "nostore", False):
            return          # nobody reads psi / lambda after this pass: the sign is never applied
        sg.parts[0].append(f'  sgn ^= __float_as_uint({expr}) & 0x80000000u; asm volatile("" : "+s"(sgn));')

    def pairs_of(self, J: int):
        B = 1 << J
        return [(_ins0(g, J), _ins0(g, J) | B) for g in range(self.NR // 2)]

    def calls8(self, fn: str, J: int, vec: str, tail: str) -> List[str]:
        """fn(x0, y0, ..., x7, y7, tail) over all amplitude pairs of register bit J, eight pairs per call."""
        pr = self.pairs_of(J)
        if len(pr) % 8:
            raise Unsupported("fewer than 8 amplitude pairs per thread")
        out = []
        for g in range(0, len(pr), 8):
            args = ", ".join(f"{self.A(x, vec)}, {self.A(y, vec)}" for x, y in pr[g:g + 8])
            out.append(f"  {fn}({args}, {tail});")
        return out

    def load_v2(self, out: List[str], name: str, off, count: int, tab: str = "ptab"):
        """const v2f name_i = ((KV2)(tab + off))[i], i < count"""
        p = self.fresh("tp")
        out.append(f"  const KV2 {p} = (KV2)({tab} + {off});")
        for i in range(count):
            out.append(f"  const v2f {name}_{i} = {p}[{i}];")

    def slot_ptr(self, slot: int) -> Tuple[str, int]:
        slot = _u32(slot)
        if slot & P.CONST_FLAG:
            return "ctab", slot & ~P.CONST_FLAG
        return "ptab", slot

    def rename_perm(self, ja: int, jb: int, kind: int):
        """CNOT (kind 1: control ja, 2: control jb) / SWAP (3) between register bits: swap the variable names."""
        A, B = 1 << ja, 1 << jb
        for g in range(self.NR // 4):
            r0 = _ins0(_ins0(g, ja), jb)
            if kind == 1:
                x, y = r0 | A, r0 | A | B
            elif kind == 2:
                x, y = r0 | B, r0 | A | B
            else:
                x, y = r0 | B, r0 | A
            self.reg[x], self.reg[y] = self.reg[y], self.reg[x]

    def header(self, kname: str, params: str) -> List[str]:
        return [
            "// GENERATED by tcmi/specialize.py -- one pass of one plan, straight-line (see that file).",
            "#include <hip/hip_runtime.h>",
            "#include <stdint.h>",
            '#include "tcmi_dev.h"',
            "using namespace tcmi;",
            "typedef float v2f __attribute__((ext_vector_type(2)));",
            "typedef float v4f __attribute__((ext_vector_type(4)));",
            "typedef const v2f TCMI_K* KV2;",
            "typedef const float TCMI_K* KF;",
            "#define LDS_AS __attribute__((address_space(3)))",
            '#include "tcmi_vm2_asm.inc"',
            "",
            f'extern "C" __global__ __launch_bounds__({1 << self.LT}, {self.waves_per_eu()}) void {kname}({params}) {{',
        ]

    def waves_per_eu(self) -> int:
        # waves per SIMD the register budget is sized for: 4 (128 VGPRs) with up to 16 amplitudes per vector and thread,
        # 2 (256) beyond that
        return int(self.opts.get("waves", min(4 if self.R * len(self.vectors) <= 5 else 2, 1024 >> self.LT)))


# ======================================================================================================
#  forward gate pass  (mirror of pass2_kernel, csrc/tcmi_vm2.hip)
# ======================================================================================================
class _Forward(_Emitter):
    def __init__(self, words, opts=None):
        super().__init__(words, ["a"], opts)
        if self.R < 4:
            raise Unsupported("R < 4")

    def g1m(self, q: int) -> int:
        w = self.w
        mk, base = int(w[q + 1]), int(w[q + 2])
        for J in range(self.R):
            if not (mk >> J) & 1:
                continue
            kind = (mk >> (8 + 2 * J)) & 3
            sh = (mk >> (P.SHEAR_SHIFT + J)) & 1
            c = self.fresh("c")
            off = base + 8 * J
            sg = self.seg(f"gate on register bit {J}")
            if sh:
                self.load_v2(sg.loads, c, off, 2)
                self.sign_update(sg, f"{c}_1.x")
                if kind == 2:
                    # two shears + the third unless the builder chose the two-shear form for this batch element (the
                    # skip is inside the asm body: the pass stays one basic block)
                    calls = self.calls8("vm2_shear23_8_rx", J, "a", f"{c}_0, __float_as_uint({c}_1.y)")
                    sg.parts[0] += calls[:1]
                    sg.new_part().extend(calls[1:])
                elif kind == 1:
                    calls = self.calls8("vm2_shear8_real", J, "a", f"{c}_0")
                    sg.parts[0] += calls[:1]
                    sg.new_part().extend(calls[1:])
                else:
                    raise Unsupported("shear form of a general gate")
            else:
                self.load_v2(sg.loads, c, off, 4)
                fn = {0: "vm2_gate8_gen", 1: "vm2_gate8_real", 2: "vm2_gate8_rx"}.get(kind)
                if fn is None:
                    raise Unsupported("gate class 3")
                calls = self.calls8(fn, J, "a", f"{c}_0, {c}_1, {c}_2, {c}_3")
                sg.parts[0] += calls[:1]
                sg.new_part().extend(calls[1:])
        return q + 3

    def wave_variant(self, out: List[str], q0: int, nsel: int) -> str:
        """v = sum_k parity(wave's thread index & m_k) << k   (wave-uniform)"""
        v = self.fresh("tv")
        out.append(f"  const uint32_t {v}w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wg_base | {self.tphys}));")
        terms = [f"((__builtin_popcount({v}w & {_u32(self.w[q0 + k2]):#x}u) & 1u) << {k2})" for k2 in range(nsel)]
        out.append(f"  const uint32_t {v} = {' | '.join(terms) if terms else '0u'};")
        return v

    def table_mul(self, pre: List[str], tabexpr: str, fn: str = "vm2_cmul8s", vecs: Sequence[str] = ("a",)):
        """a[r] *= table[r]: two segments of 16 entries (32 scalar registers each)."""
        p = self.fresh("tp")
        CH = min(16, self.NR)
        for h0 in range(0, self.NR, CH):
            sg = self.seg(f"table entries {h0}..{h0 + CH - 1}")
            if h0 == 0:
                sg.loads += pre
                sg.loads.append(f"  const KV2 {p} = (KV2)({tabexpr});")
            t = self.fresh("t")
            for i in range(CH):
                sg.loads.append(f"  const v2f {t}_{i} = {p}[{h0 + i}];")
            first = True
            for h in range(0, CH, 8):
                for vec in vecs:
                    amps = ", ".join(self.A(h0 + h + i, vec) for i in range(8))
                    tabs = ", ".join(f"{t}_{h + i}" for i in range(8))
                    (sg.parts[0] if first else sg.new_part()).append(f"  {fn}({amps}, {tabs});")
                    first = False

    def diagb_apply(self, sg: Seg, J: int, e: str, vecs: Sequence[str] = ("a",)):
        pr = self.pairs_of(J)
        for vec in vecs:
            for g in range(0, len(pr), 4):
                lo = ", ".join(self.A(x, vec) for x, _ in pr[g:g + 4])
                hi = ", ".join(self.A(y, vec) for _, y in pr[g:g + 4])
                sg.parts[-1].append(f"  vm2_cmul44v({lo}, {hi}, {e});")
                if g == 0 and vec == vecs[0]:
                    sg.new_part()

    def op(self, q: int) -> int:
        w = self.w
        op = int(w[q])
        NR = self.NR
        if op == P.OP_G1M:
            return self.g1m(q)
        if op == P.OP_DIAGC:
            self.table_mul([], f"ptab + {int(w[q + 1])}")
            return q + 2
        if op == P.OP_DIAGCW:
            nsel = int(w[q + 2])
            pre: List[str] = []
            v = self.wave_variant(pre, q + 3, nsel)
            self.table_mul(pre, f"ptab + {int(w[q + 1])} + {2 * NR} * {v}")
            return q + 6
        if op == P.OP_DIAGB:
            J, m, slot = int(w[q + 1]), _u32(w[q + 2]), int(w[q + 3])
            sg = self.seg(f"DIAGB bit {J}")
            e, t = self.fresh("e"), self.fresh("b")
            sg.loads.append(f"  const float {t}c = ptab[{slot}], {t}s = ptab[{slot + 1}];")
            sg.parts[0].append(f"  v2f {e}; {e}.x = {t}c; {e}.y = (__builtin_popcount((wg_base | {self.tphys}) & {m:#x}u) & 1) "
                               f"? -{t}s : {t}s;")
            self.diagb_apply(sg, J, e)
            return q + 4
        if op == P.OP_DIAGB2:
            J, m1, m2, slot = int(w[q + 1]), _u32(w[q + 2]), _u32(w[q + 3]), int(w[q + 4])
            sg = self.seg(f"DIAGB2 bit {J}")
            t = self.fresh("b")
            self.load_v2(sg.loads, t, slot, 4)
            e = self.fresh("e")
            p = sg.parts[0]
            p.append(f"  v2f {e};")
            p.append(f"  {{ const uint32_t ti = wg_base | {self.tphys}; const bool s1 = __builtin_popcount(ti & {m1:#x}u) & 1, "
                     f"s2 = __builtin_popcount(ti & {m2:#x}u) & 1;")
            p.append(f"    {e}.x = s2 ? (s1 ? {t}_3.x : {t}_2.x) : (s1 ? {t}_1.x : {t}_0.x);")
            p.append(f"    {e}.y = s2 ? (s1 ? {t}_3.y : {t}_2.y) : (s1 ? {t}_1.y : {t}_0.y); }}")
            self.diagb_apply(sg, J, e)
            return q + 5
        if op == P.OP_G2:
            jak, jb, slot = int(w[q + 1]), int(w[q + 2]), int(w[q + 3])
            ja, kind = jak & 0xFF, jak >> 8
            if kind:
                self.rename_perm(ja, jb, kind)
                return q + 4
            tab, off = self.slot_ptr(slot)
            sg = self.seg(f"G2 bits {ja},{jb}")
            m = self.fresh("m")
            self.load_v2(sg.loads, m, off, 16, tab)
            A, B = 1 << ja, 1 << jb
            ms = ", ".join(f"{m}_{i}" for i in range(16))
            for g in range(0, NR // 4, 2):
                r0, r1 = _ins0(_ins0(g, ja), jb), _ins0(_ins0(g + 1, ja), jb)
                quad = lambda r: f"{self.A(r)}, {self.A(r | B)}, {self.A(r | A)}, {self.A(r | A | B)}"  # noqa: E731
                (sg.parts[0] if g == 0 else sg.new_part()).append(f"  vm2_g2x2({quad(r0)}, {quad(r1)}, {ms});")
            return q + 4
        if op == P.OP_DIAG:
            return self.diag_generic(q)
        raise Unsupported(f"forward op {op}")

    def diag_generic(self, q: int) -> int:
        """General phase polynomial (thread-only terms, many register-x-thread terms: the final flush of a plan): per-thread
        phases in turns, hardware sin / cos.  Same arithmetic, term by term, as the interpreting kernel (pass2_kernel,
        OP_DIAG): double sums over the terms in descriptor order, float phases per register index."""
        w, R, NR = self.w, self.R, self.NR
        nA, nB, nC, base = int(w[q + 1]), int(w[q + 2]), int(w[q + 3]), int(w[q + 4])
        qq = q + 5
        mA = [_u32(w[qq + e]) for e in range(nA)]
        qq += nA
        mB = [_u32(w[qq + e]) for e in range(nB)]
        jB = [int(w[qq + nB + e]) for e in range(nB)]
        qq += 2 * nB
        mC = [_u32(w[qq + e]) for e in range(nC)]
        sg = self.seg("generic phase polynomial")
        p = sg.parts[0]
        t = self.fresh("dg")
        p.append(f"  const uint32_t {t}i = wg_base | {self.tphys};")
        p.append(f"  double {t}phi = 0.0;")
        for e in range(nA):
            p.append(f"  {{ const double c = (double)ptab[{base + e}]; {t}phi += (__builtin_popcount({t}i & {mA[e]:#x}u) & 1) ? -c : c; }}")
        for j in range(R):
            p.append(f"  double {t}c{j} = 0.0;")
        for e in range(nB):
            p.append(f"  {{ const double c = (double)ptab[{base + nA + e}]; "
                     f"{t}c{jB[e]} += (__builtin_popcount({t}i & {mB[e]:#x}u) & 1) ? -c : c; }}")
        p.append(f"  const float {t}p0 = (float)({t}phi - rint({t}phi));")
        for j in range(R):
            p.append(f"  const float {t}f{j} = (float)({t}c{j} - rint({t}c{j}));")
        for e in range(nC):
            p.append(f"  const float {t}k{e} = ptab[{base + nA + nB + e}];")
        for h in range(0, NR, 8):
            part = sg.new_part()
            es = []
            for i in range(8):
                r = h + i
                terms = [f"{t}p0"] + [("-" if (r >> j) & 1 else "+") + f" {t}f{j}" for j in range(R)]
                # the interpreter adds the C terms one after the other to the running float phase
                expr = "(" * (R + nC) + terms[0]
                for x in terms[1:]:
                    expr += f" {x})"
                for e in range(nC):
                    sgn = "-" if bin(r & mC[e]).count("1") & 1 else "+"
                    expr += f" {sgn} {t}k{e})"
                ev = f"{t}e{r}"
                part.append(f"  v2f {ev}; {{ float sn, cs; sincos_turns<float>({expr}, &sn, &cs); {ev}.x = cs; {ev}.y = sn; }}")
                es.append(ev)
            part.append("  vm2_cmul8v(" + ", ".join(self.A(h + i) for i in range(8)) + ", " + ", ".join(es) + ");")
        return qq + nC

    def source(self, kname: str) -> str:
        NR = self.NR
        params = ("v2f* __restrict__ state, long long state_stride, const float* __restrict__ ctab_g, "
                  "const float* __restrict__ ptab_g, long long ptab_stride, uint32_t live_mask, uint32_t umask")
        from_src = bool(self.opts.get("src"))
        if from_src:
            # "src" variant (tcmi_spec_run_pass_from): the tile is read from ANOTHER batch -- state b of this pass starts as
            # scale[b] * src[b >> src_shift] -- and written to ``state``: the replicate-and-weight step of a cut
            # half-circuit (executor._HalfBatch) without a launch and a round trip of its own
            params += (", const v2f* __restrict__ src, long long src_stride, uint32_t src_shift, uint32_t has_scale, "
                       "const v2f* __restrict__ scale")
        else:
            self.opts["umask_arg"] = 1
        pro = ["  extern __shared__ __attribute__((aligned(16))) char lb_[];",
               "  char LDS_AS* const lb = (char LDS_AS*)lb_;",
               "  state += (long long)blockIdx.y * state_stride;",
               "  const KF ctab = (KF)ctab_g; (void)ctab;",
               "  const KF ptab = (KF)(ptab_g + (long long)blockIdx.y * ptab_stride);"]
        pre, loop = self.index_lines()
        pro += pre + loop
        pro.append("  v2f " + ", ".join(f"a{r}" for r in range(NR)) + ";")
        pro.append("  uint32_t sgn = 0u;")
        rd0 = self.rounds[0]
        sg = self.seg("tile load")
        self.tphys = self.fresh("tph")
        if int(self.opts.get("prio", 0)) & 1:      # a young workgroup's loads go out ahead of the older ones' arithmetic
            sg.parts[0].append("  __builtin_amdgcn_s_setprio(3);")
        self.thread_xor(sg.parts[0], self.tphys, rd0.thr_phys)
        if from_src:
            sg.parts[0].append("  const v2f* __restrict__ src_b = src + (long long)(blockIdx.y >> src_shift) * src_stride;")
            self.tile_io(sg.parts[0], rd0, False, {"a": "src_b"}, self.tphys)
        else:
            self.tile_io(sg.parts[0], rd0, False, {"a": "state"}, self.tphys)
        if int(self.opts.get("prio", 0)) & 1:
            sg.parts[0].append("  __builtin_amdgcn_s_setprio(0);")
        if from_src:
            sc = self.fresh("sc")
            p_ = sg.new_part()
            p_.append(f"  v2f {sc} = v2f{{1.f, 0.f}};")
            p_.append(f"  if (has_scale) {sc} = ((KV2)scale)[blockIdx.y];")
            for h in range(0, NR, 8):
                p_.append("  vm2_cmul8s(" + ", ".join(self.A(h + i) for i in range(8)) + ", " + ", ".join([sc] * 8) + ");")
        for k, rd in enumerate(self.rounds):
            q = rd.ops_at
            for _ in range(rd.nops):
                q = self.op(q)
            if q != rd.ops_at + rd.nwords:
                raise Unsupported("descriptor length mismatch")
            if k == self.nrounds - 1:
                break
            self.exchange(k, [("a", "x"), ("a", "y")], 4)
        last = self.rounds[-1]
        sg = self.seg("sign + tile store")
        p = sg.parts[0]
        for r in range(0, NR, 16):
            p.append("  vm2_negate16_if(" + ", ".join(self.A(r + i) for i in range(16)) + ", sgn);")
        self.tile_io(p, last, True, {"a": "state"}, self.tphys)
        return "\n".join(self.header(kname, params) + pro + self.linear() + ["  }", "}"]) + "\n"

    def lds_bytes(self) -> int:
        return 4 << self.T


def forward_source(words, kname: str = "tcmi_spec_pass", opts=None) -> Tuple[str, dict]:
    """HIP source of the straight-line kernel of one forward gate pass + its launch geometry."""
    e = _Forward(words, opts)
    src = e.source(kname)
    return src, {"kind": "forward", "n": e.n, "T": e.T, "LT": e.LT, "lds": e.lds_bytes(), "src": bool(e.opts.get("src"))}


# ======================================================================================================
#  cache, compiler driver, loader
# ======================================================================================================
_SRC_DIGEST = None
_LOCK = threading.Lock()
_LOADED: Dict[tuple, "SpecKernel"] = {}
STATS = {"compiled": 0, "cache_hits": 0, "unsupported": 0, "compile_s": 0.0}


MIN_N = int(os.environ.get("TCMI_SPEC_MIN_N", "22"))   # 'auto': plans below this size are never compiled (cached ones still load)


def mode() -> str:
    """TCMI_SPECIALIZE: '0' = never, '1' = compile at the first use of a plan, 'auto' (default) = use cached kernels at once,
    compile missing ones when a plan turns out to be hot (TCMI_SPEC_HOT calls, default 3) and big enough to pay."""
    return os.environ.get("TCMI_SPECIALIZE", "auto")


def _support_digest() -> bytes:
    global _SRC_DIGEST
    if _SRC_DIGEST is None:
        h = hashlib.blake2b(digest_size=16)
        for f in ("tcmi_vm2_asm.inc", "tcmi_dev.h", "tcmi_vm.h"):
            with open(os.path.join(CSRC, f), "rb") as fh:
                h.update(fh.read())
        h.update(" ".join(HIPCC_FLAGS[:-1]).encode())   # not the include path
        _SRC_DIGEST = h.digest()
    return _SRC_DIGEST


def pass_digest(src: str) -> str:
    """Cache key of a generated kernel: its source text + the included bodies + the compiler flags."""
    h = hashlib.blake2b(digest_size=16)
    h.update(_support_digest())
    h.update(src.encode())
    return h.hexdigest()


class SpecKernel:
    """A loaded plan-specialised kernel: ``handle`` for tcmi_spec_run_*; geometry in ``meta``."""

    def __init__(self, handle, meta, path):
        self.handle, self.meta, self.path = handle, meta, path


_EMITTERS = {}
# Emitter options a measuring script may override for EVERY kernel it has generated from here on (scripts/ set this dict
# themselves; there is no environment switch): tuning choices that do not change results -- "waves", "prio", "xcd", "ntl",
# "nts", "single_reads", "single8", "own_slots", "nosb".  Part of the generated text, hence of the cache key.
EXTRA_OPTS: Dict[str, int] = {}
try:        # TCMI_KNOBS="spec.xcd=0,spec.waves=3": the same from the environment (tcmi/_knobs.py)
    from . import _knobs as _KN

    EXTRA_OPTS.update({k[5:]: int(v) for k, v in _KN.VALUES.items() if k.startswith("spec.")})
except ImportError:     # the generator also runs stand-alone (scripts)
    pass


_SHORT = {"forward": "fwd", "adjoint": "adj"}


def _source(kind: str, words, opts, index: int = 0) -> Tuple[str, dict]:
    """Source + meta of pass ``index`` of a plan.  The kernel is NAMED after its pass and its text --
    ``tcmi_spec_fwd_p<k>_<digest8>`` / ``tcmi_spec_adj_p<k>_<digest8>`` -- so that a rocprofv3 kernel trace separates the
    passes of a plan (and the plans of one process) instead of one row for every generated kernel."""
    opts = dict(opts or {})
    opts.update(EXTRA_OPTS)
    generic = f"tcmi_spec_{kind}"
    src, meta = _EMITTERS[kind](words, generic, opts)
    name = f"tcmi_spec_{_SHORT.get(kind, kind)}_p{int(index)}_{pass_digest(src)[:8]}"
    src = src.replace(generic, name)
    meta["kernel"] = name
    meta["pass_index"] = int(index)
    meta["arith"] = pass_arithmetic(src)
    return src, meta


_BODY_OPS: Optional[Dict[str, Dict[str, float]]] = None


def _body_ops() -> Dict[str, Dict[str, float]]:
    """Packed-f32 instruction counts of the hand-written bodies (csrc/tcmi_vm2_asm.inc), by mnemonic, as EXECUTED: the
    two-or-three-shear bodies skip part of their text at run time (the builder's flag bit), counted here in the two-shear
    form the plans of this package ask for (``shear2``) -- the lower bound of the arithmetic."""
    global _BODY_OPS
    if _BODY_OPS is None:
        import re

        text = open(os.path.join(CSRC, "tcmi_vm2_asm.inc")).read()
        out: Dict[str, Dict[str, float]] = {}
        for m in re.finditer(r"void (vm2_\w+)\(.*?\n\}\n", text, re.S):
            body = m.group(0)
            ops = {k: float(len(re.findall(r'"' + k + r" ", body))) for k in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")}
            out[m.group(1)] = ops
        # executed, not written: shear23 = 16 of its 24 FMAs in the two-shear form, shear23l = 16 of the 40 it holds
        for name in ("vm2_shear23_8_rx", "vm2_shear23l_8_rx"):
            if name in out:
                out[name] = {"v_pk_fma_f32": 16.0, "v_pk_mul_f32": 0.0, "v_pk_add_f32": 0.0}
        _BODY_OPS = out
    return _BODY_OPS


def pass_arithmetic(src: str) -> dict:
    """Arithmetic of one generated pass per WAVE, counted from its source: packed-f32 instructions of the asm bodies it
    calls (two-shear forms), the packed adds of the Walsh transforms written as plain code, and the wave folds of the
    gradient events (29 VALU instructions each, tcmi_dev.h wave_fold8).  ``flops`` per wave = 64 lanes x (4 per packed
    FMA, 2 per packed multiply / add); ``valu_instructions`` = what the VALU must issue at least (a packed op holds the
    issue port for 4 clocks; exchange addressing, selects and moves come on top)."""
    import re

    ops = _body_ops()
    tot = {"v_pk_fma_f32": 0.0, "v_pk_mul_f32": 0.0, "v_pk_add_f32": 0.0}
    for name, cnt in re.findall(r"\b(vm2_\w+)\(", src) and [(n_, src.count(n_ + "(")) for n_ in set(re.findall(r"\b(vm2_\w+)\(", src))]:
        for k, v in ops.get(name, {}).items():
            tot[k] += v * cnt
    # Walsh transform lines: "const v2f a = lo + hi;" / "... = lo - hi;" = one packed add each
    tot["v_pk_add_f32"] += float(len(re.findall(r"const v2f \w+ = \w+ [+-] \w+;", src)))
    tot["v_pk_fma_f32"] += float(src.count("__builtin_elementwise_fma("))      # Pauli-sum terms folded into a sweep pass
    folds = src.count("wave_fold8(")
    pk = tot["v_pk_fma_f32"] + tot["v_pk_mul_f32"] + tot["v_pk_add_f32"]
    flops = 64.0 * (4.0 * tot["v_pk_fma_f32"] + 2.0 * tot["v_pk_mul_f32"] + 2.0 * tot["v_pk_add_f32"])
    return {"pk_fma": tot["v_pk_fma_f32"], "pk_mul": tot["v_pk_mul_f32"], "pk_add": tot["v_pk_add_f32"],
            "wave_folds": folds, "valu_instructions": pk + 29.0 * folds, "flops": flops}


def have_compiler() -> bool:
    return os.path.exists(HIPCC) or shutil.which("hipcc") is not None


def _compile(src: str, out_path: str, keep_source: bool):
    """Compile one generated kernel into ``out_path`` -- ONE writer per code object and cache directory: the processes
    that share a cache (the ranks of a job on one node, all of which want the same passes at the same moment) agree
    through ``<out_path>.lock`` (O_EXCL) on who runs hipcc; the others wait for the file to appear.  A lock whose owner
    died (pid gone, or older than TCMI_SPEC_LOCK_STALE_S = 600 s) is taken over."""
    import time

    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    lock = out_path + ".lock"
    stale_s = float(os.environ.get("TCMI_SPEC_LOCK_STALE_S", "600"))
    while True:
        if os.path.exists(out_path):
            return
        try:
            fd = os.open(lock, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o644)
            os.write(fd, str(os.getpid()).encode())
            os.close(fd)
            break
        except FileExistsError:
            try:
                st = os.stat(lock)
                with open(lock) as fh:
                    owner = int(fh.read().strip() or "0")
                dead = owner > 0 and owner != os.getpid() and not os.path.exists(f"/proc/{owner}")
                if dead or time.time() - st.st_mtime > stale_s:
                    os.remove(lock)
                    continue
            except (OSError, ValueError):
                pass          # the owner finished (or is writing its pid) between the two calls: look again
            time.sleep(0.05)
    tmp_base = f"{out_path}.{os.getpid()}.{threading.get_ident()}"
    hip = tmp_base + ".hip"
    tmp_out = tmp_base + ".hsaco"
    try:
        if os.path.exists(out_path):      # written between the check and the lock
            return
        with open(hip, "w") as fh:
            fh.write(src)
        cc = HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")
        r = subprocess.run([cc] + HIPCC_FLAGS + [hip, "-o", tmp_out], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(tmp_out):
            raise RuntimeError(f"hipcc failed on a plan-specialised kernel:\n{r.stderr[-2000:]}")
        os.replace(tmp_out, out_path)     # atomic: a reader sees the whole file or none
        if keep_source:
            os.replace(hip, out_path[:-6] + ".hip")
    finally:
        for f in (hip, tmp_out, lock):
            if os.path.exists(f):
                try:
                    os.remove(f)
                except OSError:
                    pass


def _compile_workers(njobs: int) -> int:
    """hipcc processes this process may run side by side: the host's cores are shared by the ranks of the node
    (LOCAL_WORLD_SIZE of the launcher), one core stays free for the launch threads."""
    try:
        local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        local = 1
    return max(1, min(njobs, ((os.cpu_count() or 2) - 1) // local, 32))


def _user_cache_dir() -> Optional[str]:
    """Per-user cache (``$XDG_CACHE_HOME/tcmi/plancache``, default ``~/.cache``) for trees that are read-only.  Code
    objects found here are loaded and LAUNCHED, so the directory is only used when it belongs to this user and nobody
    else can write to it (created 0700; a directory somebody else prepared is ignored)."""
    base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    d = os.path.join(base, "tcmi", "plancache")
    try:
        os.makedirs(d, mode=0o700, exist_ok=True)
        st = os.stat(d)
    except OSError:
        return None
    if st.st_uid != os.getuid() or (st.st_mode & 0o022):
        return None
    return d


def _cache_dirs() -> List[str]:
    """In-tree cache first (it travels with the built library), the per-user directory when the tree is read-only."""
    u = _user_cache_dir()
    return [CACHE_DIR] + ([u] if u else [])


def _find(dg: str) -> Optional[str]:
    for d in _cache_dirs():
        p = os.path.join(d, dg + ".hsaco")
        if os.path.exists(p):
            return p
    return None


def _writable_dir() -> str:
    for d in _cache_dirs():
        try:
            os.makedirs(d, exist_ok=True)
            if os.access(d, os.W_OK):
                return d
        except OSError:
            continue
    raise RuntimeError("no writable directory for plan-specialised kernels")


def prepare(kind: str, descs: Sequence, opts: Optional[dict] = None, compile_missing: bool = True,
            workers: Optional[int] = None) -> List[Optional[Tuple[str, dict]]]:
    """(code-object path, meta) per descriptor, None where the pass is not specialisable (or not compiled yet and
    ``compile_missing`` is off).  Needs no GPU: __graft_entry__.build() pre-compiles the bench plans with it."""
    import time

    out: List[Optional[Tuple[str, dict]]] = [None] * len(descs)
    jobs = []
    for i, d in enumerate(descs):
        try:
            src, meta = _source(kind, d, opts, i)
        except Unsupported:
            STATS["unsupported"] += 1
            continue
        dg = pass_digest(src)
        p = _find(dg)
        if p is not None:
            STATS["cache_hits"] += 1
            out[i] = (p, meta)
        elif compile_missing and have_compiler():
            jobs.append((i, src, meta, os.path.join(_writable_dir(), dg + ".hsaco")))
    if jobs:
        t0 = time.perf_counter()
        keep = bool(os.environ.get("TCMI_SPEC_KEEP"))
        nw = workers or _compile_workers(len(jobs))
        with ThreadPoolExecutor(max_workers=nw) as ex:
            futs = [ex.submit(_compile, src, path, keep) for _, src, _, path in jobs]
            for f in futs:
                f.result()
        for i, _src, meta, path in jobs:
            out[i] = (path, meta)
        STATS["compiled"] += len(jobs)
        STATS["compile_s"] += time.perf_counter() - t0
    return out


def load(path: str, meta: dict) -> SpecKernel:
    import ctypes

    from . import _lib

    import torch

    key = (path, torch.cuda.current_device())      # a module belongs to the device it was loaded on
    with _LOCK:
        k = _LOADED.get(key)
        if k is None:
            h = ctypes.c_void_p()
            _lib.check(_lib.lib().tcmi_spec_load(path.encode(), meta["kernel"].encode(), int(meta["lds"]), ctypes.byref(h)),
                       "tcmi_spec_load")
            if meta.get("src"):        # the launchers check that a handle matches the argument buffer they build
                _lib.check(_lib.lib().tcmi_spec_set_flags(h, 1), "tcmi_spec_set_flags")
            k = SpecKernel(h, meta, path)
            _LOADED[key] = k
        return k


class PassSet:
    """The specialised kernels of one list of pass descriptors (a forward plan or a reverse sweep), resolved lazily:
    cached code objects are loaded at the first call, missing ones are compiled once the plan is hot."""

    def __init__(self, kind: str, descs: Sequence, n_exec: int, opts: Optional[dict] = None):
        self.kind, self.descs, self.n_exec, self.opts = kind, [np.asarray(d) for d in descs], n_exec, opts
        self.kernels: List[Optional[SpecKernel]] = [None] * len(self.descs)
        self.calls = 0
        self.state = "new"       # new -> cached (hits loaded, misses pending) -> done
        self.hot = int(os.environ.get("TCMI_SPEC_HOT", "3"))

    def get(self) -> List[Optional[SpecKernel]]:
        m = mode()
        if m == "0":
            return [None] * len(self.descs)
        if self.state == "done":
            return self.kernels
        try:
            import torch

            if torch.cuda.is_current_stream_capturing():   # no module loads / compiler runs inside a hipGraph capture
                return self.kernels
        except Exception:  # noqa: BLE001
            pass
        self.calls += 1
        try:
            if self.state == "new":
                self._resolve(compile_missing=False)
                self.state = "cached"
                if all(k is not None for k in self.kernels):
                    self.state = "done"
                    return self.kernels
            if m == "1" or (self.calls >= self.hot and self.n_exec >= MIN_N):
                self._resolve(compile_missing=True)
                self.state = "done"
        except Exception as e:  # noqa: BLE001
            # specialisation is an optimisation: a compiler that fails, a cache that cannot be written or a code object
            # that does not load must never take the computation down -- the interpreting kernels run instead
            import warnings

            warnings.warn(f"tcmi: plan-specialised {self.kind} kernels unavailable ({type(e).__name__}: {str(e)[:200]}); "
                          f"using the interpreting kernels", RuntimeWarning)
            self.state = "done"
        return self.kernels

    def _resolve(self, compile_missing: bool) -> None:
        for i, r in enumerate(prepare(self.kind, self.descs, self.opts, compile_missing=compile_missing)):
            if r is not None and self.kernels[i] is None:
                try:
                    self.kernels[i] = load(*r)
                except Exception as e:  # noqa: BLE001
                    # a truncated file in the per-user cache is dropped (recompiled next time); never on a transient error
                    # (out of memory ...) and never from the in-tree cache build() shipped
                    u = _user_cache_dir()
                    if u and os.path.dirname(os.path.abspath(r[0])) == os.path.abspath(u) and any(
                            t in str(e).lower() for t in ("invalid image", "invalid kernel file", "invalid device function",
                                                          "no kernel image", "not found")):
                        try:
                            os.remove(r[0])
                        except OSError:
                            pass
                    raise


_EMITTERS["forward"] = forward_source


def adjoint_opts(cfg) -> dict:
    """Emitter options of a reverse-sweep plan (part of the cache key)."""
    return {"shear2": bool(cfg.shear2)}


def precompile_circuit(c, adjoint: bool = True, forward: bool = True, fold_x=None, fold_z=None, nterms: int = 0,
                       fold_pairs=None) -> dict:
    """Compile (into the cache) the specialised kernels of the plans the executor will choose for circuit ``c``: the
    forward passes and, with ``adjoint``, the reverse sweeps a value_and_grad may run (short and full gate list, last pass
    with and without write-back).  ``fold_x`` = [(qubit, weight)] / ``fold_z`` = [(qubits, weight)] / ``nterms``: also the
    sweep in which those single-X terms and Z-only strings of a Pauli-sum energy of ``nterms`` terms are born (executor.
    fold_setup); ``fold_pairs`` = [((qubit a, qubit b), (kind a, kind b), weight)]: its two-factor strings (kind 0 = X,
    1 = Y).  Host work only -- no GPU needed (hipcc cross-compiles)."""
    from . import cons
    from . import executor as X

    gates, nparams = c._gate_records(), len(c._params)
    n_exec, cfg, plan, eg = X.choose_plan(c._nqubits, gates, nparams, cons.dtypestr, cons._plan_options)
    res = {"n_exec": n_exec, "forward": None, "adjoint": None}
    if cfg.gen < 2 or cons.dtypestr != "complex64":
        return res
    if forward:
        res["forward"] = [r is not None for r in prepare("forward", plan.descs)]
        # wavefunction by cut contraction: the two half-circuit batches have plans of their own (prefix / suffix)
        spec = X.choose_cut(c._nqubits, gates, nparams, cons.dtypestr, plan)
        # (a cut whose last crossing gate is applied by the join kernel carries the plain cut along: joins on the exact-f32
        # kernel run that one)
        res["cut_halves"] = [] if spec is not None else None
        for spec in ([] if spec is None else [spec] + ([spec.plain] if getattr(spec, "plain", None) is not None else [])):
            nb = len(spec.bonds)
            radices = [len(b.terms) for b in spec.bonds]
            for nq, hg in ((spec.n_left, spec.left), (c._nqubits - spec.n_left, spec.right)):
                s_, cut, _K = X._HalfBatch.split_point(hg, nparams, nb, radices)
                for isub, sub in enumerate([hg] if s_ == 0 else [hg[:cut], hg[cut:]]):
                    _, hcfg, hplan, _ = X.choose_plan(nq, sub, nparams + nb, cons.dtypestr, cons._plan_options)
                    if hcfg.gen >= 2:
                        res["cut_halves"] += [r is not None for r in prepare("forward", hplan.descs)]
                        if isub == 1:      # the suffix reads its input states from the prefix batch (tcmi_spec_run_pass_from)
                            res["cut_halves"] += [r is not None for r in prepare("forward", hplan.descs[:1], {"src": 1})]
    if adjoint and "adjoint" in _EMITTERS:
        for full, zero in ((False, False), (True, False), (True, True)):
            r = X.choose_adjoint_plan(eg, n_exec, cons.dtypestr, full, zero)    # zero: the sweep of a psi from |0...0>
            if r is None:
                continue
            acfg, ap = r
            if acfg.gen < 2:
                continue
            descs = [np.asarray(d) for d in ap.descs]
            out = [x is not None for x in prepare("adjoint", descs, adjoint_opts(acfg))]
            if descs:      # the sweep of a traced value_and_grad keeps its last tile to itself (either plan: the full one
                # is the cheaper sweep for a psi that came from |0...0>, executor._adjoint_from_zero)
                last = descs[-1].copy()
                last[6] = last[6] | P.FLAG_NOSTORE
                out.append(prepare("adjoint", [last], adjoint_opts(acfg))[0] is not None)
            res["adjoint" if not full else ("adjoint_zero_start" if zero else "adjoint_full")] = out
        if fold_x or fold_pairs:
            plans = {}

            def get_plan(full):
                if full not in plans:
                    r_ = X.choose_adjoint_plan(eg, n_exec, cons.dtypestr, full, full)
                    if r_ is None:
                        r_ = X.choose_adjoint_plan(eg, n_exec, cons.dtypestr, True, False)
                    plans[full] = {"plan": r_[1], "cfg": r_[0]}
                return plans[full]

            adj0, _m, _f = X.pick_adjoint_from_zero(eg, n_exec, get_plan)
            pad = n_exec - c._nqubits
            xw = [(k, n_exec - 1 - (int(q) + pad), float(w)) for k, (q, w) in enumerate(fold_x or [])]
            dw = [(len(xw) + k, sum(1 << (n_exec - 1 - (int(q) + pad)) for q in qs), float(w))
                  for k, (qs, w) in enumerate(fold_z or [])]
            pw = [(len(xw) + len(dw) + k, n_exec - 1 - (int(qa) + pad), n_exec - 1 - (int(qb) + pad), float(w), int(ka), int(kb))
                  for k, ((qa, qb), (ka, kb), w) in enumerate(fold_pairs or [])]
            fr = X.fold_plan_host(eg, n_exec, adj0, nparams, xw, dw, nterms, pw) if adj0["cfg"].gen >= 2 else None
            if fr is not None:
                descs = [np.asarray(d) for d in fr[0].descs]
                descs[-1] = descs[-1].copy()
                descs[-1][6] = descs[-1][6] | P.FLAG_NOSTORE
                res["adjoint_fold"] = [x is not None for x in prepare("adjoint", descs, adjoint_opts(adj0["cfg"]))]
    return res


# ======================================================================================================
#  reverse-sweep pass  (mirror of adjoint2_kernel, csrc/tcmi_adjoint2.hip)
# ======================================================================================================
class _Adjoint(_Forward):
    """psi (a*) is un-computed and lambda (l*) propagated through U^dagger gate by gate; every parametrised gate and
    diagonal term leaves one gradient EVENT = a per-lane partial sum.  Events are static here, so
      * they are reduced over the wave four at a time (wave_sum4_uniform) and parked in lanes of accumulator registers
        (v_writelane) -- no LDS atomics, no slot bookkeeping at run time;
      * the Walsh-Hadamard transform of w = Im(conj(lambda) psi) is written as plain float code: only the outputs some
        event reads are live, the compiler drops the rest of the butterfly;
      * the accumulators of the workgroup's waves meet in LDS once, at the end of the pass: one f64 atomic per event."""

    def __init__(self, words, opts=None):
        _Emitter.__init__(self, words, ["a", "l"], opts)
        if self.R < 4:
            raise Unsupported("R < 4")
        self.events: List[int] = []          # gradient slot of event e
        self.pending: List[Tuple[str, int]] = []   # (per-lane float expression, slot) not reduced yet
        self.nostore = bool(self.flags & P.FLAG_NOSTORE)

    # ---- gradient events ---------------------------------------------------------------------------------
    EVB = 8     # events reduced together (wave_fold8)

    def event(self, sg: Seg, expr: str, slot: int):
        v = self.fresh("gv")
        sg.parts[-1].append(f"  const float {v} = {expr};")
        self.pending.append((v, slot))
        if len(self.pending) == self.EVB:
            self.reduce_pending(sg.parts[-1])

    def reduce_pending(self, out: List[str]):
        """Eight pending per-lane sums -> one folded register -> parked in lane group (batch % 8) of an accumulator:
        lane e % 64 of gacc[e / 64] is event e (tcmi_dev.h wave_fold8 / park8)."""
        if not self.pending:
            return
        names = [v for v, _ in self.pending] + ["0.f"] * (self.EVB - len(self.pending))
        e0 = len(self.events)
        assert e0 % self.EVB == 0
        r = self.fresh("gr")
        out.append(f"  const float {r} = wave_fold8({', '.join(names)}, (int)lane);")
        g_ = (e0 // 8) % 8       # lane group of this batch of eight: row g_ >> 1, banks 2 (g_ & 1) and 2 (g_ & 1) + 1
        out.append(f'  asm("v_add_f32_dpp %0, %1, %0 quad_perm:[0,1,2,3] row_mask:{1 << (g_ >> 1):#x} bank_mask:{3 << (2 * (g_ & 1)):#x}" '
                   f': "+v"(gacc{e0 // 64}) : "v"({r}));')
        for _v, slot in self.pending:
            self.events.append(slot)
        while len(self.events) % self.EVB:
            self.events.append(-1)        # padding lane: nothing to add
        self.pending = []

    # ---- ops ---------------------------------------------------------------------------------------------
    def g1m(self, q: int) -> int:
        w, R = self.w, self.R
        mk, ubase, kmask, kbase = int(w[q + 1]), int(w[q + 2]), int(w[q + 3]), int(w[q + 4])
        for J in range(R):
            if not (mk >> J) & 1:
                continue
            kd = (mk >> (8 + 2 * J)) & 3
            sh = (mk >> (P.SHEAR_SHIFT + J)) & 1
            sg = self.seg(f"U^dagger on register bit {J}")
            u = self.fresh("u")
            pr = self.pairs_of(J)
            if (kmask >> J) & 1:
                k = self.fresh("k")
                self.load_v2(sg.loads, k, kbase + 8 * J, 4)
                fn = {2: "vm2_grad4_rx", 1: "vm2_grad4_real", 0: "vm2_grad4_gen"}.get(kd)
                if fn is None:
                    raise Unsupported("generator class 3")
                p = sg.parts[-1]
                c0, c1 = self.fresh("gc"), self.fresh("gc")
                p.append(f"  v2f {c0}, {c1};")
                for g in range(0, len(pr), 4):
                    av = ", ".join(f"{self.A(x, 'a')}, {self.A(y, 'a')}" for x, y in pr[g:g + 4])
                    lv = ", ".join(f"{self.A(x, 'l')}, {self.A(y, 'l')}" for x, y in pr[g:g + 4])
                    kk = f", {k}_0, {k}_1, {k}_2, {k}_3" if kd == 0 else ""
                    if g == 0 and kd == 0:
                        p.append(f"  {c0} = v2f{{0.f, 0.f}}; {c1} = v2f{{0.f, 0.f}};")
                    # the first group starts the two accumulators (v_pk_mul), the others add to them
                    p.append(f"  {fn if (g or kd == 0) else fn.replace('grad4_', 'grad4i_')}({av}, {lv}{kk}, {c0}, {c1});")
                s0, s1 = f"({c0}.x + {c0}.y)", f"({c1}.x + {c1}.y)"
                if kd == 2:      # K = i kappa X, kappa = Im K01
                    expr = f"-{k}_1.y * (({s0}) + ({s1}))"
                elif kd == 1:    # real antisymmetric K
                    expr = f"{k}_1.x * ({s0}) + {k}_2.x * ({s1})"
                else:
                    expr = f"({s0}) + ({s1})"
                self.event(sg, expr, int(w[q + 5 + J]))
                sg.new_part()
            if sh:
                self.load_v2(sg.loads, u, ubase + 8 * J, 2)
                self.sign_update(sg, f"{u}_1.x")
                fn = {2: "vm2_shear8_rx", 1: "vm2_shear8_real"}.get(kd)
                if fn is None:
                    raise Unsupported("shear form of a general gate")
                tail = f"{u}_0"
            else:
                self.load_v2(sg.loads, u, ubase + 8 * J, 4)
                fn = {0: "vm2_gate8_gen", 1: "vm2_gate8_real", 2: "vm2_gate8_rx"}.get(kd)
                if fn is None:
                    raise Unsupported("gate class 3")
                tail = f"{u}_0, {u}_1, {u}_2, {u}_3"
            for vec in ("a", "l"):
                f2, t2 = fn, tail
                if sh and kd == 2 and self.opts.get("shear2"):
                    # the builder may have chosen the two-shear form for this batch element (flag word of the record):
                    # psi gets two shears, lambda the two in the other order; the skip is inside the asm bodies
                    f2 = "vm2_shear23_8_rx" if vec == "a" else "vm2_shear23l_8_rx"
                    t2 = f"{u}_0, __float_as_uint({u}_1.y)"
                for cl in self.calls8(f2, J, vec, t2):
                    sg.new_part().append(cl)
        return q + 5 + R

    def diagf(self, q: int) -> int:
        w, R, NR = self.w, self.R, self.NR
        cslot, hasC, nB, nA, nsel = (_i32(w[q + 1]), int(w[q + 2]), int(w[q + 3]), int(w[q + 4]), int(w[q + 5]))
        gsc = [_i32(w[q + 9 + k]) for k in range(NR)]
        qq = q + 9 + NR
        Bs = [(int(w[qq + 4 * e]), _u32(w[qq + 4 * e + 1]), _i32(w[qq + 4 * e + 2]), _i32(w[qq + 4 * e + 3])) for e in range(nB)]
        qq += 4 * nB
        As = [(_u32(w[qq + 2 * e]), _i32(w[qq + 2 * e + 1])) for e in range(nA)]
        qn = qq + 2 * nA
        need_w = ((hasC & 1) and any(g >= 0 for g in gsc[1:])) or any(b[3] >= 0 for b in Bs) or any(a_[1] >= 0 for a_ in As)
        sg = self.seg("diagonal flush: gradients")
        p = sg.parts[-1]
        wn = None
        if need_w:
            # w[r] = Im(conj(lambda[r]) psi[r]); its Walsh-Hadamard transform holds every term's signed sum.  Plain float
            # code: the outputs nobody reads (and the butterflies that only feed them) are dropped by the compiler.
            wn = self.fresh("w")
            for h in range(0, NR, 8):
                ts = [f"{wn}t{h + i}" for i in range(8)]
                p.append("  v2f " + ", ".join(ts) + ";")
                p.append("  vm2_cross8(" + ", ".join(self.A(h + i, "a") for i in range(8)) + ", "
                         + ", ".join(self.A(h + i, "l") for i in range(8)) + ", " + ", ".join(ts) + ");")
            # the transform runs on the (re, im) PAIRS (packed adds: two transforms for the price of one), the difference
            # t.x - t.y is taken only of the outputs that are read.  Only the butterflies that feed an output some event
            # reads are written (the source is what executes: pass_arithmetic counts it).
            wanted = set()
            if hasC & 1:
                wanted |= {k3 for k3 in range(1, NR) if gsc[k3] >= 0}
            wanted |= {1 << b[0] for b in Bs if b[3] >= 0}
            if any(a_[1] >= 0 for a_ in As):
                wanted.add(0)
            need = [set() for _ in range(R + 1)]      # need[j]: indices whose value after stage j - 1 is read
            need[R] = set(wanted)
            for j in reversed(range(R)):
                for r in need[j + 1]:
                    need[j] |= {r & ~(1 << j), r | (1 << j)}
            cur = [f"{wn}t{r}" for r in range(NR)]
            for j in range(R):
                nxt = list(cur)
                for r in sorted(need[j + 1]):
                    lo, hi = cur[r & ~(1 << j)], cur[r | (1 << j)]
                    nm = f"{wn}_{j}_{r}"
                    p.append(f"  const v2f {nm} = {lo} {'-' if (r >> j) & 1 else '+'} {hi};")
                    nxt[r] = nm
                cur = nxt
            cur = [f"({c}.x - {c}.y)" for c in cur]
            W = cur
            if hasC & 1:
                for k3 in range(1, NR):
                    if gsc[k3] >= 0:
                        self.event(sg, W[k3], gsc[k3])
            tix = self.fresh("ti")
            if Bs or As:
                p = sg.parts[-1]
                p.append(f"  const uint32_t {tix} = wg_base | {self.tphys};")
            for (jj, mask, slot, gs) in Bs:
                if gs >= 0:
                    self.event(sg, f"(__builtin_popcount({tix} & {mask:#x}u) & 1) ? -{W[1 << jj]} : {W[1 << jj]}", gs)
            for (mask, gs) in As:
                if gs >= 0:
                    self.event(sg, f"(__builtin_popcount({tix} & {mask:#x}u) & 1) ? -{W[0]} : {W[0]}", gs)
        # conjugate table multiply of both vectors
        if cslot >= 0:
            pre: List[str] = []
            v = self.wave_variant(pre, q + 6, nsel) if nsel > 0 else "0u"
            twin = (hasC >> 1) & 1
            base = f"ptab + {cslot} + {2 * NR} * {v}"
            if not twin:
                self.table_mul(pre, base, "vm2_cmul8s_conj", ("a", "l"))
            else:
                self.table_mul(pre, base, "vm2_cmul8s_conj", ("a",))
                self.table_mul([], f"{base} + {2 * (NR << nsel)}", "vm2_cmul8s_conj", ("l",))
        # register-x-thread terms with their own factor: the inverse phase (cs - i ys on z = +1) on both vectors
        for (jj, mask, slot, gs) in Bs:
            if slot < 0:
                continue
            sg2 = self.seg(f"DIAGF B term bit {jj}")
            e, t = self.fresh("e"), self.fresh("b")
            sg2.loads.append(f"  const float {t}c = ptab[{slot}], {t}s = ptab[{slot + 1}];")
            sg2.parts[0].append(f"  v2f {e}; {e}.x = {t}c; {e}.y = (__builtin_popcount((wg_base | {self.tphys}) & {mask:#x}u) & 1) "
                                f"? {t}s : -{t}s;")
            self.diagb_apply(sg2, jj, e, ("a", "l"))
        return qn

    def xfold(self, q: int) -> int:
        """{OP_XFOLD, j, cslot, gslot}: lambda[r] += c psi[r ^ (1 << j)] -- the term (c / 2) X_q of a Pauli-sum cotangent
        2 sum_t w_t P_t |psi>, born in registers instead of arriving from a pass of tcmi_apply_pauli_sum_tiled (plan.py
        fold_rounds) -- and its energy (c / 2) <X_q> = c sum_pairs Re(conj(psi_x) psi_y) as one gradient event."""
        w = self.w
        J, kind, cslot, gslot = int(w[q + 1]) & 0xFF, int(w[q + 1]) >> 8, int(w[q + 2]), _i32(w[q + 3])
        tab, off = self.slot_ptr(cslot)
        sg = self.seg(f"{'XY'[kind]} fold on register bit {J}")
        c, e = self.fresh("xc"), self.fresh("xe")
        sg.loads.append(f"  const float {c} = {tab}[{off}];")
        p = sg.parts[0]
        pr = self.pairs_of(J)
        if kind == 0:
            # energy: c sum_pairs Re(conj(a_x) a_y) = c sum (x.x y.x + x.y y.y)
            p.append(f"  v2f {e} = {self.A(pr[0][0], 'a')} * {self.A(pr[0][1], 'a')};")
            for x, y in pr[1:]:
                p.append(f"  {e} = __builtin_elementwise_fma({self.A(x, 'a')}, {self.A(y, 'a')}, {e});")
            p = sg.new_part()
            for x, y in pr:
                p.append(f"  {self.A(x, 'l')} = __builtin_elementwise_fma(v2f{{{c}, {c}}}, {self.A(y, 'a')}, {self.A(x, 'l')}); "
                         f"{self.A(y, 'l')} = __builtin_elementwise_fma(v2f{{{c}, {c}}}, {self.A(x, 'a')}, {self.A(y, 'l')});")
            self.event(sg, f"{c} * ({e}.x + {e}.y)", gslot)
        else:
            # Y: (Y a)_x = -i a_y, (Y a)_y = +i a_x; energy: c sum_pairs Re(conj(a_x) (-i a_y)) = c sum (x.x y.y - x.y y.x)
            p.append(f"  v2f {e} = {self.A(pr[0][0], 'a')} * {self.A(pr[0][1], 'a')}.yx;")
            for x, y in pr[1:]:
                p.append(f"  {e} = __builtin_elementwise_fma({self.A(x, 'a')}, {self.A(y, 'a')}.yx, {e});")
            p = sg.new_part()
            for x, y in pr:
                p.append(f"  {self.A(x, 'l')} = __builtin_elementwise_fma(v2f{{{c}, -{c}}}, {self.A(y, 'a')}.yx, {self.A(x, 'l')}); "
                         f"{self.A(y, 'l')} = __builtin_elementwise_fma(v2f{{-{c}, {c}}}, {self.A(x, 'a')}.yx, {self.A(y, 'l')});")
            self.event(sg, f"{c} * ({e}.x - {e}.y)", gslot)
        return q + 4

    def xfold2(self, q: int) -> int:
        """{OP_XFOLD2, ja | jb << 8 | ka << 16 | kb << 17, cslot, gslot}: lambda[r] += c f(r) psi[r ^ (1 << ja) ^ (1 << jb)] --
        the two-factor string (c / 2) P_a P_b (P = X or Y) of a Pauli-sum cotangent born in registers (plan.py fold_rounds:
        both bits are register bits of this round) -- with f(r) = P_a[x_a, 1 - x_a] P_b[x_b, 1 - x_b] in {1, -1, -i, +i},
        and its energy (c / 2) Re <psi| P_a P_b |psi> as one gradient event: per group of four amplitudes the two partner
        products (00 <-> 11, 01 <-> 10), real parts for XX / YY, imaginary parts for XY / YX."""
        w = self.w
        w1 = int(w[q + 1])
        ja, jb, ka, kb = w1 & 0xFF, (w1 >> 8) & 0xFF, (w1 >> 16) & 1, (w1 >> 17) & 1
        cslot, gslot = int(w[q + 2]), _i32(w[q + 3])
        tab, off = self.slot_ptr(cslot)
        sg = self.seg(f"{'XY'[ka]}{'XY'[kb]} fold on register bits {ja}, {jb}")
        c, e = self.fresh("pc"), self.fresh("pe")
        sg.loads.append(f"  const float {c} = {tab}[{off}];")
        pm = {0: {(0, 1): 1, (1, 0): 1}, 1: {(0, 1): -1j, (1, 0): 1j}}        # X / Y entries [x, 1 - x]
        bases = [r for r in range(self.NR) if not (r >> ja) & 1 and not (r >> jb) & 1]
        p = sg.parts[0]
        imag = (ka != kb)
        first = True
        for b in bases:
            r00, r11 = b, b | (1 << ja) | (1 << jb)
            r01, r10 = b | (1 << jb), b | (1 << ja)            # (x_a, x_b) = (0, 1), (1, 0)
            for (x, y, xa, xb) in ((r00, r11, 0, 0), (r01, r10, 0, 1)):
                f = pm[ka][(xa, 1 - xa)] * pm[kb][(xb, 1 - xb)]        # the factor of destination x (source y)
                # contribution of the partner products x <-> y to Re <psi|P|psi>: 2 Re(f conj(a_x) a_y)
                if not imag:
                    term = f"{self.A(x, 'a')} * {self.A(y, 'a')}"            # Re(conj(x) y) = sum of the two components
                    sgn = "" if f.real > 0 else "-"
                else:
                    term = f"{self.A(x, 'a')} * {self.A(y, 'a')}.yx"         # Im(conj(x) y) = component x minus component y
                    sgn = "" if (f * 1j).real > 0 else "-"                 # f = -i: +Im, f = +i: -Im
                if first:
                    p.append(f"  v2f {e} = {sgn}({term});")
                    first = False
                elif sgn:
                    p.append(f"  {e} = __builtin_elementwise_fma(-{self.A(x, 'a')}, {term.split(' * ')[1]}, {e});")
                else:
                    p.append(f"  {e} = __builtin_elementwise_fma({self.A(x, 'a')}, {term.split(' * ')[1]}, {e});")
        p = sg.new_part()
        for b in bases:
            for xa in (0, 1):
                for xb in (0, 1):
                    dst = b | (xa << ja) | (xb << jb)
                    src = dst ^ (1 << ja) ^ (1 << jb)
                    f = pm[ka][(xa, 1 - xa)] * pm[kb][(xb, 1 - xb)]
                    if f == 1:
                        cf, av = f"v2f{{{c}, {c}}}", self.A(src, 'a')
                    elif f == -1:
                        cf, av = f"v2f{{-{c}, -{c}}}", self.A(src, 'a')
                    elif f == -1j:       # -i (x + i y) = y - i x
                        cf, av = f"v2f{{{c}, -{c}}}", self.A(src, 'a') + ".yx"
                    else:                # +i (x + i y) = -y + i x
                        cf, av = f"v2f{{-{c}, {c}}}", self.A(src, 'a') + ".yx"
                    p.append(f"  {self.A(dst, 'l')} = __builtin_elementwise_fma({cf}, {av}, {self.A(dst, 'l')});")
        self.event(sg, f"{c} * ({e}.x {'-' if imag else '+'} {e}.y)", gslot)
        return q + 4

    def dfold(self, q: int) -> int:
        """{OP_DFOLD, nterms, gslot, (thread-side Z mask, register mask, cslot) * nterms}: lambda[r] += D[r] psi[r] with
        D = sum_t c_t (-1)^{parity(index & zmask_t)} -- the Z-only strings of the cotangent 2 sum_t w_t P_t |psi> (c_t = 2 w_t),
        born in registers -- and their energy 1/2 sum_r D[r] |psi[r]|^2 as one gradient event.  Per thread: one signed
        coefficient per term, summed per register mask; D[r] = sum over the masks in use of +-that sum (plain float code)."""
        w, NR = self.w, self.NR
        nt, gslot = int(w[q + 1]), _i32(w[q + 2])
        sg = self.seg("diagonal strings of the cotangent")
        p = sg.parts[0]
        t = self.fresh("df")
        p.append(f"  const uint32_t {t}i = wg_base | {self.tphys};")
        by_mask: Dict[int, List[str]] = {}
        for e in range(nt):
            zm, rm, cslot = _u32(w[q + 3 + 3 * e]), int(w[q + 4 + 3 * e]), int(w[q + 5 + 3 * e])
            tab, off = self.slot_ptr(cslot)
            c = f"{t}c{e}"
            sg.loads.append(f"  const float {c} = {tab}[{off}];")
            if zm:
                p.append(f"  const float {t}s{e} = (__builtin_popcount({t}i & {zm:#x}u) & 1) ? -{c} : {c};")
                by_mask.setdefault(rm, []).append(f"{t}s{e}")
            else:
                by_mask.setdefault(rm, []).append(c)
        for rm, names in by_mask.items():
            p.append(f"  const float {t}m{rm} = " + " + ".join(names) + ";")
        p = sg.new_part()
        acc = f"{t}e"
        p.append(f"  v2f {acc} = v2f{{0.f, 0.f}};")
        for r in range(NR):
            terms = [("-" if bin(r & rm).count("1") & 1 else "+") + f" {t}m{rm}" for rm in by_mask]
            expr = " ".join(terms)
            expr = expr[2:] if expr.startswith("+ ") else expr
            p.append(f"  {{ const float d_ = {expr}; const v2f dd_ = v2f{{d_, d_}}; "
                     f"{acc} = __builtin_elementwise_fma(dd_, {self.A(r, 'a')} * {self.A(r, 'a')}, {acc}); "
                     f"{self.A(r, 'l')} = __builtin_elementwise_fma(dd_, {self.A(r, 'a')}, {self.A(r, 'l')}); }}")
        self.event(sg, f"0.5f * ({acc}.x + {acc}.y)", gslot)
        return q + 3 + 3 * nt

    def op(self, q: int) -> int:
        op = int(self.w[q])
        if op == P.OP_G1M:
            return self.g1m(q)
        if op == P.OP_DIAGF:
            return self.diagf(q)
        if op == P.OP_XFOLD:
            return self.xfold(q)
        if op == P.OP_DFOLD:
            return self.dfold(q)
        if op == P.OP_XFOLD2:
            return self.xfold2(q)
        raise Unsupported(f"backward op {op}")

    def lds_bytes(self) -> int:
        return max(8 << self.T, 4 * 64 * (1 << (self.LT - 6)) * max(1, (len(self.events) + 63) // 64))

    def source(self, kname: str) -> str:
        NR = self.NR
        params = ("v2f* __restrict__ psi, v2f* __restrict__ lam, long long state_stride, const float* __restrict__ ctab_g, "
                  "const float* __restrict__ ptab_g, long long ptab_stride, double* __restrict__ gout, "
                  "long long gout_stride, int gcopies, uint32_t live_mask, long long gcopy_stride")
        pro = ["  extern __shared__ __attribute__((aligned(16))) char lb_[];",
               "  char LDS_AS* const lb = (char LDS_AS*)lb_;",
               "  psi += (long long)blockIdx.y * state_stride;",
               "  lam += (long long)blockIdx.y * state_stride;",
               "  gout += (long long)blockIdx.y * gout_stride + (long long)(blockIdx.x % (unsigned)gcopies) * gcopy_stride;",
               "  const KF ctab = (KF)ctab_g; (void)ctab; (void)live_mask;",
               "  const KF ptab = (KF)(ptab_g + (long long)blockIdx.y * ptab_stride);"]
        pre, loop = self.index_lines()
        pro += pre
        pro.append("  const uint32_t lane = tid & 63u;")
        pro.append("  //GACC")
        pro += loop
        pro.append("  v2f " + ", ".join(f"a{r}" for r in range(NR)) + ";")
        pro.append("  v2f " + ", ".join(f"l{r}" for r in range(NR)) + ";")
        pro.append("  uint32_t sgn = 0u;")
        rd0 = self.rounds[0]
        sg = self.seg("tile load")
        self.tphys = self.fresh("tph")
        if int(self.opts.get("prio", 0)) & 1:
            sg.parts[0].append("  __builtin_amdgcn_s_setprio(3);")
        self.thread_xor(sg.parts[0], self.tphys, rd0.thr_phys)
        if self.flags & P.FLAG_LAMBDA_ZERO:
            # every term of the cotangent is born in this pass (OP_XFOLD / OP_DFOLD): lambda is not read
            self.tile_io(sg.parts[0], rd0, False, {"a": "psi"}, self.tphys, vecs=["a"])
            for r in range(NR):
                sg.parts[0].append(f"  {self.A(r, 'l')} = v2f{{0.f, 0.f}};")
        else:
            self.tile_io(sg.parts[0], rd0, False, {"a": "psi", "l": "lam"}, self.tphys)
        if int(self.opts.get("prio", 0)) & 1:
            sg.parts[0].append("  __builtin_amdgcn_s_setprio(0);")
        for k, rd in enumerate(self.rounds):
            q = rd.ops_at
            for _ in range(rd.nops):
                q = self.op(q)
            if q != rd.ops_at + rd.nwords:
                raise Unsupported("descriptor length mismatch")
            if k == self.nrounds - 1:
                break
            self.exchange(k, [("a", ""), ("l", "")], 8)
        sg = self.seg("gradient sums")
        self.reduce_pending(sg.parts[0])
        nev = len(self.events)
        nacc = (nev + 63) // 64
        NW = 1 << (self.LT - 6)
        p = post = []       # after the tile loop: the accumulators hold the sums over all of this workgroup's tiles
        if nev:
            # the waves' accumulators meet in LDS (the exchange buffer is free now): one f64 atomic per event
            p.append("  __syncthreads();")
            p.append("  { float LDS_AS* const gl = (float LDS_AS*)lb;")
            for k in range(nacc):
                p.append(f"    gl[({k * NW}u + (tid >> 6)) * 64u + lane] = gacc{k};")
            p.append("    __syncthreads();")
            p.append(f"    for (uint32_t e = tid; e < {nev}u; e += {1 << self.LT}u) {{")
            p.append("      float s = 0.f;")
            p.append(f"      for (uint32_t w_ = 0; w_ < {NW}u; ++w_) s += gl[((e >> 6) * {NW}u + w_) * 64u + (e & 63u)];")
            p.append("      const int sl = tcmi_spec_slots[e];")
            p.append("      if (sl >= 0) atomicAdd(gout + sl, (double)s);")
            p.append("    }")
            p.append("  }")
        if not self.nostore:
            sg = self.seg("sign + tile store")
            p = sg.parts[0]
            for vec in ("a", "l"):
                for r in range(0, NR, 16):
                    p.append("  vm2_negate16_if(" + ", ".join(self.A(r + i, vec) for i in range(16)) + ", sgn);")
            self.tile_io(p, self.rounds[-1], True, {"a": "psi", "l": "lam"}, self.tphys)
        body = self.linear()
        i = pro.index("  //GACC")
        pro[i:i + 1] = [f"  float gacc{k} = 0.f;" for k in range(nacc)]
        body = body + ["  }"] + post
        head = self.header(kname, params)
        if nev:
            # before the kernel: the events' gradient slots
            i = head.index("")
            head[i:i] = [f"__constant__ int tcmi_spec_slots[{nev}] = {{{', '.join(str(s) for s in self.events)}}};"]
        return "\n".join(head + pro + body + ["}"]) + "\n"


def adjoint_source(words, kname: str = "tcmi_spec_pass", opts=None) -> Tuple[str, dict]:
    """HIP source of the straight-line kernel of one reverse-sweep pass + its launch geometry."""
    e = _Adjoint(words, opts)
    src = e.source(kname)
    return src, {"kind": "adjoint", "n": e.n, "T": e.T, "LT": e.LT, "lds": e.lds_bytes(), "events": len(e.events)}


_EMITTERS["adjoint"] = adjoint_source
