#!/usr/bin/env python3
"""Per-kernel instruction statistics of a hipcc -S listing: totals, register-copy counts and the basic blocks
that hold whole-array copies.  Usage: isa_stats.py file.s [kernel-name-substring]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\n(_Z\w+):[^\n]*\n", s):
    name = m.group(1)
    if pat not in name:
        continue
    end = s.find("s_endpgm", m.end())
    if end < 0:
        continue
    body = s[m.end():end]
    ins = [l.strip() for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    c = Counter(x.split()[0] for x in ins)
    print(name)
    print("   instructions", len(ins), " v_mov_b64", c["v_mov_b64_e32"], " v_mov_b32", c["v_mov_b32_e32"],
          " v_pk_*", sum(v for k, v in c.items() if k.startswith("v_pk")),
          " other VALU", sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith(("v_pk", "v_mov"))),
          " branches", sum(v for k, v in c.items() if "branch" in k),
          " scratch", sum(v for k, v in c.items() if k.startswith("scratch")))
    blocks = re.split(r"\n(\.LBB\d+_\d+):", body)
    for j in range(1, len(blocks), 2):
        b = blocks[j + 1]
        mv = b.count("v_mov_b64") + b.count("v_mov_b32")
        if mv > 8:
            print("      block", blocks[j], "moves", mv, "v_pk", b.count("v_pk_"))
