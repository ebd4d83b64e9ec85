#!/usr/bin/env python3
"""Instruction mix per basic block of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only).
usage: isa_mix.py listing.s name-substring [min block size]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2]
min_size = int(sys.argv[3]) if len(sys.argv) > 3 else 100
for m in re.finditer(r"^(_Z\w+):", s, re.M):
    if want not in m.group(1):
        continue
    body = s[m.end():s.index("s_endpgm", m.end())]
    parts = re.split(r"\n(\.LBB\d+_\d+):", body)
    print(m.group(1))
    for name, blk in zip(["entry"] + parts[1::2], [parts[0]] + parts[2::2]):
        ins = [l.split()[0] for l in blk.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
        kinds, valu = collections.Counter(), collections.Counter()
        for i in ins:
            if i.startswith("v_mfma"):
                kinds["mfma"] += 1
            elif i.startswith("v_"):
                kinds["valu"] += 1
                valu[i] += 1
            elif i.startswith("ds_"):
                kinds["lds"] += 1
            elif i.startswith(("global_", "buffer_", "scratch_")):
                kinds["vmem"] += 1
            elif i.startswith("s_waitcnt"):
                kinds["wait"] += 1
            elif i.startswith("s_barrier"):
                kinds["barrier"] += 1
            elif i.startswith("s_nop"):
                kinds["nop"] += 1
            else:
                kinds["salu"] += 1
        if len(ins) >= min_size:
            print(" ", name, len(ins), dict(kinds))
            print("     ", valu.most_common(30))
