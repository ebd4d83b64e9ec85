#!/usr/bin/env python3
"""Issue order of a kernel's large basic blocks as one letter per instruction (M matrix, v vector, r/w LDS read/write,
G/S global load/store, | waitcnt, B barrier, n nop, s scalar) — shows whether matrix and vector work interleave.
usage: isa_pattern.py listing.s name-substring [min block size]"""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2]
min_size = int(sys.argv[3]) if len(sys.argv) > 3 else 150
for m in re.finditer(r"^(_Z\w+):", s, re.M):
    if want not in m.group(1):
        continue
    body = s[m.end():s.index("s_endpgm", m.end())]
    parts = re.split(r"\n(\.LBB\d+_\d+):", body)
    print(m.group(1))
    for name, blk in zip(["entry"] + parts[1::2], [parts[0]] + parts[2::2]):
        ins = [l.split()[0] for l in blk.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
        if len(ins) < min_size:
            continue
        seq = ""
        for i in ins:
            seq += ("M" if i.startswith("v_mfma") else "v" if i.startswith("v_") else "r" if i.startswith("ds_read")
                    else "w" if i.startswith("ds_") else "G" if i.startswith("global_load") else "S"
                    if i.startswith("global_") else "|" if i.startswith("s_waitcnt") else "B"
                    if i.startswith("s_barrier") else "n" if i.startswith("s_nop") else "s")
        print(" ", name, len(ins))
        for k in range(0, len(seq), 120):
            print("     ", seq[k:k + 120])
