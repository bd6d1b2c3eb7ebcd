#!/usr/bin/env python3
"""Instruction mix of the basic blocks of one kernel in a `hipcc -S --cuda-device-only` listing.
usage: asm_loops.py file.s <kernel-name-substring> [min-instructions]"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
label, block = "entry", []
def flush():
    ins = [l.split()[0] for l in block]
    if len(ins) >= minins:
        c = Counter(ins)
        print("%s: %d instr; %s" % (label, len(ins), ", ".join("%s %d" % kv for kv in c.most_common(14))))
for l in lines[start + 1:end]:
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        flush(); label, block = m.group(1), []
    elif t and not t.startswith((".", ";", "//")):
        block.append(t)
flush()
