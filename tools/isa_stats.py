"""Instruction-class census of a kernel in a hipcc -S listing. Usage: python tools/isa_stats.py file.s kernel_substring"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
c, ops = collections.Counter(), collections.Counter()
loop_depth = 0
for l in lines[start + 1:end]:
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m:
        continue
    op = m.group(1)
    cls = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else
           "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
    c[cls] += 1
    ops[op] += 1
print(dict(c))
print(ops.most_common(30))
