#!/usr/bin/env python3
"""Static instruction counts between the GMARK comments of a tick_group_kernel instantiation in hipcc's -S output
(sai2b_group_tick.hpp). Usage: count_group_phases.py file.s [G] — counts every instruction textually between two
consecutive marks, so a phase with branches shows the sum over its paths."""
import re
import sys

path, G = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "16")
txt = open(path).read()
m = re.search(r"^_ZN5sai2b17tick_group_kernelILi%sELb0E.*?s_endpgm" % G, txt, re.S | re.M)
body = m.group(0).splitlines()
cur, counts, order = "start", {}, []
for ln in body:
    ln = ln.strip()
    mm = re.match(r"; GMARK (\w+)", ln)
    if mm:
        cur = mm.group(1)
        if cur not in counts:
            counts[cur] = [0, 0, 0]
            order.append(cur)
        continue
    if not ln or ln.startswith((";", ".", "//")) or ln.endswith(":"):
        continue
    c = counts.setdefault(cur, [0, 0, 0])
    if cur not in order:
        order.append(cur)
    c[0] += 1
    c[1] += "dpp" in ln and "v_fmac" in ln
    c[2] += ln.startswith(("scratch_", "v_accvgpr"))
print(f"{'after mark':<20}{'instrs':>8}{'dpp fma':>9}{'spill mv':>9}")
for k in order:
    print(f"{k:<20}{counts[k][0]:>8}{counts[k][1]:>9}{counts[k][2]:>9}")
print("total", sum(v[0] for v in counts.values()))
