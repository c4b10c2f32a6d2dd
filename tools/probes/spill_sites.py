#!/usr/bin/env python3
"""spill_sites.py <file.s> [substring]: for every kernel whose mangled name contains the substring, list its scratch
(spill) instructions with the number of MFMAs that precede them -- i.e. whether a spill sits in the K-loop, where a scratch
access joins the vmcnt queue the LDS-DMA waits are counted on, or only around the epilogue."""
import sys, re
s = open(sys.argv[1]).read()
sub = sys.argv[2] if len(sys.argv) > 2 else "8phase"
for m in re.finditer(r"^(_Z\w+):[^\n]*\n", s, re.M):
    name = m.group(1)
    if sub not in name: continue
    j = s.index(".Lfunc_end", m.end())
    body = s[m.end():j].split("\n")
    nm = 0; rows = []
    for l in body:
        if "v_mfma" in l: nm += 1
        if "scratch_" in l: rows.append((nm, l.strip().split(";")[0]))
    total = nm
    print(name[-40:], "mfma", total, "spill ops", len(rows))
    for nmf, l in rows: print("   after mfma %4d: %s" % (nmf, l))
