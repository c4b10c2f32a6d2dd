#!/usr/bin/env python3
"""timeline_chains.py <kernel_trace.csv> [steps]: per hardware queue of a rocprofv3 --kernel-trace of `bench.py` (concurrent streams), over the
last 60 % of the trace: kernels, summed duration, idle time between consecutive kernels of the queue (dependent launches: the chain's gaps),
and the kernel names that make up the queue's time (count, total ms, mean us) -- i.e. what each encoder's chain is made of and where it waits."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = []
for r in rows:
    n = r["Kernel_Name"]
    if "merv::" not in n:
        continue
    q = r.get("Queue_Id", "?") + "/" + r.get("Stream_Id", "?")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, q, int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 1) or 1)))
ev.sort()
t0, t1 = ev[0][0], ev[-1][1]
cut = t0 + (t1 - t0) * 4 // 10
ev = [e for e in ev if e[0] >= cut]
t0, t1 = ev[0][0], max(e[1] for e in ev)
span = t1 - t0
print("span %.2f ms, %d kernels%s" % (span / 1e6, len(ev), (", %.3f ms per step" % (span / 1e6 / steps)) if steps else ""))


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("merv::(anonymous namespace)::", "").replace("merv::", "")
    return n[:110]


byq = collections.defaultdict(list)
for e in ev:
    byq[e[3]].append(e)
for q, es in sorted(byq.items(), key=lambda kv: -sum(e[1] - e[0] for e in kv[1])):
    es.sort()
    busy = sum(e[1] - e[0] for e in es)
    gaps = [max(0, b[0] - a[1]) for a, b in zip(es, es[1:])]
    small = sum(g for g in gaps if g < 50000)
    print("\nqueue/stream %s: %d kernels, busy %.2f ms (%.1f %% of span), gaps < 50 us between consecutive kernels: %.2f ms (mean %.2f us over %d)"
          % (q, len(es), busy / 1e6, 100.0 * busy / span, small / 1e6, small / 1e3 / max(1, sum(1 for g in gaps if g < 50000)), sum(1 for g in gaps if g < 50000)))
    agg = collections.defaultdict(lambda: [0, 0, 0])
    for s, e, n, _, grid, wg in es:
        a = agg[(short(n), grid // max(wg, 1))]
        a[0] += 1; a[1] += e - s
    for (n, blocks), (c, tot, _) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %6.2f ms  %5d x %7.1f us  blocks %5d  %s" % (tot / 1e6, c, tot / 1e3 / c, blocks, n))
