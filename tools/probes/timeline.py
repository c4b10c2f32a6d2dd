#!/usr/bin/env python3
"""Reads a rocprofv3 kernel_trace.csv of `bench.py` (concurrent encoder streams) and reports, for the timed steps: wall span,
time with no kernel running, time with exactly one kernel running, and the sum of kernel durations per class."""
import csv
import sys
import collections

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    if "merv::" not in n:
        continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
ev.sort()
# keep the last 60 % of the trace (steady state: skip set-up and warm-up)
t0, t1 = ev[0][0], ev[-1][1]
cut = t0 + (t1 - t0) * 4 // 10
ev = [e for e in ev if e[0] >= cut]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, _ in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = collections.Counter()
cur, last = 0, t0
for t, d in pts:
    busy[min(cur, 4)] += t - last
    last = t
    cur += d
span = t1 - t0
print("span %.2f ms over %d kernels" % (span / 1e6, len(ev)))
for k in sorted(busy):
    print("  %s kernel(s) running: %5.1f %%" % (k if k < 4 else ">=4", 100.0 * busy[k] / span))
cls = collections.Counter()
for s, e, n in ev:
    key = "gemm 8-phase" if "8phase" in n else "gemm other" if "gemm_bf16" in n else "attention" if "attn_kernel" in n and "temporal" not in n else "temporal attn" if "temporal" in n else "layernorm/stats" if ("layernorm" in n or "stats" in n) else "other"
    cls[key] += e - s
tot = sum(cls.values())
print("sum of kernel durations %.2f ms = %.2f x the span" % (tot / 1e6, tot / span))
for k, v in cls.most_common():
    print("  %-16s %6.2f ms  %4.1f %% of the span" % (k, v / 1e6, 100.0 * v / span))

# ---- where the idle time is: gaps (no kernel running) by length, and the kernels around the longest ones
gaps = []
cur, last_end = 0, None
active = 0
for t, d in pts:
    if active == 0 and last_end is not None and t > last_end:
        gaps.append((t - last_end, last_end, t))
    active += d
    if active == 0:
        last_end = t
buckets = collections.Counter()
for g, _, _ in gaps:
    b = "<2us" if g < 2000 else "2-5us" if g < 5000 else "5-20us" if g < 20000 else "20-200us" if g < 200000 else ">200us"
    buckets[b] += g
print("idle time by gap length:", {k: "%.2f ms" % (v / 1e6) for k, v in buckets.items()}, "gaps:", len(gaps))
ends = {e: n for s, e, n in ev}
starts = {s: n for s, e, n in ev}
for g, a, b in sorted(gaps, reverse=True)[:8]:
    print("  gap %.1f us after %s before %s" % (g / 1e3, ends.get(a, "?")[30:80], starts.get(b, "?")[30:80]))

# ---- per class: how long its kernels are the ONLY thing running (a launch on a few CUs then idles the rest of the chip)
def klass(n):
    return ("gemm 8-phase" if "8phase" in n else "gemm other" if "gemm_bf16" in n else "attention" if "attn_kernel" in n and "temporal" not in n
            else "temporal attn" if "temporal" in n else "layernorm/stats" if ("layernorm" in n or "stats" in n) else "other")
pts2 = []
for i, (s, e, n) in enumerate(ev):
    pts2.append((s, 1, i)); pts2.append((e, -1, i))
pts2.sort()
running, last = set(), t0
alone = collections.Counter()
for t, d, i in pts2:
    if len(running) == 1:
        alone[klass(ev[next(iter(running))][2])] += t - last
    last = t
    if d == 1:
        running.add(i)
    else:
        running.discard(i)
print("time a class runs ALONE on the chip:", {k: "%.2f ms (%.1f %% of the span)" % (v / 1e6, 100.0 * v / span) for k, v in alone.most_common()})
small = [(e - s) for s, e, n in ev if klass(n) == "gemm other"]
print("gemm other: %d launches, mean %.1f us" % (len(small), sum(small) / len(small) / 1e3))
