#!/bin/bash
# Runs on the GPU box (via gpurun). Produces, under gpurun_out/round/:
#   bench.json            default bench.py line (concurrent encoder streams, roofline leg, parity + cpu baseline, e2e)
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats of `bench.py --sequential --no-prof --steps 6 --warmup 2` (8 identical
#                         steps; durations comparable with the event-timed roofline legs, which are also sequential)
#   kernel_roofline.json  tools/kernel_roofline.py: bench.json's roofline.by_kernel joined with that trace
#   pmc_gemm_traffic.json HBM bytes per GEMM launch from separate FETCH_SIZE / WRITE_SIZE passes (gfx950: FETCH_SIZE x2)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --sequential --steps 6 --warmup 2 --no-prof --no-cpu-baseline --no-e2e > $OUT/trace_bench.json 2> $OUT/trace.err
cat $OUT/trace_bench.json
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 $R/tools/kernel_roofline.py $OUT/bench.json $OUT/kernel_stats.csv 8 $OUT/kernel_roofline.json
bash $R/tools/pmc_gemm_round.sh > $OUT/pmc_gemm.log 2>&1; cp $R/gpurun_out/pmc_round/pmc_gemm.json $OUT/pmc_gemm.json
bash $R/tools/pmc_attn.sh > $OUT/pmc_attn.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --sequential --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-prof > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --sequential --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-prof > /dev/null 2> $OUT/pmc_write.err
python3 - $OUT <<'PY'
import csv, sys, json, glob, collections
out = sys.argv[1]
def collect(d, counter):
    f = glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True)[0]
    tot = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        k = "gemm" if "gemm_bf16" in r["Kernel_Name"] else ("attn" if "attn_kernel" in r["Kernel_Name"] and "temporal" not in r["Kernel_Name"] else None)
        if k is None: continue
        tot[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    return {k: (tot[k], len(n[k])) for k in tot}
fe, wr = collect("pmc_fetch", "FETCH_SIZE"), collect("pmc_write", "WRITE_SIZE")
res = {}
for k in fe:
    f_kb, nf = fe[k]; w_kb, nw = wr.get(k, (0.0, 1))
    res[k] = {"launches": nf, "fetch_size_kb_per_launch_raw": f_kb / nf, "write_size_kb_per_launch": w_kb / max(nw, 1),
              "hbm_bytes_per_launch": (2.0 * f_kb / nf + w_kb / max(nw, 1)) * 1024.0}
STEPS = 3  # 1 warm-up + 2 timed steps, no roofline leg (--no-prof)
g = res.get("gemm", {})
doc = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --sequential --steps 2 --warmup 1 "
               "--no-prof` (3 steps); gfx950: FETCH_SIZE counts 128-B requests at 64 B, so reads are doubled (MI355X_MICROARCH.md, "
               "HBM). One GEMM call of the library may be two kernel launches (eight-phase part + remaining rows): "
               "hbm_bytes_per_step sums all GEMM kernels of a step; bench.py divides it by the GEMM calls per step.",
       "steps_in_run": STEPS, "videos_per_step": 16, "hbm_bytes_per_launch": g.get("hbm_bytes_per_launch"),
       "hbm_bytes_per_step": (g.get("hbm_bytes_per_launch", 0.0) * g.get("launches", 0) / STEPS) if g else None, "kernels": res}
json.dump(doc, open(f"{out}/pmc_gemm_traffic.json", "w"), indent=1)
print(json.dumps(doc)[:600])
PY
head -12 $OUT/kernel_stats.csv | cut -c1-160
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +5M -delete
