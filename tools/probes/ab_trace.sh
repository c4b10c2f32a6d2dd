#!/bin/bash
export MERV_HIP_LIB_AB=1  # tolerant binding for a previous build (merv_amd/_lib.py)
# ab_trace.sh LIB...: rocprofv3 --kernel-trace --stats of `bench.py --sequential --no-prof` per library -> gpurun_out/ab_trace/<lib>.csv (+ top GEMM rows printed)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ab_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  export MERV_HIP_LIB=$R/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 $R/bench.py --sequential --steps 6 --warmup 2 --no-prof --no-cpu-baseline --no-e2e > $OUT/$name.json 2> $OUT/$name.err
  cp $(find $OUT/$name -name "*kernel_stats.csv" | head -1) $OUT/$name.csv
  rm -rf $OUT/$name
  echo "== $lib"; python3 - $OUT/$name.csv <<'PY'
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "merv::" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    n = re.sub(r"^void |merv::\(anonymous namespace\)::|merv::|\(merv::\w+\)$", "", r["Name"])
    print(f'{float(r["TotalDurationNs"])/8e6:8.3f} ms/step {int(r["Calls"])//8:4d} calls/step avg {float(r["AverageNs"])/1e3:8.1f} us  {n[:90]}')
PY
done
