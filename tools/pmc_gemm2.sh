#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
V=${1:-3}
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/a -- python3 $R/tools/gemm_bench.py 8 $V > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/b -- python3 $R/tools/gemm_bench.py 8 $V > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- python3 $R/tools/gemm_bench.py 8 $V > $OUT/c.log 2>&1
tail -2 $OUT/a.log
for d in a b c; do
f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r['Kernel_Name']
    if 'gemm' not in k: continue
    key = (k[-42:], r['Grid_Size'])
    agg[key][r['Counter_Name']] += float(r['Counter_Value']); n[key].add(r['Dispatch_Id'])
for key, d in agg.items():
    nd = len(n[key])
    print(key, 'dispatches', nd, {c: f'{v/nd:.4g}' for c, v in d.items()})
PY
done
