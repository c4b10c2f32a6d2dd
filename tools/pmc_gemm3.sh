#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc3
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
V=${1:-4}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $R/tools/gemm_bench.py 8 $V > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --output-format csv -d $OUT/b -- python3 $R/tools/gemm_bench.py 8 $V > $OUT/b.log 2>&1
for d in a b; do
f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r['Kernel_Name']
    if 'gemm' not in k: continue
    key = (k[-30:], r['Grid_Size'])
    agg[key][r['Counter_Name']] += float(r['Counter_Value']); n[key].add(r['Dispatch_Id'])
for key, d in list(agg.items())[:3]:
    nd = len(n[key])
    print(key, 'dispatches', nd, {c: f'{v/nd:.4g}' for c, v in sorted(d.items())})
PY
done
