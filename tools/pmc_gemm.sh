#!/bin/bash
# PMC counters for the GEMM micro-benchmark (own run, no tracing domains besides kernel-trace).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
V=${1:-3}
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $R/tools/gemm_bench.py 8 $V > $OUT/a.log 2>&1
tail -3 $OUT/a.log
f=$(find $OUT/a -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name']
    if 'gemm' not in k: continue
    key = (k[-40:], r['Grid_Size'])
    agg[key][r['Counter_Name']] += float(r['Counter_Value'])
for key, d in agg.items():
    print(key)
    for c, v in sorted(d.items()): print('   ', c, f'{v:.4g}')
    if d.get('SQ_LDS_IDX_ACTIVE'): print('    conflict frac', d['SQ_LDS_BANK_CONFLICT']/d['SQ_LDS_IDX_ACTIVE'])
PY
