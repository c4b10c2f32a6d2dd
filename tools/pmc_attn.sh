#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_attn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -- python3 $R/tools/attn_bench.py 8 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/b -- python3 $R/tools/attn_bench.py 8 > $OUT/b.log 2>&1
tail -3 $OUT/b.log
for d in a b; do
f=$(find $OUT/$d -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r['Kernel_Name']
    if 'attn_kernel' not in k: continue
    key = (k[40:90], r['Grid_Size'])
    agg[key][r['Counter_Name']] += float(r['Counter_Value']); n[key].add(r['Dispatch_Id'])
for key, d in agg.items():
    nd = len(n[key])
    print(key, 'dispatches', nd, {c: f'{v/nd:.4g}' for c, v in sorted(d.items())})
PY
done
