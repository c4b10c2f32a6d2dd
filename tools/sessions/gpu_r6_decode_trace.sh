#!/bin/bash
# rocprofv3 kernel trace of bench.py's e2e leg on the final code: per-kernel durations of the decode step's launches (and the prefill's)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/decode_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-prof > $O/bench.json 2> $O/trace.err
echo "rc $?"; tail -c 400 $O/bench.json
f=$(find $O/trace -name "*kernel_stats.csv" | head -1); echo "stats: $f"
python3 - "$f" $O/decode_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
hdr, body = rows[0], rows[1:]
keep = [r for r in body if any(k in r[0] for k in ("gemv", "decode_attn", "oproj_merge", "greedy_advance", "sample_advance", "rmsnorm", "prefill_", "silu_mul", "rope_cache"))]
csv.writer(open(sys.argv[2], "w")).writerows([hdr] + keep)
for r in keep: print(r[0][:110], r[1], r[3])
PY
