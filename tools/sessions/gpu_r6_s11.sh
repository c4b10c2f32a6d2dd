#!/bin/bash
# Round 6, session 11: decode step with W_o prefetched into the o-projection's L2s by the split-attention launch (A/B), decode tests.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s11
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 600 python3 tools/probes/decode_prefetch_probe.py > $OUT/decode_prefetch.txt 2> $OUT/decode_prefetch.err; tail -1 $OUT/decode_prefetch.txt > $OUT/decode_prefetch.json; cat $OUT/decode_prefetch.txt; tail -3 $OUT/decode_prefetch.err
timeout 900 python3 -m pytest tests/test_decode_gpu.py tests/test_generate_gpu.py -q -x > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dec -- python3 $R/tools/probes/decode_prefetch_probe.py > /dev/null 2> $OUT/trace_dec.err
cp $(find $OUT/trace_dec -name "*kernel_stats.csv" | head -1) $OUT/decode_kernel_stats.csv; find $OUT -name "*kernel_trace.csv" -delete
grep -E "decode_attn_split|oproj_merge|gemv" $OUT/decode_kernel_stats.csv | cut -c1-200 | head -12
