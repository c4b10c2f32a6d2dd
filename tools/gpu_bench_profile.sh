#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel-trace stats of the same command.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python3 bench.py --steps ${STEPS:-5} --warmup 2 ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err; cat $OUT/bench.json
python3 bench.py --steps ${STEPS:-5} --warmup 2 --sequential --no-cpu-baseline > $OUT/bench_seq.json 2>> $OUT/bench.err
cat $OUT/bench_seq.json
python3 bench.py --steps ${STEPS:-5} --warmup 2 --no-prof --no-cpu-baseline > $OUT/bench_noprof.json 2>> $OUT/bench.err
cat $OUT/bench_noprof.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
tail -2 $OUT/prof.err; cat $OUT/prof_bench.json
find $OUT/prof -name "*kernel_stats*" | head; f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
# keep only the small summaries
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
