#!/bin/bash
# fc2-shaped launches with the nt cache policy on their A pieces (MERV_GEMM_A_NT=1) against the default, one library, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do for v in 0 1; do
  echo "== rep $rep MERV_GEMM_A_NT=$v"
  MERV_GEMM_A_NT=$v python3 tools/gemm_bench.py 16 0 2>/dev/null | grep "fc2"
  MERV_GEMM_A_NT=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('ms_per_step', d['ms_per_step'], 'gemm frac', r['frac'], 'gemm_ms', r['gemm_ms_per_step'], ' | '.join('%s %.2f' % (k['name'][:24], k['ms_per_step']) for k in r['by_kernel'][:3]))
"
done; done
