#!/bin/bash
# whole-call rate of the CLI gz -> gz against the size of the stream decoder's first batch (RD_GZS_FIRST): tools/gzs_first_ab.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/ab
OUT=gpurun_out/ab/gzs_first.txt
: > $OUT
for rep in 1; do
for first in 6291456 25165824 67108864; do
  for mode in "" "--single-end"; do
    python tools/e2e_bench.py --bench-legs --legs gz_to_gz,seqlike_gz_to_gz --env RD_GZS_FIRST=$first $mode 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
for k in ('gz_to_gz','seqlike_gz_to_gz'):
    v=j[k]; print('first %9d %-12s %-18s %.2f M reads/s  steady %.2f M  cores %.2f' % ($first, '${mode:-paired}', k, v['reads_per_s']/1e6, (v['reads_per_s_after_first_chunk'] or 0)/1e6, v['host_cores_busy']))" | tee -a $OUT
  done
done
done
