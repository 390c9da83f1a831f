#!/bin/bash
# CLI gz -> gz with the stream decoder's kernels at high (-1, the default) or normal (0) stream priority: tools/gzs_prio_ab.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/ab
OUT=gpurun_out/ab/gzs_prio.txt
: > $OUT
for prio in -1 0 -1 0; do
  for mode in "" "--single-end"; do
    python tools/e2e_bench.py --bench-legs --legs gz_to_gz,seqlike_gz_to_gz --env RD_GZS_PRIORITY=$prio $mode 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
for k in ('gz_to_gz','seqlike_gz_to_gz'):
    v=j[k]; print('priority %2d %-12s %-18s %.2f M reads/s  steady %.2f M  cores %.2f' % ($prio, '${mode:-paired}', k, v['reads_per_s']/1e6, (v['reads_per_s_after_first_chunk'] or 0)/1e6, v['host_cores_busy']))" | tee -a $OUT
  done
done
