#!/bin/bash
# refine_ab.sh - same-box A/B of where the float64 pass runs in bench.py's timed region: the deferred pass (round 4: candidates
# recorded by the recurrence epilogue, rd_sync_results on the post stream) against the round-3 form (rd_refine, a scan of all logits,
# on the side stream). Three interleaved pairs, 20 steps each.   tools/refine_ab.sh > profiles/r04_refine_ab.txt
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
F="--steps 20 --warmup 3 --no-cpu-baseline --no-alt --no-encoder --no-e2e --traffic off"
for i in 1 2 3; do
  for mode in deferred deferred_normal_prio scan; do
    X=""; [ $mode = scan ] && X="--refine-scan"
    P=high; [ $mode = deferred_normal_prio ] && P=normal
    RD_REFINE_STREAM_PRIORITY=$P python bench.py $F $X 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$mode', 'run $i', 'reads/s %.0f' % j['value'], 'ms_per_step %.3f' % j['ms_per_step'], 'launch_ms %.3f' % j['roofline']['avg_launch_ms'], 'kernel_only %.0f' % j['config']['kernel_only_reads_per_s'])"
  done
done
