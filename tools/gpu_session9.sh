#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s9
mkdir -p $O
cd $R
T0=$(date +%s)
(timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -40) > $O/pytest.log
cat $O/pytest.log
echo "gpu test suite wall seconds: $(( $(date +%s) - T0 ))" | tee -a $O/pytest.log
cp gpurun_out/parity_report.json $O/ 2>/dev/null
for extra in "" "--inline-refine"; do
  timeout 600 python bench.py --no-cpu-baseline --no-alt --no-encoder --traffic off $extra 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('bench [$extra]', j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'], j['config']['kernel_only_reads_per_s'], j['config']['device_path_over_kernel_only'])" | tee -a $O/bench_ab.txt
done
bash tools/profile_round.sh r02 > $O/profile.log 2>&1
tail -5 $O/profile.log
