#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s12
mkdir -p $O
cd $R
T0=$(date +%s)
(timeout 2400 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -20) > $O/r02_gpu_tests.txt
echo "gpu test suite wall seconds: $(( $(date +%s) - T0 ))" >> $O/r02_gpu_tests.txt
cat $O/r02_gpu_tests.txt
cp gpurun_out/parity_report.json $O/r02_parity_report.json 2>/dev/null
(timeout 600 python tools/outlier_probe.py) > $O/r02_outlier.json 2> $O/err.txt
bash tools/profile_round.sh r02 > $O/profile.log 2>&1
cp $R/gpurun_out/prof_r02/* $O/ 2>/dev/null
tail -22 $O/r02_summary.txt
