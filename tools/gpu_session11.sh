#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s11
mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "refine or labels_equal" 2>&1 | tail -6) | tee $O/pytest.log
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_s11 -o b -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt --no-encoder --traffic off --inline-refine) > $O/trace.log 2>&1
find /tmp/p_s11 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
grep "rd_refine\|rd_lstm" $O/kernel_stats.csv | cut -c1-60,200-400
rm -f $O/trace.log
