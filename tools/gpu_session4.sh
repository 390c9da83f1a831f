#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s4
mkdir -p $O
cd $R
(timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -25) > $O/pytest.log
cat $O/pytest.log
cp gpurun_out/parity_report.json $O/ 2>/dev/null
(timeout 300 python bench.py --steps 5 --no-cpu-baseline --no-alt --no-encoder --traffic off 2>/dev/null | grep "^{") > $O/bench_refine.json
python - <<PY
import json
j=json.load(open("$O/bench_refine.json")); print(j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["config"]["kernel_only_reads_per_s"])
PY
