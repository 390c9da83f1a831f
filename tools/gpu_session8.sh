#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s8
mkdir -p $O
cd $R
(timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $O/pytest.log
cat $O/pytest.log
cp gpurun_out/parity_report.json $O/ 2>/dev/null
T0=$(date +%s.%N)
timeout 900 python bench.py 2> $O/bench.err | grep "^{" > $O/bench.json
echo "default bench wall: $(echo "$(date +%s.%N) - $T0" | bc) s" | tee $O/bench.time
python - <<PY
import json
j=json.load(open("$O/bench.json")); print("default bench", j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["config"]["kernel_only_reads_per_s"], j["config"]["device_path_over_kernel_only"], j["roofline"]["traffic"])
print(j.get("cpu_baseline",{}).get("value"), j.get("cpu_baseline",{}).get("encode_plus_model_reads_per_s"), j.get("parity_sample"))
PY
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_s8 -o b -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt --no-encoder --traffic off) > $O/trace.log 2>&1
find /tmp/p_s8 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
head -12 $O/kernel_stats.csv
rm -f $O/trace.log
