#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s6
mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -15) > $O/pytest.log
cat $O/pytest.log
cp gpurun_out/parity_report.json $O/ 2>/dev/null
(RD_HIP_LIB=$R/ribodetector_amd/csrc/librd_hip_diag.so timeout 900 python tools/acc_experiment.py --reads 1048576 --oracle-reads 0 --variants t32_acc0,t32_acc16_ops24,t32_acc32_creg,t32_acc48_ops24_creg,t32_acc0,t32_acc48_ops24_creg) > $O/perf_variants.json 2> $O/perf.err
python - <<PY
import json
j=json.load(open("$O/perf_variants.json"))
for k,v in j["variants"].items():
    a=v["vs_f64"]; print("%-26s ms %.3f rms %.3g p9999 %.3g max %.3g >5e-5 %d lab %d" % (k, v["ms"], a["rms"], a["p9999"], a["max"], a["n_over_5e-5"], v["labels_vs_f64"]["mismatches"]))
PY
for v in t32_acc0 t32_acc16_ops24 t32_acc48_ops24_creg t32_acc0 t32_acc48_ops24_creg; do
  RD_HIP_LIB=$R/ribodetector_amd/csrc/librd_hip_diag.so timeout 300 python bench.py --steps 8 --variant $v --resident-only --no-cpu-baseline --no-alt --no-encoder --traffic off 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v', j['value'], j['roofline']['avg_launch_ms'], j['ms_per_step'])"
done | tee $O/bench_variants.txt
(timeout 300 python bench.py --steps 5 --no-cpu-baseline --no-alt --no-encoder --traffic off 2>/dev/null | grep "^{") > $O/bench_refine.json
python - <<PY
import json
j=json.load(open("$O/bench_refine.json")); print("product+refine", j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["config"]["kernel_only_reads_per_s"])
PY
