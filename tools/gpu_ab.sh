#!/bin/bash
# same-box A/B of diagnostic kernel variants: tools/gpu_ab.sh v1 v2 ...   (needs librd_hip_diag.so: python __graft_entry__.py --diag)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab
mkdir -p $O
cd $R
export RD_HIP_LIB=$R/ribodetector_amd/csrc/librd_hip_diag.so
VARS=${@:-auto mfma_f32}
(timeout 600 python tools/acc_experiment.py --reads 262144 --oracle-reads 0 --variants $(echo $VARS | tr ' ' ',')) > $O/acc.json 2> $O/acc.err
python - <<PY
import json
j=json.load(open("$O/acc.json"))
for k,v in j["variants"].items():
    a=v["vs_f64"]; print("%-28s rms %.4g max %.4g lab %d" % (k, a["rms"], a["max"], v["labels_vs_f64"]["mismatches"]))
PY
: > $O/ab.txt
for rep in 1 2; do
for v in $VARS; do
  timeout 300 python bench.py --steps 8 --variant $v --resident-only --inline-refine --no-cpu-baseline --no-alt --no-encoder --traffic off 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%-28s %.2f M reads/s  launch %.3f ms  step %.3f ms' % ('$v', j['value']/1e6, j['roofline']['avg_launch_ms'], j['ms_per_step']))" | tee -a $O/ab.txt
done
done
