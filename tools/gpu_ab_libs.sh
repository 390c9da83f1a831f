#!/bin/bash
# same-box A/B of two builds of the product library: tools/gpu_ab_libs.sh libA.so libB.so
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/ab
: > gpurun_out/ab/ab_libs.txt
for rep in 1 2 3; do
for lib in "$@"; do
  RD_HIP_LIB=$R/$lib timeout 300 python bench.py --steps 8 --resident-only --inline-refine --no-cpu-baseline --no-alt --no-encoder --traffic off 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%-44s %.2f M reads/s  launch %.3f ms  step %.3f ms' % ('$lib', j['value']/1e6, j['roofline']['avg_launch_ms'], j['ms_per_step']))" | tee -a gpurun_out/ab/ab_libs.txt
done
done
