#!/bin/bash
# Package power and shader clock (rocm-smi, 1 sample/s) while bench.py runs one kernel variant for ~20 s.
#   tools/power_trace.sh [variant ...]      (default: auto mfma_f32 mfma_f32_diag_mfmaonly)
# Writes gpurun_out/power_trace.txt ; copy into profiles/ afterwards.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
export RD_HIP_LIB=$R/ribodetector_amd/csrc/librd_hip_diag.so    # the *_diag_* variants exist only in the diagnostic build
O=$R/gpurun_out/power_trace.txt
mkdir -p $R/gpurun_out
VARS=${@:-auto mfma_f32 mfma_f32_diag_mfmaonly}
: > $O
for v in $VARS; do
  steps=300; [ "$v" = "mfma_f32" ] && steps=80
  echo "## variant $v: python bench.py --variant $v --steps $steps --warmup 2 --no-cpu-baseline --no-alt" >> $O
  python $R/bench.py --variant $v --steps $steps --warmup 2 --no-cpu-baseline --no-alt --no-encoder --traffic off --resident-only > /tmp/pt_$v.json 2>/dev/null &
  pid=$!
  t=0
  while kill -0 $pid 2>/dev/null; do
    s=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk clock level|Socket Graphics Package Power|Average Graphics Package Power" | sed -E 's/.*sclk clock level: [0-9]+: //; s/.*Power \(W\): //' | tr '\n' ' ')
    echo "t=$t $s" >> $O
    t=$((t+1)); sleep 1
  done
  python -c "import json,sys; j=json.load(open('/tmp/pt_$v.json')); print('# result: %.2f M reads/s, %.2f ms per launch' % (j['value']/1e6, j['roofline']['avg_launch_ms']))" >> $O 2>/dev/null
done
cat $O | tail -120
