#!/bin/bash
# same-box A/B of the stream decoder's workgroup shape (sections per workgroup = waves per workgroup: -DRD_GZS_WAVES=4 | 8 | 16):
#   tools/gzs_waves_ab.sh ribodetector_amd/csrc/librd_hip.so tools/ab/librd_hip_w8.so tools/ab/librd_hip_w16.so
# per library: rd_gz_stream_inflate alone (tools/fq_bench.py) and the CLI gz -> gz, paired-end and single-end (tools/e2e_bench.py)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/ab
OUT=gpurun_out/ab/gzs_waves.txt
: > $OUT
for rep in 1 2; do
for lib in "$@"; do
  export RD_HIP_LIB=$R/$lib
  python tools/fq_bench.py --out /tmp/fq_ab.json > /dev/null 2>&1
  python - "$lib" <<'PY' | tee -a $OUT
import json, sys
j = json.load(open("/tmp/fq_ab.json"))
k = [v for n, v in j["kernels"].items() if "stream" in n][0]
print("%-40s stream inflate alone %.2f ms" % (sys.argv[1], k["ms"]))
PY
  for mode in "" "--single-end"; do
    python tools/e2e_bench.py --bench-legs --legs gz_to_gz --no-seqlike $mode 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['gz_to_gz']; print('%-40s gz->gz %-12s %.2f M reads/s  steady %.2f M  cores %.2f' % ('$lib', '${mode:-paired}', j['reads_per_s']/1e6, (j['reads_per_s_after_first_chunk'] or 0)/1e6, j['host_cores_busy']))" | tee -a $OUT
  done
done
done
