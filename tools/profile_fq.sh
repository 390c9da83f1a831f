#!/bin/bash
# the device-ingest part of tools/profile_round.sh alone (kernel table + counters of tools/fq_bench.py): tools/profile_fq.sh r06
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_$TAG
(rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$TAG/fq -o x -- python $R/tools/fq_bench.py) > $O/fq.log 2>&1
find /tmp/pf_$TAG/fq -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats_fq.csv \;
grep -h "^{" $O/fq.log | head -1 > $O/${TAG}_line_fq.json
python $R/tools/fq_bench.py --out $O/${TAG}_fq_bench.json > /dev/null 2>&1
tail -3 $O/fq.log
