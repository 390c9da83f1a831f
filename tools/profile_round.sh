#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01
# writes small text/CSV summaries to gpurun_out/prof_<tag>/ ; copy them into profiles/ afterwards.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt"
rm -rf /tmp/p_$TAG
(rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG/stats -o bench -- $B) > $O/stats.log 2>&1
find /tmp/p_$TAG/stats -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  (rocprofv3 --pmc $c --output-format csv -d /tmp/p_$TAG/pmc_$c -o bench -- $B) > $O/pmc_$c.log 2>&1
done
(rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/p_$TAG/pmc_sq -o bench -- $B) > $O/pmc_sq.log 2>&1
(rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --output-format csv -d /tmp/p_$TAG/pmc_lds -o bench -- $B) > $O/pmc_lds.log 2>&1
python $R/tools/prof_summarize.py /tmp/p_$TAG/stats /tmp/p_$TAG/pmc_FETCH_SIZE /tmp/p_$TAG/pmc_WRITE_SIZE /tmp/p_$TAG/pmc_sq /tmp/p_$TAG/pmc_lds | grep "^##\|rd_" > $O/${TAG}_summary.txt 2>&1
cd $R
(timeout 900 python bench.py 2>&1 | grep "^{") > $O/${TAG}_bench.json
grep -h "^{" $O/stats.log | head -1 > $O/${TAG}_bench_under_rocprof.json
rm -f $O/*.log
ls -la $O
