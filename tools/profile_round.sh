#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02
# writes small text/CSV summaries to gpurun_out/prof_<tag>/ ; copy them into profiles/ afterwards.
# Counter passes (--pmc) are their own runs, never combined with a trace domain.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_$TAG
BF="--steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-encoder --no-e2e --traffic off"
stats() {   # stats <name> <command...>: kernel trace + stats of a command
  local name=$1; shift
  (rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG/$name -o x -- "$@") > $O/$name.log 2>&1
  find /tmp/p_$TAG/$name -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats_$name.csv \;
  grep -h "^{" $O/$name.log | head -1 > $O/${TAG}_line_$name.json
}
pmc() {     # pmc <name> <counters...> -- <command...>
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
  shift
  (rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d /tmp/p_$TAG/$name -o x -- "$@") > $O/$name.log 2>&1
}
# 1. the metric's workload (pe100, default kernel) and the other BASELINE configs / the exact-fp32 kernel
stats pe100 python $R/bench.py $BF
for wl in se100 pe150 var300; do stats $wl python $R/bench.py $BF --workload $wl; done
stats pe100_mfma_f32 python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-encoder --no-e2e --traffic off --variant mfma_f32
# 2. HBM traffic and SQ counters of the recurrence kernel (resident inputs, everything on one stream: no kernel runs beside the one counted)
for c in FETCH_SIZE WRITE_SIZE; do pmc pmc_$c $c -- python $R/bench.py $BF --resident-only --inline-refine; done
pmc pmc_sq GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -- python $R/bench.py $BF --resident-only --inline-refine
pmc pmc_lds SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SALU -- python $R/bench.py $BF --resident-only --inline-refine
# 3. the HBM-bound kernels: standalone encoders, pair fusion, counters
stats encoders python $R/tools/encoder_bench.py
for c in FETCH_SIZE WRITE_SIZE; do pmc enc_pmc_$c $c -- python $R/tools/encoder_bench.py; done
# 4. device gzip (round 4): kernel table of the rd_gz_* kernels on a 2^20-record chunk, and their HBM traffic
stats gz python $R/tools/gz_bench.py
python $R/tools/gz_bench.py --inflate-sweep --out $O/${TAG}_gz_bench.json > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do pmc gz_pmc_$c $c -- python $R/tools/gz_bench.py; done
pmc gz_pmc_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -- python $R/tools/gz_bench.py
pmc gz_pmc_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -- python $R/tools/gz_bench.py
# 5. device ingest (round 5): FASTQ index / gather / select / copy kernels and the single-stream inflate on a 2^20-record batch
stats fq python $R/tools/fq_bench.py
python $R/tools/fq_bench.py --out $O/${TAG}_fq_bench.json > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do pmc fq_pmc_$c $c -- python $R/tools/fq_bench.py --no-stream --reps 2; done
python $R/tools/prof_summarize.py --tag $TAG --out $O /tmp/p_$TAG > $O/${TAG}_summary.txt 2>&1
cd $R
T0=$(date +%s)
(timeout 900 python bench.py --full-out $O/${TAG}_bench_full.json 2>/dev/null | grep "^{") > $O/${TAG}_bench.json
echo "default bench.py wall seconds: $(( $(date +%s) - T0 ))" >> $O/${TAG}_summary.txt
rm -f $O/*.log
ls -la $O
