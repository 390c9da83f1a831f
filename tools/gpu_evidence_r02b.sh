#!/bin/bash
# Round-2 evidence, second part (after the staging rewrite of the default kernel): the microbenchmarks behind the energy / issue model
# of DESIGN.md §3.1, the occupancy sweep, the 16x16x32 timing diagnosis, the refreshed rocprofv3 round profile and the GPU test suite.
# One gpurun call; results land in gpurun_out/ev_r02b/ (copy what is to be kept into profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ev_r02b
mkdir -p $O
cd $R
DIAG=$R/ribodetector_amd/csrc/librd_hip_diag.so
(echo "## tools/ubench/mfma_clock"; timeout 120 tools/ubench/mfma_clock; echo; echo "## tools/ubench/issue_cycles"; timeout 120 tools/ubench/issue_cycles; echo;
 echo "## tools/ubench/op_energy 3"; timeout 300 tools/ubench/op_energy 3; echo; echo "## tools/ubench/mfma_energy 3"; timeout 400 tools/ubench/mfma_energy 3) > $O/r02_ubench.txt 2>&1
cat $O/r02_ubench.txt
(RD_HIP_LIB=$DIAG timeout 600 python tools/occupancy_sweep.py --variants mfma_f16x3_t32 mfma_f16x3_t32_diag_mfmaonly mfma_f16x3_t32_diag_nobarrier t32_diag_mfma16 | tail -1) > $O/r02_occupancy_sweep.json 2>$O/occ.err
: > $O/r02_variants_ab2.txt
for v in t32_acc112_hlskip t32_acc240_sharedrcp t32_diag_mfma16 mfma_f16x3_t32_diag_mfmaonly t32_acc240_sharedrcp t32_diag_mfma16; do
  RD_HIP_LIB=$DIAG timeout 300 python bench.py --steps 8 --variant $v --resident-only --inline-refine --no-cpu-baseline --no-alt --no-encoder --traffic off 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%-30s %.2f M reads/s  launch %.3f ms  step %.3f ms' % ('$v', j['value']/1e6, j['roofline']['avg_launch_ms'], j['ms_per_step']))" >> $O/r02_variants_ab2.txt
done
cat $O/r02_variants_ab2.txt
bash tools/profile_round.sh r02 > $O/profile.log 2>&1
cp $R/gpurun_out/prof_r02/* $O/ 2>/dev/null
(timeout 900 python tools/acc_experiment.py --reads 1048576 --oracle-reads 300000 --variants auto,mfma_f32) > $O/r02_acc_100_final.json 2> $O/acc.err
T0=$(date +%s)
(timeout 2400 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -20) > $O/r02_gpu_tests.txt
echo "gpu test suite wall seconds: $(( $(date +%s) - T0 ))" >> $O/r02_gpu_tests.txt
cat $O/r02_gpu_tests.txt
cp gpurun_out/parity_report.json $O/r02_parity_report.json 2>/dev/null
ls $O | head -60
