#!/bin/bash
# Round-2 evidence, one gpurun call: accuracy experiments, sensitivity probes, kernel A/Bs, the rocprofv3 round profile, the GPU test
# suite. Results land in gpurun_out/ev_r02/ (copy what is to be kept into profiles/).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ev_r02
mkdir -p $O
cd $R
DIAG=$R/ribodetector_amd/csrc/librd_hip_diag.so
(RD_HIP_LIB=$DIAG timeout 900 python tools/acc_experiment.py --reads 1048576 --oracle-reads 300000 \
   --variants auto,mfma_f32,t32_acc0,t32_acc1_4prod,t32_acc2_smallfirst,t32_acc4_exparg,t32_acc8_newton,t32_acc15_all,t32_acc16_ops24,t32_acc32_creg,t32_acc48_ops24_creg,t32_acc112_hlskip,t32_acc240_sharedrcp,t32_acc752_onercp,t32_acc248_sharedrcp_newton) > $O/r02_acc_100.json 2> $O/acc.err
for L in 150 300; do
  (timeout 900 python tools/acc_experiment.py --reads 524288 --len $L --oracle-reads 100000 --variants auto,mfma_f32,simple) > $O/r02_acc_$L.json 2>> $O/acc.err
done
(timeout 600 python tools/sens_probe.py --reads 4194304 --seed 99) > $O/r02_sens_100.json 2>> $O/acc.err
(timeout 600 python tools/outlier_probe.py) > $O/r02_outlier.json 2>> $O/acc.err
: > $O/r02_variants_ab.txt
for v in t32_acc0 t32_acc16_ops24 t32_acc32_creg t32_acc48_ops24_creg t32_acc112_hlskip t32_acc240_sharedrcp t32_acc0 t32_acc240_sharedrcp; do
  RD_HIP_LIB=$DIAG timeout 300 python bench.py --steps 8 --variant $v --resident-only --no-cpu-baseline --no-alt --no-encoder --traffic off 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('%-24s %.2f M reads/s  launch %.3f ms  step %.3f ms' % ('$v', j['value']/1e6, j['roofline']['avg_launch_ms'], j['ms_per_step']))" >> $O/r02_variants_ab.txt
done
cat $O/r02_variants_ab.txt
: > $O/r02_bench_refine_ab.txt
for extra in "" "--inline-refine" "" "--inline-refine"; do
  timeout 600 python bench.py --no-cpu-baseline --no-alt --no-encoder --traffic off $extra 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('bench.py %-16s %.2f M reads/s  %.2f ms/step  launch %.2f ms  kernel-only %.2f M  ii/i %.4f' % ('$extra', j['value']/1e6, j['ms_per_step'], j['roofline']['avg_launch_ms'], j['config']['kernel_only_reads_per_s']/1e6, j['config']['device_path_over_kernel_only']))" >> $O/r02_bench_refine_ab.txt
done
cat $O/r02_bench_refine_ab.txt
(timeout 900 python tools/e2e_bench.py --reads 8000000) > $O/r02_e2e.json 2> $O/e2e.err
bash tools/profile_round.sh r02 > $O/profile.log 2>&1
cp $R/gpurun_out/prof_r02/* $O/ 2>/dev/null
T0=$(date +%s)
(timeout 2400 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -20) > $O/r02_gpu_tests.txt
echo "gpu test suite wall seconds: $(( $(date +%s) - T0 ))" >> $O/r02_gpu_tests.txt
cat $O/r02_gpu_tests.txt
cp gpurun_out/parity_report.json $O/r02_parity_report.json 2>/dev/null
tail -3 $O/acc.err
ls $O | head -60
