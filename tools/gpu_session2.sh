#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s2
mkdir -p $O
cd $R
export RD_HIP_LIB=$R/ribodetector_amd/csrc/librd_hip_diag.so
(timeout 1500 python tools/acc_experiment.py --reads 1048576 --oracle-reads 300000) > $O/acc_100.json 2> $O/acc_100.err
tail -5 $O/acc_100.err
