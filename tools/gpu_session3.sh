#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s3
mkdir -p $O
cd $R
(timeout 600 python tools/sens_probe.py --reads 4194304 --seed 99) > $O/sens_100.json 2> $O/sens.err
(timeout 600 python tools/sens_probe.py --reads 2097152 --seed 7 --rrna-frac 0.3) > $O/sens_100_r30.json 2>> $O/sens.err
tail -3 $O/sens.err
