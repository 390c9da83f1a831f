#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/s10
mkdir -p $O
cd $R
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "encoders or abi" 2>&1 | tail -8) | tee $O/pytest.log
timeout 300 python tools/encoder_bench.py 2>/dev/null | tee $O/encoder_bench.json
