#!/bin/bash
# gz_overlap_trace.sh - a kernel trace of the CLI writing .gz outputs (plain -> gz, paired-end): where the device deflate runs
# relative to the recurrence kernels (post stream, beside the next chunk's recurrences).   tools/gz_overlap_trace.sh > profiles/r04_gz_overlap_trace.txt
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
D=$(mktemp -d /dev/shm/rd_tr.XXXXXX)
cd "$R"
python - "$D" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from ribodetector_amd import synth
d = sys.argv[1]
n = 3 << 20
for m, seed in ((1, 2000), (2, 7000)):
    a, o, l = synth.reads_torch(n, 100, seed=seed, device="cuda:0")
    synth.fastq_image_torch(a, o, l, mate=m).cpu().numpy().tofile(os.path.join(d, "r_%d.fq" % m))
PY
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rocprofv3 --kernel-trace --output-format csv -d $D/tr -o x -- python -m ribodetector_amd.detect -l 100 -i $D/r_1.fq $D/r_2.fq -o $D/o1.fq.gz $D/o2.fq.gz -r $D/q1.fq.gz $D/q2.fq.gz -e rrna > $D/run.log 2>&1
tail -2 $D/run.log | cut -c1-200 1>&2
cd "$R"
echo "# CLI, paired-end plain -> gz, 3 Mi pairs (three chunks): kernel trace around a device-deflate launch (tools/trace_window.py)."
echo "# q = HSA queue: the recurrence kernels run on the main stream's queue, rd_gz_* on the post stream's - beside the next chunk's recurrences."
PYTHONPATH=$R python tools/trace_window.py $D/tr rd_gz_deflate
rm -rf $D
