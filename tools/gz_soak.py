#!/usr/bin/env python
"""Soak of the device gunzip / gzip path: the CLI, BGZF -> .gz, 2 x 2 Mi sequencer-like records, over and over for `seconds`, every
output hashed (the files must be the same every time; any member that fails its CRC-32 / ISIZE check on the device ends a run with an
error), then the inflate kernel alone, 60 launches over 40 MB of members. Run two at once (different tags) to have two processes share
the GPU.   python tools/gz_soak.py <tag> <seconds>"""
import os, sys, time, json, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ribodetector_amd import detect, synth
from ribodetector_amd.gz import DeviceGzip, DeviceGunzip, eof_block
tag = sys.argv[1]
d = "/dev/shm/soak_" + tag; os.makedirs(d, exist_ok=True)
n = 2 << 20
files = []
dg = DeviceGzip("cuda:0")
for m, seed in ((1, 11), (2, 12)):
    a, o, _ = synth.reads_numpy(n, 100, seed=seed)
    p = os.path.join(d, "r_%d.fq" % m)
    synth.write_fastq_realistic(p, a, o, m, seed=seed)
    t = torch.from_numpy(np.fromfile(p, dtype=np.uint8)).cuda()
    nl = torch.nonzero(t == 10).flatten()
    rs = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), nl[3::4] + 1])
    out, info = dg.compress_selected(t, rs, torch.zeros(rs.numel() - 1, dtype=torch.int8, device="cuda"), 0)
    torch.cuda.synchronize()
    with open(p + ".gz", "wb") as fh:
        fh.write(out[: int(info[0])].cpu().numpy().tobytes()); fh.write(eof_block())
    files.append(p + ".gz")
    del t, nl, rs, out
shas = set()
t0 = time.time(); runs = 0; errs = 0
while time.time() - t0 < float(sys.argv[2]):
    outs = [os.path.join(d, "o%d.fq.gz" % k) for k in (1, 2)]
    try:
        p = detect.main(["-l", "100", "-i", *files, "-o", *outs, "-e", "rrna"])
        h = hashlib.sha1(open(outs[0], "rb").read()).hexdigest() + hashlib.sha1(open(outs[1], "rb").read()).hexdigest()
        shas.add(h)
    except BaseException as e:
        errs += 1
        print("ERROR", repr(e)[:300], flush=True)
    runs += 1
# and the inflate kernel alone on many members, repeatedly
comp = np.fromfile(files[0], dtype=np.uint8)
du = DeviceGunzip("cuda:0")
bad = 0
for it in range(60):
    nm, consumed, ob, _ = du.index(comp[: 40 << 20], min(len(comp), 40 << 20))
    try:
        du.inflate(comp, consumed, nm, ob)
    except ValueError as e:
        bad += 1; print("INFLATE ERROR", e, flush=True)
print(json.dumps({"tag": tag, "cli_runs": runs, "cli_errors": errs, "distinct_outputs": len(shas), "inflate_launches": 60, "inflate_errors": bad}))
import shutil; shutil.rmtree(d)
