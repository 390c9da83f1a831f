#!/usr/bin/env python
"""The chunk reader alone, no classification: 4 Mi sequencer-like records as a plain file, as BGZF through the device inflate (with the
feeder's stage times: RD_FEED_TRACE) and as BGZF through the host's member decoder - records/s and CPU seconds of each. Shows what
the reader can deliver when nothing else wants the GPU (inside a run the inflate batches wait for the gaps between the recurrence
launches: data_loader/fastx_parser.py:_DeviceInflateFeeder keeps two of them in flight for that reason).   python tools/feed_probe.py"""
import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ribodetector_amd import synth
from ribodetector_amd.data_loader import fastx_parser as fx
from ribodetector_amd.gz import DeviceGzip, eof_block
d = "/dev/shm/feedprobe"; os.makedirs(d, exist_ok=True)
n = 4 << 20
a, o, _ = synth.reads_numpy(n, 100, seed=1)
p = os.path.join(d, "r.fq")
synth.write_fastq_realistic(p, a, o, 1, seed=1)
dg = DeviceGzip("cuda:0")
t = torch.from_numpy(np.fromfile(p, dtype=np.uint8)).cuda()
nl = torch.nonzero(t == 10).flatten()
rs = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), nl[3::4] + 1])
out, info = dg.compress_selected(t, rs, torch.zeros(rs.numel() - 1, dtype=torch.int8, device="cuda"), 0)
torch.cuda.synchronize()
with open(p + ".gz", "wb") as fh:
    fh.write(out[: int(info[0])].cpu().numpy().tobytes()); fh.write(eof_block())
size = os.path.getsize(p)
res = {}
def read_all(path):
    t0 = time.perf_counter(); c0 = time.process_time(); k = 0
    for c in fx.get_seq_chunks(path, chunk_size=1 << 20, first_chunk=1 << 17):
        k += len(c.seq_len)
    return {"s": round(time.perf_counter() - t0, 3), "GB_per_s": round(size / (time.perf_counter() - t0) / 1e9, 2), "cpu_s": round(time.process_time() - c0, 3), "records": k}
for rep in range(2):
    res["plain_%d" % rep] = read_all(p)
    os.environ["RD_FEED_TRACE"] = "1"
    res["bgzf_dev_%d" % rep] = read_all(p + ".gz")
    os.environ["RD_DEVICE_INFLATE"] = "0"
    res["bgzf_host_%d" % rep] = read_all(p + ".gz")
    del os.environ["RD_DEVICE_INFLATE"]
print(json.dumps(res, indent=1))
import shutil; shutil.rmtree(d)
