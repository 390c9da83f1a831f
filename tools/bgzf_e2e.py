import os, sys, time, json, shutil, tempfile
sys.path.insert(0, os.getcwd())
import torch
from ribodetector_amd import detect, synth
d = tempfile.mkdtemp(prefix="rd_bg_", dir="/dev/shm")
n = 4 << 20
files = []
for m, seed in ((1, 2000), (2, 7000)):
    a, o, l = synth.reads_torch(n, 100, seed=seed, device="cuda:0")
    p = os.path.join(d, "r_%d.fq" % m)
    synth.fastq_image_torch(a, o, l, mate=m).cpu().numpy().tofile(p)
    files.append(p)
# BGZF inputs = this build's own .gz outputs: run plain -> gz with everything labelled into one file (ensure rrna puts ~all in non-rRNA)... simpler: write members with the device
from ribodetector_amd.gz import DeviceGzip, eof_block
import numpy as np
dg = DeviceGzip("cuda:0")
for p in files:
    t = torch.from_numpy(np.fromfile(p, dtype=np.uint8)).cuda()
    nl = torch.nonzero(t == 10).flatten()
    rs = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), nl[3::4] + 1])
    out, info = dg.compress_selected(t, rs, torch.zeros(rs.numel() - 1, dtype=torch.int8, device="cuda"), 0)
    torch.cuda.synchronize()
    with open(p + ".gz", "wb") as fh:
        fh.write(out[: int(info[0])].cpu().numpy().tobytes()); fh.write(eof_block())
res = {}
for tag, env in (("host_inflate", "0"), ("device_inflate", "1"), ("host_inflate_2", "0"), ("device_inflate_2", "1")):
    os.environ["RD_DEVICE_INFLATE"] = env
    outs = [os.path.join(d, "%s_o%d.fq.gz" % (tag, k)) for k in (1, 2)]
    t0, c0 = time.perf_counter(), time.process_time()
    p = detect.main(["-l", "100", "-i", files[0] + ".gz", files[1] + ".gz", "-o", *outs, "-e", "rrna"])
    dt, cpu = time.perf_counter() - t0, time.process_time() - c0
    res[tag] = {"reads_per_s": 2 * n / dt, "seconds": dt, "host_cores_busy": cpu / dt, "main_thread_s": p._stage_s, "rrna": p.num_rrna}
    for o in outs: os.remove(o)
print(json.dumps(res))
shutil.rmtree(d)
