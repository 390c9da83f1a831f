import sys, os, time, json, gzip
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ribodetector_amd import synth, detect
d = "/dev/shm/rd_pf"; os.makedirs(d, exist_ok=True)
n = 6000000
files = []
for mate in (1, 2):
    a, o, l = synth.reads_torch(n, 100, seed=mate, device="cuda")
    p = os.path.join(d, "r_%d.fq" % mate)
    synth.fastq_image_torch(a, o, l, mate=mate, style="seqlike").cpu().numpy().tofile(p)
    synth.pgzip_file(p, p + ".gz", 6)
    files.append(p + ".gz")
    del a, o, l
plain = [f[:-3] for f in files]
for rep in range(4):
    t0 = time.perf_counter()
    pr = detect.main(["-l", "100", "-i", *(files if rep < 3 else plain), "-o", d + "/o1.fq.gz", d + "/o2.fq.gz", "-e", "rrna"], log_level="WARNING")
    dt = time.perf_counter() - t0
    fc = pr._first_chunk
    print(json.dumps({"wall": round(dt, 3), "timing": {k: v for k, v in pr.timing.items() if k != "ingest"}, "start": time.strftime("%H:%M:%S"), "first_chunk_labels_at_s_after_detect_start": None,
                      "feeders": {k: {x: (v["feeder"].get(x) if x != "thread_started_at" else round(v["feeder"].get(x, 0) - pr.timing["run_started_at"], 4)) for x in ("thread_started_at", "first_batch_submitted_at_s", "batches_framed_at_s", "read", "wait_slot", "submit", "batches")} for k, v in pr.ingest.items()}}))
import shutil; shutil.rmtree(d)
