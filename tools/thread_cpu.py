#!/usr/bin/env python
"""Which threads of the CLI use the host: one plain -> plain run on 2 x 8 Mi records in tmpfs with /proc/<pid>/task/*/stat sampled every
20 ms - CPU seconds per thread (user / sys), beside the per-role figures the CLI records itself (Predictor.thread_cpu_s). Readers:
parse + copy into the pinned chunk; writers: pwritev into the page cache (sys); one thread of the HIP runtime wakes for completion
signals; the main thread sleeps between polls.   python tools/thread_cpu.py"""
import os, sys, time, threading, json
sys.path.insert(0, os.getcwd())
import torch
from ribodetector_amd import detect, synth
d = "/dev/shm/thrcpu"; os.makedirs(d, exist_ok=True)
n = 8 << 20
files = []
for m, seed in ((1, 2000), (2, 7000)):
    a, o, l = synth.reads_torch(n, 100, seed=seed, device="cuda:0")
    p = os.path.join(d, "r_%d.fq" % m)
    synth.fastq_image_torch(a, o, l, mate=m).cpu().numpy().tofile(p)
    files.append(p)
del a, o, l
torch.cuda.empty_cache()
def run(tag):
    outs = [os.path.join(d, "%s_o%d.fq" % (tag, k)) for k in (1, 2)]
    return detect.main(["-l", "100", "-i", *files, "-o", *outs, "-e", "rrna"])
run("warm")
samples = {}
stop = False
def sampler():
    pid = os.getpid()
    while not stop:
        for tid in os.listdir("/proc/%d/task" % pid):
            try:
                st = open("/proc/%d/task/%s/stat" % (pid, tid)).read()
                comm = st[st.index("(") + 1:st.rindex(")")]
                f = st[st.rindex(")") + 2:].split()
                samples[tid] = (comm, (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK"), int(f[11]) / os.sysconf("SC_CLK_TCK"), int(f[12]) / os.sysconf("SC_CLK_TCK"))
            except Exception:
                pass
        time.sleep(0.02)
base = {}
pid = os.getpid()
for tid in os.listdir("/proc/%d/task" % pid):
    st = open("/proc/%d/task/%s/stat" % (pid, tid)).read()
    f = st[st.rindex(")") + 2:].split()
    base[tid] = (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK")
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); c0 = time.process_time()
p = run("timed")
dt = time.perf_counter() - t0; cpu = time.process_time() - c0
names = {str(t.native_id): t.name for t in threading.enumerate()}
stop = True; th.join()
rows = sorted(((v[1] - base.get(k, 0.0), v[0], k, v[2], v[3]) for k, v in samples.items()), reverse=True)
print("wall", round(dt, 3), "cpu", round(cpu, 3), "reads/s", round(2 * n / dt / 1e6, 2), p.thread_cpu_s)
for r in rows[:14]:
    print(round(r[0], 3), r[1], r[2], names.get(r[2], ""), "user", round(r[3], 2), "sys", round(r[4], 2))
import shutil; shutil.rmtree(d)
