#!/usr/bin/env python
"""End-to-end CLI rate on this machine's GPU: FASTQ file(s) in -> classified FASTQ file(s) out, wall clock around
detect.main() (model load, parsing, H2D, kernels, D2H, writing). python tools/e2e_bench.py [--reads 4000000]"""
import argparse
import gzip
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from ribodetector_amd import detect, synth      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4000000)
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="rde2e", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    files = {}
    for mate, seed in ((1, 1), (2, 2)):
        arena, off, _ = synth.reads_numpy(a.reads, 100, seed=seed)
        p = os.path.join(d, "r_%d.fq" % mate)
        synth.write_fastq_realistic(p, arena, off, mate, seed=seed)
        with open(p, "rb") as fi, gzip.open(p + ".gz", "wb", compresslevel=5) as fo:
            shutil.copyfileobj(fi, fo, 1 << 24)
        files[mate] = p
    out = {"reads_per_file": a.reads}

    def run(tag, inputs, outputs, extra=(), n=None):
        t, c = time.perf_counter(), time.process_time()
        p = detect.main(["-l", "100", "-i", *inputs, "-o", *outputs, *extra])      # default chunking: 1 Mi reads per chunk
        dt, cpu = time.perf_counter() - t, time.process_time() - c
        out[tag] = {"reads_per_s": len(inputs) * (n or a.reads) / dt, "seconds": dt, "host_cores_busy": round(cpu / dt, 2), "rrna": p.num_rrna, "non_rrna": p.num_nonrrna,
                    "main_thread_s": {k: round(v, 3) for k, v in p._stage_s.items()}, "timing": {k: round(v, 4) for k, v in p.timing.items()}}

    o = lambda n: os.path.join(d, n)     # noqa: E731
    run("first_call_se_plain", [files[1]], [o("w.fq")])        # includes HIP start-up, model load, first pinned allocations
    run("first_call_pe_plain", [files[1], files[2]], [o("w1.fq"), o("w2.fq")], ["-e", "rrna"])
    run("se_plain_to_plain", [files[1]], [o("a.fq")])
    run("se_gz_to_plain", [files[1] + ".gz"], [o("b.fq")])
    run("se_gz_to_gz", [files[1] + ".gz"], [o("c.fq.gz")])
    run("pe_plain_to_plain", [files[1], files[2]], [o("d1.fq"), o("d2.fq")], ["-e", "rrna"])
    run("pe_gz_to_plain", [files[1] + ".gz", files[2] + ".gz"], [o("e1.fq"), o("e2.fq")], ["-e", "rrna"])
    run("pe_gz_to_gz", [files[1] + ".gz", files[2] + ".gz"], [o("f1.fq.gz"), o("f2.fq.gz")], ["-e", "rrna"])
    # round 4: .gz outputs are deflated on the GPU by default; the host's libdeflate writer for comparison, plain -> gz (GPU-bound:
    # recurrence + deflate), and gz -> gz with every usable core given to the inflate of the inputs (-t)
    run("se_plain_to_gz", [files[1]], [o("g.fq.gz")])
    run("pe_plain_to_gz", [files[1], files[2]], [o("h1.fq.gz"), o("h2.fq.gz")], ["-e", "rrna"])
    cores = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = min(cores, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    run("pe_gz_to_gz_t%d" % cores, [files[1] + ".gz", files[2] + ".gz"], [o("i1.fq.gz"), o("i2.fq.gz")], ["-e", "rrna", "-t", str(cores)])
    # BGZF inputs = the .gz files the device wrote above (non-rRNA mates of pe_gz_to_gz): members inflated on the GPU (the default for
    # such files) or by the host's member decoder
    with gzip.open(o("f1.fq.gz"), "rb") as fh:
        nb = sum(chunk.count(b"\n") for chunk in iter(lambda: fh.read(1 << 24), b"")) // 4
    out["bgzf_reads_per_file"] = nb
    bg = [o("f1.fq.gz"), o("f2.fq.gz")]
    run("pe_bgzf_to_gz", bg, [o("m1.fq.gz"), o("m2.fq.gz")], ["-e", "rrna"], n=nb)
    run("pe_bgzf_to_plain", bg, [o("n1.fq"), o("n2.fq")], ["-e", "rrna"], n=nb)
    os.environ["RD_DEVICE_INFLATE"] = "0"
    run("pe_bgzf_to_gz_host_inflate", bg, [o("p1.fq.gz"), o("p2.fq.gz")], ["-e", "rrna"], n=nb)
    del os.environ["RD_DEVICE_INFLATE"]
    import hashlib

    def sha(name):      # of the text: .gz files through Python's gzip (every member's CRC-32 and ISIZE checked on the way)
        h = hashlib.sha1()
        with (gzip.open(o(name), "rb") if name.endswith(".gz") else open(o(name), "rb")) as fh:
            for chunk in iter(lambda: fh.read(1 << 24), b""):
                h.update(chunk)
        return h.hexdigest()
    # the same records whatever the route: plain -> plain / gz -> gz (device deflate) of the sequencer-like inputs; BGZF -> plain (device
    # inflate) / BGZF -> gz (device inflate + deflate) / BGZF -> gz with the host's inflate
    out["same_text"] = {"plain_to_plain == gz_to_gz": sha("d1.fq") == sha("f1.fq.gz"),
                        "bgzf_to_plain == bgzf_to_gz == bgzf_to_gz_host_inflate": len({sha("n1.fq"), sha("m1.fq.gz"), sha("p1.fq.gz")}) == 1,
                        "second mates likewise": sha("d2.fq") == sha("f2.fq.gz") and len({sha("n2.fq"), sha("m2.fq.gz"), sha("p2.fq.gz")}) == 1}
    for f in ("m1.fq.gz", "m2.fq.gz", "n1.fq", "n2.fq", "p1.fq.gz", "p2.fq.gz"):
        os.remove(o(f))
    os.environ["RD_DEVICE_GZIP"] = "0"
    run("se_gz_to_gz_host_deflate", [files[1] + ".gz"], [o("j.fq.gz")])
    run("pe_gz_to_gz_host_deflate", [files[1] + ".gz", files[2] + ".gz"], [o("k1.fq.gz"), o("k2.fq.gz")], ["-e", "rrna"])
    run("pe_plain_to_gz_host_deflate", [files[1], files[2]], [o("l1.fq.gz"), o("l2.fq.gz")], ["-e", "rrna"])
    del os.environ["RD_DEVICE_GZIP"]
    out["output_bytes"] = {n: os.path.getsize(o(n)) for n in ("f1.fq.gz", "k1.fq.gz", "d1.fq") if os.path.exists(o(n))}
    shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
