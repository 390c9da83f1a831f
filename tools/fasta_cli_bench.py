#!/usr/bin/env python
"""The CLI on FASTA input: the device reader (rd_fasta_index: the batch re-written and indexed on the GPU, the round-5 default) against the
host parser (RD_DEVICE_FASTA=0), same file, alternating calls, outputs compared. 8 Mi single-end reads of 100 bp as two-line FASTA in
tmpfs, plain output.          python tools/fasta_cli_bench.py [--reads N] [--out file.json]"""
import argparse
import hashlib
import json
import os
import shutil
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=8 << 20)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import numpy as np
    from ribodetector_amd import detect, synth
    n = a.reads
    arena, off, lens = synth.reads_numpy(n, 100, seed=5)
    rec = np.empty((n, 112), dtype=np.uint8)
    rec[:, 0], rec[:, 1] = ord(">"), ord("r")
    idx = np.arange(n)
    for k in range(8):
        rec[:, 9 - k] = ord("0") + (idx // 10 ** k) % 10
    rec[:, 10] = 10
    rec[:, 11:111] = arena.reshape(n, 100)
    rec[:, 111] = 10
    d = "/dev/shm/fa_cli_bench"
    os.makedirs(d, exist_ok=True)
    p = os.path.join(d, "in.fasta")
    rec.tofile(p)
    runs, sha = {"1": [], "0": []}, {}
    try:
        for it, mode in enumerate(("1", "0", "1", "0", "1", "0", "1", "0")):
            os.environ["RD_DEVICE_FASTA"] = mode
            o = os.path.join(d, "out_%s.fasta" % mode)
            t0, c0 = time.time(), time.process_time()
            pr = detect.main(["-l", "100", "-i", p, "-o", o])
            dt = time.time() - t0
            if it >= 2:        # (the first call of either mode warms the process up)
                runs[mode].append({"seconds": round(dt, 4), "reads_per_s": round(n / dt), "host_cores_busy": round((time.process_time() - c0) / dt, 2),
                                   "path": pr.ingest.get("in.fasta", {}).get("path", "host")})
            sha[mode] = hashlib.sha1(open(o, "rb").read()).hexdigest()
    finally:
        os.environ.pop("RD_DEVICE_FASTA", None)
        shutil.rmtree(d, ignore_errors=True)
    med = lambda xs, k: sorted(x[k] for x in xs)[len(xs) // 2]
    out = {"what": "ribodetector CLI, %d single-end reads of 100 bp, two-line FASTA in tmpfs -> plain FASTA; median of 3 calls after a warm one" % n,
           "device_reader": {"reads_per_s": med(runs["1"], "reads_per_s"), "host_cores_busy": med(runs["1"], "host_cores_busy"), "calls": runs["1"]},
           "host_parser": {"reads_per_s": med(runs["0"], "reads_per_s"), "host_cores_busy": med(runs["0"], "host_cores_busy"), "calls": runs["0"]},
           "outputs_identical": sha["0"] == sha["1"]}
    print(json.dumps(out))
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
