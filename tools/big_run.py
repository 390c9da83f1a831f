#!/usr/bin/env python
"""Stability run: the CLI on a large single-end file (an 8 M-read sequencer-like FASTQ concatenated k times), reporting the
rate, peak host RSS, peak pinned/device memory. python tools/big_run.py [--copies 6]"""
import argparse
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
    ap.add_argument("--copies", type=int, default=6)
    ap.add_argument("--reads", type=int, default=8000000)
    a = ap.parse_args()
    import torch
    d = tempfile.mkdtemp(prefix="rdbig", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    arena, off, _ = synth.reads_numpy(a.reads, 100, seed=11)
    one = os.path.join(d, "one.fq")
    synth.write_fastq_realistic(one, arena, off, 1, seed=11)
    big = os.path.join(d, "big.fq")
    with open(big, "wb") as fo:
        for _ in range(a.copies):
            with open(one, "rb") as fi:
                shutil.copyfileobj(fi, fo, 1 << 26)
    os.remove(one)
    n = a.reads * a.copies
    del arena, off
    import gc
    import threading
    import psutil
    gc.collect()
    proc = psutil.Process()
    rss0, peak, stop = proc.memory_info().rss, [0], [False]

    def sample():
        while not stop[0]:
            peak[0] = max(peak[0], proc.memory_info().rss)
            time.sleep(0.05)
    th = threading.Thread(target=sample, daemon=True)
    th.start()
    t = time.perf_counter()
    p = detect.main(["-l", "100", "-i", big, "-o", os.path.join(d, "out.fq"), "-r", os.path.join(d, "rrna.fq")])
    dt = time.perf_counter() - t
    stop[0] = True
    th.join()
    out_bytes = os.path.getsize(os.path.join(d, "out.fq")) + os.path.getsize(os.path.join(d, "rrna.fq"))
    res = {"reads": n, "input_GB": os.path.getsize(big) / 1e9, "seconds": dt, "reads_per_s": n / dt,
           "classified": p.num_read, "rrna": p.num_rrna, "non_rrna": p.num_nonrrna,
           "outputs_equal_input_size": out_bytes == os.path.getsize(big),
           "rss_before_GB": rss0 / 1e9, "peak_rss_during_run_GB": peak[0] / 1e9,
           "peak_device_GB": torch.cuda.max_memory_allocated() / 1e9, "main_thread_s": p._stage_s}
    shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
