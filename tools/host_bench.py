#!/usr/bin/env python
"""Host-side ingest / output rates of librd_host.so on this machine (no GPU needed):
    python tools/host_bench.py [--reads 1000000]
prints reads/s for the FASTQ reader on plain and gzip input, next to Python's gzip + the numpy parser and zlib's gzread
speed class (python gzip module), and for the label-partitioned writer (plain and .gz)."""
import argparse
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from ribodetector_amd import synth                                   # noqa: E402
from ribodetector_amd.data_loader import fastx_parser as fx          # noqa: E402


def rate(n, fn, reps=3):
    best = 1e30
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t)
    return n / best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1000000)
    a = ap.parse_args()
    import torch  # noqa: F401  (the reader allocates pinned tensors when a GPU is present)
    d = tempfile.mkdtemp(prefix="rdhost")
    arena, off, _ = synth.reads_numpy(a.reads, 100, seed=1)
    plain, gz = os.path.join(d, "r.fq"), os.path.join(d, "r.fq.gz")
    synth.write_fastq_realistic(plain, arena, off, 1, seed=1)
    with open(plain, "rb") as fi, gzip.open(gz, "wb", compresslevel=5) as fo:
        fo.write(fi.read())
    out = {"reads": a.reads, "plain_MB": os.path.getsize(plain) / 1e6, "gz_MB": os.path.getsize(gz) / 1e6}

    def consume(path):
        return lambda: sum(len(c.seq_len) for c in fx.get_seq_chunks(path, 1 << 18))

    out["reader_plain_reads_per_s"] = rate(a.reads, consume(plain))
    out["reader_gz_reads_per_s"] = rate(a.reads, consume(gz))            # parallel decoder for files >= 16 MB (csrc/rd_pgzip.h)
    os.environ["RD_GZ_THREADS"] = "0"
    out["reader_gz_sequential_reads_per_s"] = rate(a.reads, consume(gz))
    del os.environ["RD_GZ_THREADS"]
    # the decoders alone: decompressed MB/s, sequential vs parallel by thread count
    import ctypes as C
    from ribodetector_amd import _native as N
    L = N.host_lib()
    raw_n = os.path.getsize(plain)
    buf = np.empty(raw_n + 16, dtype=np.uint8)
    n = C.c_int64(0)
    out["inflate_sequential_MB_per_s"] = rate(raw_n / 1e6, lambda: L.rd_host_gunzip(gz.encode(), buf.ctypes.data, raw_n + 16, C.byref(n)))
    for th in (2, 4, 7, 8, 12):
        st = (C.c_int64 * 4)()
        out["inflate_parallel_%d_MB_per_s" % th] = rate(
            raw_n / 1e6, lambda: L.rd_host_gunzip_parallel(gz.encode(), buf.ctypes.data, raw_n + 16, C.byref(n), th, 4 << 20, st))
        out["inflate_parallel_%d_stats" % th] = list(st)
    out["python_gzip_read_reads_per_s"] = rate(a.reads, lambda: gzip.open(gz, "rb").read(), reps=1)
    chunk = next(fx.get_seq_chunks(plain, a.reads))
    labels = (np.arange(a.reads) % 10 == 0).astype(np.int8)
    for name in ("o.fq", "o.fq.gz"):
        def write(path=os.path.join(d, name)):
            w = fx.open_for_write(path)
            w.write_selected(chunk, labels, 0)
            w.close()
        out["writer_%s_reads_per_s" % ("gz" if name.endswith("gz") else "plain")] = rate(int((labels == 0).sum()), write, reps=2)
    # this build's own .gz output read back: its members carry their size, the reader decodes them in parallel (indexed mode)
    own = os.path.join(d, "o.fq.gz")
    n_own = int((labels == 0).sum())
    out["reader_own_gz_output_reads_per_s"] = rate(n_own, consume(own))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
