#!/usr/bin/env python
"""Where does the time of gz_shard.prepare() go? W thread-ranks on the one GPU of this box (collectives = barriers, as in
tests/test_gpu_gz_range.py) over two sequencer-like mate files, per phase; beside it the streaming device reader of one rank over the same
files (text framed, nothing classified).      python tools/gz_range_probe.py [--reads 4000000] [--worlds 1,2,8]"""
import argparse
import gzip
import json
import os
import sys
import tempfile
import threading
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


class Ranks:
    def __init__(self, world):
        self.world, self.bar = world, threading.Barrier(world, timeout=600)
        self.slots, self.sh = [None] * world, [None] * world

    def all_gather(self, rank):
        def f(obj):
            self.slots[rank] = obj
            self.bar.wait()
            out = list(self.slots)
            self.bar.wait()
            return out
        return f

    def shift(self, rank):
        def f(buf):
            self.sh[rank] = None if buf is None else buf.clone()
            self.bar.wait()
            got = self.sh[rank + 1] if rank + 1 < self.world else None
            self.bar.wait()
            return got
        return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4000000)
    ap.add_argument("--worlds", default="1,2,8")
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    import torch
    from ribodetector_amd import synth
    from ribodetector_amd.data_loader import device_reader as dr
    from ribodetector_amd.data_loader import gz_shard as gs
    d = tempfile.mkdtemp(prefix="rd_gzr_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    out = {"reads_per_file": a.reads}
    try:
        paths = []
        for mate, seed in ((1, 1), (2, 2)):
            arena, off, _ = synth.reads_numpy(a.reads, 100, seed=seed)
            p = os.path.join(d, "r_%d.fq" % mate)
            synth.write_fastq_realistic(p, arena, off, mate, seed=seed)
            with open(p, "rb") as fi, open(p + ".gz", "wb") as fo:
                fo.write(gzip.compress(fi.read(), 6))
            os.remove(p)
            paths.append(p + ".gz")
        out["compressed_bytes"] = [os.path.getsize(p) for p in paths]
        dev = "cuda:0"
        # the streaming reader of one rank
        for rep in range(a.repeat):
            t0 = time.perf_counter()
            n = 0

            def drain(p):
                nonlocal n
                for c in dr.get_seq_chunks_device(p, chunk_size=1 << 20, first_chunk=1 << 17, device=dev):
                    c.ready.synchronize()
                    n += c.n
            th = [threading.Thread(target=drain, args=(p,)) for p in paths]
            [t.start() for t in th]
            [t.join() for t in th]
            out.setdefault("stream_reader_s", []).append(round(time.perf_counter() - t0, 3))
        for w in [int(x) for x in a.worlds.split(",")]:
            for rep in range(a.repeat):
                G = Ranks(w)
                res = [None] * w

                def rank(r):
                    torch.cuda.set_device(torch.device(dev))
                    t0 = time.perf_counter()
                    rr, why = gs.prepare(paths, r, w, dev, [False, False], G.all_gather(r), G.shift(r))
                    t1 = time.perf_counter()
                    assert rr is not None, why
                    n = 0
                    for p, x in zip(paths, rr):
                        for c in dr.get_seq_chunks_device(p, chunk_size=1 << 20, first_chunk=1 << 17, byte_range=x, device=dev):
                            c.ready.synchronize()
                            n += c.n
                    res[r] = {"prepare_s": round(t1 - t0, 3), "chunks_s": round(time.perf_counter() - t1, 3), "records": n, "phases": rr[0].stats["phases_s"],
                              "decode_s": [x.stats["decode_s"] for x in rr]}
                th = [threading.Thread(target=rank, args=(r,)) for r in range(w)]
                t0 = time.perf_counter()
                [t.start() for t in th]
                [t.join() for t in th]
                out.setdefault("world_%d" % w, []).append({"wall_s": round(time.perf_counter() - t0, 3), "records": sum(x["records"] for x in res), "rank0": res[0],
                                                             "rank_last": res[-1]})
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_gz_range_probe.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
