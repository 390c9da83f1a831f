#!/usr/bin/env python
"""Soak of the RANGE decoder (data_loader/gz_shard.py over rd_gz_range_decode / rd_gz_range_resolve: a single-stream .gz shared by W ranks):
the random FASTQ files of tools/gzs_soak.py - every zlib level / strategy / memLevel / window size, flushes, several members, padding - each
prepared by W in {2, 3, 5, 8} thread-ranks on the one GPU (collectives = barriers) with random batch sizes, then read through
get_seq_chunks_device. THE properties: every rank takes the same decision; if the ranks accept the file, their chunks concatenate to
gzip.decompress()'s text (line ends as LF); a refusal is counted with its reason (the CLI then takes the one-decode path).
    python tools/gzr_soak.py <seconds> [seed] [out.json]"""
import gzip
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np
import torch

from gzs_soak import deflate_member, fastq_text
from ribodetector_amd import gz
from ribodetector_amd.data_loader import device_reader as dr
from ribodetector_amd.data_loader import gz_shard as gs

DEV = "cuda:0"


class Ranks:
    def __init__(self, world):
        self.world, self.bar = world, threading.Barrier(world, timeout=300)
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
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out = sys.argv[3] if len(sys.argv) > 3 else None
    rng = np.random.default_rng(seed)
    d = "/dev/shm/gzr_soak_%d" % seed
    os.makedirs(d, exist_ok=True)
    p, p2 = os.path.join(d, "x.fastq.gz"), os.path.join(d, "x_2.fastq.gz")
    os.environ["RD_GZ_SHARD_MIN"] = "2048"
    rec = {"files": 0, "accepted": 0, "refused": 0, "mismatches": 0, "disagreements": 0, "errors": 0, "text_bytes": 0, "by_world": {}, "refusal_reasons": {},
           "members_in_accepted": 0, "seed": seed}
    BATCH0 = gz.DeviceStreamGunzip.BATCH
    t0 = time.time()
    while time.time() - t0 < seconds:
        parts = []
        nm = int(rng.choice([1, 1, 1, 2, 3]))
        for _ in range(nm):
            parts.append(deflate_member(fastq_text(rng), rng)[0])
            if rng.random() < 0.1:
                parts.append(deflate_member(b"", rng)[0])
        blob = b"".join(parts) + (bytes(int(rng.integers(1, 600))) if rng.random() < 0.1 else b"")
        want = gzip.decompress(blob).replace(b"\r\n", b"\n")
        open(p, "wb").write(blob)
        # every third file gets a MATE: the same records with longer headers, reversed bases and qualities, compressed with its own settings -
        # the two files' compressed positions drift apart, the ranks must still hold the same record indices of both
        paths, wants = [p], [want]
        if rec["files"] % 3 == 2 and want.count(b"\n") % 4 == 0 and want:
            ls = want.split(b"\n")[:-1]
            m2 = b"".join(b"%s mate=2 of this pair\n%s\n+\n%s\n" % (ls[i], ls[i + 1][::-1], ls[i + 3][::-1]) for i in range(0, len(ls), 4))
            open(p2, "wb").write(deflate_member(m2, rng)[0])
            paths.append(p2)
            wants.append(m2)
        world = int(rng.choice([2, 3, 5, 8]))
        gz.DeviceStreamGunzip.BATCH = int(rng.choice([1 << 16, 1 << 18, 1 << 20])) if rng.random() < 0.6 else BATCH0
        G = Ranks(world)
        res, errs = [None] * world, []
        chunk = int(rng.choice([1000, 65536]))

        def rank(r):
            try:
                torch.cuda.set_device(torch.device(DEV))
                rr, why = gs.prepare(paths, r, world, DEV, [False] * len(paths), G.all_gather(r), G.shift(r))
                if rr is None:
                    res[r] = (None, why)
                    return
                out_f, counts = [], []
                for f, q in enumerate(paths):
                    texts, nrec = [], 0
                    for c in dr.get_seq_chunks_device(q, chunk_size=chunk, byte_range=rr[f], device=DEV):
                        texts.append(c.to_host()[0].tobytes())
                        nrec += c.n
                    out_f.append(b"".join(texts))
                    counts.append(nrec)
                if len(set(counts)) != 1:
                    raise AssertionError("rank %d holds %s records of the mates" % (r, counts))
                res[r] = (out_f, None)
            except BaseException as e:      # noqa: BLE001
                errs.append(repr(e))
                G.bar.abort()
        th = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join() for t in th]
        rec["files"] += 1
        bw = rec["by_world"].setdefault(str(world), {"files": 0, "accepted": 0})
        bw["files"] += 1
        real = [e for e in errs if "BrokenBarrier" not in e]
        if errs:
            # a malformed FASTQ tail (CR LF files cut inside a record ...) raises in the reader like on one rank: only unexpected errors count
            if real and not all(("truncated" in e or "does not start with" in e or "different numbers of records" in e) for e in real):
                rec["errors"] += 1
                rec.setdefault("error_examples", []).append(real[0][:300])
            continue
        dec = {x[0] is None for x in res}
        if len(dec) != 1:
            rec["disagreements"] += 1
            continue
        if res[0][0] is None:
            rec["refused"] += 1
            why = res[0][1]
            key = why.split(" (rank")[0].split(" in the batch")[0][:70]
            rec["refusal_reasons"][key] = rec["refusal_reasons"].get(key, 0) + 1
            if "per rank" not in key and "is empty" not in key and "slot holds" not in key and len(rec.setdefault("other_refusals", [])) < 40:
                import zlib as _z
                ends, q = [], 0          # where the members end (file offsets), by zlib
                while q < len(blob) and blob[q:].strip(b"\0"):
                    dd = _z.decompressobj(31)
                    dd.decompress(blob[q:])
                    q = len(blob) - len(dd.unused_data)
                    ends.append(q)
                    while q < len(blob) and blob[q] == 0:
                        q += 1
                rec["other_refusals"].append({"why": why, "world": world, "size": len(blob), "member_ends": ends, "cuts": gs.range_bounds(len(blob), world),
                                              "batch": gz.DeviceStreamGunzip.BATCH})
            if len({x[1] for x in res}) != 1:
                rec["disagreements"] += 1
            continue
        rec["accepted"] += 1
        bw["accepted"] += 1
        rec["members_in_accepted"] += nm
        rec["paired_accepted"] = rec.get("paired_accepted", 0) + (len(paths) == 2)
        for f in range(len(paths)):
            got = b"".join(x[0][f] for x in res)
            rec["text_bytes"] += len(got)
            if got != wants[f] and got != wants[f] + b"\n":
                rec["mismatches"] += 1
                rec.setdefault("mismatch_examples", []).append({"world": world, "file": f, "want": len(wants[f]), "got": len(got), "batch": gz.DeviceStreamGunzip.BATCH})
    rec["seconds"] = round(time.time() - t0, 1)
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    s = json.dumps(rec, indent=1)
    if out:
        open(out, "w").write(s)
    print(s)
    sys.exit(1 if rec["mismatches"] or rec["disagreements"] or rec["errors"] else 0)


if __name__ == "__main__":
    main()
