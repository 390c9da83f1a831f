#!/usr/bin/env python
"""Soak of the single-stream gzip decoder on the device (csrc/rd_inflate_stream.hpp behind data_loader/device_reader.py, the DEFAULT for
.gz FASTQ on one rank): random FASTQ-like files - read lengths from 30 bp to 30 kb, header styles of four platforms, qualities from
2-level bins to uniform noise, LF / CR LF, N runs, low-complexity stretches - compressed with every zlib level / strategy / memLevel /
window size, cut by flushes, glued from several members, read through the device reader with random batch sizes. THE property: the
text delivered equals gzip.decompress()'s (line ends as LF), whichever way it went (device, host fallback before the first batch, host resume behind a
failed batch); the record counts how often each way was taken.        python tools/gzs_soak.py <seconds> [seed] [out.json]"""
import gzip
import json
import os
import struct
import sys
import time
import zlib

sys.path.insert(0, os.getcwd())
import numpy as np
import torch

from ribodetector_amd import gz
from ribodetector_amd.data_loader import device_reader as dr

DEV = "cuda:0"


def fastq_text(rng):
    n = int(rng.choice([1, 3, 50, 2000, 20000, 60000]))
    kind = int(rng.integers(0, 4))
    lo, hi = [(30, 50), (100, 100), (60, 300), (1000, 30000)][int(rng.integers(0, 4))]
    if hi > 1000:
        n = min(n, 800)
    qual = int(rng.integers(0, 4))
    eol = b"\r\n" if rng.random() < 0.15 else b"\n"
    out = []
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    for i in range(n):
        L = int(rng.integers(lo, hi + 1))
        if rng.random() < 0.05:
            s = np.full(L, acgt[int(rng.integers(0, 4))], dtype=np.uint8)       # homopolymer
        else:
            s = acgt[rng.integers(0, 4, L)]
            if rng.random() < 0.1:
                a = int(rng.integers(0, L)); s[a:a + int(rng.integers(1, 40))] = ord("N")
        if qual == 0:
            q = np.array([35, 44, 58, 70], dtype=np.uint8)[rng.integers(0, 4, L)]       # binned
        elif qual == 1:
            q = rng.integers(33, 74, L, dtype=np.uint8)
        elif qual == 2:
            q = np.full(L, 73, dtype=np.uint8)
        else:
            q = np.clip(70 - np.abs(rng.normal(0, 6, L)).astype(np.int64) - np.arange(L) * 20 // max(L, 1), 35, 73).astype(np.uint8)
        if kind == 0:
            h = b"@A00123:45:HXXXXDSXY:%d:%d:%d:%d 1:N:0:ACGTACGT+TTGCAATG" % (1 + i % 4, 1101 + i // 1000, 1000 + i * 7 % 30000, 1000 + i * 13 % 30000)
        elif kind == 1:
            h = b"@SRR1234567.%d %d length=%d" % (i + 1, i + 1, L)
        elif kind == 2:
            h = b"@%08x-%04x-%04x-%04x-%012x runid=%040x read=%d ch=%d" % (int(rng.integers(0, 1 << 32)), i & 0xffff, 0x4abc, 0x8def, i * 977, 12345678901234567890, i, i % 512)
        else:
            h = b"@r%d" % i
        plus = b"+" + (h[1:] if rng.random() < 0.02 else b"")
        out.append(h + eol + s.tobytes() + eol + plus + eol + q.tobytes() + eol)
    return b"".join(out)


def deflate_member(data, rng):
    level = int(rng.integers(1, 10))
    strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY, zlib.Z_FIXED][int(rng.integers(0, 7))]
    if rng.random() < 0.03:
        level = 0
    mem = int(rng.integers(1, 10))
    wbits = int(rng.integers(9, 16))
    flags = int(rng.choice([0, 8, 8, 4 | 8 | 16 | 2, 16]))
    co = zlib.compressobj(level, zlib.DEFLATED, -wbits, mem, strategy)
    if rng.random() < 0.3 and len(data) > 10:
        step = int(rng.choice([997, 32768, 131072, 1 << 20]))
        mode = zlib.Z_FULL_FLUSH if rng.random() < 0.5 else zlib.Z_SYNC_FLUSH
        body = b"".join(co.compress(data[i:i + step]) + co.flush(mode) for i in range(0, len(data), step)) + co.flush()
    else:
        body = co.compress(data) + co.flush()
    hdr = b"\x1f\x8b\x08" + bytes([flags]) + b"\0\0\0\0\x02\xff"
    if flags & 4:
        hdr += struct.pack("<H", 7) + b"EXTRA!!"
    if flags & 8:
        hdr += b"reads_1.fastq\0"
    if flags & 16:
        hdr += b"a comment\0"
    if flags & 2:
        hdr += struct.pack("<H", zlib.crc32(hdr) & 0xffff)
    return hdr + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff), dict(level=level, strategy=strategy, mem=mem, wbits=wbits, flags=flags)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    out = sys.argv[3] if len(sys.argv) > 3 else None
    rng = np.random.default_rng(seed)
    d = "/dev/shm/gzs_soak_%d" % seed
    os.makedirs(d, exist_ok=True)
    p = os.path.join(d, "x.fastq.gz")
    rec = {"files": 0, "device": 0, "fallback": 0, "resumed_on_host": 0, "mismatches": 0, "errors": 0, "text_bytes": 0, "batches": 0, "fallback_reasons": {}, "fallback_examples": [], "seed": seed}
    FIRST0, BATCH0 = dr.DeviceFeeder.FIRST, gz.DeviceStreamGunzip.BATCH
    t0 = time.time()
    while time.time() - t0 < seconds:
        parts, metas = [], []
        for _ in range(int(rng.choice([1, 1, 1, 2, 3]))):
            m, meta = deflate_member(fastq_text(rng), rng)
            parts.append(m); metas.append(meta)
            if rng.random() < 0.1:
                parts.append(deflate_member(b"", rng)[0])
        blob = b"".join(parts) + (bytes(int(rng.integers(1, 600))) if rng.random() < 0.1 else b"")
        want = gzip.decompress(blob).replace(b"\r\n", b"\n")       # (the reader strips line ends like the reference's parser: CR LF -> LF)
        open(p, "wb").write(blob)
        if rng.random() < 0.6:       # small batches: state carried across many of them
            dr.DeviceFeeder.FIRST = int(rng.choice([1 << 14, 1 << 16, 1 << 18]))
            gz.DeviceStreamGunzip.BATCH = int(rng.choice([1 << 16, 1 << 18, 1 << 20]))
        else:
            dr.DeviceFeeder.FIRST, gz.DeviceStreamGunzip.BATCH = FIRST0, BATCH0
        st = {}
        try:
            assert dr.device_ingest_kind(p) == "stream"
            got = b"".join(c.to_host()[0].tobytes() for c in dr.get_seq_chunks_device(p, chunk_size=int(rng.choice([500, 20000, 200000])), device=DEV, stats=st))
        except BaseException as e:       # noqa: BLE001 - the soak records everything
            rec["errors"] += 1
            print("ERROR", repr(e)[:300], metas, len(blob), flush=True)
            open(os.path.join(d, "bad_%d.gz" % rec["files"]), "wb").write(blob)
            rec["files"] += 1
            continue
        # the reader delivers whole records; a text whose last line lacks its newline is delivered as it is
        if got != want:
            rec["mismatches"] += 1
            print("MISMATCH", len(got), len(want), metas, st.get("feeder"), flush=True)
            open(os.path.join(d, "bad_%d.gz" % rec["files"]), "wb").write(blob)
        f = st.get("feeder", {})
        rec["files"] += 1
        rec["text_bytes"] += len(want)
        rec["batches"] += int(f.get("batches", 0))
        if "fallback" in f:
            rec["fallback"] += 1
            k = str(f["fallback"])[:60]
            rec["fallback_reasons"][k] = rec["fallback_reasons"].get(k, 0) + 1
            if len(rec["fallback_examples"]) < 40:
                rec["fallback_examples"].append({"why": k, "members": metas, "gz_bytes": len(blob), "text_bytes": len(want)})
        elif "resumed_on_host" in f:
            rec["resumed_on_host"] += 1
        else:
            rec["device"] += 1
    rec["seconds"] = round(time.time() - t0, 1)
    if not rec["mismatches"] and not rec["errors"]:
        import shutil
        shutil.rmtree(d)
    print(json.dumps({k: v for k, v in rec.items() if k != "fallback_examples"}))
    if out:
        json.dump(rec, open(out, "w"), indent=1)
    return 1 if rec["mismatches"] or rec["errors"] else 0


if __name__ == "__main__":
    sys.exit(main())
