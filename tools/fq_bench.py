#!/usr/bin/env python
"""The device-ingest kernels on one 2^20-record batch of FASTQ text (218 B per record, the text the CLI's e2e legs read): time by
events on the launch stream, achieved GB/s against the 8 TB/s of HBM (PCIe for the copy kernels).
    rd_fastq_index   (begin, count, scan, fill, check, verdict, sample)   algorithmic bytes: the text once + 4 B per line
    rd_fastq_gather  (the whole batch into a chunk)                        the text read + written, 20 B of index per record
    rd_select_pack   (both label files)                                    the text read + written
    rd_copy_bytes    pinned -> HBM and back                                PCIe
    rd_gz_stream_inflate  (the same text as ONE zlib level-6 stream)       text out; per-kernel split in the rocprofv3 table
python tools/fq_bench.py [--records 1048576] [--out file.json] [--no-stream]"""
import argparse
import ctypes as C
import json
import os
import sys
import zlib

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np      # noqa: E402
import torch            # noqa: E402
from ribodetector_amd import _native as N, gz, synth      # noqa: E402
from ribodetector_amd.data_loader import device_reader as dr      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=1 << 20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-stream", action="store_true")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    n = a.records
    arena, off, lens = synth.reads_torch(n, 100, seed=1, device=dev)
    text = synth.fastq_image_torch(arena, off, lens, mate=1)
    tb = int(text.numel())
    st = torch.cuda.current_stream(dev)
    L = N.lib()
    ix = dr.FastqIndexer(dev, st)

    def timed(fn, reps=a.reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    out = {"records": n, "text_bytes": tb, "hbm_peak_GBps": 8000.0, "kernels": {}}

    def put(name, ms, nbytes, what, peak=8000.0):
        out["kernels"][name] = {"ms": ms, "GBps": nbytes / ms / 1e6, "frac_of_peak": nbytes / ms / 1e6 / peak, "algorithmic_bytes": what}
    buf = ix.alloc_text(tb)
    buf[dr.PAD:dr.PAD + tb] = text
    batch = [None]

    def do_index():
        ix.prev = None
        batch[0] = ix.index(buf, dr.PAD, dr.PAD + tb, final=True, chain=False)
    ms = timed(do_index)
    b = ix.finish(batch[0])
    assert b.n == n and b.status == 0, (b.n, b.status)
    put("rd_fastq_index", ms, tb + 16 * n, "text once + 4 B per line")
    ms = timed(lambda: ix.gather([(b, 0, n)]))
    put("rd_fastq_gather", ms, 2 * tb + 20 * n, "text read + written, 20 B of index per record")
    ch = ix.gather([(b, 0, n)])
    lab = (torch.arange(n, device=dev) % 50 == 0).to(torch.int8)
    ds = gz.DeviceSelect(dev)
    ms = timed(lambda: [ds.pack_selected(ch.dev[0], ch.dev[3], lab, v, slot=v) for v in (0, 1)])
    put("rd_select_pack (both label files)", ms, 2 * tb + 2 * 9 * n, "text read + written, 9 B of index per record and file")
    # FASTA (rd_fasta_index: the batch re-written as header / joined upper-case sequence): the same reads as 60-column FASTA
    nfa = min(n, 1 << 18)
    a_np, o_np = arena.cpu().numpy(), off.cpu().numpy()
    fa = b"".join(b">read%d\n" % i + b"\n".join(a_np[o_np[i]:o_np[i] + 100].tobytes()[q:q + 60] for q in (0, 60)) + b"\n" for i in range(nfa))
    fa = fa * (n // nfa)
    fb = len(fa)
    fx_ = dr.FastaIndexer(dev, st)
    fbuf = fx_.alloc_text(fb)
    fbuf[dr.PAD:dr.PAD + fb] = torch.from_numpy(np.frombuffer(fa, dtype=np.uint8).copy()).to(dev)
    fbatch = [None]

    def do_fa():
        fx_.prev = None
        fbatch[0] = fx_.index(fbuf, dr.PAD, dr.PAD + fb, final=True, chain=False)
    ms = timed(do_fa)
    fbt = fx_.finish(fbatch[0])
    assert fbt.n == n and fbt.status == 0, (fbt.n, fbt.status)
    put("rd_fasta_index (60-column FASTA, %d lines)" % (3 * n), ms, 2 * fb, "text read + re-written")
    ms = timed(lambda: fx_.gather([(fbt, 0, n)]))
    put("rd_fasta_gather", ms, 2 * fbt.norm_end + 20 * n, "normalised text read + written, 20 B of index per record")
    del fbuf, fbt, fbatch
    pin = torch.empty(tb, dtype=torch.uint8, pin_memory=True)
    pin.copy_(text)
    dst = torch.empty(tb, dtype=torch.uint8, device=dev)
    ms = timed(lambda: N.copy_bytes(dst, pin, tb, st))
    put("rd_copy_bytes pinned -> HBM", ms, tb, "the bytes (PCIe 5 x16: 64 GB/s)", peak=64.0)
    ms = timed(lambda: N.copy_bytes(pin, dst, tb, st))
    put("rd_copy_bytes HBM -> pinned", ms, tb, "the bytes (PCIe 5 x16: 64 GB/s)", peak=64.0)
    for wgs in (1, 2, 4, 16, 32):      # (default 8: above) workgroups of 1,024 threads - how few CUs keep the link busy
        ms = timed(lambda: N.copy_bytes(dst, pin, tb, st, workgroups=wgs))
        put("rd_copy_bytes pinned -> HBM, %d workgroups" % wgs, ms, tb, "the bytes", peak=64.0)
        ms = timed(lambda: N.copy_bytes(pin, dst, tb, st, workgroups=wgs))
        put("rd_copy_bytes HBM -> pinned, %d workgroups" % wgs, ms, tb, "the bytes", peak=64.0)
    ms = timed(lambda: dst.copy_(pin, non_blocking=True))
    put("hipMemcpyAsync pinned -> HBM (SDMA, for comparison)", ms, tb, "the bytes", peak=64.0)
    if not a.no_stream:
        raw = text.cpu().numpy().tobytes()
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        blob = co.compress(raw) + co.flush()
        dsg = gz.DeviceStreamGunzip(dev, st)
        dsg.BATCH = max(dsg.BATCH, (len(blob) + dsg.SECTION) // dsg.SECTION * dsg.SECTION)
        src = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).pin_memory()
        tx = torch.empty(dsg.text_cap(len(blob)), dtype=torch.uint8, device=dev)
        hl = gz.gzip_header_len(blob)
        res = [None]

        def do_stream():
            dsg.carry = dsg.win = None
            res[0] = dsg.submit(src, len(blob), len(blob), hl * 8, True, tx)
        ms = timed(do_stream, reps=5)
        r = dsg.finish(res[0])
        assert r["status"] == 0 and r["final"] and r["n_text"] == tb and r["crc"] == (zlib.crc32(raw) & 0xffffffff), r
        put("rd_gz_stream_inflate (zlib level 6, one batch of %d sections)" % r["n_sections"], ms, tb, "text produced")
        out["stream"] = {"compressed_bytes": len(blob), "sections": r["n_sections"], "section_bytes": dsg.SECTION}
        # round 6: the second half of the same stream as a RANGE - decoded without the text in front of it (rd_gz_range_decode: symbols and the
        # range's map), then symbols -> bytes with the window the first half leaves (rd_gz_range_resolve); the text must be zlib's
        half = (len(blob) // 2) // dsg.SECTION * dsg.SECTION
        drg = gz.DeviceRangeGunzip(dev, st)
        drg.BATCH = dsg.BATCH
        src2 = torch.from_numpy(np.frombuffer(blob[half:], dtype=np.uint8).copy()).pin_memory()
        tk = [None]

        def do_range():
            drg.carry = drg.map = None
            tk[0] = drg.submit(src2, len(blob) - half, len(blob) - half, gz.GZS_SEARCH, True)
        ms = timed(do_range, reps=5)
        r2 = drg.finish(tk[0])
        assert r2["status"] == 0 and r2["final"], r2
        n2 = r2["n_text"]
        drg.carry = drg.map = None
        first = drg.submit(torch.from_numpy(np.frombuffer(blob[:half + dsg.SLACK], dtype=np.uint8).copy()).pin_memory(), min(len(blob), half + dsg.SLACK), half, hl * 8, False)
        r1 = drg.finish(first)          # (the first half, for its map: the window in front of the second)
        assert r1["status"] == 0 and half * 8 + r2["first_start"] == r1["next_start"] and r1["n_text"] + n2 == tb, (r1, r2)
        win = gz.apply_map(drg.map.cpu().numpy(), np.zeros(32768, dtype=np.uint8))
        put("rd_gz_range_decode (the stream's second half, window unknown: %d sections)" % r2["n_sections"], ms, n2, "text produced, as 16-bit symbols")
        win_dev = torch.from_numpy(win).to(dev)
        rstate = torch.zeros(8, dtype=torch.int64, device=dev)
        out8 = torch.empty(n2 + 64, dtype=torch.uint8, device=dev)

        def do_resolve():
            rstate.zero_()
            drg.resolve(tk[0]["sym"], n2, win_dev, 32768, out8, rstate)
        ms = timed(do_resolve, reps=5)
        assert out8[:n2].cpu().numpy().tobytes() == raw[tb - n2:], "range text differs from zlib's"
        put("rd_gz_range_resolve (symbols -> bytes + CRC-32)", ms, 3 * n2, "2 B symbol read + 1 B written per text byte")
    s = json.dumps(out)
    if a.out:
        open(a.out, "w").write(json.dumps(out, indent=1))
    print(s)


if __name__ == "__main__":
    main()
