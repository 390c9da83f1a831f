"""ONE DEFLATE stream inflated on the device (C ABI rd_gz_stream_inflate, csrc/rd_inflate_stream.hpp; RD_DEVICE_INFLATE=members keeps the host's) and
the reader path built on it (data_loader/device_reader.py: DeviceFeeder._run_stream).

What it replaces for plain .gz files - the format sequencers write: gzip.open(path, 'rt') of the reference
(data_loader/seq_encoder.py:21-39). The property: the text equals zlib's, byte for byte - for every compression level and strategy of
FASTQ text, streams flushed the way pigz flushes them, members that span many batches (state carried on the device), headers with every
optional field, further members and padding behind the first; CRC-32 and ISIZE are checked; what the decoder cannot take (binary
payloads: no block start passes the text filter; 250:1 text: a section outgrows its slot) goes to zlib on the host before anything
has been delivered; damage is an error with zlib's message, never bytes."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flags=0, mem=8, wbits=15):
    """one gzip member built by hand so that header flags and deflate strategies can be chosen (as tests/test_inflate.py)"""
    co = zlib.compressobj(level, zlib.DEFLATED, -wbits, mem, strategy)
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
    return hdr + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff)


def fastq_bytes(n, seed=3):
    from ribodetector_amd import synth
    arena, off, _ = synth.reads_numpy(n, (60, 150), seed=seed)
    b = arena.tobytes()
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        s = b[off[i]:off[i + 1]]
        q = bytes(rng.integers(35, 74, len(s), dtype=np.uint8))
        out.append(b"@read.%d/1 lane=3\n%s\n+\n%s\n" % (i, s, q))
    return b"".join(out)


@pytest.fixture(scope="module")
def fastq():
    return fastq_bytes(60000)          # ~18 MB of text


def _inflate(blob, batch=None, section=None, check=True):
    """the first member of `blob` through gz.DeviceStreamGunzip, batch by batch: (text, states)"""
    from ribodetector_amd import gz
    dev = torch.device(DEV)
    dg = gz.DeviceStreamGunzip(dev, torch.cuda.Stream(dev))
    if batch:
        dg.BATCH = batch
    if section:
        dg.SECTION = section
    buf = np.frombuffer(blob, dtype=np.uint8)
    pos, first, out, states = 0, gz.gzip_header_len(blob) * 8, [], []
    while True:
        data = min(dg.BATCH, len(buf) - pos)
        valid = min(data + dg.SLACK, len(buf) - pos)
        at_eof = pos + valid >= len(buf)
        src = torch.from_numpy(buf[pos:pos + valid].copy()).pin_memory()
        tx = torch.empty(dg.text_cap(data), dtype=torch.uint8, device=dev)
        r = dg.finish(dg.submit(src, valid, data, first, at_eof, tx))
        states.append(r)
        if r["status"]:
            return None, states
        out.append(tx[: r["n_text"]].cpu().numpy().tobytes())
        if r["final"]:
            end = pos + (r["end_bit"] + 7) // 8
            r["trailer_ok"] = int.from_bytes(blob[end:end + 4], "little") == r["crc"] and int.from_bytes(blob[end + 4:end + 8], "little") == r["total_len"] & 0xffffffff
            assert r["trailer_ok"] or not check
            return b"".join(out), states
        pos += data                     # (at_eof only says that the bytes behind the sections reach the end of the file: the next batch
        first = 0xffffffff              # takes them as its sections)
        assert pos < len(buf)


@pytest.mark.parametrize("level,strategy", [(1, zlib.Z_DEFAULT_STRATEGY), (5, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (6, zlib.Z_FILTERED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)])
def test_fastq_text_equals_zlib(fastq, level, strategy):
    """one batch and many batches (1 MiB of compressed bytes each: the window, the CRC and the next start travel on the device)"""
    blob = member(fastq, level, strategy, flags=8)
    for batch in (None, 1 << 20):
        got, st = _inflate(blob, batch=batch)
        assert got == fastq, (level, strategy, batch, st[-1])
        assert sum(s["n_sections"] for s in st) >= (3 if strategy != zlib.Z_HUFFMAN_ONLY else 1)      # really in parallel
        assert batch is None or len(st) >= 3


def test_sections_of_every_size_and_flushed_streams(fastq):
    """section sizes from 4 KiB (smaller than a block: sections without a block start are taken over by the one before) to 256 KiB;
    streams cut by sync / full flushes the way pigz writes them (empty stored blocks between the pieces)"""
    blob = member(fastq[:6000000], 6)
    for section in (4096, 8192, 65536, 262144):
        got, st = _inflate(blob, section=section)
        assert got == fastq[:6000000], (section, st[-1])
    data = fastq[:5000000]
    for mode, step in ((zlib.Z_SYNC_FLUSH, 131072), (zlib.Z_FULL_FLUSH, 131072), (zlib.Z_SYNC_FLUSH, 3000)):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = b"".join(co.compress(data[i:i + step]) + co.flush(mode) for i in range(0, len(data), step)) + co.flush()
        blob = b"\x1f\x8b\x08\0\0\0\0\0\0\xff" + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))
        assert gzip.decompress(blob) == data
        got, st = _inflate(blob)
        assert got == data, (mode, step, st[-1])
    # small windows (wbits 9) and little memory: other block sizes, same text
    got, st = _inflate(member(data, 6, mem=1, wbits=9))
    assert got == data, st[-1]


def test_what_the_decoder_cannot_take_is_reported_not_guessed(fastq):
    """a stream whose sections hold no text block start is ONE section's work (it outgrows its slot), damage ends in a status: the
    caller (the reader, below) hands such files to zlib"""
    rng = np.random.default_rng(1)
    got, st = _inflate(member(rng.integers(0, 256, 3 << 20, dtype=np.uint8).tobytes(), 6))
    assert got is None and st[-1]["status"] != 0, st[-1]
    data = b"@read\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n" * 400000
    got, st = _inflate(member(data, 6))
    assert got is None and st[-1]["status"] in (3, 6)
    bad = bytearray(member(fastq[:3000000], 6))
    bad[len(bad) // 2] ^= 0x20
    got, st = _inflate(bytes(bad), check=False)
    assert got is None or (got != fastq[:3000000] and not st[-1]["trailer_ok"])      # (a flipped literal decodes: the CRC-32 catches it)


def _reader_text(path, chunk=20000, stats=None):
    from ribodetector_amd.data_loader import device_reader as dr
    assert dr.device_ingest_kind(path) == "stream"
    return b"".join(c.to_host()[0].tobytes() for c in dr.get_seq_chunks_device(path, chunk_size=chunk, device=DEV, stats=stats))


def test_reader_on_single_stream_gz_equals_the_host_reader(tmp_path, fastq, monkeypatch):
    from ribodetector_amd import gz
    from ribodetector_amd.data_loader import device_reader as dr
    from ribodetector_amd.data_loader import fastx_parser as fx
    monkeypatch.setenv("RD_DEVICE_INFLATE", "stream")
    a, b = fastq[:9000000], fastq[9000000:]          # (members cut anywhere: their texts concatenate)
    cases = {"plain_member": member(fastq, 6), "every_header_field": member(fastq, 5, flags=4 | 8 | 16 | 2),
             "members_and_padding": member(a, 6, flags=8) + member(b"", 6) + member(b, 9) + bytes(512),
             "python_gzip": gzip.compress(fastq, 6), "tiny": member(b"@r\nACGT\n+\nIIII\n", 6), "empty": member(b"", 6)}
    for name, blob in cases.items():
        p = str(tmp_path / (name + ".fastq.gz"))
        open(p, "wb").write(blob)
        want = gzip.decompress(blob)
        for first, batch in ((None, None), (1 << 18, 1 << 20)):
            if first:
                monkeypatch.setattr(dr.DeviceFeeder, "FIRST", first)
                monkeypatch.setattr(gz.DeviceStreamGunzip, "BATCH", batch)
            st = {}
            assert _reader_text(p, stats=st) == want, (name, first)
            assert "fallback" not in st["feeder"], (name, st)
            if name == "members_and_padding":          # (round 6: every member on the device, not only the first)
                assert st["feeder"]["members"] == 3, st["feeder"]
        monkeypatch.undo()
        monkeypatch.setenv("RD_DEVICE_INFLATE", "stream")
    # chunk for chunk what the host reader delivers
    p = str(tmp_path / "plain_member.fastq.gz")
    dev = [(c.n, c.to_host()) for c in dr.get_seq_chunks_device(p, chunk_size=25000, first_chunk=1000, device=DEV)]
    monkeypatch.setenv("RD_DEVICE_INFLATE", "0")
    host = list(fx.get_seq_chunks(p, chunk_size=25000, first_chunk=1000))
    assert [n for n, _ in dev] == [len(c.seq_len) for c in host]
    for (n, (text, rs, so, sl)), c in zip(dev, host):
        assert text.tobytes() == c.buf[c.rec_start[0]:c.rec_start[-1]].tobytes() and np.array_equal(sl, c.seq_len) and np.array_equal(so, c.seq_off)


def test_reader_hands_unsuitable_and_damaged_files_to_zlib(tmp_path, fastq, monkeypatch):
    """binary-ish payloads and 250:1 text fall back to the host before anything is delivered (same text); damage is an error"""
    monkeypatch.setenv("RD_DEVICE_INFLATE", "stream")
    rep = b"@read\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n" * 300000
    small = fastq_bytes(6000, seed=5)
    for name, blob, want in (("rep", member(rep, 6), rep), ("stored", member(small, 0), small), ("fixed", member(small, 6, zlib.Z_FIXED), small)):
        p = str(tmp_path / (name + ".fastq.gz"))
        open(p, "wb").write(blob)
        st = {}
        assert _reader_text(p, stats=st) == want, name
    good = member(fastq_bytes(12000, seed=6), 6)
    for name, blob, msg in (("flipped", good[:len(good) // 2] + bytes([good[len(good) // 2] ^ 0x41]) + good[len(good) // 2 + 1:], "."),
                            ("crc", good[:-8] + bytes([good[-8] ^ 1]) + good[-7:], "CRC check failed|incorrect data check"),
                            ("isize", good[:-4] + struct.pack("<I", 3999), "Incorrect length|incorrect length"),
                            ("truncated", good[:-3000], "ended before the end-of-stream marker|incomplete"),
                            ("garbage_tail", good + b"not a gzip member at all", ".")):
        p = str(tmp_path / (name + ".fastq.gz"))
        open(p, "wb").write(blob)
        with pytest.raises(ValueError, match=msg):
            _reader_text(p)


def test_c_abi_argument_errors():
    from ribodetector_amd import _native as N
    L = N.lib()
    t = torch.zeros(1 << 16, dtype=torch.uint8, device=DEV)
    assert L.rd_gz_stream_inflate(None, 0, 0, 0, 16384, 1 << 16, 0, None, 0, 0, None, None, None, 0, None, None, 0, None) != 0 and b"null" in L.rd_last_error()
    assert L.rd_gz_stream_inflate(N.ptr(t), 1 << 16, 1 << 16, 1 << 15, 16384, 1 << 16, 0, None, 0, 1, None, N.ptr(t), N.ptr(t), 1 << 16, N.ptr(t), N.ptr(t), 1 << 16, None) != 0
    assert L.rd_gz_stream_workspace_bytes(-1, 16384, 1 << 16, 0) == 0 and L.rd_gz_stream_workspace_bytes(1 << 20, 16384, 1 << 16, 1 << 23) > (1 << 20) // 16384 * (1 << 17)


def test_reader_resumes_on_the_host_behind_a_batch_it_cannot_take(tmp_path, fastq, monkeypatch):
    """FASTQ, then 36 MB of one record repeated (250:1: a section outgrows its slot), then FASTQ again - ONE member. The batches in
    front are decoded on the device and delivered; from the batch that fails on, zlib continues at that batch's first bit (the
    compressed bytes shifted to a byte boundary) with the window the device left as its dictionary; trailer checked; same text"""
    from ribodetector_amd import gz
    from ribodetector_amd.data_loader import device_reader as dr
    monkeypatch.setenv("RD_DEVICE_INFLATE", "stream")
    monkeypatch.setattr(dr.DeviceFeeder, "FIRST", 1 << 18)
    monkeypatch.setattr(gz.DeviceStreamGunzip, "BATCH", 1 << 20)
    rep = b"@read\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n" * 350000
    data = fastq + rep + fastq_bytes(8000, seed=9)
    for flags, tail in ((0, b""), (8, member(b"@x\nAC\n+\nII\n", 6) + bytes(100))):
        p = str(tmp_path / "mixed.fastq.gz")
        open(p, "wb").write(member(data, 6, flags=flags) + tail)
        st = {}
        got = _reader_text(p, chunk=50000, stats=st)
        assert got == data + (b"@x\nAC\n+\nII\n" if tail else b""), (len(got), len(data))
        assert "resumed_on_host" in st["feeder"] and "fallback" not in st["feeder"] and st["feeder"]["batches"] >= 3, st["feeder"]
    # damage behind the point of resumption is zlib's to report
    blob = bytearray(member(data, 6))
    blob[-3000] ^= 0x10
    open(p, "wb").write(bytes(blob))
    with pytest.raises(ValueError):
        _reader_text(p, chunk=50000)
