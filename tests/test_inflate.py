"""The reader's own gzip/DEFLATE decoder (ribodetector_amd/csrc/rd_inflate.h) against zlib/Python's gzip - which is what
the reference reads .gz input with (data_loader/fastx_parser.py, seq_encoder.py:75-92): identical bytes for every block
type and framing feature, and an error (not silent truncation) for damaged files."""
import ctypes as C
import gzip
import struct
import zlib

import numpy as np
import pytest

from ribodetector_amd import _native as N
from ribodetector_amd import synth
from ribodetector_amd.data_loader import fastx_parser as fx


def gunzip(path, cap):
    L = N.host_lib()
    out = np.empty(max(cap, 1), dtype=np.uint8)
    n = C.c_int64(0)
    rc = L.rd_host_gunzip(str(path).encode(), out.ctypes.data, cap, C.byref(n))
    return rc, out[: n.value].tobytes(), L.rd_host_last_error().decode()


def member(data, level=5, strategy=zlib.Z_DEFAULT_STRATEGY, flags=0, mem=8, wbits=15):
    """one gzip member built by hand so that header flags and deflate strategies can be chosen"""
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
    arena, off, _ = synth.reads_numpy(n, (60, 150), seed=seed)
    b = arena.tobytes()
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        s = b[off[i]:off[i + 1]]
        q = bytes(rng.integers(35, 74, len(s), dtype=np.uint8))
        out.append(b"@read.%d/1 lane=3\n%s\n+\n%s\n" % (i, s, q))
    return b"".join(out)


RNG = np.random.default_rng(11)
PAYLOADS = {
    "empty": b"",
    "one": b"A",
    "fastq": fastq_bytes(20000),                                           # ~6 MB: several output blocks, input refills
    "random": RNG.integers(0, 256, 3 << 20, dtype=np.uint8).tobytes(),     # incompressible: stored blocks / long codes
    "zeros": bytes(5 << 20),                                               # distance-1 matches of length 258
    "period3": b"ACG" * 900001,                                            # overlapping copies with distance < 8
    "period7": b"ACGTTGA" * 400000,
    "skewed": bytes(RNG.choice(np.arange(256, dtype=np.uint8), 2 << 20, p=np.r_[0.7, np.full(255, 0.3 / 255)])),  # long Huffman codes
}


@pytest.mark.parametrize("name", list(PAYLOADS))
@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (5, zlib.Z_DEFAULT_STRATEGY),
                                            (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)])
def test_inflate_matches_zlib(tmp_path, name, level, strategy):
    data = PAYLOADS[name]
    p = tmp_path / "x.gz"
    p.write_bytes(member(data, level, strategy))
    assert gzip.decompress(p.read_bytes()) == data                         # the fixture itself is valid gzip
    rc, got, err = gunzip(p, len(data) + 16)
    assert rc == 0, err
    assert got == data


def test_gzip_framing_features(tmp_path):
    a, b, c = PAYLOADS["fastq"][:300000], b"second member\n" * 1000, PAYLOADS["random"][:70000]
    blob = member(a, 5, flags=4 | 8 | 16 | 2) + member(b"", 6) + member(b, 9, flags=8) + bytes(37) + member(c, 1) + bytes(512)
    p = tmp_path / "multi.gz"
    p.write_bytes(blob)
    assert gzip.decompress(blob) == a + b + c                              # Python's gzip: members concatenate, zero padding skipped
    rc, got, err = gunzip(p, len(a + b + c) + 1)
    assert rc == 0, err
    assert got == a + b + c
    # small windows / memory levels change the block structure, not the format
    p.write_bytes(member(PAYLOADS["fastq"], 7, mem=1, wbits=9))
    rc, got, err = gunzip(p, len(PAYLOADS["fastq"]))
    assert rc == 0 and got == PAYLOADS["fastq"], err
    # gzip module / command line produce the same thing
    p.write_bytes(gzip.compress(a, 6))
    assert gunzip(p, len(a))[1] == a


def test_damaged_files_are_errors(tmp_path):
    data = PAYLOADS["fastq"][:2000000]
    good = member(data, 5)
    p = tmp_path / "bad.gz"
    for cut in (5, 11, 200, len(good) // 2, len(good) - 9, len(good) - 8, len(good) - 1):
        p.write_bytes(good[:cut])
        with pytest.raises(Exception):
            gzip.decompress(good[:cut])                                    # the reference's reader fails on these too
        rc, _, err = gunzip(p, len(data) + 16)
        assert rc != 0 and "ended before the end-of-stream marker" in err, (cut, err)
    crc_bad = bytearray(good)
    crc_bad[-6] ^= 0x10
    p.write_bytes(bytes(crc_bad))
    rc, _, err = gunzip(p, len(data) + 16)
    assert rc != 0 and "CRC check failed" in err
    size_bad = bytearray(good)
    size_bad[-2] ^= 0x01
    p.write_bytes(bytes(size_bad))
    rc, _, err = gunzip(p, len(data) + 16)
    assert rc != 0 and "Incorrect length" in err
    rng = np.random.default_rng(5)
    detected = 0
    for _ in range(40):                                                    # a flipped bit anywhere in the body is caught
        x = bytearray(good)
        x[int(rng.integers(10, len(good) - 8))] ^= 1 << int(rng.integers(0, 8))
        p.write_bytes(bytes(x))
        rc, got, err = gunzip(p, len(data) + 1024)
        detected += rc != 0
        assert rc != 0 or got == data
    assert detected == 40
    p.write_bytes(good + b"trailing garbage")
    rc, _, err = gunzip(p, len(data) + 16)
    assert rc != 0 and "Not a gzipped file" in err
    p.write_bytes(good)
    rc, _, err = gunzip(p, len(data) - 1)                                  # caller buffer too small
    assert rc != 0 and "exceeds the buffer" in err


def test_random_deflate_streams_never_crash(tmp_path):
    """arbitrary bytes after a valid header: any outcome but a crash / out-of-bounds access is fine"""
    rng = np.random.default_rng(9)
    p = tmp_path / "fuzz.gz"
    for i in range(300):
        body = rng.integers(0, 256, int(rng.integers(1, 4000)), dtype=np.uint8).tobytes()
        p.write_bytes(b"\x1f\x8b\x08\0\0\0\0\0\x02\xff" + body)
        rc, got, err = gunzip(p, 1 << 20)
        assert rc in (0, -1)
    good = member(PAYLOADS["fastq"][:100000], 6)
    for i in range(300):                                                   # mutate a valid stream: exercises deeper states
        x = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            x[int(rng.integers(10, len(x)))] = int(rng.integers(0, 256))
        p.write_bytes(bytes(x))
        rc, got, err = gunzip(p, 1 << 20)
        assert rc in (0, -1)


def test_reader_reports_truncated_gzip(tmp_path):
    arena, off, _ = synth.reads_numpy(50000, 100, seed=8)
    fq = tmp_path / "r.fq.gz"
    synth.write_fastq(str(fq), arena, off, 1)
    whole = fq.read_bytes()
    assert sum(len(c.seq_len) for c in fx.get_seq_chunks(str(fq), 8192)) == 50000
    cut = tmp_path / "cut.fq.gz"
    cut.write_bytes(whole[: len(whole) * 2 // 3])
    with pytest.raises(ValueError, match="ended before the end-of-stream marker"):
        for _ in fx.get_seq_chunks(str(cut), 8192):
            pass


def test_match_must_not_reach_into_previous_member(tmp_path):
    """a distance that points before the start of its own gzip member is invalid (zlib: 'invalid distance too far back'),
    even when an earlier member left bytes in the window"""
    import struct
    import zlib
    first = member(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ" * 40, 5)
    # second member: fixed-Huffman block whose first symbol is a match (length 3, distance 1) - there is no history yet
    bits = []

    def put(v, n, msb=False):
        for i in (range(n - 1, -1, -1) if msb else range(n)):
            bits.append((v >> i) & 1)
    put(1, 1)
    put(1, 2)                    # BFINAL, BTYPE = 01 (fixed)
    put(0b0000001, 7, msb=True)  # length code 257 (length 3)
    put(0, 5, msb=True)          # distance code 0 (distance 1)
    put(0, 7, msb=True)          # end of block
    while len(bits) % 8:
        bits.append(0)
    body = bytes(sum(b << k for k, b in enumerate(bits[i:i + 8])) for i in range(0, len(bits), 8))
    with pytest.raises(zlib.error):
        zlib.decompressobj(-15).decompress(body)
    bad = first + b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff" + body + struct.pack("<II", 0, 3)
    p = tmp_path / "cross.gz"
    p.write_bytes(bad)
    rc, _, err = gunzip(p, 1 << 16)
    assert rc != 0 and "distance" in err.lower(), err


# ---- the parallel decoder (csrc/rd_pgzip.h): same bytes and same errors as the sequential one -----------------------------------
def pgunzip(path, cap, threads=4, section=65536):
    L = N.host_lib()
    out = np.empty(max(cap, 1), dtype=np.uint8)
    n = C.c_int64(0)
    st = (C.c_int64 * 4)()
    rc = L.rd_host_gunzip_parallel(str(path).encode(), out.ctypes.data, cap, C.byref(n), threads, section, st)
    return rc, out[: n.value].tobytes(), L.rd_host_last_error().decode(), {"used": st[0], "dropped": st[1], "batches": st[2], "fell_back": st[3]}


@pytest.mark.parametrize("level", [1, 5, 6, 9])
@pytest.mark.parametrize("threads,section", [(2, 16384), (4, 65536), (3, 300000), (8, 1 << 20)])
def test_parallel_decoder_matches_zlib_on_fastq(tmp_path, level, threads, section):
    data = PAYLOADS["fastq"]
    p = tmp_path / "x.gz"
    p.write_bytes(member(data, level))
    rc, got, err, st = pgunzip(p, len(data) + 16, threads, section)
    assert rc == 0, err
    assert got == data
    if section in (65536, 300000):   # several sections were really decoded with an unknown window and stitched; nothing fell back
        assert st["used"] >= 4 and st["fell_back"] == 0, st
    # (16 KiB sections are shorter than a deflate block: no block start within a batch, the sequential decoder takes over - same bytes)


@pytest.mark.parametrize("name", list(PAYLOADS))
def test_parallel_decoder_on_every_payload(tmp_path, name):
    """binary payloads have no block start the search accepts (it wants text): the sequential decoder finishes them - same bytes"""
    data = PAYLOADS[name]
    p = tmp_path / "x.gz"
    for level, strategy in ((1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY)):
        p.write_bytes(member(data, level, strategy))
        rc, got, err, st = pgunzip(p, len(data) + 16, 4, 32768)
        assert rc == 0, (name, level, err)
        assert got == data, (name, level, st)


def test_parallel_decoder_framing_and_windows(tmp_path):
    a, b, c = PAYLOADS["fastq"][:3000000], b"second member\n" * 1000, PAYLOADS["random"][:70000]
    blob = member(a, 5, flags=4 | 8 | 16 | 2) + member(b"", 6) + member(b, 9, flags=8) + bytes(37) + member(c, 1) + bytes(512)
    p = tmp_path / "multi.gz"
    p.write_bytes(blob)
    rc, got, err, st = pgunzip(p, len(a + b + c) + 1, 4, 65536)
    assert rc == 0, err
    assert got == a + b + c and st["used"] >= 4          # first member in parallel, the others by the sequential decoder
    # a text that refers back across every section boundary (long repeats: the window really is needed), small windows, tiny files
    rep = (PAYLOADS["fastq"][:20000] * 150)
    for data, kw in ((rep, {}), (PAYLOADS["fastq"], {"mem": 1, "wbits": 9}), (b"", {}), (b"A", {}), (b"@r\nACGT\n+\nIIII\n", {})):
        p.write_bytes(member(data, 6, **kw))
        rc, got, err, st = pgunzip(p, len(data) + 16, 4, 40000)
        assert rc == 0 and got == data, (err, st)


def test_parallel_decoder_bounds_its_buffers_on_extremely_compressible_text(tmp_path):
    """a 250:1 text: a section would decode to more symbols than the decoder allows itself - the sequential decoder takes over"""
    data = b"@read\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n" * 1100000
    p = tmp_path / "rep.gz"
    p.write_bytes(member(data, 6))
    assert p.stat().st_size * 150 < len(data)
    rc, got, err, st = pgunzip(p, len(data) + 16, 4, 131072)
    assert rc == 0 and got == data, (err, st)
    assert st["fell_back"] == 1


def test_parallel_decoder_reports_damage_like_the_sequential_one(tmp_path):
    data = PAYLOADS["fastq"][:2000000]
    good = member(data, 5)
    p = tmp_path / "bad.gz"
    cases = [good[:cut] for cut in (5, 11, 200, len(good) // 2, len(good) - 9, len(good) - 8, len(good) - 1)]
    x = bytearray(good); x[-6] ^= 0x10; cases.append(bytes(x))               # CRC
    x = bytearray(good); x[-2] ^= 0x01; cases.append(bytes(x))               # ISIZE
    cases.append(good + b"trailing garbage")
    rng = np.random.default_rng(6)
    for _ in range(30):
        x = bytearray(good)
        x[int(rng.integers(10, len(good) - 8))] ^= 1 << int(rng.integers(0, 8))
        cases.append(bytes(x))
    for blob in cases:
        p.write_bytes(blob)
        rc0, got0, err0 = gunzip(p, len(data) + 1024)
        rc1, got1, err1, st = pgunzip(p, len(data) + 1024, 4, 65536)
        assert rc1 == rc0 and err1 == err0, (len(blob), err0, err1, st)
        if rc0 == 0:
            assert got1 == got0
        else:   # what precedes the error is handed out too; how far that is depends on the decoder's buffering, the bytes do not
            assert got0.startswith(got1) or got1.startswith(got0)


def test_parallel_decoder_fuzz_never_crashes(tmp_path):
    rng = np.random.default_rng(10)
    p = tmp_path / "fuzz.gz"
    good = member(PAYLOADS["fastq"][:400000], 6)
    for i in range(150):
        x = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            x[int(rng.integers(10, len(x)))] = int(rng.integers(0, 256))
        p.write_bytes(bytes(x))
        rc, got, err, st = pgunzip(p, 1 << 21, 3, 20000)
        rc0, got0, err0 = gunzip(p, 1 << 21)
        assert rc == rc0 and (rc0 != 0 or got == got0), (i, err, err0)


def test_reader_uses_the_parallel_decoder_for_large_files(tmp_path, monkeypatch):
    """records through the reader with the parallel decoder forced on a small file (thresholds are test knobs)"""
    arena, off, _ = synth.reads_numpy(60000, 100, seed=8)
    fq = tmp_path / "r.fq.gz"
    synth.write_fastq(str(fq), arena, off, 1)
    monkeypatch.setenv("RD_GZ_THREADS", "0")
    seq = [(c.buf[: c.rec_start[-1]].tobytes(), c.seq_len.copy()) for c in fx.get_seq_chunks(str(fq), 8192)]
    monkeypatch.setenv("RD_GZ_THREADS", "3")
    monkeypatch.setenv("RD_GZ_PARALLEL_MIN", "0")
    monkeypatch.setenv("RD_GZ_SECTION", "50000")
    par = [(c.buf[: c.rec_start[-1]].tobytes(), c.seq_len.copy()) for c in fx.get_seq_chunks(str(fq), 8192)]
    assert len(seq) == len(par) and all(a[0] == b[0] and (a[1] == b[1]).all() for a, b in zip(seq, par))
    cut = tmp_path / "cut.fq.gz"
    cut.write_bytes(fq.read_bytes()[: fq.stat().st_size * 2 // 3])
    with pytest.raises(ValueError, match="ended before the end-of-stream marker"):
        for _ in fx.get_seq_chunks(str(cut), 8192):
            pass


def bgzf(data, block=60000):
    """BGZF (SAM/BAM spec 4.1): members of <= 64 KiB with their size in a 'BC' extra subfield, closed by the empty EOF member"""
    out = []
    for i in list(range(0, len(data), block)) + [None]:
        piece = b"" if i is None else data[i:i + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = co.compress(piece) + co.flush()
        size = 18 + len(body) + 8
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<HccHH", 6, b"B", b"C", 2, size - 1) + body +
                   struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece)))
    return b"".join(out)


def test_members_that_carry_their_size_are_decoded_in_parallel(tmp_path):
    """BGZF files and this build's own .gz outputs ('R','D' subfield, 32-bit size): the members are walked without decoding"""
    data = PAYLOADS["fastq"]
    p = tmp_path / "x.gz"
    blob = bgzf(data)
    assert gzip.decompress(blob) == data
    p.write_bytes(blob)
    rc, got, err, st = pgunzip(p, len(data) + 16, 4, 65536)
    assert rc == 0 and got == data, err
    assert st["used"] == len(data) // 60000 + 2 and st["fell_back"] == 0, st      # every member, incl. the empty EOF member
    rc, got, err = gunzip(p, len(data) + 16)                                      # and the sequential decoder reads the same
    assert rc == 0 and got == data
    # this build's writer
    arena, off, _ = synth.reads_numpy(30000, 100, seed=4)
    src = tmp_path / "in.fq"
    synth.write_fastq(str(src), arena, off, 1)
    chunk = next(fx.get_seq_chunks(str(src), 30000))
    w = fx.open_for_write(str(tmp_path / "out.fq.gz"))
    for _ in range(3):                                                            # > 4 MiB of records: several members
        w.write_selected(chunk, np.zeros(30000, dtype=np.int8), 0)
    w.close()
    out = (tmp_path / "out.fq.gz").read_bytes()
    want = src.read_bytes() * 3
    assert gzip.decompress(out) == want                                           # any gzip reader skips the extra field
    assert out[3] == 4 and out[12:16] == b"RD\x04\x00"
    rc, got, err, st = pgunzip(tmp_path / "out.fq.gz", len(want) + 16, 4, 1 << 20)
    assert rc == 0 and got == want and st["used"] >= 3 and st["fell_back"] == 0, (err, st)
    # a member without the subfield after sized ones, and zero padding between members: the sequential decoder continues
    tail = b"plain member\n" * 5000
    p.write_bytes(blob + bytes(100) + member(tail, 6))
    rc, got, err, st = pgunzip(p, len(data) + len(tail) + 16, 4, 65536)
    assert rc == 0 and got == data + tail, err
    # damage inside a sized member, a size that lies, a truncated file: the same error as the sequential decoder
    for mutate in (lambda b: b[:40000] + bytes([b[40000] ^ 0x55]) + b[40001:], lambda b: b[:16] + b"\xff\xff" + b[18:], lambda b: b[: len(b) // 2],
                   lambda b: b[:-20]):
        p.write_bytes(mutate(blob))
        rc0, got0, err0 = gunzip(p, len(data) + 16)
        rc1, got1, err1, st = pgunzip(p, len(data) + 16, 4, 65536)
        assert rc1 == rc0 and err1 == err0, (err0, err1, st)
        assert got0.startswith(got1) or got1.startswith(got0)


def test_parallel_decoder_on_flushed_streams_like_pigz(tmp_path):
    """pigz compresses 128 KiB pieces and joins them with empty stored blocks (sync flush), -i with full flushes; a member can also
    hold many tiny blocks: all one DEFLATE stream"""
    data = PAYLOADS["fastq"]
    p = tmp_path / "x.gz"
    for mode, step in ((zlib.Z_SYNC_FLUSH, 131072), (zlib.Z_FULL_FLUSH, 131072), (zlib.Z_SYNC_FLUSH, 3000)):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = b"".join(co.compress(data[i:i + step]) + co.flush(mode) for i in range(0, len(data), step)) + co.flush()
        blob = b"\x1f\x8b\x08\0\0\0\0\0\0\xff" + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))
        assert gzip.decompress(blob) == data
        p.write_bytes(blob)
        for threads, section in ((4, 65536), (3, 200000)):
            rc, got, err, st = pgunzip(p, len(data) + 16, threads, section)
            assert rc == 0 and got == data, (mode, step, err, st)
            if step > 3000:
                assert st["used"] >= 4 and st["fell_back"] == 0, st


def test_sequential_decoder_on_pigz_streams_larger_than_its_input_buffer(tmp_path):
    """round 6: a pigz-made member (sync flush = an empty stored block every piece) of many input buffers. A block header inside the
    last KiB of a buffer made the decoder refill with whole bytes still in its bit buffer; the stored block behind it then handed
    'unread bytes' back that had left the buffer: "Compressed file ended ..." in the middle of a good file - about once per GB of
    pigz output, in the decoder every fallback ends in."""
    arena, off, _ = synth.reads_numpy(200000, 100, seed=9)
    src = tmp_path / "t.fq"
    synth.write_fastq_realistic(str(src), arena, off, 1, seed=9)
    data = src.read_bytes()
    for chunk in (300000, 131072, 70001):
        p = tmp_path / ("pigz_%d.fq.gz" % chunk)
        synth.pgzip_file(str(src), str(p), level=6, chunk=chunk)
        rc, got, err = gunzip(p, len(data) + 16)
        assert rc == 0 and got == data, (chunk, err, len(got))
