"""gzip members inflated on the device (C ABI rd_gz_inflate_members, csrc/rd_inflate_dev.hpp) and the reader path built on it.

What it replaces for files whose members say how long they are (BGZF; every .gz this build's CLI writes): gzip.open(path, 'rt') of
the reference (data_loader/seq_encoder.py:21-39). The property: the device's output equals zlib's, byte for byte, for every DEFLATE
block type and any number of blocks per member - and a damaged member is reported (CRC-32 and ISIZE are checked on the device)."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, kind="BC"):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    d = co.compress(data) + co.flush()
    if kind == "BC":
        n = 18 + len(d) + 8
        assert n <= 65536
        hdr = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", n - 1)
    else:   # this build's host writer: 'R','D', 32-bit size
        n = 20 + len(d) + 8
        hdr = b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x08\x00RD\x04\x00" + struct.pack("<I", n - 1)
    return hdr + d + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def _inflate(blob):
    from ribodetector_amd.gz import DeviceGunzip
    dg = DeviceGunzip(DEV)
    buf = np.frombuffer(blob, dtype=np.uint8).copy()
    n, consumed, out_bytes, streaming = dg.index(buf, len(buf))
    assert consumed == len(buf) and not streaming
    return dg.inflate(buf, consumed, n, out_bytes).cpu().numpy().tobytes(), n


def _payloads(rng):
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(rng.choice(list(b"ACGT"), 100).astype(np.uint8)), b"F" * 100) for i in range(250))
    return {"fastq": fq, "random": bytes(rng.integers(0, 256, 40000, dtype=np.uint8)), "zeros": b"\0" * 60000, "one": b"x",
            "run_then_text": b"A" * 300 + fq[:3000], "binaryish": bytes(rng.choice(list(range(0, 256, 7)), 50000).astype(np.uint8)),
            "short_period": (b"ACGTTGCA" * 5000)[:39999], "fib": b"".join(bytes([65 + k]) * f for k, f in enumerate([1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597, 2584, 4181, 6765]))}


@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                            (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (6, zlib.Z_FILTERED)])
def test_every_block_type_equals_zlib(level, strategy):
    rng = np.random.default_rng(level * 10 + strategy)
    pay = _payloads(rng)
    names = sorted(pay)
    blob = b"".join(_member(pay[k], level, strategy) for k in names)
    got, n = _inflate(blob)
    assert n == len(names) and got == b"".join(pay[k] for k in names)


def test_many_blocks_per_member_and_large_members():
    """'R','D' members of several hundred KB: zlib cuts them into many blocks (stored, fixed and dynamic ones mixed by the data)"""
    rng = np.random.default_rng(7)
    parts = []
    for i in range(6):
        chunk = b"".join([bytes(rng.integers(0, 256, 30000, dtype=np.uint8)), b"ACGT" * 20000, bytes(rng.choice(list(b"ACGTN\n"), 150000).astype(np.uint8)),
                          b"\n".join(b"line %d of member %d" % (k, i) for k in range(4000))])
        parts.append(chunk)
    blob = b"".join(_member(p, 6 if i % 2 else 1, kind="RD") for i, p in enumerate(parts))
    got, n = _inflate(blob)
    assert n == 6 and got == b"".join(parts)


def test_long_codes_far_distances_and_long_runs():
    """what leaves the fast paths of the kernel: codewords longer than the 10-bit table (a 200-symbol alphabet with geometric
    frequencies: zlib hands out 11 ... 15 bit codes), distances up to the full 32 KiB window, matches of 258 with distance 1 ... 7,
    matches that straddle the 64-position rounds, a member that is one long run"""
    rng = np.random.default_rng(42)
    p = 0.93 ** np.arange(200)
    skew = bytes(rng.choice(200, 60000, p=p / p.sum()).astype(np.uint8))
    block = bytes(rng.integers(0, 256, 3000, dtype=np.uint8))
    far = block + bytes(rng.choice(list(b"AC"), 29000).astype(np.uint8)) + block + bytes(rng.choice(list(b"GT"), 700).astype(np.uint8)) + block
    runs = b"".join(bytes(rng.integers(65, 70, int(rng.integers(1, 8)), dtype=np.uint8)) * int(rng.integers(40, 400)) for _ in range(150))
    mixed = b"".join(block[i:i + int(rng.integers(3, 70))] + bytes(rng.choice(list(b"ACGT"), int(rng.integers(0, 30))).astype(np.uint8))
                     for i in rng.integers(0, 2900, 900))
    pay = [skew, far, runs[:65000], mixed[:65000], b"Q" * 65280]
    for level, strategy in ((6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_HUFFMAN_ONLY), (1, zlib.Z_DEFAULT_STRATEGY)):
        blob = b"".join(_member(x, level, strategy) for x in pay)
        got, n = _inflate(blob)
        assert n == len(pay) and got == b"".join(pay), (level, strategy)


def test_random_damage_never_yields_wrong_bytes():
    """400 members with 1 ... 4 bytes of their DEFLATE data overwritten at random: every one is either reported or (the damage hit
    bits that do not matter) decodes to the original bytes - and the launch returns (no walk without progress, nothing out of bounds)"""
    from ribodetector_amd import _native as N
    from ribodetector_amd.gz import DeviceGunzip
    rng = np.random.default_rng(5)
    pay = _payloads(rng)
    names = [k for k in sorted(pay) if len(pay[k]) > 2000]
    members, plain = [], []
    for i in range(400):
        k = names[i % len(names)]
        m = bytearray(_member(pay[k], (1, 6, 9)[i % 3], (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY)[(i // 3) % 3]))
        for _ in range(int(rng.integers(1, 5))):
            m[int(rng.integers(18, len(m) - 8))] = int(rng.integers(0, 256))
        members.append(bytes(m))
        plain.append(pay[k])
    dg = DeviceGunzip(DEV)
    buf = np.frombuffer(b"".join(members), dtype=np.uint8).copy()
    n, consumed, out_bytes, _ = dg.index(buf, len(buf))
    assert n == 400 and consumed == len(buf)
    try:
        dg.inflate(buf, consumed, n, out_bytes)
    except ValueError:
        pass
    torch.cuda.synchronize()
    st = dg._status[:n].cpu().numpy()
    text = dg._text_dev[:out_bytes].cpu().numpy().tobytes()
    off, reported = 0, 0
    for i in range(n):
        if st[i] == 0:
            assert text[off:off + len(plain[i])] == plain[i], i
        else:
            assert 1 <= st[i] <= 9
            reported += 1
        off += len(plain[i])
    assert reported > 300


def test_member_table_entries_outside_the_buffers_are_refused():
    from ribodetector_amd import _native as N
    lib = N.lib()
    data = b"ACGT" * 1000
    blob = np.frombuffer(_member(data), dtype=np.uint8).copy()
    comp = torch.from_numpy(blob).to(DEV)
    text = torch.zeros(len(data), dtype=torch.uint8, device=DEV)
    rows = np.array([(18, 0, len(blob) - 26, len(data)),            # the real one
                     (18, 0, len(blob), len(data)),                 # data + trailer past the end of comp
                     (18, 1, len(blob) - 26, len(data)),            # output past the end of text
                     (-4, 0, 100, 10), (18, -1, 100, 10), (18, 0, -5, 10), (18, 0, 100, -1)],
                    dtype=np.dtype([("i", "<i8"), ("o", "<i8"), ("il", "<i4"), ("ol", "<i4")]))
    tab = torch.from_numpy(rows.view(np.uint8)).to(DEV)
    st = torch.full((len(rows),), 77, dtype=torch.int32, device=DEV)
    N.check(lib.rd_gz_inflate_members(N.ptr(comp), comp.numel(), N.ptr(tab), len(rows), N.ptr(text), text.numel(), N.ptr(st),
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rd_gz_inflate_members")
    torch.cuda.synchronize()
    assert st.cpu().tolist() == [0, 10, 10, 10, 10, 10, 10] and text.cpu().numpy().tobytes() == data


def test_output_of_the_device_deflate_round_trips_on_the_device():
    from ribodetector_amd import synth
    from ribodetector_amd.gz import DeviceGzip
    a, o, l = synth.reads_torch(200000, 100, seed=3, device=DEV)
    text = synth.fastq_image_torch(a, o, l, mate=1)
    rs = torch.arange(200001, dtype=torch.int64, device=DEV) * 218
    out, info = DeviceGzip(DEV).compress_selected(text, rs, torch.zeros(200000, dtype=torch.int8, device=DEV), 0)
    torch.cuda.synchronize()
    comp = out[: int(info[0])].cpu().numpy().tobytes()
    got, n = _inflate(comp)
    assert n == int(info[2]) and got == text.cpu().numpy().tobytes()


def test_damage_is_reported():
    """a flipped byte in the DEFLATE data, in the CRC, in ISIZE; a truncated member: each is an error naming the member - never bytes"""
    from ribodetector_amd.gz import DeviceGunzip
    rng = np.random.default_rng(1)
    good = [_member(bytes(rng.choice(list(b"ACGT\n"), 20000).astype(np.uint8))) for _ in range(5)]
    for where in ("data", "crc"):
        bad = bytearray(good[2])
        bad[len(bad) // 2 if where == "data" else len(bad) - 7] ^= 0x10
        with pytest.raises(ValueError, match="gzip member 2"):
            _inflate(b"".join(good[:2]) + bytes(bad) + b"".join(good[3:]))
    bad = bytearray(good[1])
    bad[-4:] = struct.pack("<I", 19999)                                    # ISIZE one short: the index believes it, the data does not fit
    with pytest.raises(ValueError, match="gzip member 1"):
        _inflate(good[0] + bytes(bad))
    dg = DeviceGunzip(DEV)
    buf = np.frombuffer(good[0] + good[1][:-100], dtype=np.uint8).copy()
    n, consumed, out_bytes, streaming = dg.index(buf, len(buf))
    assert n == 1 and consumed == len(good[0])                              # the incomplete member is left for the next batch
    plain = np.frombuffer(gzip.compress(b"hello"), dtype=np.uint8).copy()   # a member without a size subfield: not for this path
    assert dg.index(plain, len(plain))[3] is True


@pytest.mark.parametrize("fmt", ["fq", "fa"])
def test_reader_with_device_inflate_equals_the_host_reader(tmp_path, fmt, monkeypatch):
    """RD_DEVICE_INFLATE=1: a BGZF file goes file -> GPU (one wave per member) -> parser; same chunks as the host's inflate gives"""
    from ribodetector_amd import synth
    from ribodetector_amd.data_loader import fastx_parser as fx
    arena, off, lens = synth.reads_numpy(120000, (40, 150), seed=4)
    seqs = synth.as_strings(arena, off)
    if fmt == "fq":
        text = "".join("@read%d some text\n%s\n+\n%s\n" % (i, s, "F" * len(s)) for i, s in enumerate(seqs)).encode()
    else:
        text = "".join(">read%d\n%s\n" % (i, s) for i, s in enumerate(seqs)).encode()
    path = str(tmp_path / ("in.%s.gz" % ("fastq" if fmt == "fq" else "fasta")))
    with open(path, "wb") as fh:
        for i in range(0, len(text), 65280):
            fh.write(_member(text[i:i + 65280]))
        fh.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))

    def read_all():
        return [(c.buf[c.rec_start[0]:c.rec_start[-1]].tobytes(), c.seq_off.copy(), c.seq_len.copy()) for c in fx.get_seq_chunks(path, chunk_size=25000)]
    monkeypatch.setenv("RD_DEVICE_INFLATE", "0")
    host = read_all()
    monkeypatch.delenv("RD_DEVICE_INFLATE")                                 # the default: BGZF files go through the device
    assert fx.device_inflate_wanted(path)
    dev = read_all()
    assert len(host) == len(dev) == 5
    for h, d in zip(host, dev):
        assert h[0] == d[0] and np.array_equal(h[1], d[1]) and np.array_equal(h[2], d[2])
    # a damaged member surfaces as the reader's error, after the chunks before it
    raw = bytearray(open(path, "rb").read())
    raw[len(raw) // 2] ^= 0x55
    open(path, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="gzip member"):
        read_all()


def test_members_without_a_size_behind_bgzf_blocks_go_to_the_host(tmp_path, monkeypatch):
    """`cat a.bgzf.gz b.gz`: the BGZF blocks are inflated on the GPU, the plain gzip members behind them by zlib on the host, in order"""
    from ribodetector_amd.data_loader import fastx_parser as fx
    recs = ["@r%d\n%s\n+\n%s\n" % (i, "ACGT" * 20, "F" * 80) for i in range(9000)]
    a, b, c = "".join(recs[:5000]).encode(), "".join(recs[5000:7000]).encode(), "".join(recs[7000:]).encode()
    path = str(tmp_path / "mixed.fastq.gz")
    with open(path, "wb") as fh:
        for i in range(0, len(a), 65280):
            fh.write(_member(a[i:i + 65280]))
        fh.write(gzip.compress(b) + gzip.compress(c))
    monkeypatch.delenv("RD_DEVICE_INFLATE", raising=False)
    assert fx.device_inflate_wanted(path)
    got = b"".join(ch.buf[ch.rec_start[0]:ch.rec_start[-1]].tobytes() for ch in fx.get_seq_chunks(path, chunk_size=4000))
    assert got == a + b + c
    with open(path, "ab") as fh:
        fh.write(gzip.compress(b"@x\nAC\n+\nFF\n")[:-6])               # and a truncated member at the very end is an error, not silence
    with pytest.raises(ValueError, match="ended before the end-of-stream marker"):
        list(fx.get_seq_chunks(path, chunk_size=4000))
