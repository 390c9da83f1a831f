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
