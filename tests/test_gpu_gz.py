"""Device-side gzip of the label-partitioned records (C ABI rd_gz_compress_selected, csrc/rd_deflate.hpp).

What it replaces: the reference writes the records of one label, in input order, through gzip.open(path, 'wt', compresslevel=5)
(reference detect.py:485-492,729-741). The properties that make a replacement right are size-independent:
  * round trip: zlib / gzip decompress the output to EXACTLY the selected records, in input order (every member carries its CRC-32
    and ISIZE, so a wrong byte anywhere is an error, not a mismatch);
  * framing: a sequence of complete BGZF members (18-byte header with the 'B','C' subfield holding the member's size - 1);
  * ratio: within 10 % of zlib level 5 on FASTQ (the round-4 target).
"""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _records(rng, n, kind):
    recs = []
    for i in range(n):
        if kind == "fastq":
            L = int(rng.integers(30, 160))
            seq = bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8))
            q = bytes(rng.choice(list(b"F:,#"), L, p=[0.9, 0.06, 0.03, 0.01]).astype(np.uint8))
            recs.append(b"@A00123:45:HXXXXXXX:1:%d:%d:%d 1:N:0:ACGT\n%s\n+\n%s\n" % (1101 + i // 4000, 1000 + i % 3000, 1000 + i // 7, seq, q))
        elif kind == "random":
            recs.append(bytes(rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8)))
        elif kind == "runs":
            recs.append(bytes([int(rng.integers(65, 70))]) * int(rng.integers(1, 5000)) + b"\n")
        elif kind == "tiny":
            recs.append(bytes(rng.choice(list(b"AC\n"), int(rng.integers(1, 4))).astype(np.uint8)))
        elif kind == "huge":
            recs.append(bytes(rng.choice(list(b"ACGT"), int(rng.integers(60000, 200000))).astype(np.uint8)) + b"\n")
        elif kind == "fib":
            # 23 byte values with Fibonacci frequencies, shuffled: the optimal Huffman code of a member is deeper than DEFLATE's 15
            # bits, so the length-limiting repair (zlib's gen_bitlen step) is what makes the block decodable
            fib = [1, 1]
            while len(fib) < 23:
                fib.append(fib[-1] + fib[-2])
            data = np.concatenate([np.full(f, 33 + j, dtype=np.uint8) for j, f in enumerate(fib)])
            rng.shuffle(data)
            recs.append(data.tobytes())
    return recs


def _run(dg, recs, labels, label):
    text = np.frombuffer(b"".join(recs), dtype=np.uint8)
    rs = np.zeros(len(recs) + 1, dtype=np.int64)
    rs[1:] = np.cumsum([len(r) for r in recs])
    t = torch.from_numpy(text.copy()).to(DEV) if len(text) else torch.zeros(1, dtype=torch.uint8, device=DEV)[:0]
    out, info = dg.compress_selected(t, torch.from_numpy(rs).to(DEV), torch.from_numpy(labels).to(DEV), label)
    torch.cuda.synchronize()
    nb, plain, members, _ = (int(x) for x in info.cpu().tolist())
    comp = out[:nb].cpu().numpy().tobytes()
    want = b"".join(r for r, l in zip(recs, labels) if l == label)
    return comp, want, plain, members


def _check_members(comp, members, plain):
    """walk the BGZF headers; every member inflates on its own to ISIZE bytes with the CRC-32 it carries"""
    p, k, total = 0, 0, 0
    while p < len(comp):
        assert comp[p:p + 4] == b"\x1f\x8b\x08\x04" and comp[p + 10:p + 16] == b"\x06\x00BC\x02\x00", (p, comp[p:p + 18])
        bsize = struct.unpack("<H", comp[p + 16:p + 18])[0] + 1
        assert bsize <= 65536
        body = zlib.decompress(comp[p + 18:p + bsize - 8], -15)
        crc, isize = struct.unpack("<II", comp[p + bsize - 8:p + bsize])
        assert isize == len(body) <= 65280 and crc == (zlib.crc32(body) & 0xffffffff)
        total += isize
        p += bsize
        k += 1
    assert p == len(comp) and k == members and total == plain


@pytest.mark.parametrize("kind", ["fastq", "random", "runs", "tiny", "huge", "fib"])
def test_round_trip_and_framing(kind):
    from ribodetector_amd.gz import DeviceGzip
    rng = np.random.default_rng(hash(kind) % 1000)
    dg = DeviceGzip(DEV)
    n = {"fastq": 5000, "random": 400, "runs": 300, "tiny": 20000, "huge": 9, "fib": 6}[kind]
    recs = _records(rng, n, kind)
    labels = rng.choice(np.array([0, 1, -1], dtype=np.int8), n, p=[0.6, 0.3, 0.1])
    for label in (0, 1, -1, 5):
        comp, want, plain, members = _run(dg, recs, labels, label)
        assert plain == len(want) and members == (len(want) + 65279) // 65280
        assert (gzip.decompress(comp) if comp else b"") == want, (kind, label)
        _check_members(comp, members, plain)


def test_sizes_around_the_member_boundary_and_empty_inputs():
    from ribodetector_amd.gz import DeviceGzip
    dg = DeviceGzip(DEV)
    rng = np.random.default_rng(3)
    for total in (0, 1, 7, 8, 9, 63, 64, 65, 8159, 8160, 8161, 8223, 16319, 16320, 16321, 65279, 65280, 65281, 2 * 65280, 2 * 65280 + 1, 200001):
        body = bytes(rng.choice(list(b"ACGTN\n"), total).astype(np.uint8))
        recs = [body[i:i + 997] for i in range(0, total, 997)] or [b""]
        labels = np.ones(len(recs), dtype=np.int8)
        comp, want, plain, members = _run(dg, recs, labels, 1)
        assert want == body and plain == total
        assert (gzip.decompress(comp) if comp else b"") == body, total
        _check_members(comp, members, plain)
    # no record at all
    out, info = dg.compress_selected(torch.zeros(0, dtype=torch.uint8, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV),
                                     torch.zeros(0, dtype=torch.int8, device=DEV), 0)
    torch.cuda.synchronize()
    assert info.cpu().tolist() == [0, 0, 0, 0]
    # a record table that does not describe the text (records past its end): flagged, nothing compressed, nothing overrun
    t = torch.from_numpy(np.frombuffer(b"ACGT\n" * 100, dtype=np.uint8).copy()).to(DEV)
    rs = torch.tensor([0, 400, 900, 5000], dtype=torch.int64, device=DEV)
    out, info = dg.compress_selected(t, rs, torch.ones(3, dtype=torch.int8, device=DEV), 1)
    torch.cuda.synchronize()
    assert info.cpu().tolist() == [0, 0, 0, 1]


def test_fuzz_small_inputs_of_every_alphabet():
    """300 small chunks: alphabets of 1 ... 256 symbols, skewed and flat, with planted repeats at every distance class, uint8 labels -
    each must inflate (zlib checks every code, CRC-32 and ISIZE) to exactly the selected records"""
    from ribodetector_amd.gz import DeviceGzip
    dg = DeviceGzip(DEV)
    rng = np.random.default_rng(2026)
    for case in range(int(os.environ.get("RD_GZ_FUZZ_CASES", "300"))):      # (3,000 run once per change of the kernel: tools/README.md)
        k = int(rng.choice([1, 2, 3, 4, 5, 16, 64, 256]))
        nrec = int(rng.integers(1, 60))
        recs = []
        for _ in range(nrec):
            L = int(rng.integers(0, 4000))
            p = rng.dirichlet(np.full(k, float(rng.choice([0.05, 0.5, 5.0]))))
            b = bytearray(rng.choice(k, L, p=p).astype(np.uint8) + int(rng.integers(0, 256 - k + 1)))
            for _ in range(int(rng.integers(0, 6))):                       # plant repeats: distance 1 ... 20000, length 3 ... 400
                if len(b) > 16 and recs:
                    src = recs[int(rng.integers(0, len(recs)))]
                    if len(src) > 8:
                        a0 = int(rng.integers(0, len(src) - 4))
                        piece = src[a0:a0 + int(rng.integers(3, 400))]
                        at = int(rng.integers(0, len(b)))
                        b[at:at + len(piece)] = piece
            recs.append(bytes(b))
        labels = rng.integers(0, 2, nrec).astype(np.uint8)                 # rd_classify's label type
        for label in (0, 1):
            comp, want, plain, members = _run(dg, recs, labels, label)
            assert (gzip.decompress(comp) if comp else b"") == want, case
            _check_members(comp, members, plain)


def test_the_eof_block_is_an_empty_member():
    from ribodetector_amd.gz import eof_block
    e = eof_block()
    assert len(e) == 28 and gzip.decompress(e) == b"" and e == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def test_ratio_against_zlib_level_5_and_rate(report):
    """sequencer-like FASTQ (Illumina headers, binned qualities) and the bench's own FASTQ: compressed size within 10 % of zlib
    level 5 (one stream, what the reference's gzip.open writes); the rate is recorded"""
    from ribodetector_amd import synth
    from ribodetector_amd.gz import DeviceGzip
    import tempfile
    dg = DeviceGzip(DEV)
    rep = {}
    with tempfile.TemporaryDirectory() as d:
        a, o, _ = synth.reads_numpy(200000, 100, seed=11)
        p1, p2 = os.path.join(d, "real.fq"), os.path.join(d, "syn.fq")
        synth.write_fastq_realistic(p1, a, o, mate=1, seed=1)
        synth.write_fastq(p2, a[: o[80000]], o[:80001], mate=1)
        for name, path in (("sequencer_like", p1), ("bench_fastq", p2)):
            data = open(path, "rb").read()
            arr = np.frombuffer(data, dtype=np.uint8)
            nl = np.flatnonzero(arr == 10)
            rs = np.concatenate([[0], nl[3::4] + 1]).astype(np.int64)
            n = len(rs) - 1
            t, r = torch.from_numpy(arr.copy()).to(DEV), torch.from_numpy(rs).to(DEV)
            lab = torch.zeros(n, dtype=torch.int8, device=DEV)
            out, info = dg.compress_selected(t, r, lab, 0)
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(5):
                out, info = dg.compress_selected(t, r, lab, 0)
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / 5
            nb = int(info[0])
            comp = out[:nb].cpu().numpy().tobytes()
            assert gzip.decompress(comp) == data
            z5 = len(zlib.compress(data, 5))
            rep[name] = {"bytes": len(data), "device_gzip_bytes": nb, "zlib5_bytes": z5, "ratio_vs_zlib5": nb / z5, "ms": ms,
                         "GB_per_s": len(data) / ms / 1e6, "reads_per_s": n / ms * 1e3}
            assert nb <= 1.10 * z5, rep[name]
    report["device_gzip"] = rep
