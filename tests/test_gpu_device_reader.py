"""The FASTQ record index on the device (C ABI rd_fastq_index / rd_fastq_gather / rd_fastq_strip_mark, csrc/rd_fastq_index.hpp) and
the reader built on it (data_loader/device_reader.py): text that is inflated / copied onto the GPU is framed there.

What it replaces: the parser's FASTQ state machine (reference data_loader/fastx_parser.py:15-37). The property: the device's chunks -
record text, rec_start, seq_off, seq_len - equal the host reader's (csrc/rd_host.cpp, itself pinned to the reference's parser by
tests/test_fastx.py) bit for bit, on the reference-made parser.json text and on fuzzed files (CR LF, '@' / '+' inside quality lines,
'+name' lines, no final newline, empty sequences, blank tails, lower case) with batch boundaries falling on every byte; malformed
streams are reported like the host reader reports them, after the records before the damage."""
import io
import os
import struct
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _member(data, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9)
    d = co.compress(data) + co.flush()
    n = 18 + len(d) + 8
    assert n <= 65536
    return b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", n - 1) + d + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data))


def _bgzf(text, block=65280, rng=None):
    out, i = [], 0
    while i < len(text):
        k = block if rng is None else int(rng.integers(1, block + 1))
        out.append(_member(text[i:i + k]))
        i += k
    return b"".join(out) + EOF


def _host_chunks(path, chunk, first=None, errors=None):
    """errors: a list that receives the reader's error message instead of the exception (the chunks before it are returned)"""
    from ribodetector_amd.data_loader import fastx_parser as fx
    old = os.environ.get("RD_DEVICE_INFLATE")
    os.environ["RD_DEVICE_INFLATE"] = "0"
    out = []
    try:
        for c in fx.get_seq_chunks(path, chunk_size=chunk, first_chunk=first):
            out.append((c.buf[c.rec_start[0]:c.rec_start[-1]].tobytes(), np.asarray(c.rec_start) - c.rec_start[0], np.asarray(c.seq_off) - c.rec_start[0],
                        np.asarray(c.seq_len)))
        return out
    except ValueError as e:
        if errors is None:
            raise
        errors.append(str(e).split(": ", 1)[-1])
        return out
    finally:
        os.environ.pop("RD_DEVICE_INFLATE")
        if old is not None:
            os.environ["RD_DEVICE_INFLATE"] = old


def _dev_chunks(path, chunk, first=None, byte_range=None, stats=None, errors=None):
    from ribodetector_amd.data_loader import device_reader as dr
    out = []
    try:
        for c in dr.get_seq_chunks_device(path, chunk_size=chunk, first_chunk=first, byte_range=byte_range, device=DEV, stats=stats):
            text, rs, so, sl = c.to_host()
            assert len(rs) == c.n + 1 and len(so) == len(sl) == c.n == len(c.seq_len) and rs[0] == 0 and rs[-1] == len(text)
            out.append((text.tobytes(), rs, so, sl))
    except ValueError as e:
        if errors is None:
            raise
        errors.append(str(e))
    return out


def _same(host, dev, records_only=False):
    if records_only:         # a stream that ends in an error: the same records in front of it (the chunking of the tail may differ)
        def cat(cs):
            return (b"".join(c[0] for c in cs), np.concatenate([c[3] for c in cs]) if cs else np.zeros(0, np.int32),
                    np.concatenate([c[2][: len(c[3])] - c[1][: len(c[3])] for c in cs]) if cs else np.zeros(0, np.int64))
        h, d = cat(host), cat(dev)
        assert h[0] == d[0] and np.array_equal(h[1], d[1]) and np.array_equal(h[2], d[2])
        return
    assert len(host) == len(dev), (len(host), len(dev))
    for k, (h, d) in enumerate(zip(host, dev)):
        assert h[0] == d[0], "chunk %d: record text differs" % k
        for a, b, what in ((h[1], d[1], "rec_start"), (h[2], d[2], "seq_off"), (h[3], d[3], "seq_len")):
            assert np.array_equal(a, b), "chunk %d: %s differs" % (k, what)


@pytest.fixture
def small_batches(monkeypatch):
    """batch sizes of a few hundred bytes: record, line and carry boundaries fall on every byte position of a small file"""
    from ribodetector_amd.data_loader import device_reader as dr

    def set_sizes(first, full, members=None):
        monkeypatch.setattr(dr.DeviceFeeder, "PLAIN_FIRST", first)
        monkeypatch.setattr(dr.DeviceFeeder, "PLAIN_BATCH", full)
        monkeypatch.setattr(dr.DeviceFeeder, "FIRST", first)
        if members:
            monkeypatch.setattr(dr.DeviceFeeder, "MAX_MEMBERS", members)
    return set_sizes


def test_reference_parser_text(golden, tmp_path, small_batches):
    """tests/golden/parser.json: the text the reference's own seq_parser was run on - the device's records are the reference's records"""
    g = golden.json("parser")
    text = g["fastq_text"].encode()
    want = ["\n".join(r) + "\n" for r in g["fastq_records"]]
    p = str(tmp_path / "ref.fastq")
    open(p, "wb").write(text)
    for first, full in ((1 << 20, 1 << 20), (64, 64), (7, 13), (1, 1)):
        small_batches(first, full)
        dev = _dev_chunks(p, 1000)
        assert b"".join(d[0] for d in dev).decode() == "".join(want)
        rs, so, sl = dev[0][1], dev[0][2], dev[0][3]
        for i, r in enumerate(g["fastq_records"]):
            assert dev[0][0][rs[i]:rs[i + 1]].decode() == "\n".join(r) + "\n" and dev[0][0][so[i]:so[i] + sl[i]].decode() == r[1]
        _same(_host_chunks(p, 1000), dev)


def _fuzz_text(rng, nrec):
    qual = b"!\"#$%&'()*+,-./0123456789:;<=>?@ABCDEFGHIJ"
    crlf = rng.random() < 0.25
    nl = b"\r\n" if crlf else b"\n"
    recs = []
    for i in range(nrec):
        L = int(rng.choice([0, 1, 2, 30, 100, 151, 300]))
        seq = bytes(rng.choice(list(b"ACGTNacgtn"), L).astype(np.uint8))
        q = bytes(rng.choice(list(qual), L).astype(np.uint8))
        if L and rng.random() < 0.2:
            q = (b"@" if rng.random() < 0.5 else b"+") + q[1:]                 # quality lines that start like a header / a '+' line
        plus = b"+" if rng.random() < 0.7 else b"+name %d" % i
        hdr = b"@r%d" % i + (b" 1:N:0:ACGT" if rng.random() < 0.5 else b"")
        trail = [b"", b"", b"", b""]
        if not crlf and rng.random() < 0.03:
            trail[int(rng.integers(0, 4))] = bytes(rng.choice(list(b" \t\r"), int(rng.integers(1, 4))).astype(np.uint8))   # stray trailing blanks
        recs.append(hdr + trail[0] + nl + seq + trail[1] + nl + plus + trail[2] + nl + q + trail[3] + nl)
    text = b"".join(recs)
    tail = rng.random()
    if tail < 0.2 and text:
        text = text[:-len(nl)]                                # no final newline
    elif tail < 0.35:
        text += nl * int(rng.integers(1, 4))                  # up to three blank lines behind the last record
    elif tail < 0.4:
        text += b" \t" + nl                                   # a line of blanks
    return text


def test_fuzz_plain_and_bgzf_against_the_host_reader(tmp_path, small_batches):
    """200 files (RD_TEST_FULL=1: 500): the chunks of the device reader == the chunks of the host reader, plain and BGZF, with small random
    batch sizes and chunk sizes (the records of a chunk come from up to dozens of batches; the carry is exercised at every offset)"""
    from conftest import FULL
    rng = np.random.default_rng(11)
    n_files = n_err = 0
    iters = 250 if FULL else 100
    for it in range(iters):
        nrec = int(rng.choice([0, 1, 2, 3, 5, 17, 100, 400, 1500]))
        text = _fuzz_text(rng, nrec)
        first = int(rng.choice([1, 3, 50, 700, 5000, 1 << 20]))
        small_batches(first, max(first, int(rng.choice([1, 9, 333, 4096, 1 << 20]))), members=int(rng.choice([1, 2, 7, 4096])))
        chunk, fc = int(rng.choice([1, 2, 7, 64, 1000, 100000])), (None if rng.random() < 0.5 else int(rng.choice([1, 5, 100])))
        p = str(tmp_path / "f.fastq")
        open(p, "wb").write(text)
        he = []
        host = _host_chunks(p, chunk, fc, errors=he)        # (a file whose last record lost its empty quality line with the final newline
        n_err += len(he)                                    # is truncated for both readers: same records before, same message)
        pz = str(tmp_path / "f.fastq.gz")
        open(pz, "wb").write(_bgzf(text, block=int(rng.choice([40, 700, 65280])), rng=rng if rng.random() < 0.5 else None))
        for q in (p, pz):
            de = []
            dev = _dev_chunks(q, chunk, fc, errors=de)
            assert de == he, (q, de, he)
            _same(host, dev)
            n_files += 1
    assert n_files == 2 * iters and n_err < 25


def test_large_file_many_batches(tmp_path):
    """a file of several default-sized batches (12 MB first, doubling): chunks of exactly the scheduled sizes, text == the file"""
    from ribodetector_amd import synth
    arena, off, lens = synth.reads_numpy(700000, (40, 150), seed=3)
    p = str(tmp_path / "big.fastq")
    synth.write_fastq(p, arena, off, 1)
    st = {}
    dev = _dev_chunks(p, 200000, first=1 << 17, stats=st)
    assert [len(d[3]) for d in dev] == [131072, 200000, 200000, 168928] and st["feeder"]["batches"] >= 3 and st["indexer"]["stripped"] == 0
    assert b"".join(d[0] for d in dev) == open(p, "rb").read()
    _same(_host_chunks(p, 200000, 1 << 17), dev)
    # a byte range (the multi-rank CLI's share of a plain file): both ends on record boundaries
    from ribodetector_amd.data_loader import fastx_parser as fx
    a, b = fx.find_record_start(p, 30_000_000), fx.find_record_start(p, 90_000_000)
    part = _dev_chunks(p, 1 << 20, byte_range=(a, b))
    assert b"".join(d[0] for d in part) == open(p, "rb").read()[a:b]


def test_malformed_streams_are_reported_like_the_host_reader_reports_them(tmp_path, small_batches):
    from ribodetector_amd.data_loader import device_reader as dr
    good = b"".join(b"@r%d\nACGT\n+\nFFFF\n" % i for i in range(50))
    cases = {"header": (good + b"r50\nAC\n+\nFF\n" + good, "does not start with '@'", 50),
             "truncated": (good + b"@x\nAC\n+\n", "truncated FASTQ record", 50),
             "truncated_one_line": (good + b"@x", "truncated FASTQ record", 50),
             "four_blank_lines": (good + b"\n\n\n\n", "does not start with '@'", 50),
             "blank_header_crlf": (b"\r\n" + good, "does not start with '@'", 0)}
    for name, (text, msg, n_before) in cases.items():
        for first in (1 << 20, 37):
            for chunk in (1000, 16, 10, 1):
                small_batches(first, first)
                p = str(tmp_path / (name + ".fastq"))
                open(p, "wb").write(text)
                he, de = [], []
                host = _host_chunks(p, chunk, errors=he)                     # the host reader's verdict and the chunks in front of it ...
                dev = _dev_chunks(p, chunk, errors=de)                       # ... are the device reader's
                assert len(he) == 1 and msg in he[0] and de == he, (name, first, chunk, he, de)
                _same(host, dev)
                assert sum(len(c[3]) for c in dev) == (n_before // chunk) * chunk
    # a damaged BGZF member: named, never bytes
    raw = bytearray(_bgzf(good * 40, block=700))
    at, k = 0, 0
    while k < 20:                                             # the 21st member: one bit of its CRC-32
        at += struct.unpack_from("<H", raw, at + 16)[0] + 1
        k += 1
    raw[at + struct.unpack_from("<H", raw, at + 16)[0] + 1 - 8] ^= 0x01
    p = str(tmp_path / "dmg.fastq.gz")
    open(p, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="gzip member|not a gzip member|member size"):
        _dev_chunks(p, 1000)
    # a record longer than the carry pad (round 5: refused, "set RD_DEVICE_PARSE=0") is framed with a pad grown to hold it - and a
    # TRUNCATED record behind such a one is still reported
    small_batches(1 << 20, 1 << 20)
    monkey_pad = dr.PAD
    try:
        dr.PAD = 4096
        big = b"@long\n" + b"A" * 9000 + b"\n+\n" + b"F" * 9000 + b"\n"
        p = str(tmp_path / "long.fastq")
        open(p, "wb").write(good + big + good)
        small_batches(1000, 1000)
        _same(_host_chunks(p, 1000), _dev_chunks(p, 1000))
        open(p, "wb").write(good + big + good[:-40])
        eh, ed = [], []
        _same(_host_chunks(p, 7, errors=eh), _dev_chunks(p, 7, errors=ed), records_only=True)
        assert eh and ed and "truncated" in ed[0]
    finally:
        dr.PAD = monkey_pad


def test_c_abi_argument_errors():
    from ribodetector_amd import _native as N
    L = N.lib()
    t = torch.zeros(4096, dtype=torch.uint8, device=DEV)
    assert L.rd_fastq_index(None, 0, 0, None, None, 0, None, 0, None, None, 0, None) != 0 and b"null" in L.rd_last_error()
    s = torch.zeros(8, dtype=torch.int64, device=DEV)
    le = torch.zeros(64, dtype=torch.int32, device=DEV)
    ws = torch.zeros(4096, dtype=torch.uint8, device=DEV)
    assert L.rd_fastq_index(N.ptr(t), 10, 5, None, None, 0, N.ptr(le), 64, N.ptr(s), N.ptr(ws), 4096, None) != 0      # end < pad
    assert L.rd_fastq_index(N.ptr(t), 0, 100, N.ptr(t), None, 0, N.ptr(le), 64, N.ptr(s), N.ptr(ws), 4096, None) != 0  # prev_text without prev
    assert L.rd_fastq_index_workspace_bytes(-1) == 0 and L.rd_fastq_index_workspace_bytes(1 << 20) >= 4 * 65
    assert L.rd_select_pack(None, 10, None, None, 5, 300, None, 0, N.ptr(s), None, 0, None) != 0 and b"int8" in L.rd_last_error()


def test_select_pack_equals_a_host_gather():
    """rd_select_pack: the records of one label as one text == b''.join of those records, for random labels and ragged records"""
    from ribodetector_amd.gz import DeviceSelect
    rng = np.random.default_rng(2)
    for n in (1, 2, 255, 256, 257, 5000, 70000):
        lens = rng.integers(0, 400, n)
        rs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        text = rng.integers(32, 127, int(rs[-1]) + 1, dtype=np.uint8)
        labels = rng.choice(np.array([0, 1, -1], dtype=np.int8), n, p=[0.6, 0.3, 0.1])
        ds = DeviceSelect(DEV)
        T, R, Lb = torch.from_numpy(text).to(DEV), torch.from_numpy(rs).to(DEV), torch.from_numpy(labels).to(DEV)
        for lab in (0, 1, -1):
            out, info = ds.pack_selected(T, R, Lb, lab)
            info = info.cpu().numpy()
            want = b"".join(text[rs[i]:rs[i + 1]].tobytes() for i in range(n) if labels[i] == lab)
            assert info[3] == 0 and info[1] == len(want) and out[: len(want)].cpu().numpy().tobytes() == want


def test_short_lines_overflow_the_line_table_and_are_framed_again(tmp_path, small_batches):
    """the line table of a batch holds window / 4 lines; records like '@\\n\\n+\\n\\n' (1.5 bytes per line) overflow it: the batch - and
    every batch the producer had chained to it meanwhile - is framed again with a full-size table, behind the batch finished last"""
    text = b"".join((b"@\n\n+\n\n" if i % 3 else b"@r%d\nA\n+\nF\n" % i) for i in range(60000))
    p = str(tmp_path / "tiny.fastq")
    open(p, "wb").write(text)
    for first, full in ((1 << 20, 1 << 20), (30000, 50000)):
        small_batches(first, full)
        st = {}
        dev = _dev_chunks(p, 7000, stats=st)
        assert st["indexer"]["reframed"] >= (1 if first > 100000 else 3), st
        assert b"".join(d[0] for d in dev) == text
        _same(_host_chunks(p, 7000), dev)


# ---- FASTA on the device (rd_fasta_index / rd_fasta_gather; reference data_loader/fastx_parser.py:39-55) -----------------------------------

def _fuzz_fasta(rng, nrec):
    """FASTA as files have it: multi-line sequences of every width, lower case, CR LF, indented lines, blank lines anywhere, records
    without a sequence, headers with blanks, '>' inside lines, a last line without its newline"""
    crlf = rng.random() < 0.25
    nl = b"\r\n" if crlf else b"\n"
    out = []
    if rng.random() < 0.2:
        out.append(nl * int(rng.integers(1, 3)) if rng.random() < 0.5 else b"  \t" + nl)      # blank lines in front of the first header
    if rng.random() < 0.15:                                # sequence lines in front of the first header (the reference glues them to the
        for _ in range(int(rng.integers(1, 4))):           # first record; a file without any header: one record with an empty header)
            out.append(bytes(rng.choice(list(b"ACGTacgt"), int(rng.integers(1, 90))).astype(np.uint8)) + nl)
    for i in range(nrec):
        hdr = (b"  " if rng.random() < 0.05 else b"") + b">s%d" % i + (b" some text > here" if rng.random() < 0.3 else b"") + (b" \t" if rng.random() < 0.1 else b"")
        out.append(hdr + nl)
        L = int(rng.choice([0, 0, 1, 7, 60, 100, 151, 511, 512, 1000, 2500]))      # (lines of 512+ bytes are copied by a workgroup)
        seq = bytes(rng.choice(list(b"ACGTNacgtnRYKM"), L).astype(np.uint8))
        width = int(rng.choice([1, 10, 60, 70, 80, 100000]))
        for o in range(0, L, width):
            line = seq[o:o + width]
            if rng.random() < 0.03:
                line = b" " + line + b"\t"
            out.append(line + nl)
            if rng.random() < 0.03:
                out.append(nl)                                     # a blank line inside a sequence
        if rng.random() < 0.05:
            out.append(nl)
    text = b"".join(out)
    if text and rng.random() < 0.25:
        text = text[:-len(nl)]
    return text


def test_fasta_reference_parser_text(golden, tmp_path, small_batches):
    """tests/golden/parser.json: the FASTA text the reference's own seq_parser was run on - the device's records are the reference's"""
    g = golden.json("parser")
    text = g["fasta_text"].encode()
    want = "".join("\n".join(r) + "\n" for r in g["fasta_records"])
    p = str(tmp_path / "ref.fasta")
    open(p, "wb").write(text)
    for first, full in ((1 << 20, 1 << 20), (64, 64), (7, 13), (1, 1)):
        small_batches(first, full)
        dev = _dev_chunks(p, 1000)
        assert b"".join(d[0] for d in dev).decode() == want
        rs, so, sl = dev[0][1], dev[0][2], dev[0][3]
        for i, r in enumerate(g["fasta_records"]):
            assert dev[0][0][rs[i]:rs[i + 1]].decode() == "\n".join(r) + "\n" and dev[0][0][so[i]:so[i] + sl[i]].decode() == r[1]
        _same(_host_chunks(p, 1000), dev)


def test_fasta_fuzz_plain_and_bgzf_against_the_host_reader(tmp_path, small_batches):
    """120 files (RD_TEST_FULL=1: 300): the device's FASTA chunks == the host reader's (normalised text, rec_start, seq_off, seq_len), plain
    and BGZF, small random batch and chunk sizes: the carry (the raw text from the last header line on) is exercised at every offset"""
    from conftest import FULL
    rng = np.random.default_rng(12)
    n_files = 0
    iters = 150 if FULL else 60
    for it in range(iters):
        nrec = int(rng.choice([0, 1, 2, 3, 5, 17, 100, 400]))
        text = _fuzz_fasta(rng, nrec)
        first = int(rng.choice([1, 3, 50, 700, 5000, 1 << 20]))
        small_batches(first, max(first, int(rng.choice([1, 9, 333, 4096, 1 << 20]))), members=int(rng.choice([1, 2, 7, 4096])))
        chunk, fc = int(rng.choice([1, 2, 7, 64, 1000, 100000])), (None if rng.random() < 0.5 else int(rng.choice([1, 5, 100])))
        p = str(tmp_path / "f.fasta")
        open(p, "wb").write(text)
        host = _host_chunks(p, chunk, fc)
        pz = str(tmp_path / "f.fasta.gz")
        open(pz, "wb").write(_bgzf(text, block=int(rng.choice([40, 700, 65280])), rng=rng if rng.random() < 0.5 else None))
        for q in (p, pz):
            _same(host, _dev_chunks(q, chunk, fc))
            n_files += 1
    assert n_files == 2 * iters


def test_fasta_large_file_and_what_stays_with_the_host(tmp_path):
    from ribodetector_amd import synth
    from ribodetector_amd.data_loader import device_reader as dr
    arena, off, lens = synth.reads_numpy(300000, (40, 150), seed=5)
    p = str(tmp_path / "big.fasta")
    with open(p, "wb") as fh:                      # 60-column FASTA
        for i in range(len(lens)):
            s = arena[off[i]:off[i] + lens[i]].tobytes()
            fh.write(b">read%d\n" % i + b"\n".join(s[o:o + 60] for o in range(0, len(s), 60)) + b"\n")
    st = {}
    dev = _dev_chunks(p, 100000, first=1 << 15, stats=st)
    assert [len(d[3]) for d in dev] == [32768, 65536, 100000, 100000, 1696] and st["feeder"]["batches"] >= 2
    _same(_host_chunks(p, 100000, 1 << 15), dev)
    assert dr.device_ingest_kind(p) == "plain"
    # sequence in front of the first header: the reference glues it to the first record; no header at all: one record, empty header
    q = str(tmp_path / "lead.fasta")
    open(q, "wb").write(b"ACGT\nac\n>r1\nGG\n>r2\nT\n")
    assert dr.device_ingest_kind(q) == "plain" and _dev_chunks(q, 10)[0][0] == b">r1\nACGTACGG\n>r2\nT\n"
    open(q, "wb").write(b"ACGT\n\nac\n")
    assert _dev_chunks(q, 10)[0][0] == b"\nACGTAC\n"
    _same(_host_chunks(q, 10), _dev_chunks(q, 10))
    os.environ["RD_DEVICE_FASTA"] = "0"
    try:
        assert dr.device_ingest_kind(p) is None
    finally:
        del os.environ["RD_DEVICE_FASTA"]


def test_fasta_single_stream_gz_and_a_record_longer_than_the_pad(tmp_path, small_batches):
    """a FASTA .gz that is ONE gzip stream goes through the stream decoder and the same re-writing; a record that does not fit the carry
    pad is refused with the way out named"""
    import gzip
    from ribodetector_amd.data_loader import device_reader as dr
    rng = np.random.default_rng(21)
    text = _fuzz_fasta(rng, 3000)
    p, pz = str(tmp_path / "s.fasta"), str(tmp_path / "s.fasta.gz")
    open(p, "wb").write(text)
    with gzip.open(pz, "wb", compresslevel=6) as fh:
        fh.write(text)
    host = _host_chunks(p, 700)
    assert dr.device_ingest_kind(pz) in ("stream", None)
    _same(host, _dev_chunks(pz, 700))
    small_batches(1000, 1000)
    pad = dr.PAD
    try:
        # (round 6) a record longer than the pad - here 4 KiB, batches of 1,000 bytes, so the carry grows over a dozen batches - is
        # framed again with a pad that holds it: the reference's parser joins lines without bound (fastx_parser.py:39-55)
        dr.PAD = 4096
        q = str(tmp_path / "long.fasta")
        open(q, "wb").write(b">a\nACGT\n>long\n" + b"ACGT" * 3000 + b"\n>b\nGG\n" + b">long2 wrapped\n" + (b"acgtn" * 12 + b"\n") * 200 + b">c\nT\n")
        st = {}
        dev = _dev_chunks(q, 1000, stats=st)
        _same(_host_chunks(q, 1000), dev)
        assert st["indexer"]["regrown"] >= 2
        fq = str(tmp_path / "long.fq")
        rec = lambda k, n: b"@r%d\n%s\n+\n%s\n" % (k, b"ACGT" * n, b"IIII" * n)      # noqa: E731
        open(fq, "wb").write(rec(0, 10) + rec(1, 2500) + rec(2, 5) + rec(3, 4000) + rec(4, 7))
        st = {}
        dev = _dev_chunks(fq, 3, stats=st)
        _same(_host_chunks(fq, 3), dev)
        assert st["indexer"]["regrown"] >= 2
    finally:
        dr.PAD = pad


def test_records_longer_than_the_real_pad(tmp_path):
    """a FASTA file with a 40 MiB record (a contig) and a FASTQ file with a 20 MiB read, batches and pad at their product sizes: the
    default reader frames them - same chunks as the host reader - where round 5 ended the run with 'set RD_DEVICE_PARSE=0'"""
    rng = np.random.default_rng(5)
    big = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 40 << 20)].tobytes()
    fa = str(tmp_path / "contigs.fasta")
    with open(fa, "wb") as fh:
        fh.write(b">small1\nACGTACGT\n>contig_40MiB\n" + b"\n".join(big[o:o + 60] for o in range(0, len(big), 60)) + b"\n>small2\nGGCC\n")
    st = {}
    dev = _dev_chunks(fa, 1000, stats=st)
    host = _host_chunks(fa, 1000)
    _same(host, dev)
    assert st["indexer"]["regrown"] >= 1 and len(dev[0][3]) == 3 and int(dev[0][3][1]) == 40 << 20
    fq = str(tmp_path / "ultralong.fq")
    n = 20 << 20
    with open(fq, "wb") as fh:
        fh.write(b"@a\nACGT\n+\nIIII\n" * 1000 + b"@ultra\n" + big[:n] + b"\n+\n" + b"F" * n + b"\n" + b"@z\nGG\n+\nII\n" * 1000)
    st = {}
    dev = _dev_chunks(fq, 1500, stats=st)
    _same(_host_chunks(fq, 1500), dev)
    assert sum(len(d[3]) for d in dev) == 2001
    # ... and through the CLI: the default (device) reader writes what the host reader writes, no switch needed
    from ribodetector_amd import detect
    for inp, ext in ((fa, "fa"), (fq, "fq")):
        outs = {}
        for mode in ("device", "host"):
            os.environ["RD_INGEST"] = mode
            try:
                o, r = str(tmp_path / ("%s.non.%s" % (mode, ext))), str(tmp_path / ("%s.rrna.%s.gz" % (mode, ext)))
                p = detect.main(["-l", "100", "-i", inp, "-o", o, "-r", r, "--chunk_size", "1", "-m", "3"], log_level="WARNING")
                import gzip
                outs[mode] = (open(o, "rb").read(), gzip.open(r, "rb").read(), p.num_read)
                assert (mode == "device") == bool(p.ingest and all(v["path"] == "device" for v in p.ingest.values()))
            finally:
                del os.environ["RD_INGEST"]
        assert outs["device"] == outs["host"] and outs["device"][2] == (3 if ext == "fa" else 2001)
        assert len(outs["device"][0]) + len(outs["device"][1]) > (40 << 20 if ext == "fa" else 40 << 20)


def test_fasta_share_keeps_its_last_record_without_a_sequence(tmp_path):
    """a rank's share of a FASTA file that goes on behind it: the share's last record counts even when it has no sequence (the next
    rank starts with the next header); the end of the FILE drops such a record, like the reference - both as the host reader does"""
    from ribodetector_amd.data_loader import fastx_parser as fx
    text = b">a\nAC\n>b\n>c\nGG\n>z\n"
    p = str(tmp_path / "s.fasta")
    open(p, "wb").write(text)
    cut = text.index(b">c")
    for rng_ in ((0, cut), (cut, len(text)), None):
        host = [(c.buf[c.rec_start[0]:c.rec_start[-1]].tobytes(), np.asarray(c.seq_len)) for c in fx.get_seq_chunks(p, chunk_size=100, byte_range=rng_)]
        dev = _dev_chunks(p, 100, byte_range=rng_)
        assert [h[0] for h in host] == [d[0] for d in dev] and all(np.array_equal(h[1], d[3]) for h, d in zip(host, dev)), rng_
    assert _dev_chunks(p, 100, byte_range=(0, cut))[0][0] == b">a\nAC\n>b\n\n" and _dev_chunks(p, 100)[0][0] == b">a\nAC\n>b\n\n>c\nGG\n"


def test_fasta_short_lines_overflow_the_tables_and_are_framed_again(tmp_path, small_batches):
    """FASTA whose lines average under 8 bytes (the line / record tables of a batch hold window / 8 entries): the batch - and what was
    chained to it - is framed again with full-size tables"""
    text = b"".join((b">\nA\n" if i % 3 else b">r%d\nac\nG\n" % i) for i in range(60000))
    p = str(tmp_path / "tiny.fasta")
    open(p, "wb").write(text)
    for first, full in ((1 << 20, 1 << 20), (30000, 50000)):
        small_batches(first, full)
        st = {}
        dev = _dev_chunks(p, 7000, stats=st)
        assert st["indexer"]["reframed"] >= 1, st
        _same(_host_chunks(p, 7000), dev)
