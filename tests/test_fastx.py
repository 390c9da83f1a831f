"""Host ingest (no GPU): record semantics of the reference parser (golden parser.json produced by the reference's
seq_parser) and the vectorised chunk reader / label-partitioned writer built on top."""
import gzip
import io
import os

import numpy as np
import pytest

from ribodetector_amd import synth
from ribodetector_amd.data_loader import fastx_parser as fx


@pytest.fixture(autouse=True, params=["mapped", "buffered"])
def _plain_reader_mode(request, monkeypatch):
    """every test of this file runs twice: plain files parsed out of the mapped file (the default) and through the buffered reader
    (RD_READER_MMAP=0: pipes, and what gzip input always uses)"""
    monkeypatch.setenv("RD_READER_MMAP", "1" if request.param == "mapped" else "0")


def test_seq_parser_matches_reference(golden):
    g = golden.json("parser")
    assert [list(r) for r in fx.seq_parser(io.StringIO(g["fastq_text"]), "fastq")] == g["fastq_records"]
    assert [list(r) for r in fx.seq_parser(io.StringIO(g["fasta_text"]), "fasta")] == g["fasta_records"]


def test_get_seq_format():
    assert fx.get_seq_format("a.fq") == "fq" and fx.get_seq_format("a.fastq.gz") == "fqgz"
    assert fx.get_seq_format("x/y.fa") == "fa" and fx.get_seq_format("y.fna.gz") == "fagz"
    with pytest.raises(ValueError):
        fx.get_seq_format("reads.txt")
    with pytest.raises(ValueError):
        fx.get_seq_format("reads.fq.bz2")


READERS = [fx.get_seq_chunks, fx.get_seq_chunks_numpy]     # native (librd_host.so) and the numpy cross-check


def _write(path, chunk, mask):
    w = fx.open_for_write(path)
    w.write_selected(chunk, mask.astype(np.int8), 1)
    w.close()
    op = gzip.open if path.endswith("gz") else open
    with op(path, "rb") as fh:
        return fh.read()


def _records(chunk):
    b = chunk.buf.tobytes()
    return [b[chunk.seq_off[i]:chunk.seq_off[i] + chunk.seq_len[i]].decode() for i in range(len(chunk.seq_len))]


@pytest.mark.parametrize("reader", READERS)
@pytest.mark.parametrize("suffix", ["fq", "fq.gz"])
def test_fastq_chunks(tmp_path, suffix, golden, reader):
    arena, off, lens = synth.reads_numpy(1000, (30, 150), seed=2)
    p = str(tmp_path / ("r." + suffix))
    synth.write_fastq(p, arena, off, mate=1)
    want = synth.as_strings(arena, off)
    got, total = [], 0
    for c in reader(p, chunk_size=333):
        assert len(c.seq_len) <= 333 and c.verbatim
        got += _records(c)
        total += len(c.seq_len)
        # full records round-trip verbatim
        assert fx.select_records(c, np.ones(len(c.seq_len), bool)) == c.buf[c.rec_start[0]:c.rec_start[-1]].tobytes()
    assert total == 1000 and got == want
    # against the reference-semantics generator
    op = gzip.open if suffix.endswith("gz") else open
    with op(p, "rt") as fh:
        ref = list(fx.seq_parser(fh, "fastq"))
    assert [r[1] for r in ref] == want
    # label partition: selecting a mask yields exactly those records, in order, newline terminated
    chunks = list(reader(p, chunk_size=1000))
    mask = np.random.default_rng(1).random(1000) < 0.3
    want = "".join("\n".join(r) + "\n" for r, m in zip(ref, mask) if m)
    assert fx.select_records(chunks[0], mask).decode() == want
    assert fx.select_records(chunks[0], np.zeros(1000, bool)) == b""
    # native writer, plain and gzip (level 5 like the reference), same bytes
    assert _write(str(tmp_path / "sel.fq"), chunks[0], mask).decode() == want
    assert _write(str(tmp_path / "sel.fq.gz"), chunks[0], mask).decode() == want
    assert _write(str(tmp_path / "none.fq"), chunks[0], np.zeros(1000, bool)) == b""


@pytest.mark.parametrize("reader", READERS)
def test_fastq_edge_framing(tmp_path, golden, reader):
    g = golden.json("parser")
    p = str(tmp_path / "odd.fastq")
    # CRLF endings, trailing blanks, '@' as first quality char, no final newline
    text = g["fastq_text"].replace("\n", "\r\n").rstrip("\r\n")
    with open(p, "w", newline="") as fh:
        fh.write(text)
    cs = list(reader(p, chunk_size=2))
    assert [len(c.seq_len) for c in cs] == [2, 1]
    assert sum((_records(c) for c in cs), []) == [r[1] for r in g["fastq_records"]]
    out = b"".join(fx.select_records(c, np.ones(len(c.seq_len), bool)) for c in cs).decode()
    assert out == "".join("\n".join(r) + "\n" for r in g["fastq_records"])
    with open(p, "w") as fh:
        fh.write("@a\nACGT\n+\n")
    with pytest.raises(ValueError):
        list(reader(p, 10))
    with open(p, "w") as fh:
        fh.write("a\nACGT\n+\nIIII\n")
    with pytest.raises(ValueError):
        list(reader(p, 10))
    if reader is fx.get_seq_chunks:                # trailing blank lines (the reference crashes on them) are not a record
        with open(p, "w") as fh:
            fh.write("@a\nACGT\n+\nIIII\n\n\n")
        assert sum(len(c.seq_len) for c in reader(p, 10)) == 1


@pytest.mark.parametrize("reader", READERS)
def test_fasta_chunks(tmp_path, golden, reader):
    g = golden.json("parser")
    p = str(tmp_path / "x.fa")
    open(p, "w").write(g["fasta_text"])
    cs = list(reader(p, chunk_size=2))
    assert sum((_records(c) for c in cs), []) == [r[1] for r in g["fasta_records"]]
    out = b"".join(fx.select_records(c, np.ones(len(c.seq_len), bool)) for c in cs).decode()
    assert out == "".join("\n".join(r) + "\n" for r in g["fasta_records"])


def test_paired_chunks(tmp_path):
    a1, o1, _ = synth.reads_numpy(100, 50, seed=1)
    a2, o2, _ = synth.reads_numpy(100, 60, seed=2)
    p1, p2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    synth.write_fastq(p1, a1, o1, 1)
    synth.write_fastq(p2, a2, o2, 2)
    n = 0
    for c1, c2 in fx.get_pairedread_chunks(p1, p2, chunk_size=64):
        assert len(c1.seq_len) == len(c2.seq_len)
        n += len(c1.seq_len)
    assert n == 100


def test_native_reader_small_buffers_and_growth(tmp_path):
    """records that do not fit the first buffer estimate: the reader hands back what fits, the wrapper grows and continues"""
    rng = np.random.default_rng(3)
    recs = []
    for i in range(300):
        L = int(rng.integers(1, 3000))
        s = "".join("ACGT"[k] for k in rng.integers(0, 4, L))
        recs.append(("@r%d some description" % i, s, "+", "I" * L))
    p = str(tmp_path / "big.fastq.gz")
    with gzip.open(p, "wt") as fh:
        for r in recs:
            fh.write("\n".join(r) + "\n")
    r = fx.NativeReader(p, est_record_bytes=8)       # deliberately tiny estimate
    got = []
    while True:
        c = r.read(77)
        if c is None:
            break
        assert len(c.seq_len) == 77 or r.eof
        b = c.buf.tobytes()
        for i in range(len(c.seq_len)):
            got.append(tuple(b[c.rec_start[i]:c.rec_start[i + 1]].decode().rstrip("\n").split("\n")))
            assert b[c.seq_off[i]:c.seq_off[i] + c.seq_len[i]].decode() == got[-1][1]
    assert got == recs
    with pytest.raises(ValueError):
        fx.NativeReader(str(tmp_path / "reads.txt"))


def test_gzip_writer_members(tmp_path):
    """gz output = concatenated independent gzip members (parallel level-5 compression): must round-trip, also when empty
    and when the selection spans several 4 MiB members and several write calls."""
    arena, off, lens = synth.reads_numpy(30000, 150, seed=8)
    p = str(tmp_path / "big.fq")
    synth.write_fastq(p, arena, off, mate=1)
    chunks = list(fx.get_seq_chunks(p, chunk_size=7000))
    out = str(tmp_path / "sel.fastq.gz")
    w = fx.open_for_write(out)
    want = b""
    rng = np.random.default_rng(0)
    for c in chunks:
        lab = (rng.random(len(c.seq_len)) < 0.8).astype(np.int8)
        w.write_selected(c, lab, 1)
        want += fx.select_records(c, lab == 1)
    w.close()
    assert len(want) > (5 << 20)              # spans more than one 4 MiB member
    with gzip.open(out, "rb") as fh:
        assert fh.read() == want
    empty = str(tmp_path / "empty.fq.gz")
    w = fx.open_for_write(empty)
    w.write_selected(chunks[0], np.zeros(len(chunks[0].seq_len), np.int8), 1)
    w.close()
    with gzip.open(empty, "rb") as fh:
        assert fh.read() == b""


def test_writer_appends_members_made_elsewhere(tmp_path):
    """rd_writer_write_members: complete gzip members made elsewhere (the GPU's BGZF blocks; here: Python's zlib) are appended as they
    are, in order with what the host path wrote before and after; the file is closed with BGZF's end-of-file marker; a plain
    output refuses them."""
    import ctypes as C
    import zlib
    arena, off, lens = synth.reads_numpy(4000, 100, seed=9)
    p = str(tmp_path / "in.fq")
    synth.write_fastq(p, arena, off, mate=1)
    c = next(fx.get_seq_chunks(p, chunk_size=4000))
    all1 = np.ones(len(c.seq_len), np.int8)
    text = fx.select_records(c, all1 == 1)

    def member(b):   # what a BGZF writer emits for one block of bytes
        co = zlib.compressobj(5, zlib.DEFLATED, -15)
        d = co.compress(b) + co.flush()
        n = 18 + len(d) + 8
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + (n - 1).to_bytes(2, "little") + d +
                (zlib.crc32(b) & 0xffffffff).to_bytes(4, "little") + len(b).to_bytes(4, "little"))
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    out = str(tmp_path / "mixed.fq.gz")
    w = fx.open_for_write(out)
    w.write_selected(c, all1, 1)                          # host path first (buffered: less than a 4 MiB member) ...
    mem = b"".join(member(text[i:i + 60000]) for i in range(0, len(text), 60000))
    buf = C.create_string_buffer(mem, len(mem))
    w.write_members(C.addressof(buf), len(mem))           # ... then members made elsewhere: the buffered records come first
    w.write_members(None, 0)                              # (nothing to append is fine)
    w.write_selected(c, all1, 1)
    w.close()
    raw = open(out, "rb").read()
    assert raw.endswith(eof) and raw.count(mem) == 1
    with gzip.open(out, "rb") as fh:
        assert fh.read() == text * 3
    back = b"".join(ch.buf[ch.rec_start[0]:ch.rec_start[-1]].tobytes() for ch in fx.get_seq_chunks(out, chunk_size=3000))
    assert back == text * 3                               # this build's reader walks the mixed member kinds
    only = str(tmp_path / "only.fq.gz")
    w = fx.open_for_write(only)
    w.write_members(None, 0)
    w.close()
    assert open(only, "rb").read() == eof and gzip.open(only, "rb").read() == b""
    plain = fx.open_for_write(str(tmp_path / "plain.fq"))
    with pytest.raises(ValueError, match="not a gzip output"):
        plain.write_members(C.addressof(buf), len(mem))
    plain.close()


def test_gzip_writer_zlib_fallback(tmp_path):
    """RD_HOST_ZLIB=1 (or a machine without libdeflate.so.0) takes the zlib path: same decompressed bytes, and the
    library's own reader/decoder reads both back."""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    arena, off, _ = synth.reads_numpy(40000, 120, seed=9)
    src = str(tmp_path / "in.fq")
    synth.write_fastq(src, arena, off, mate=1)
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from ribodetector_amd.data_loader import fastx_parser as fx\n"
            "w = fx.open_for_write(sys.argv[2])\n"
            "for c in fx.get_seq_chunks(sys.argv[1], 9000):\n"
            "    w.write_selected(c, (np.arange(len(c.seq_len)) %% 3 != 0).astype(np.int8), 1)\n"
            "w.close()\n") % root
    outs = {}
    for tag, env in (("auto", {}), ("zlib", {"RD_HOST_ZLIB": "1"})):
        out = str(tmp_path / ("o_%s.fq.gz" % tag))
        subprocess.run([sys.executable, "-c", code, src, out], check=True, env=dict(os.environ, **env), timeout=300)
        with gzip.open(out, "rb") as fh:
            outs[tag] = fh.read()
        got = b"".join(c.buf[c.rec_start[0]:c.rec_start[-1]].tobytes() for c in fx.get_seq_chunks(out, 5000))
        assert got == outs[tag]
    assert outs["auto"] == outs["zlib"] and outs["auto"].count(b"\n") == 4 * (40000 - 13334)


def test_plain_writer_vectored(tmp_path):
    """plain output is written with pwritev straight from the chunk buffer, 1,024 runs per call: right bytes for dense,
    sparse and alternating selections (tens of thousands of runs), across several calls on one file"""
    arena, off, _ = synth.reads_numpy(90000, 150, seed=10)
    p = str(tmp_path / "big.fq")
    synth.write_fastq(p, arena, off, mate=1)
    chunks = list(fx.get_seq_chunks(p, chunk_size=45000))
    rng = np.random.default_rng(1)
    for name, pick in (("dense", lambda n: rng.random(n) < 0.97), ("sparse", lambda n: rng.random(n) < 0.4),
                       ("alternating", lambda n: np.arange(n) % 2 == 0), ("all", lambda n: np.ones(n, bool))):
        out = str(tmp_path / ("sel_%s.fq" % name))
        w = fx.open_for_write(out)
        want = b""
        for c in chunks:
            lab = pick(len(c.seq_len)).astype(np.int8)
            w.write_selected(c, lab, 1)
            want += fx.select_records(c, lab == 1)
        w.close()
        assert len(want) > (8 << 20) or name == "sparse"
        with open(out, "rb") as fh:
            assert fh.read() == want, name


def test_long_record_with_small_chunk(tmp_path):
    """one record far larger than the first buffer (a 120 kb FASTA sequence, a 300 kb FASTQ read) with a tiny chunk_size: the
    reader reports the bytes it needs and the wrapper grows - the reference parser has no record-size limit"""
    rng = np.random.default_rng(11)
    big = "".join("ACGT"[k] for k in rng.integers(0, 4, 120000))
    fa = str(tmp_path / "long.fa")
    with open(fa, "w") as fh:
        fh.write(">short1\nACGT\n>big one\n")
        for i in range(0, len(big), 60):
            fh.write(big[i:i + 60].lower() + "\n")
        fh.write(">short2\nGGCC\n")
    recs = []
    for c in fx.get_seq_chunks(fa, chunk_size=2):
        b = c.buf.tobytes()
        recs += [b[c.seq_off[i]:c.seq_off[i] + c.seq_len[i]].decode() for i in range(len(c.seq_len))]
    assert recs == ["ACGT", big, "GGCC"]
    fq = str(tmp_path / "long.fq")
    big2 = big * 2 + big[:60000]
    with open(fq, "w") as fh:
        fh.write("@a\nAC\n+\nII\n@b\n%s\n+\n%s\n@c\nGT\n+\nII\n" % (big2, "I" * len(big2)))
    r = fx.NativeReader(fq, est_record_bytes=8)
    c = r.read(3)
    assert list(c.seq_len) == [2, len(big2), 2] and r.read(3) is None
    assert c.buf.tobytes()[c.seq_off[1]:c.seq_off[1] + c.seq_len[1]].decode() == big2


def test_writer_threads_follow_set_threads(tmp_path):
    """rd_writer_open takes the thread count in force when it is called: -t must be applied before the writers are opened"""
    from ribodetector_amd import _native as N
    L = N.host_lib()
    try:
        for t in (3, 1, 7):
            L.rd_host_set_threads(t)
            w = fx.open_for_write(str(tmp_path / ("t%d.fq.gz" % t)))
            assert w.threads == t
            w.close()
    finally:
        L.rd_host_set_threads(0)
    w = fx.open_for_write(str(tmp_path / "auto.fq.gz"))
    assert 1 <= w.threads <= 32
    w.close()


def _plan_all(paths, world):
    """plan_ranges for every rank of `world`, with an in-process all_gather (threads + barrier)"""
    import threading
    bar = threading.Barrier(world)
    slots = [None] * world
    out = [None] * world
    errs = []

    def run(rank):
        def all_gather(obj):
            slots[rank] = obj
            bar.wait()
            got = list(slots)
            bar.wait()
            return got
        try:
            out[rank] = fx.plan_ranges(paths, rank, world, all_gather)
        except BaseException as e:     # noqa: BLE001
            errs.append(e)
            bar.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]
    return out


def _range_records(path, byte_range=None):
    recs = []
    for c in fx.get_seq_chunks(path, chunk_size=997, byte_range=byte_range):
        b = c.buf.tobytes()
        recs += [b[c.rec_start[i]:c.rec_start[i + 1]] for i in range(len(c.seq_len))]
    return recs


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
def test_byte_range_sharding_of_mate_files(tmp_path, world):
    """multi-rank ingest: the ranks' byte ranges tile each file, hold the same record indices in both mates (whose byte
    positions differ: R2 has longer, varying headers) and parse to exactly the records of the whole file - also with quality
    lines that start with '@' or '+', the classic FASTQ re-synchronisation trap"""
    rng = np.random.default_rng(world)
    n = 4000
    r1, r2 = str(tmp_path / "m_1.fastq"), str(tmp_path / "m_2.fastq")
    with open(r1, "w") as f1, open(r2, "w") as f2:
        for i in range(n):
            L1, L2 = int(rng.integers(30, 150)), int(rng.integers(30, 150))
            s1 = "".join("ACGTN"[k] for k in rng.integers(0, 5, L1))
            s2 = "".join("ACGTN"[k] for k in rng.integers(0, 5, L2))
            q1 = "".join("@+I#F"[k] for k in rng.integers(0, 5, L1))          # qualities full of '@' and '+', also in column 0
            q2 = "".join("@+I#F"[k] for k in rng.integers(0, 5, L2))
            f1.write("@r%d/1\n%s\n+\n%s\n" % (i, s1, q1))
            f2.write("@r%d/2 %s\n%s\n+r%d\n%s\n" % (i, "x" * int(rng.integers(0, 80)), s2, i, q2))
    whole = [_range_records(r1), _range_records(r2)]
    assert len(whole[0]) == n == len(whole[1])
    plans = _plan_all([r1, r2], world)
    for f, path in enumerate((r1, r2)):
        assert plans[0][f][0] == 0 and plans[-1][f][1] == os.path.getsize(path)
        for a, b in zip(plans[:-1], plans[1:]):
            assert a[f][1] == b[f][0]                                          # the ranges tile the file
        parts = [_range_records(path, byte_range=pl[f]) for pl in plans]
        assert sum(parts, []) == whole[f]
        if f == 1:
            assert [len(p) for p in parts] == [len(_range_records(r1, byte_range=pl[0])) for pl in plans]   # same records per rank in both mates
    if world > 1:
        per = [len(_range_records(r1, byte_range=pl[0])) for pl in plans]
        assert max(per) < 1.5 * n / world + 8
    # single file, FASTA with multi-line sequences
    fa = str(tmp_path / "s.fa")
    with open(fa, "w") as fh:
        for i in range(700):
            s = "".join("ACGT"[k] for k in rng.integers(0, 4, int(rng.integers(1, 400))))
            fh.write(">s%d\n" % i + "\n".join(s[k:k + 60] for k in range(0, len(s), 60)) + "\n")
    plans = [fx.plan_ranges([fa], r, world) for r in range(world)]
    assert sum([_range_records(fa, byte_range=pl[0]) for pl in plans], []) == _range_records(fa)


def test_byte_range_sharding_edge_cases(tmp_path):
    """fewer records than ranks, an empty file, mates of different record counts, gzip input refused"""
    p1, p2 = str(tmp_path / "t_1.fq"), str(tmp_path / "t_2.fq")
    open(p1, "w").write("@a\nAC\n+\nII\n@b\nGT\n+\nII\n")
    open(p2, "w").write("@a\nACCC\n+\nIIII\n@b\nGTTTT\n+\n@@@@@\n")
    plans = _plan_all([p1, p2], 5)
    got1 = sum([_range_records(p1, byte_range=pl[0]) for pl in plans], [])
    got2 = sum([_range_records(p2, byte_range=pl[1]) for pl in plans], [])
    assert got1 == _range_records(p1) and got2 == _range_records(p2)
    for pl in plans:
        assert len(_range_records(p1, byte_range=pl[0])) == len(_range_records(p2, byte_range=pl[1]))
    empty = str(tmp_path / "e.fq")
    open(empty, "w").close()
    assert fx.plan_ranges([empty], 1, 3) == [(0, 0)] and _range_records(empty, byte_range=(0, 0)) == []
    open(p2, "a").write("@c\nA\n+\nI\n")
    with pytest.raises(ValueError, match="different numbers of records"):
        _plan_all([p1, p2], 2)
    gz = str(tmp_path / "z.fq.gz")
    with gzip.open(gz, "wt") as fh:
        fh.write("@a\nAC\n+\nII\n")
    assert fx.file_info(gz)[1] and not fx.file_info(p1)[1]
    with pytest.raises(ValueError, match="gzip"):
        fx.NativeReader(gz, byte_range=(0, 10))


def test_byte_range_sharding_crlf_and_trailing_blank_lines(tmp_path):
    """CRLF line ends, a last line without terminator in one mate and blank lines at the end of the other: the ranks'
    ranges still hold the same record indices and parse to the whole file's records"""
    rng = np.random.default_rng(9)
    p1, p2 = str(tmp_path / "c_1.fq"), str(tmp_path / "c_2.fq")
    r1, r2 = [], []
    for i in range(300):
        L = int(rng.integers(1, 90))
        s = "".join("ACGT"[k] for k in rng.integers(0, 4, L))
        r1.append("@a%d\r\n%s\r\n+\r\n%s" % (i, s, "I" * L))
        r2.append("@b%d xx\n%s\n+\n%s" % (i, s[::-1], "@" * L))
    open(p1, "w", newline="").write("\r\n".join(r1))                 # no terminator after the last line
    open(p2, "w", newline="").write("\n".join(r2) + "\n\n\n\n")     # three blank lines at the end (what the reader tolerates)
    whole1, whole2 = _range_records(p1), _range_records(p2)
    assert len(whole1) == 300 == len(whole2)
    for world in (2, 3, 7):
        plans = _plan_all([p1, p2], world)
        got1 = [_range_records(p1, byte_range=pl[0]) for pl in plans]
        got2 = [_range_records(p2, byte_range=pl[1]) for pl in plans]
        assert sum(got1, []) == whole1 and sum(got2, []) == whole2
        assert [len(g) for g in got1] == [len(g) for g in got2]


def test_byte_range_fasta_empty_sequence_at_a_cut(tmp_path):
    """the FASTA end-of-file quirk - a last record with an empty sequence is dropped (reference
    fastx_parser.py:39-55 yields at end of file only `if seq`) - applies at the TRUE end of the file only. A range that ends
    before the file does is followed by the next rank's header, where the reference yields the record whatever its sequence is.
    CRLF FASTA with empty-sequence records sprinkled in, so that cuts land right behind some of them."""
    rng = np.random.default_rng(15)
    for trial in range(6):
        recs = []
        for i in range(64):
            if rng.random() < 0.35 and i != 63:
                recs.append(">r%d\r\n\r\n" % i)                                  # header, blank line: empty sequence
            else:
                s = "".join("ACGTn"[k] for k in rng.integers(0, 5, int(rng.integers(1, 50))))
                recs.append(">r%d\r\n%s\r\n" % (i, s))
        if trial % 2:
            recs.append(">last\r\n\r\n")                                         # dropped by the whole-file parse AND by the last range
        p1, p2 = str(tmp_path / ("e%d_1.fa" % trial)), str(tmp_path / ("e%d_2.fa" % trial))
        open(p1, "w", newline="").write("".join(recs))
        open(p2, "w", newline="").write("".join(r.replace(">r", ">mate") for r in recs))
        whole = _range_records(p1)
        assert len(whole) == 64                                                  # the trailing '>last' never counts
        for world in (2, 3, 8):
            plans = [fx.plan_ranges([p1], r, world) for r in range(world)]
            assert sum([_range_records(p1, byte_range=pl[0]) for pl in plans], []) == whole, (trial, world)
            plans = _plan_all([p1, p2], world)
            per1 = [len(_range_records(p1, byte_range=pl[0])) for pl in plans]
            per2 = [len(_range_records(p2, byte_range=pl[1])) for pl in plans]
            assert per1 == per2 and sum(per1) == 64, (trial, world, per1, per2)


def test_count_records_counts_an_empty_last_fastq_record(tmp_path):
    """'@b\\n\\n+\\n\\n' at the end of a FASTQ file is a record for the reader, so the record count that aligns
    the mates' cuts must count it too (trimmed data: one mate may end with an empty read)"""
    from ribodetector_amd import _native as N
    import ctypes as C
    p1, p2 = str(tmp_path / "t_1.fq"), str(tmp_path / "t_2.fq")
    open(p1, "w").write("@a\nACGT\n+\nIIII\n@b\n\n+\n\n")
    open(p2, "w").write("@a\nAC\n+\nII\n@b\nG\n+\nI\n")
    assert len(_range_records(p1)) == 2
    n = C.c_int64(-1)
    for path, want in ((p1, 2), (p2, 2)):
        N.host_check(N.host_lib().rd_host_count_records(path.encode(), -1, 0, os.path.getsize(path), C.byref(n)), "count")
        assert n.value == want
    open(p1, "a").write("\n\n")                                                   # blank remainder of fewer than four lines: tolerated
    N.host_check(N.host_lib().rd_host_count_records(p1.encode(), -1, 0, os.path.getsize(p1), C.byref(n)), "count")
    assert n.value == 2 and len(_range_records(p1)) == 2
    for world in (2, 3):
        plans = _plan_all([p1, p2], world)                                       # round 2: 'paired-end files have different numbers of records'
        assert [len(_range_records(p1, byte_range=pl[0])) for pl in plans] == [len(_range_records(p2, byte_range=pl[1])) for pl in plans]


@pytest.mark.parametrize("workers", [2, 3])
def test_parallel_segment_reader_equals_sequential(tmp_path, workers):
    """get_seq_chunks_parallel (several reader threads over record-aligned byte segments of a plain file, small first chunks)
    delivers the records of the sequential reader, in order: FASTQ with qualities full of '@' and '+', CRLF, multi-line FASTA
    with empty-sequence records, files smaller than one segment, an empty file, and inside a byte range of the multi-rank CLI."""
    rng = np.random.default_rng(workers)
    fq = str(tmp_path / "p.fastq")
    with open(fq, "w", newline="") as fh:
        for i in range(6000):
            L = int(rng.integers(1, 200))
            s = "".join("ACGTN"[k] for k in rng.integers(0, 5, L))
            q = "".join("@+I#F"[k] for k in rng.integers(0, 5, L))
            fh.write("@r%d %s\r\n%s\r\n+\r\n%s\r\n" % (i, "x" * int(rng.integers(0, 60)), s, q))
    fa = str(tmp_path / "p.fa")
    with open(fa, "w") as fh:
        for i in range(3000):
            s = "".join("ACGT"[k] for k in rng.integers(0, 4, int(rng.integers(0, 300))))
            fh.write(">s%d\n" % i + "\n".join(s[k:k + 50] for k in range(0, len(s), 50)) + "\n")
        fh.write(">tail\nACGT\n")
    tiny = str(tmp_path / "t.fq")
    open(tiny, "w").write("@a\nAC\n+\nII\n")
    empty = str(tmp_path / "e.fq")
    open(empty, "w").close()

    def recs(gen):
        out = []
        for c in gen:
            b = c.buf.tobytes()
            out += [b[c.rec_start[i]:c.rec_start[i + 1]] for i in range(len(c.seq_len))]
        return out
    for path in (fq, fa, tiny, empty):
        want = recs(fx.get_seq_chunks(path, chunk_size=700))
        for chunk, first in ((700, 64), (5000, 1 << 18), (97, 1)):
            segs = fx.plan_segments(path, chunk, first_chunk=first)
            assert all(a[1] == b[0] for a, b in zip(segs[:-1], segs[1:])) and (not segs or (segs[0][0] == 0 and segs[-1][1] == os.path.getsize(path)))
            got = recs(fx.get_seq_chunks_parallel(path, chunk_size=chunk, workers=workers, first_chunk=first))
            assert got == want, (path, chunk, first, len(got), len(want))
    # inside a rank's byte range
    for world in (2, 3):
        for r in range(world):
            br = fx.plan_ranges([fq], r, world)[0]
            assert recs(fx.get_seq_chunks_parallel(fq, chunk_size=500, byte_range=br, workers=workers, first_chunk=32)) == \
                recs(fx.get_seq_chunks(fq, chunk_size=500, byte_range=br))
    # a parser error inside a worker reaches the consumer
    bad = str(tmp_path / "bad.fq")
    open(bad, "w").write("@a\nAC\n+\nII\n" * 4000 + "@b\nAC\n+\n")          # truncated last record
    with pytest.raises(ValueError, match="truncated"):
        recs(fx.get_seq_chunks_parallel(bad, chunk_size=300, workers=workers, first_chunk=16))


def test_chunk_schedule_pairs_up_mate_files(tmp_path):
    """fx.chunk_schedule: ONE list of chunk sizes for both mate readers - small first and last chunks, chunk_size in between - so
    that chunk i of R1 and chunk i of R2 hold the same records whatever their byte sizes; the records are those of the plain
    sequential read; a schedule that ends early (the record count is an estimate) is continued with small chunks"""
    rng = np.random.default_rng(4)
    r1, r2 = str(tmp_path / "s_1.fq"), str(tmp_path / "s_2.fq")
    n = 30000
    with open(r1, "w") as f1, open(r2, "w") as f2:
        for i in range(n):
            L1, L2 = int(rng.integers(30, 150)), int(rng.integers(30, 150))
            f1.write("@r%d/1\n%s\n+\n%s\n" % (i, "A" * L1, "I" * L1))
            f2.write("@r%d/2 %s\n%s\n+\n%s\n" % (i, "x" * int(rng.integers(0, 90)), "C" * L2, "I" * L2))
    for chunk, first in ((4096, 256), (1 << 20, 512), (100, 100)):
        sched = fx.chunk_schedule(r1, chunk, first_chunk=first)
        assert sched and sched[0] == min(first, chunk) and max(sched) <= chunk and sched[-1] <= max(sched)
        assert abs(sum(sched) - n) < 0.1 * n                                       # an estimate from the head of the file
        c1 = [len(c.seq_len) for c in fx.get_seq_chunks(r1, chunk, schedule=sched)]
        c2 = [len(c.seq_len) for c in fx.get_seq_chunks(r2, chunk, schedule=sched)]
        assert c1 == c2 and sum(c1) == n
        assert c1[:len(sched) - 1] == sched[:len(c1)][:len(sched) - 1][:len(c1[:len(sched) - 1])]   # the schedule is followed while the file lasts
    short = [len(c.seq_len) for c in fx.get_seq_chunks(r1, 4096, schedule=[1000, 500])]        # ends early: chunks of its last entry
    assert short[:2] == [1000, 500] and set(short[2:-1]) == {500} and sum(short) == n
    assert fx.chunk_schedule(str(tmp_path / "s_1.fq"), 4096, byte_range=(0, 0)) == []


def _bgzf_file(src, dst, blk):
    """src re-framed as BGZF with members of `blk` input bytes, empty members sprinkled in, BGZF's end-of-file block last"""
    import struct
    import zlib
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    data = open(src, "rb").read()
    with open(dst, "wb") as fh:
        for i in range(0, len(data), blk):
            piece = data[i:i + blk]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            d = co.compress(piece) + co.flush()
            fh.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(d) + 25) + d
                     + struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece)))
            if i % (3 * blk) == 0:
                fh.write(eof)
        fh.write(eof)


def test_bgzf_view_gives_the_plain_files_answers(tmp_path):
    """fx.BgzfView (positions in the decompressed stream of a BGZF file; the member index comes from the headers alone): size, text,
    find_record_start, count_records, skip_records equal the librd_host.so helpers on the plain file - and so do the ranges plan_ranges
    makes of them for mate files, for 1 ... 5 ranks"""
    import threading
    a1, o1, _ = synth.reads_numpy(6000, (40, 160), seed=51)
    a2, o2, _ = synth.reads_numpy(6000, (40, 160), seed=52)
    p1, p2 = str(tmp_path / "p_1.fq"), str(tmp_path / "p_2.fq")
    synth.write_fastq(p1, a1, o1, 1)
    synth.write_fastq(p2, a2, o2, 2, prefix="the_second_mate_has_longer_headers")
    g1, g2 = str(tmp_path / "r_1.fq.gz"), str(tmp_path / "r_2.fq.gz")
    _bgzf_file(p1, g1, 5000)
    _bgzf_file(p2, g2, 65280)
    assert fx.bgzf_all_the_way(g1) and not fx.bgzf_all_the_way(p1)
    rng = np.random.default_rng(3)
    for plain, gzp in ((p1, g1), (p2, g2)):
        v = fx.BgzfView(gzp)
        raw = open(plain, "rb").read()
        assert v.size == len(raw) and v.text(0, v.size) == raw and v.text(777, 70001) == raw[777:70001]
        for pos in list(rng.integers(0, v.size, 40)) + [0, 1, v.size - 1, v.size, v.size + 5]:
            assert v.find_record_start(pos) == fx.find_record_start(plain, pos), pos
        for _ in range(12):
            a = fx.find_record_start(plain, int(rng.integers(0, v.size)))
            b = fx.find_record_start(plain, int(rng.integers(a, v.size + 1)))
            assert v.count_records(a, b) == fx.count_records(plain, a, b)
            k = int(rng.integers(0, 2000))
            assert v.skip_records(a, k) == fx.skip_records(plain, a, k)
        c0, c1, drop = v.file_span(12345, v.size - 999)
        assert 0 <= c0 < c1 <= os.path.getsize(gzp) and 0 <= drop < 65280

    def plan(paths, views_of, world):
        res, store, bar = [None] * world, {}, threading.Barrier(world)

        def work(r):
            calls = [0]

            def ag(obj):
                k = calls[0]
                calls[0] += 1
                store.setdefault(k, [None] * world)[r] = obj
                bar.wait()
                out = list(store[k])
                bar.wait()
                return out
            res[r] = fx.plan_ranges(paths, r, world, ag, views=views_of() if views_of else None)
        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join() for t in th]
        return [[tuple(x) for x in r] for r in res]
    for world in (1, 2, 3, 5):
        assert plan([p1, p2], None, world) == plan([g1, g2], lambda: [fx.BgzfView(g1), fx.BgzfView(g2)], world)
    # a plain gzip member behind the blocks: the index walk refuses (such a file goes to the one-decoder path)
    import gzip
    with open(g1, "ab") as fh:
        fh.write(gzip.compress(b"@x\nAC\n+\nFF\n"))
    with pytest.raises(ValueError, match="without a size subfield"):
        fx.BgzfView(g1)


def test_feed_reader_equals_the_file_reader(tmp_path):
    """rd_reader_open_feed: the text arrives in pieces of any size from another thread (what the device inflate does with a BGZF
    file's text); the chunks equal the file reader's; an error given to rd_reader_feed_end surfaces after the records fed before it"""
    import ctypes as C
    import threading
    from ribodetector_amd import _native as N
    L = N.host_lib()
    arena, off, _ = synth.reads_numpy(5000, (30, 150), seed=9)
    p = str(tmp_path / "r.fq")
    synth.write_fastq(p, arena, off, 1)
    raw = open(p, "rb").read()
    want = [(c.buf[c.rec_start[0]:c.rec_start[-1]].tobytes(), c.seq_len.copy()) for c in fx.get_seq_chunks(p, chunk_size=700)]

    def run(pieces, error=b""):
        h = C.c_void_p()
        N.host_check(L.rd_reader_open_feed(0, C.byref(h)), "rd_reader_open_feed")

        def feeder():
            at = 0
            for k in pieces:
                piece = np.frombuffer(raw[at:at + k], dtype=np.uint8).copy()
                if len(piece) and L.rd_reader_feed(h, piece.ctypes.data, len(piece)) != 0:
                    return
                at += k
            L.rd_reader_feed_end(h, error)
        th = threading.Thread(target=feeder)
        th.start()
        got, err = [], None
        buf = np.zeros(1 << 20, dtype=np.uint8)
        rs, so, sl = np.zeros(701, np.int64), np.zeros(700, np.int64), np.zeros(700, np.int32)
        n, nb = C.c_int64(0), C.c_int64(0)
        try:
            while True:
                rc = L.rd_reader_next(h, 700, buf.ctypes.data, buf.size, rs.ctypes.data, so.ctypes.data, sl.ctypes.data, C.byref(n), C.byref(nb))
                if rc < 0:
                    err = L.rd_host_last_error().decode()
                    break
                if n.value:
                    got.append((buf[: nb.value].tobytes(), sl[: n.value].copy()))
                if rc == 1:
                    break
        finally:
            L.rd_reader_close(h)
            th.join()
        return got, err
    rng = np.random.default_rng(1)
    for pieces in ([len(raw)], [1] * 300 + [len(raw)], list(rng.integers(1, 70000, 4000)), [9 << 20]):
        got, err = run(pieces)
        # (chunks are "up to 700 records": compare the concatenation and the record lengths)
        assert err is None and b"".join(g[0] for g in got) == b"".join(w[0] for w in want)
        assert np.array_equal(np.concatenate([g[1] for g in got]), np.concatenate([w[1] for w in want]))
    cut = raw.index(b"\n@", len(raw) // 2) + 1
    got, err = run([cut], error=b"gzip member 7: CRC check failed")
    assert err == "gzip member 7: CRC check failed" and b"".join(g[0] for g in got) == raw[:cut]


def test_bgzf_view_on_fasta_gives_the_plain_files_answers(tmp_path):
    """BGZF FASTA (round 5): records start at '>' lines; multi-line sequences, blank lines, an empty-sequence record. The view's
    find_record_start / count_records / skip_records equal the librd_host.so helpers on the plain file"""
    rng = np.random.default_rng(8)
    recs = []
    for i in range(3000):
        L = int(rng.choice([0, 5, 60, 61, 200, 700]))
        seq = "".join(rng.choice(list("ACGTN"), L))
        lines = [seq[k:k + 60] for k in range(0, L, 60)]
        recs.append(">seq%d some description\n" % i + "".join(l + "\n" for l in lines) + ("\n" if i % 97 == 0 else ""))
    raw = "".join(recs).encode()
    plain, gzp = str(tmp_path / "x.fasta"), str(tmp_path / "x.fasta.gz")
    open(plain, "wb").write(raw)
    _bgzf_file(plain, gzp, 3000)
    v = fx.BgzfView(gzp)
    assert v.fasta and v.size == len(raw)
    for pos in list(rng.integers(0, v.size, 60)) + [0, 1, v.size - 1, v.size]:
        assert v.find_record_start(pos) == fx.find_record_start(plain, pos), pos
    for _ in range(25):
        a = fx.find_record_start(plain, int(rng.integers(0, v.size)))
        b = fx.find_record_start(plain, int(rng.integers(a, v.size + 1)))
        assert v.count_records(a, b) == fx.count_records(plain, a, b), (a, b)
        k = int(rng.integers(0, 500))
        assert v.skip_records(a, k) == fx.skip_records(plain, a, k), (a, k)
    # the ranges of 1 ... 5 ranks are the plain file's
    for world in (1, 2, 3, 5):
        for r in range(world):
            got = fx.plan_ranges([gzp], r, world, views=[fx.BgzfView(gzp)])
            want = fx.plan_ranges([plain], r, world)
            assert tuple(got[0]) == tuple(want[0])


def test_device_ingest_choice_for_fasta(tmp_path, monkeypatch):
    """which FASTA files the device reader (round 5: rd_fasta_index) would take: all of them when there is a GPU, none with
    RD_DEVICE_FASTA=0 or RD_DEVICE_PARSE=0 (the decision needs no GPU and is tested here)"""
    from ribodetector_amd.data_loader import device_reader as dr
    a = str(tmp_path / "a.fasta")
    open(a, "wb").write(b">r1\nACGT\n")
    import torch
    if not torch.cuda.is_available():
        assert dr.device_ingest_kind(a) is None
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    assert dr.device_ingest_kind(a) == "plain"
    monkeypatch.setenv("RD_DEVICE_FASTA", "0")
    assert dr.device_ingest_kind(a) is None
    monkeypatch.delenv("RD_DEVICE_FASTA")
    monkeypatch.setenv("RD_DEVICE_PARSE", "0")
    assert dr.device_ingest_kind(a) is None


def test_rd_ingest_folds_the_three_older_switches(monkeypatch):
    """RD_INGEST=device | members | host stands for RD_DEVICE_PARSE / RD_DEVICE_FASTA / RD_DEVICE_INFLATE; the old names still win"""
    import pytest
    from ribodetector_amd.data_loader import fastx_parser as fx
    for k in ("RD_INGEST", "RD_DEVICE_PARSE", "RD_DEVICE_FASTA", "RD_DEVICE_INFLATE"):
        monkeypatch.delenv(k, raising=False)
    assert [fx.ingest_env(k, d) for k, d in (("RD_DEVICE_PARSE", "1"), ("RD_DEVICE_FASTA", "1"), ("RD_DEVICE_INFLATE", "auto"))] == ["1", "1", "auto"]
    monkeypatch.setenv("RD_INGEST", "host")
    assert [fx.ingest_env(k, d) for k, d in (("RD_DEVICE_PARSE", "1"), ("RD_DEVICE_FASTA", "1"), ("RD_DEVICE_INFLATE", "auto"))] == ["0", "0", "0"]
    monkeypatch.setenv("RD_DEVICE_INFLATE", "members")
    assert fx.ingest_env("RD_DEVICE_INFLATE", "auto") == "members" and fx.ingest_env("RD_DEVICE_PARSE", "1") == "0"
    monkeypatch.delenv("RD_DEVICE_INFLATE")
    monkeypatch.setenv("RD_INGEST", "members")
    assert [fx.ingest_env(k, d) for k, d in (("RD_DEVICE_PARSE", "1"), ("RD_DEVICE_INFLATE", "auto"))] == ["1", "members"]
    monkeypatch.setenv("RD_INGEST", "sometimes")
    with pytest.raises(RuntimeError, match="RD_INGEST"):
        fx.ingest_env("RD_DEVICE_PARSE", "1")
