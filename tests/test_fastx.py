"""Host ingest (no GPU): record semantics of the reference parser (golden parser.json produced by the reference's
seq_parser) and the vectorised chunk reader / label-partitioned writer built on top."""
import gzip
import io
import os

import numpy as np
import pytest

from ribodetector_amd import synth
from ribodetector_amd.data_loader import fastx_parser as fx


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


def _records(chunk):
    b = chunk.buf.tobytes()
    return [b[chunk.seq_off[i]:chunk.seq_off[i] + chunk.seq_len[i]].decode() for i in range(len(chunk.seq_len))]


@pytest.mark.parametrize("suffix", ["fq", "fq.gz"])
def test_fastq_chunks(tmp_path, suffix, golden):
    arena, off, lens = synth.reads_numpy(1000, (30, 150), seed=2)
    p = str(tmp_path / ("r." + suffix))
    synth.write_fastq(p, arena, off, mate=1)
    want = synth.as_strings(arena, off)
    got, total = [], 0
    for c in fx.get_seq_chunks(p, chunk_size=333):
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
    chunks = list(fx.get_seq_chunks(p, chunk_size=1000))
    mask = np.random.default_rng(1).random(1000) < 0.3
    sel = fx.select_records(chunks[0], mask).decode()
    assert sel == "".join("\n".join(r) + "\n" for r, m in zip(ref, mask) if m)
    assert fx.select_records(chunks[0], np.zeros(1000, bool)) == b""


def test_fastq_edge_framing(tmp_path, golden):
    g = golden.json("parser")
    p = str(tmp_path / "odd.fastq")
    # CRLF endings, trailing blanks, '@' as first quality char, no final newline
    text = g["fastq_text"].replace("\n", "\r\n").rstrip("\r\n")
    with open(p, "w", newline="") as fh:
        fh.write(text)
    cs = list(fx.get_seq_chunks(p, chunk_size=2))
    assert [len(c.seq_len) for c in cs] == [2, 1]
    assert sum((_records(c) for c in cs), []) == [r[1] for r in g["fastq_records"]]
    assert not cs[0].verbatim
    out = b"".join(fx.select_records(c, np.ones(len(c.seq_len), bool)) for c in cs).decode()
    assert out == "".join("\n".join(r) + "\n" for r in g["fastq_records"])
    with open(p, "w") as fh:
        fh.write("@a\nACGT\n+\n")
    with pytest.raises(ValueError):
        list(fx.get_seq_chunks(p, 10))
    with open(p, "w") as fh:
        fh.write("a\nACGT\n+\nIIII\n")
    with pytest.raises(ValueError):
        list(fx.get_seq_chunks(p, 10))


def test_fasta_chunks(tmp_path, golden):
    g = golden.json("parser")
    p = str(tmp_path / "x.fa")
    open(p, "w").write(g["fasta_text"])
    cs = list(fx.get_seq_chunks(p, chunk_size=2))
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
