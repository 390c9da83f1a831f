"""End-to-end `ribodetector` CLI on the GPU (BASELINE configs[0] plumbing case and the paired-end modes):
output files must hold exactly the records the reference's label rules select, in input order."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _read(path):
    op = gzip.open if path.endswith("gz") else open
    with op(path, "rt") as fh:
        return fh.read()


def _fastq_text(arena, off, mate, idx):
    b = arena.tobytes()
    out = []
    for i in idx:
        s = b[off[i]:off[i + 1]].decode()
        out.append("@syn.%d/%d\n%s\n+\n%s\n" % (i, mate, s, "I" * len(s)))
    return "".join(out)


def test_cli_single_end_10k(tmp_path, oracle):
    from ribodetector_amd import detect, synth
    arena, off, lens = synth.reads_numpy(10000, 100, seed=0)          # cfg0: 10k SE 100 bp, seed 0
    inp = str(tmp_path / "in.fq")
    synth.write_fastq(inp, arena, off, mate=1)
    out, rr = str(tmp_path / "nonrrna.fq"), str(tmp_path / "rrna.fq.gz")
    p = detect.main(["-l", "100", "-i", inp, "-o", out, "-r", rr, "--chunk_size", "1", "-m", "3", "-t", "2"])
    ref = oracle.forward_packed(arena, off, lens, 100)
    lab = oracle.argmax(ref)
    margin = np.abs(ref[:, 1] - ref[:, 0])
    assert margin.min() > 2e-4                                          # no borderline read in this fixture
    assert p.num_read == 10000 and p.num_rrna == int(lab.sum()) and p.num_nonrrna == int((lab == 0).sum())
    assert p.writer_threads == [2]                                      # -t reaches the gzip writers
    assert _read(out) == _fastq_text(arena, off, 1, np.flatnonzero(lab == 0))
    assert _read(rr) == _fastq_text(arena, off, 1, np.flatnonzero(lab == 1))
    # whole-file mode gives the same files
    out2 = str(tmp_path / "nonrrna2.fq")
    detect.main(["-l", "100", "-i", inp, "-o", out2])
    assert _read(out2) == _read(out)


@pytest.mark.parametrize("ensure", ["none", "rrna", "norrna", "both"])
def test_cli_paired(tmp_path, oracle, ensure):
    from ribodetector_amd import detect, synth
    n = 3000
    a1, o1, l1 = synth.reads_numpy(n, (60, 120), seed=41, rrna_frac=0.3)
    a2, o2, l2 = synth.reads_numpy(n, (60, 120), seed=42, rrna_frac=0.3)
    i1, i2 = str(tmp_path / "r_1.fq.gz"), str(tmp_path / "r_2.fq.gz")
    synth.write_fastq(i1, a1, o1, 1)
    synth.write_fastq(i2, a2, o2, 2)
    outs = [str(tmp_path / "n1.fq"), str(tmp_path / "n2.fq")]
    rrs = [str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")]
    p = detect.main(["-l", "100", "-i", i1, i2, "-o", *outs, "-r", *rrs, "-e", ensure, "--chunk_size", "1", "-m", "3"])
    g1, g2 = oracle.forward_packed(a1, o1, l1, 100), oracle.forward_packed(a2, o2, l2, 100)
    lab = oracle.pair_fuse(g1, g2, ensure)
    # reads whose decision is numerically borderline may legitimately differ; exclude them from the exact comparison
    m1, m2 = np.abs(g1[:, 1] - g1[:, 0]), np.abs(g2[:, 1] - g2[:, 0])
    ms = np.abs((g1[:, 1] + g2[:, 1]) - (g1[:, 0] + g2[:, 0]))
    assert min(m1.min(), m2.min(), ms.min()) > 2e-4
    assert p.num_nonrrna == int((lab == 0).sum()) and p.num_rrna == int((lab == 1).sum())
    assert _read(outs[0]) == _fastq_text(a1, o1, 1, np.flatnonzero(lab == 0))
    assert _read(outs[1]) == _fastq_text(a2, o2, 2, np.flatnonzero(lab == 0))
    assert _read(rrs[1]) == _fastq_text(a2, o2, 2, np.flatnonzero(lab == 1))
    if ensure == "both":
        assert p.num_unknown == int((lab == -1).sum()) > 0
        assert _read(outs[0] + ".unclassified.gz") == _fastq_text(a1, o1, 1, np.flatnonzero(lab == -1))


@pytest.mark.parametrize("paired,chunk", [(False, 64), (True, 64), (True, 1)])
def test_cli_gz_outputs_are_deflated_on_the_device(tmp_path, paired, chunk, monkeypatch):
    """.gz outputs (reference detect.py:729-741: gzip level 5 by extension): by default the records are deflated on the GPU into BGZF
    members (csrc/rd_deflate.hpp); RD_DEVICE_GZIP=0 keeps the host's libdeflate writer. Same decompressed files either way, in
    input order; the device-written file is valid BGZF (every member carries its size, the file ends with the EOF marker), about
    the size of the host's level-5 output, and this build's own reader takes it back."""
    import struct
    from ribodetector_amd import detect, synth
    from ribodetector_amd.data_loader import fastx_parser as fx
    from ribodetector_amd.gz import eof_block
    n = 60000
    ins = []
    for m in range(2 if paired else 1):
        a, o, _ = synth.reads_numpy(n, (40, 140), seed=61 + m, rrna_frac=0.4)
        ins.append(str(tmp_path / ("r_%d.fq" % (m + 1))))
        synth.write_fastq(ins[-1], a, o, m + 1)
    res = {}
    for tag, env in (("device", None), ("host", "0")):
        if env is None:
            monkeypatch.delenv("RD_DEVICE_GZIP", raising=False)
        else:
            monkeypatch.setenv("RD_DEVICE_GZIP", env)
        outs = [str(tmp_path / ("%s_n%d.fq.gz" % (tag, m))) for m in range(len(ins))]
        rrs = [str(tmp_path / ("%s_r%d.fq.gz" % (tag, m))) for m in range(len(ins))]
        p = detect.main(["-l", "100", "-i", *ins, "-o", *outs, "-r", *rrs, "--chunk_size", str(chunk), "-m", "3"] + (["-e", "both"] if paired else []))
        files = outs + rrs + ([o + ".unclassified.gz" for o in outs] if paired else [])
        res[tag] = (files, (p.num_read, p.num_nonrrna, p.num_rrna, p.num_unknown))
    assert res["device"][1] == res["host"][1] and res["device"][1][0] == n and res["device"][1][2] > 0
    sizes = []
    for fd, fh in zip(res["device"][0], res["host"][0]):
        text = _read(fd)
        assert text == _read(fh) and len(text) > 0
        raw = open(fd, "rb").read()
        assert raw.endswith(eof_block()) and not open(fh, "rb").read().endswith(eof_block())
        pos, members = 0, 0
        while pos < len(raw):                                          # a chain of BGZF blocks, nothing else
            assert raw[pos:pos + 4] == b"\x1f\x8b\x08\x04" and raw[pos + 12:pos + 16] == b"BC\x02\x00"
            pos += struct.unpack("<H", raw[pos + 16:pos + 18])[0] + 1
            members += 1
        assert pos == len(raw) and members >= 2
        import shutil
        import subprocess
        if shutil.which("gzip"):                                       # the system's gzip accepts the file (every member's CRC-32 and ISIZE)
            assert subprocess.run(["gzip", "-t", fd], capture_output=True).returncode == 0
        import zlib
        if chunk > 1:
            assert len(raw) < 1.10 * len(zlib.compress(text.encode(), 5))  # the reference's compressor: gzip.open(..., compresslevel=5)
        sizes.append((os.path.basename(fd), len(raw), os.path.getsize(fh), len(zlib.compress(text.encode(), 5))))
        if fd.endswith(".fq.gz"):                                      # ('<out>.unclassified.gz' is not a name the readers take, reference included)
            back = b"".join(c.buf[c.rec_start[0]:c.rec_start[-1]].tobytes() for c in fx.get_seq_chunks(fd, chunk_size=7777))
            assert back.decode() == text
    print("device / host libdeflate-5 / zlib-5 bytes:", sizes)


def test_cli_gz_output_of_text_that_does_not_compress(tmp_path, monkeypatch):
    """records that barely shrink (IUPAC letters, full-range qualities: ~0.6 of their size). A chunk parsed on the host reserves half
    of the worst-case output size per file on the device: what does not fit is deflated by the host for that chunk. A chunk whose
    text lives on the device (the default for plain FASTQ since round 5) has no host copy to fall back to and reserves the full
    bound. Same file contents either way, valid gzip."""
    from ribodetector_amd import detect
    rng = np.random.default_rng(5)
    n = 30000
    iupac, quals = np.frombuffer(b"ACGTRYKMSWBDHVN-", np.uint8), np.arange(33, 127, dtype=np.uint8)
    inp = str(tmp_path / "noisy.fq")
    with open(inp, "wb") as fh:
        for i in range(n):
            L = int(rng.integers(80, 151))
            fh.write(b"@r%d\n%s\n+\n%s\n" % (i, iupac[rng.integers(0, 16, L)].tobytes(), quals[rng.integers(0, 94, L)].tobytes()))
    res = {}
    for tag, gzip_env, parse_env in (("device", None, None), ("hostparse", None, "0"), ("host", "0", None)):
        for k, v in (("RD_DEVICE_GZIP", gzip_env), ("RD_DEVICE_PARSE", parse_env)):
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, v)
        out = str(tmp_path / (tag + ".fq.gz"))
        p = detect.main(["-l", "100", "-i", inp, "-o", out, "--chunk_size", "64", "-m", "3"])
        res[tag] = (_read(out), p.num_read)
    assert res["device"] == res["host"] == res["hostparse"] and res["device"][1] == n and len(res["device"][0]) > 0
    raw = open(str(tmp_path / "hostparse.fq.gz"), "rb").read()
    assert b"RD\x04\x00" in raw[:64]                                    # the first member is the host writer's ('R','D' subfield), not a BGZF block
    assert b"BC\x02\x00" in open(str(tmp_path / "device.fq.gz"), "rb").read()[:64]      # device chunk: BGZF blocks (stored ones where nothing is gained)


def test_cli_argument_errors(tmp_path):
    from ribodetector_amd import detect
    with pytest.raises(RuntimeError):
        detect.main(["-l", "100", "-i", "a.fq", "b.fq", "-o", "x.fq"])
    with pytest.raises(RuntimeError):
        detect.main(["-l", "100", "-i", "a.fq", "-o", "x.fq", "-r", "r1.fq", "r2.fq"])


def test_cli_fasta_and_cpu_semantics(tmp_path, oracle):
    """FASTA input (multi-line, lower-case -> upper-cased by the parser, like the reference) and the --semantics cpu switch"""
    from ribodetector_amd import detect, synth
    arena, off, lens = synth.reads_numpy(2000, (50, 140), seed=77)
    seqs = synth.as_strings(arena, off)
    inp = str(tmp_path / "in.fa")
    with open(inp, "w") as fh:
        for i, s in enumerate(seqs):
            s2 = s.lower() if i % 3 == 0 else s
            fh.write(">seq%d some text\n%s\n%s\n" % (i, s2[:60], s2[60:]))
    for sem, fwd in (("gpu", oracle.forward_packed), ("cpu", oracle.forward_padded)):
        out = str(tmp_path / ("out_%s.fa" % sem))
        p = detect.main(["-l", "100", "-i", inp, "-o", out, "--semantics", sem])
        ref = fwd(arena, off, lens, 100)          # the parser upper-cases FASTA, so the original upper-case reads are what is classified
        lab = oracle.argmax(ref)
        assert np.abs(ref[:, 1] - ref[:, 0]).min() > 2e-4
        assert p.num_rrna == int(lab.sum())
        want = "".join(">seq%d some text\n%s\n" % (i, seqs[i]) for i in np.flatnonzero(lab == 0))
        assert _read(out) == want
        assert p.ingest["in.fa"]["path"] == "device"      # (round 5: FASTA is re-written and indexed on the GPU, rd_fasta_index)
    os.environ["RD_DEVICE_FASTA"] = "0"                   # the host parser: the same file
    try:
        out = str(tmp_path / "out_host.fa")
        p = detect.main(["-l", "100", "-i", inp, "-o", out])
        assert p.ingest.get("in.fa", {}).get("path") != "device" and _read(out) == _read(str(tmp_path / "out_gpu.fa"))
    finally:
        del os.environ["RD_DEVICE_FASTA"]


def _bgzf(src, dst):
    """src re-framed as BGZF (what bgzip writes): members of 65,280 input bytes + the empty end-of-file member"""
    import struct
    import zlib
    data = open(src, "rb").read()
    with open(dst, "wb") as fh:
        for i in range(0, len(data), 65280):
            piece = data[i:i + 65280]
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            d = co.compress(piece) + co.flush()
            fh.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(d) + 25) + d
                     + struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece)))
        fh.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


@pytest.mark.parametrize("suffix,world,shared", [("", 2, "1"), (".gz", 2, "1"), ("", 3, "1"), (".gz", 3, "1"), (".gz", 2, "0"), ("bgzf", 2, "1"), ("bgzf", 3, "1"), ("bgzf-gather", 2, "1"), ("bgzf-gather", 3, "0"), ("bgzf-mixed", 2, "1")])
def test_cli_two_ranks_match_one(tmp_path, suffix, world, shared):
    """torchrun x2 (both ranks on this box's one GPU, exchange over gloo) must write the same files as one process.
    Plain input: every rank parses only its own byte range (mates cut at the same record index), writes its own parts, rank 0
    joins them - each rank reads about half of the bytes. gzip input: rank 0 inflates and parses the stream ONCE into shared
    memory, the other ranks map each chunk and take their work-balanced share of its records (RD_SHARED_DECODE=0: every rank
    decodes the stream itself, the round-2 behaviour and what ranks on different nodes do); rank 0 gathers the labels and writes."""
    import re
    import socket
    import subprocess
    import sys
    from ribodetector_amd import detect, synth
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    n = 5000
    a1, o1, _ = synth.reads_numpy(n, (40, 160), seed=51, rrna_frac=0.3)
    a2, o2, _ = synth.reads_numpy(n, (40, 160), seed=52, rrna_frac=0.3)
    # BGZF inputs: every rank inflates, parses and writes the members of its own share on its GPU ("bgzf"), or - RD_BGZF_SHARD=0,
    # "bgzf-gather" - one decoding rank (or every rank for itself) inflates them all and the labels are gathered
    suffix_in = suffix
    bgzf, bgzf_shard = suffix.startswith("bgzf"), suffix == "bgzf"
    if bgzf:
        suffix = ".gz"
    i1, i2 = str(tmp_path / ("r_1.fq" + suffix)), str(tmp_path / ("r_2.fq" + suffix))
    if bgzf:
        synth.write_fastq(str(tmp_path / "p_1.fq"), a1, o1, 1)
        synth.write_fastq(str(tmp_path / "p_2.fq"), a2, o2, 2, prefix="the_second_mate_has_longer_headers")
        _bgzf(str(tmp_path / "p_1.fq"), i1)
        _bgzf(str(tmp_path / "p_2.fq"), i2)
        if suffix_in == "bgzf-mixed":      # a plain gzip member between the blocks of the first file: BGZF at both ends, but the index
            import gzip                    # walk of the sharded reader meets the member and the run falls back to one decoding rank
            raw = open(str(tmp_path / "p_1.fq"), "rb").read()
            cut = [raw.rfind(b"\n@syn.", 0, len(raw) * k // 10) + 1 for k in (4, 6)]
            for k, piece in enumerate((raw[:cut[0]], raw[cut[0]:cut[1]], raw[cut[1]:])):
                open(str(tmp_path / ("piece%d.fq" % k)), "wb").write(piece)
            _bgzf(str(tmp_path / "piece0.fq"), str(tmp_path / "piece0.gz"))
            _bgzf(str(tmp_path / "piece2.fq"), str(tmp_path / "piece2.gz"))
            with open(i1, "wb") as fh:
                fh.write(open(str(tmp_path / "piece0.gz"), "rb").read()[:-28] + gzip.compress(open(str(tmp_path / "piece1.fq"), "rb").read())
                         + open(str(tmp_path / "piece2.gz"), "rb").read())
            bgzf_shard = True              # (asked for; the run itself must decide against it)
        from ribodetector_amd.data_loader import fastx_parser as fx
        assert fx.device_inflate_wanted(i1)
    else:
        synth.write_fastq(i1, a1, o1, 1)
        synth.write_fastq(i2, a2, o2, 2, prefix="the_second_mate_has_longer_headers")
    one = [str(tmp_path / x) for x in ("a1.fq", "a2.fq.gz", "ar1.fq", "ar2.fq")]
    p = detect.main(["-l", "120", "-i", i1, i2, "-o", *one[:2], "-r", *one[2:], "-e", "both", "--chunk_size", "1", "-m", "3"])
    two = [str(tmp_path / x) for x in ("b1.fq", "b2.fq.gz", "br1.fq", "br2.fq")]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0", PYTHONPATH=root, RD_SHARED_DECODE=shared,
               RD_BGZF_SHARD="1" if bgzf_shard else "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "ribodetector_amd.detect", "-l", "120", "-i", i1, i2, "-o", *two[:2], "-r", *two[2:],
           "-e", "both", "--chunk_size", "1", "-m", "3"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("rd_%d_" % port)]   # the shared-memory chunk slots are gone
    assert p.num_read == n and p.num_unknown > 0
    for a, b in zip(one, two):
        assert _read(a) == _read(b) and len(_read(a)) > 0
    assert _read(one[0] + ".unclassified.gz") == _read(two[0] + ".unclassified.gz")
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]                      # the parts were joined and removed
    text = r.stdout + r.stderr                                                         # the logger writes to the console
    assert "Processed" in text and str(n) in text
    if suffix_in == "bgzf-mixed":
        assert "one rank decodes" in text and "parses" not in text
    elif not suffix or bgzf_shard:
        rows = re.findall(r"Rank (\d) parses (\d+), (\d+) bytes of (\d+), (\d+)", text)
        assert len(rows) == world
        for rk, b1, b2, t1, t2 in rows:
            assert 0.8 / world < int(b1) / int(t1) < 1.2 / world and 0.8 / world < int(b2) / int(t2) < 1.2 / world   # 1/W of each file per rank
        plain1, plain2 = (str(tmp_path / "p_1.fq"), str(tmp_path / "p_2.fq")) if bgzf else (i1, i2)     # (BGZF: positions in the text)
        assert sum(int(x[1]) for x in rows) == os.path.getsize(plain1) and sum(int(x[2]) for x in rows) == os.path.getsize(plain2)
    else:
        assert "parses" not in text


def test_cli_failing_rank_ends_the_job(tmp_path):
    """two ranks, plain FASTQ cut inside its last record: only rank 1's byte range holds the damage. Rank 1 reports it and leaves;
    rank 0 must not wait forever in the closing all-reduce - the launcher ends the job with a non-zero status"""
    import socket
    import subprocess
    import sys
    import time
    from ribodetector_amd import synth
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    arena, off, _ = synth.reads_numpy(20000, 100, seed=71)
    good = str(tmp_path / "g.fq")
    synth.write_fastq(good, arena, off, 1)
    blob = open(good, "rb").read()
    cut = str(tmp_path / "cut.fq")
    open(cut, "wb").write(blob[: len(blob) - 150])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0", PYTHONPATH=root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "ribodetector_amd.detect", "-l", "100", "-i", cut, "-o", str(tmp_path / "o.fq"),
           "--chunk_size", "1", "-m", "3"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and time.time() - t0 < 200
    assert "truncated FASTQ record" in r.stderr + r.stdout


def test_cli_damaged_gz_under_two_ranks_ends_the_job(tmp_path):
    """shared decode: rank 0 hits the damage, passes the error on instead of a chunk, and every rank leaves with it"""
    import socket
    import subprocess
    import sys
    import time
    from ribodetector_amd import synth
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    arena, off, _ = synth.reads_numpy(30000, 100, seed=61)
    good = str(tmp_path / "r.fq.gz")
    synth.write_fastq(good, arena, off, 1)
    blob = open(good, "rb").read()
    cut = str(tmp_path / "cut.fq.gz")
    open(cut, "wb").write(blob[: len(blob) // 2])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0", PYTHONPATH=root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "ribodetector_amd.detect", "-l", "100", "-i", cut, "-o", str(tmp_path / "o.fq"),
           "--chunk_size", "1", "-m", "3"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and time.time() - t0 < 200
    assert "ended before the end-of-stream marker" in r.stderr + r.stdout
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("rd_%d_" % port)]


def test_cli_damaged_bgzf_member_in_one_ranks_share_ends_the_job(tmp_path):
    """BGZF input sharded over two ranks, one byte flipped in a member of rank 1's share: rank 1's GPU reports the member (CRC-32 /
    code check), rank 1 leaves with the error, the job ends - no rank waits forever, no part file stays behind"""
    import socket
    import subprocess
    import sys
    import time
    from ribodetector_amd import synth
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    arena, off, _ = synth.reads_numpy(30000, 100, seed=61)
    plain = str(tmp_path / "p.fq")
    synth.write_fastq(plain, arena, off, 1)
    bad = str(tmp_path / "bad.fq.gz")
    _bgzf(plain, bad)
    blob = bytearray(open(bad, "rb").read())
    blob[len(blob) * 3 // 4] ^= 0x5a
    open(bad, "wb").write(bytes(blob))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0", PYTHONPATH=root)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "ribodetector_amd.detect", "-l", "100", "-i", bad, "-o", str(tmp_path / "o.fq"),
           "--chunk_size", "1", "-m", "3"]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and time.time() - t0 < 200
    assert "gzip member" in r.stderr + r.stdout or "size too small" in r.stderr + r.stdout or "not a gzip member" in r.stderr + r.stdout
    assert not os.path.exists(str(tmp_path / "o.fq"))


def test_cli_damaged_input_is_an_error(tmp_path):
    """a truncated .gz (or a FASTQ cut inside a record) stops the run with an error instead of writing a short output"""
    from ribodetector_amd import detect, synth
    arena, off, _ = synth.reads_numpy(30000, 100, seed=61)
    good = str(tmp_path / "r.fq.gz")
    synth.write_fastq(good, arena, off, 1)
    blob = open(good, "rb").read()
    cut = str(tmp_path / "cut.fq.gz")
    with open(cut, "wb") as fh:
        fh.write(blob[: len(blob) // 2])
    with pytest.raises(ValueError, match="ended before the end-of-stream marker"):
        detect.main(["-l", "100", "-i", cut, "-o", str(tmp_path / "o.fq"), "--chunk_size", "4", "-m", "3"])
    plain = str(tmp_path / "cut.fq")
    text = gzip.decompress(blob)
    lines = text.split(b"\n")
    with open(plain, "wb") as fh:               # ends inside the sequence line of record 10,000: 2 of 4 lines present
        fh.write(b"\n".join(lines[:40001]) + b"\n" + lines[40001][:25])
    with pytest.raises(ValueError, match="truncated FASTQ record"):
        detect.main(["-l", "100", "-i", plain, "-o", str(tmp_path / "o2.fq")])


def test_cpp_example_over_the_c_abi(tmp_path, gpu_model):
    """examples/classify_fastq (C++, no Python/torch: librd_host.so reader + librd_hip.so classifier through the two C
    ABIs) must give the labels the package gives for the same file."""
    import subprocess
    import torch
    from ribodetector_amd import synth
    from ribodetector_amd.data_loader import seq_encoder as E
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    exe = os.path.join(root, "examples", "classify_fastq")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    arena, off, lens = synth.reads_numpy(20000, (40, 140), seed=71, rrna_frac=0.3)
    fq = str(tmp_path / "r.fq.gz")
    synth.write_fastq(fq, arena, off, 1)
    lab_file = str(tmp_path / "labels.txt")
    weights = os.path.join(root, "ribodetector_amd", "data", "ribodetector_600k_variable_len70_101_epoch47.safetensors")
    r = subprocess.run([exe, weights, fq, "100", lab_file], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.array([int(x) for x in open(lab_file).read().split()], dtype=np.uint8)
    b = E.batch_from_numpy(arena, off[:-1], lens, "cuda")
    gpu_model.set_variant("auto")
    _, lab = gpu_model.classify_bytes(b.arena, b.offsets, b.lens, 100)
    torch.cuda.synchronize()
    want = lab.cpu().numpy()
    assert got.shape == want.shape and (got == want).all()
    assert "Processed 20000 sequences in total" in r.stdout
    assert "Detected %d rRNA sequences" % int(want.sum()) in r.stdout and "Detected %d non-rRNA sequences" % int((want == 0).sum()) in r.stdout


def test_cli_empty_and_tiny_inputs(tmp_path):
    """an empty file, a single read and a single pair go through the whole pipeline (no chunk, one-element chunks)"""
    from ribodetector_amd import detect
    empty = str(tmp_path / "empty.fq")
    open(empty, "w").close()
    out = str(tmp_path / "o.fq")
    p = detect.main(["-l", "100", "-i", empty, "-o", out])
    assert p.num_read == 0 and _read(out) == ""
    one = str(tmp_path / "one.fq")
    rec = "@r1\n" + "ACGT" * 25 + "\n+\n" + "I" * 100 + "\n"
    with open(one, "w") as fh:
        fh.write(rec)
    p = detect.main(["-l", "100", "-i", one, "-o", out, "-r", str(tmp_path / "r.fq")])
    assert p.num_read == 1 and p.num_nonrrna + p.num_rrna == 1
    assert _read(out) + _read(str(tmp_path / "r.fq")) == rec
    o1, o2 = str(tmp_path / "p1.fq.gz"), str(tmp_path / "p2.fq.gz")
    p = detect.main(["-l", "100", "-i", one, one, "-o", o1, o2, "-e", "both"])
    assert p.num_read == 1 and p.num_unknown == 0 and _read(o1) == _read(o2)


def test_cli_mismatched_pairs_and_fasta_gz(tmp_path, oracle):
    from ribodetector_amd import detect, synth
    arena, off, lens = synth.reads_numpy(1500, (60, 120), seed=81, rrna_frac=0.4)
    seqs = synth.as_strings(arena, off)
    fa = str(tmp_path / "in.fasta.gz")
    with gzip.open(fa, "wt") as fh:
        for i, s in enumerate(seqs):
            fh.write(">s%d\r\n%s\r\n" % (i, s))                         # CRLF line ends: stripped like the reference's rstrip/strip
    out, rr = str(tmp_path / "o.fa.gz"), str(tmp_path / "r.fa")
    p = detect.main(["-l", "100", "-i", fa, "-o", out, "-r", rr, "-e", "norrna"])
    lab = oracle.argmax(oracle.forward_packed(arena, off, lens, 100))
    assert p.num_rrna == int(lab.sum())
    assert _read(out) == "".join(">s%d\n%s\n" % (i, seqs[i]) for i in np.flatnonzero(lab == 0))
    assert _read(rr) == "".join(">s%d\n%s\n" % (i, seqs[i]) for i in np.flatnonzero(lab == 1))
    # R2 one record short
    f1, f2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    synth.write_fastq(f1, arena, off, 1)
    synth.write_fastq(f2, arena[: off[-2]], off[:-1], 2)
    with pytest.raises(ValueError, match="different numbers of records"):
        detect.main(["-l", "100", "-i", f1, f2, "-o", str(tmp_path / "x1.fq"), str(tmp_path / "x2.fq")])


@pytest.mark.parametrize("paired,ensure", [(False, "none"), (True, "rrna"), (True, "both")])
def test_cli_device_resident_ingest_writes_the_same_files(tmp_path, paired, ensure, monkeypatch):
    """round 5: FASTQ text stays on the device (plain files copied there, BGZF members inflated there; records framed by
    rd_fastq_index, selected by rd_select_pack / deflated by rd_gz_compress_selected). Every flow - plain / BGZF in, plain / gz out,
    CR LF input - writes the bytes RD_DEVICE_PARSE=0 (the host parser, the round-4 route) writes, with the same counts, and the run's
    record says which reader took each input."""
    from ribodetector_amd import detect, synth
    n = 150000
    ins = []
    for m in range(2 if paired else 1):
        a, o, _ = synth.reads_numpy(n, (40, 140), seed=70 + m, rrna_frac=0.3)
        p = str(tmp_path / ("r_%d.fq" % (m + 1)))
        synth.write_fastq(p, a, o, m + 1)
        ins.append(p)
    crlf = [p[:-3] + ".crlf.fq" for p in ins]
    for p, q in zip(ins, crlf):
        open(q, "wb").write(open(p, "rb").read().replace(b"\n", b"\r\n"))
    bg = [p + ".gz" for p in ins]
    for p, q in zip(ins, bg):
        _bgzf(p, q)
    ends = range(len(ins))

    def run(inputs, gz_out, tag):
        ext = ".fq.gz" if gz_out else ".fq"
        outs = [str(tmp_path / ("%s.non%d%s" % (tag, e, ext))) for e in ends]
        rrs = [str(tmp_path / ("%s.rr%d%s" % (tag, e, ext))) for e in ends]
        pr = detect.main(["-l", "100", "-i", *inputs, "-o", *outs, "-r", *rrs, "--chunk_size", "8", "-m", "3"] + (["-e", ensure] if paired else []))
        files = outs + rrs + ([o + ".unclassified.gz" for o in outs] if ensure == "both" else [])
        return pr, [_read(f) for f in files]
    monkeypatch.setenv("RD_DEVICE_PARSE", "0")
    want = {}
    for kind, inputs in (("plain", ins), ("bgzf", bg)):
        pr, want[kind] = run(inputs, False, "host_" + kind)
        assert not getattr(pr, "ingest", None)
    counts = (pr.num_read, pr.num_rrna, pr.num_nonrrna, pr.num_unknown)
    assert want["plain"] == want["bgzf"] and pr.num_read == n and pr.num_rrna > 0 and (ensure != "both" or pr.num_unknown > 0)
    monkeypatch.delenv("RD_DEVICE_PARSE")
    for kind, inputs in (("plain", ins), ("bgzf", bg), ("crlf", crlf)):
        for gz_out in (False, True):
            pr, got = run(inputs, gz_out, "dev_%s_%d" % (kind, gz_out))
            assert got == want["plain"], (kind, gz_out)
            assert (pr.num_read, pr.num_rrna, pr.num_nonrrna, pr.num_unknown) == counts
            assert len(pr.ingest) == len(inputs) and all(v["path"] == "device" and v["feeder"]["batches"] >= 1 for v in pr.ingest.values())
            if kind == "crlf":
                assert all(v["indexer"]["stripped"] >= 1 for v in pr.ingest.values())


def test_cli_single_stream_gz_inflated_on_the_device(tmp_path, monkeypatch):
    """the default since round 5: a plain .gz (one DEFLATE stream - what sequencers write) is decoded by the two-pass device decoder
    (csrc/rd_inflate_stream.hpp), its text framed and classified where it lands: the files of the default route (the host's parallel
    decoder), the same counts, and no host reader involved"""
    import gzip as gzmod
    from ribodetector_amd import detect, synth
    n = 120000
    ins = []
    for m in range(2):
        a, o, _ = synth.reads_numpy(n, (40, 140), seed=90 + m, rrna_frac=0.3)
        p = str(tmp_path / ("r_%d.fq" % (m + 1)))
        synth.write_fastq(p, a, o, m + 1)
        with open(p, "rb") as fi, gzmod.open(p + ".gz", "wb", compresslevel=6) as fo:
            fo.write(fi.read())
        ins.append(p + ".gz")

    def run(tag):
        outs = [str(tmp_path / ("%s.non%d.fq.gz" % (tag, e))) for e in range(2)]
        rrs = [str(tmp_path / ("%s.rr%d.fq" % (tag, e))) for e in range(2)]
        pr = detect.main(["-l", "100", "-i", *ins, "-o", *outs, "-r", *rrs, "-e", "rrna", "--chunk_size", "8", "-m", "3"])
        return pr, [_read(f) for f in outs + rrs]
    monkeypatch.setenv("RD_DEVICE_INFLATE", "members")          # the round-4 route: BGZF members on the GPU, a single stream on the host
    p0, want = run("host")
    assert not p0.ingest
    monkeypatch.delenv("RD_DEVICE_INFLATE")                     # the default since round 5
    p1, got = run("dev")
    assert got == want and (p1.num_read, p1.num_rrna, p1.num_nonrrna) == (p0.num_read, p0.num_rrna, p0.num_nonrrna) == (n, p0.num_rrna, n - p0.num_rrna)
    assert len(p1.ingest) == 2 and all(v["path"] == "device" and "fallback" not in v["feeder"] for v in p1.ingest.values())


def test_cli_bgzf_fasta_is_sharded_across_ranks(tmp_path):
    """round 5: a BGZF FASTA input is cut at '>' records in its decompressed stream like a BGZF FASTQ at '@' records: every rank
    inflates the members of its share on its GPU, parses (host parser: FASTA), classifies and writes its part; same files as one rank -
    multi-line sequences, an empty-sequence record at a cut included"""
    import re
    import socket
    import subprocess
    import sys
    from ribodetector_amd import detect, synth
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    n = 6000
    a, o, _ = synth.reads_numpy(n, (40, 260), seed=61, rrna_frac=0.3)
    b = a.tobytes()
    with open(str(tmp_path / "p.fasta"), "wb") as fh:
        for i in range(n):
            s = b[o[i]:o[i + 1]] if i % 500 else b""                     # (every 500th record has no sequence at all)
            fh.write(b">seq%d\n" % i + b"".join(s[k:k + 70] + b"\n" for k in range(0, len(s), 70)))
    inp = str(tmp_path / "in.fasta.gz")
    _bgzf(str(tmp_path / "p.fasta"), inp)
    one = [str(tmp_path / "one.non.fa"), str(tmp_path / "one.rr.fa.gz")]
    p = detect.main(["-l", "100", "-i", inp, "-o", one[0], "-r", one[1], "--chunk_size", "1", "-m", "3"])
    for world in (2, 3):
        two = [str(tmp_path / ("w%d.non.fa" % world)), str(tmp_path / ("w%d.rr.fa.gz" % world))]
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ, RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0", PYTHONPATH=root)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), "-m", "ribodetector_amd.detect", "-l", "100", "-i", inp, "-o", two[0], "-r", two[1], "--chunk_size", "1", "-m", "3"]
        r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        for x, y in zip(one, two):
            assert _read(x) == _read(y) and len(_read(x)) > 0
        rows = re.findall(r"Rank (\d) parses (\d+) bytes of (\d+) \(decompressed; BGZF members\)", r.stdout + r.stderr)
        assert len(rows) == world and sum(int(x[1]) for x in rows) == os.path.getsize(str(tmp_path / "p.fasta"))
    assert p.num_read > 0


def _torchrun(world, args, env_extra=None, timeout=900):
    import socket
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0", PYTHONPATH=root, **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "ribodetector_amd.detect"] + list(args)
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=timeout)
    return r, port


def _seqlike_fastq(path, n, seed, mate, lengths=(60, 150)):
    """FASTQ with Illumina-like headers and binned qualities as ONE gzip member (zlib level 6: what bcl2fastq / most pipelines write)"""
    from ribodetector_amd import synth
    arena, off, _ = synth.reads_numpy(n, lengths, seed=seed, rrna_frac=0.3)
    plain = path[:-3]
    synth.write_fastq_realistic(plain, arena, off, mate=mate, seed=seed)
    with open(plain, "rb") as fi, open(path, "wb") as fo:
        fo.write(gzip.compress(fi.read(), 6))
    return plain


@pytest.mark.parametrize("world", [2, 3, 8])
def test_cli_single_stream_gz_is_shared_by_the_ranks(tmp_path, world):
    """round 6: ONE single-stream .gz per mate, W ranks: every rank decodes its own compressed range on its GPU (gz_shard), frames,
    classifies and writes its part - no shared-memory decode, no label gather. Same files as one process; every rank reports its own
    range; the ranges add up to the files."""
    import re
    from ribodetector_amd import detect
    n = 80000
    i1, i2 = str(tmp_path / "r_1.fq.gz"), str(tmp_path / "r_2.fq.gz")
    _seqlike_fastq(i1, n, 71, 1)
    _seqlike_fastq(i2, n, 72, 2, lengths=(40, 120))           # (shorter mates: the two files' compressed positions drift apart)
    one = [str(tmp_path / x) for x in ("a1.fq", "a2.fq.gz", "ar1.fq.gz", "ar2.fq")]
    p = detect.main(["-l", "120", "-i", i1, i2, "-o", *one[:2], "-r", *one[2:], "-e", "both", "--chunk_size", "1", "-m", "3"])
    assert p.num_read == n and all(v["path"] == "device" for v in p.ingest.values())
    two = [str(tmp_path / x) for x in ("b1.fq", "b2.fq.gz", "br1.fq.gz", "br2.fq")]
    r, port = _torchrun(world, ["-l", "120", "-i", i1, i2, "-o", *two[:2], "-r", *two[2:], "-e", "both", "--chunk_size", "1", "-m", "3"],
                        {"RD_GZ_SHARD_MIN": "65536"})
    text = r.stdout + r.stderr
    assert r.returncode == 0, text[-3000:]
    for a, b in zip(one, two):
        assert _read(a) == _read(b) and len(_read(a)) > 0
    assert _read(one[0] + ".unclassified.gz") == _read(two[0] + ".unclassified.gz")
    assert not [f for f in os.listdir(tmp_path) if ".part" in f or ".joining" in f]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("rd_%d_" % port)]
    assert "one rank decodes" not in text
    rows = re.findall(r"Rank (\d) parses (\d+), (\d+) bytes of (\d+), (\d+) \(compressed; ranges of one DEFLATE stream\)", text)
    assert len(rows) == world
    assert sum(int(x[1]) for x in rows) == os.path.getsize(i1) and sum(int(x[2]) for x in rows) == os.path.getsize(i2)
    assert len(set(re.findall(r"Rank (\d) decoded \d+, \d+ compressed bytes into", text))) == world      # every rank decoded for itself


def test_cli_single_stream_gz_single_end_fasta_and_refusals(tmp_path):
    """single-end FASTQ and FASTA .gz under 3 ranks through the ranges; a lane-merged file (two members) is refused by every rank and read
    by the one-decode path with the same result; a cut file ends the job with zlib's message"""
    from ribodetector_amd import detect, synth
    n = 60000
    i1 = str(tmp_path / "s.fq.gz")
    plain = _seqlike_fastq(i1, n, 81, 1)
    one = [str(tmp_path / "a.fq.gz"), str(tmp_path / "ar.fq")]
    p = detect.main(["-l", "100", "-i", i1, "-o", one[0], "-r", one[1], "--chunk_size", "1", "-m", "3"])
    two = [str(tmp_path / "b.fq.gz"), str(tmp_path / "br.fq")]
    r, _ = _torchrun(3, ["-l", "100", "-i", i1, "-o", two[0], "-r", two[1], "--chunk_size", "1", "-m", "3"], {"RD_GZ_SHARD_MIN": "65536"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "ranges of one DEFLATE stream" in r.stdout + r.stderr
    assert [_read(x) for x in one] == [_read(x) for x in two] and p.num_read == n
    # FASTA, multi-line
    a, o, _ = synth.reads_numpy(30000, (40, 260), seed=61, rrna_frac=0.3)
    b = a.tobytes()
    fa = str(tmp_path / "p.fasta.gz")
    with gzip.open(fa, "wb", compresslevel=6) as fh:
        for i in range(30000):
            s = b[o[i]:o[i + 1]]
            fh.write(b">seq%d some text\n" % i + b"".join(s[k:k + 70] + b"\n" for k in range(0, len(s), 70)))
    one = [str(tmp_path / "fa.non.fa"), str(tmp_path / "fa.rr.fa.gz")]
    detect.main(["-l", "100", "-i", fa, "-o", one[0], "-r", one[1], "--chunk_size", "1", "-m", "3"])
    two = [str(tmp_path / "fb.non.fa"), str(tmp_path / "fb.rr.fa.gz")]
    r, _ = _torchrun(3, ["-l", "100", "-i", fa, "-o", two[0], "-r", two[1], "--chunk_size", "1", "-m", "3"], {"RD_GZ_SHARD_MIN": "65536"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "ranges of one DEFLATE stream" in r.stdout + r.stderr
    assert [_read(x) for x in one] == [_read(x) for x in two] and len(_read(one[0])) > 0
    # two lanes merged with cat: the member that ends inside a rank's share closes a segment there, the ranges go on
    text = open(plain, "rb").read()
    cutat = text.rfind(b"\n@", 0, len(text) // 2) + 1
    lanes = str(tmp_path / "lanes.fq.gz")
    open(lanes, "wb").write(gzip.compress(text[:cutat], 6) + gzip.compress(text[cutat:], 6))
    three = [str(tmp_path / "c.fq.gz"), str(tmp_path / "cr.fq")]
    r, port = _torchrun(3, ["-l", "100", "-i", lanes, "-o", three[0], "-r", three[1], "--chunk_size", "1", "-m", "3"], {"RD_GZ_SHARD_MIN": "65536"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "ranges of one DEFLATE stream" in r.stdout + r.stderr
    one = [str(tmp_path / "a.fq.gz"), str(tmp_path / "ar.fq")]
    assert [_read(x) for x in one] == [_read(x) for x in three]
    # stored blocks (gzip level 0): the block-start search does not take them - refused by every rank, read by the one-decode path
    lvl0 = str(tmp_path / "stored.fq.gz")
    open(lvl0, "wb").write(gzip.compress(text, 0))
    four = [str(tmp_path / "d.fq.gz"), str(tmp_path / "dr.fq")]
    r, port = _torchrun(3, ["-l", "100", "-i", lvl0, "-o", four[0], "-r", four[1], "--chunk_size", "1", "-m", "3"], {"RD_GZ_SHARD_MIN": "65536"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "one rank decodes" in r.stdout + r.stderr and "ranges of one DEFLATE stream" not in r.stdout + r.stderr
    assert [_read(x) for x in one] == [_read(x) for x in four]
    # a cut stream
    blob = open(i1, "rb").read()
    cut = str(tmp_path / "cut.fq.gz")
    open(cut, "wb").write(blob[: len(blob) * 2 // 3])
    r, port = _torchrun(2, ["-l", "100", "-i", cut, "-o", str(tmp_path / "o.fq"), "--chunk_size", "1", "-m", "3"], {"RD_GZ_SHARD_MIN": "65536"}, timeout=300)
    assert r.returncode != 0 and "ended before the end-of-stream marker" in r.stdout + r.stderr
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("rd_%d_" % port)]


def test_cli_mates_of_different_ingest_kinds(tmp_path):
    """advisor, round 5: whether the text stays on the device is decided ONCE per run - a pair whose first mate the device reader takes (a
    plain file) and whose second it does not (a gzip file under a plain name: the host reader sniffs the magic) used to crash on the first
    chunk; both go through the host reader now, and the outputs are those of two plain files"""
    from ribodetector_amd import detect, synth
    n = 30000
    a1, o1, _ = synth.reads_numpy(n, (40, 140), seed=91, rrna_frac=0.3)
    a2, o2, _ = synth.reads_numpy(n, (40, 140), seed=92, rrna_frac=0.3)
    p1, p2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")
    synth.write_fastq(p1, a1, o1, 1)
    synth.write_fastq(p2, a2, o2, 2)
    want = [str(tmp_path / x) for x in ("a1.fq", "a2.fq.gz")]
    pa = detect.main(["-l", "100", "-i", p1, p2, "-o", *want, "-e", "rrna", "--chunk_size", "1", "-m", "3"], log_level="WARNING")
    assert all(v["path"] == "device" for v in pa.ingest.values())
    disguised = str(tmp_path / "z_2.fq")                     # gzip bytes, plain name
    open(disguised, "wb").write(gzip.compress(open(p2, "rb").read(), 4))
    got = [str(tmp_path / x) for x in ("b1.fq", "b2.fq.gz")]
    pb = detect.main(["-l", "100", "-i", p1, disguised, "-o", *got, "-e", "rrna", "--chunk_size", "1", "-m", "3"], log_level="WARNING")
    assert not pb.ingest and pb.num_read == pa.num_read == n
    assert [_read(x) for x in want] == [_read(x) for x in got]
