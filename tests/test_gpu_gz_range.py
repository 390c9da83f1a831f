"""A RANGE of one DEFLATE stream decoded before the text in front of it is known (C ABI rd_gz_range_decode / rd_gz_range_resolve,
csrc/rd_inflate_stream.hpp) and the multi-rank reader built on it (data_loader/gz_shard.py).

What it replaces under W > 1 ranks: the one `gzip.open(path, 'rt')` of the reference (data_loader/seq_encoder.py:21-39,75-87) that a
single-stream .gz forces on ONE process. The properties: (1) the ranges' texts, resolved with the windows that follow from the exchanged
maps, concatenate to zlib's text byte for byte, for any number of ranges and any batch size, and the ranges' CRCs combine to the member's;
(2) the ranks' framed shares hold every record exactly once, mate files cut at the same record index; (3) what the decoder does not take
is refused by EVERY rank from the same facts (no rank goes on alone). The ranks are threads of this process here (the collectives are
barriers); tests/test_gpu_cli.py runs the same code under torchrun."""
import gzip
import os
import threading
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def fastq_bytes(n, seed=3, mate=1, long_headers=False):
    from ribodetector_amd import synth
    arena, off, _ = synth.reads_numpy(n, (60, 150), seed=seed)
    b = arena.tobytes()
    rng = np.random.default_rng(seed + 100 * mate)
    out = []
    for i in range(n):
        s = b[off[i]:off[i + 1]]
        q = bytes(rng.integers(35, 74, len(s), dtype=np.uint8))
        if i % 97 == 0:
            q = b"@" + q[1:]                      # quality lines that start with '@': the record-boundary rule must not take them
        h = b"@the_second_mate_has_much_longer_header_lines.%d/2 lane=3 tile=%d" % (i, i // 1000) if long_headers else b"@read.%d/%d" % (i, mate)
        out.append(b"%s\n%s\n+\n%s\n" % (h, s, q))
    return b"".join(out)


class Ranks:
    """W threads as ranks: all_gather / shift_to_prev as barriers over shared slots"""

    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world, timeout=300)
        self.slots = [None] * world
        self.sh = [None] * world

    def all_gather(self, rank):
        def f(obj):
            self.slots[rank] = obj
            self.bar.wait()
            out = list(self.slots)
            self.bar.wait()
            return out
        return f

    def shift(self, rank):
        def f(buf):
            self.sh[rank] = None if buf is None else buf.clone()
            self.bar.wait()
            got = self.sh[rank + 1] if rank + 1 < self.world else None
            self.bar.wait()
            return got
        return f

    def run(self, fn):
        res, err = [None] * self.world, []

        def work(r):
            try:
                torch.cuda.set_device(torch.device(DEV))
                res[r] = fn(r)
            except BaseException as e:      # noqa: BLE001
                err.append(e)
                self.bar.abort()
        th = [threading.Thread(target=work, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if err:
            real = [e for e in err if not isinstance(e, threading.BrokenBarrierError)]
            raise (real or err)[0]
        return res


def _ranges_text(path, world, monkeypatch=None):
    """P1 / X1 / P2 of gz_shard by hand, one range after the other: (text of every range, metas, crc)"""
    from ribodetector_amd import gz
    from ribodetector_amd.data_loader import gz_shard as gs
    dev = torch.device(DEV)
    size = os.path.getsize(path)
    b = gs.range_bounds(size, world)
    st = gz.acquire_stream(dev, priority=-1)
    ph = []
    for r in range(world):
        first = gz.GZS_SEARCH if r else gz.gzip_header_len(open(path, "rb").read(1 << 16)) * 8
        p = gs._Phase1(path, b[r], b[r + 1], size, first, dev, st)
        try:
            p.run()
        except ValueError as e:
            p.meta["status"], p.meta["why"] = -3, str(e)
        ph.append(p)
    metas = [[p.meta] for p in ph]
    why = gs._verdict_x1(metas, world, [size])
    if why is not None:
        return None, metas, why
    texts, crc, total = [], 0, 0
    for r in range(world):
        p = ph[r]
        window, valid = gs.start_window(metas, 0, r)
        parts = []
        for si, sg in enumerate(p.segs):
            rstate = torch.zeros(8, dtype=torch.int64, device=dev)
            win_dev = torch.from_numpy(window).to(dev) if (si == 0 and valid) else None
            seg_parts = []
            for sym, n in sg["syms"]:
                out = torch.empty(n + 64, dtype=torch.uint8, device=dev)
                if n:
                    p.dg.resolve(sym, n, win_dev, valid if si == 0 else 0, out, rstate)
                st.synchronize()
                seg_parts.append(out[:n].cpu().numpy().tobytes())
            h = rstate.cpu().numpy()
            assert int(h.view(np.uint32)[5]) == 0
            t = b"".join(seg_parts)
            assert int(h.view(np.uint64)[0]) == len(t) == sg["n_text"]
            assert int(h.view(np.uint32)[4]) == zlib.crc32(t)
            crc = gz.crc32_combine(crc, int(h.view(np.uint32)[4]), len(t))
            total += len(t)
            if sg["final"]:                       # a member ends here: its trailer holds the CRC-32 and size of everything since the last one
                assert int.from_bytes(sg["trailer"][:4], "little") == crc and int.from_bytes(sg["trailer"][4:], "little") == total & 0xffffffff
                crc, total = 0, 0
            parts.append(t)
        texts.append(b"".join(parts))
    assert total == 0                             # (the last segment of the last rank ended a member)
    gz.release_stream(st, priority=-1)
    return texts, metas, zlib.crc32(b"".join(texts))


@pytest.fixture(scope="module")
def fq(tmp_path_factory):
    d = tmp_path_factory.mktemp("gzr")
    text = fastq_bytes(120000)                       # ~36 MB of text, ~16 MB compressed
    path = str(d / "r_1.fq.gz")
    with open(path, "wb") as fh:
        fh.write(gzip.compress(text, 6))
    return path, text


@pytest.mark.parametrize("world,batch", [(1, None), (2, None), (3, 1 << 20), (5, 3 << 19), (8, None)])
def test_ranges_concatenate_to_zlibs_text(fq, world, batch, monkeypatch):
    path, text = fq
    if batch:
        monkeypatch.setenv("RD_GZS_BATCH", str(batch))       # several batches per range: the map and the carry chain on the device
    texts, metas, crc = _ranges_text(path, world)
    assert texts is not None, crc
    assert b"".join(texts) == text
    assert crc == zlib.crc32(text)
    assert all(len(t) > 0 for t in texts)
    for r in range(world - 1):                               # every range starts where the one before says the stream goes on
        assert metas[r][0]["next_abs"] == metas[r + 1][0]["first_abs"]


@pytest.mark.parametrize("world", [2, 3, 5])
def test_members_that_end_inside_a_range(tmp_path, world, monkeypatch):
    """a lane-merged file (`cat L001.fq.gz L002.fq.gz L003.fq.gz`, zero padding between two of them, an EMPTY member too): a member that ends
    inside a rank's share closes a segment (its trailer checked against the combined CRCs) and the next one is decoded from its own first
    block; the ranks' texts still concatenate to zlib's"""
    monkeypatch.setenv("RD_GZS_BATCH", str(1 << 20))
    text = fastq_bytes(90000, seed=13)
    cuts = [0, len(text) * 13 // 100, len(text) * 13 // 100, len(text) * 57 // 100, len(text)]      # (away from every rank boundary: see gz_shard._Phase1)
    blob = b"".join(gzip.compress(text[a:b], 6) + (bytes(100) if i == 1 else b"") for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])))
    path = str(tmp_path / "lanes.fq.gz")
    open(path, "wb").write(blob + bytes(3000))
    assert gzip.decompress(blob) == text
    texts, metas, crc = _ranges_text(path, world)
    assert texts is not None, crc
    assert b"".join(texts) == text
    assert sum(len(m[0]["segs"]) for m in metas) >= world + 2


@pytest.mark.parametrize("level,strategy", [(1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FILTERED)])
def test_ranges_of_other_encoders(tmp_path, level, strategy):
    text = fastq_bytes(40000, seed=11)
    co = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
    path = str(tmp_path / "x.fq.gz")
    open(path, "wb").write(co.compress(text) + co.flush())
    texts, _, crc = _ranges_text(path, 3)
    assert texts is not None, crc
    assert b"".join(texts) == text and crc == zlib.crc32(text)


def test_what_the_ranges_refuse(tmp_path):
    """stored blocks, a cut file, garbage behind the member: the verdict every rank computes from the gathered facts says no"""
    from ribodetector_amd.data_loader import gz_shard as gs
    text = fastq_bytes(30000, seed=5)
    stored = str(tmp_path / "stored.fq.gz")
    open(stored, "wb").write(gzip.compress(text, 0))
    texts, _, why = _ranges_text(stored, 2)
    assert texts is None, why
    cut = str(tmp_path / "cut.fq.gz")
    blob = gzip.compress(text, 6)
    open(cut, "wb").write(blob[: len(blob) * 3 // 4])
    texts, _, why = _ranges_text(cut, 2)
    assert texts is None, why
    junk = str(tmp_path / "junk.fq.gz")
    open(junk, "wb").write(blob + b"this is not a gzip member")
    texts, _, why = _ranges_text(junk, 2)
    assert texts is None and "zip" in why, why
    assert gs.find_record_cut(b"IIII\n@r2\nACGT\n+\nIIII\n", ord("I"), False) == 5
    assert gs.find_record_cut(b"@II\n@r2\nACGT\n+\n@III\n@r3\nAC\n+\nII\n", 10, False) == 4       # a quality line that starts with '@'
    assert gs.find_record_cut(b"ACGT\n+\nIIII\n@r", ord("A"), False) is None
    assert gs.find_record_cut(b"ACGT\n>r2 x\nAC\n", ord("A"), True) == 5


def _chunks_of(rr_list, chunk=20000):
    """the records a rank's ResidentRanges yield through the device reader: per file, list of (header, seq) ... as the raw record bytes"""
    from ribodetector_amd.data_loader import device_reader as dr
    out = []
    for path, rr in rr_list:
        recs = []
        for c in dr.get_seq_chunks_device(path, chunk_size=chunk, byte_range=rr, device=DEV):
            text, rs, so, sl = c.to_host()
            tb = text.tobytes()
            recs += [tb[rs[i]:rs[i + 1]] for i in range(c.n)]
        out.append(recs)
    return out


@pytest.mark.parametrize("world,paired", [(1, False), (2, False), (3, True), (8, True), (4, False)])
def test_ranks_shares_hold_every_record_once(tmp_path, world, paired, monkeypatch):
    """prepare() on W thread-ranks: SE - the shares concatenate to the file's records; PE - R1 and R2 (whose compressed positions drift:
    the second mate has longer headers and its own qualities) are cut at the same record index on every rank"""
    from ribodetector_amd.data_loader import gz_shard as gs
    monkeypatch.setenv("RD_GZ_SHARD_MIN", "65536")
    monkeypatch.setenv("RD_GZS_BATCH", str(2 << 20))
    n = 90000
    t1 = fastq_bytes(n, seed=21, mate=1)
    paths, texts = [str(tmp_path / "r_1.fq.gz")], [t1]
    open(paths[0], "wb").write(gzip.compress(t1, 6))
    if paired:
        t2 = fastq_bytes(n, seed=22, mate=2, long_headers=True)
        paths.append(str(tmp_path / "r_2.fq.gz"))
        texts.append(t2)
        open(paths[1], "wb").write(gzip.compress(t2, 4))
    G = Ranks(world)

    def rank(r):
        rr, why = gs.prepare(paths, r, world, DEV, [False] * len(paths), G.all_gather(r), G.shift(r))
        assert rr is not None, why
        skips = [x.skip for x in rr]
        return _chunks_of(list(zip(paths, rr))), skips
    res = G.run(rank)
    for f in range(len(paths)):
        lines = texts[f].split(b"\n")[:-1]
        fixed = [b"\n".join(lines[4 * i:4 * i + 4]) + b"\n" for i in range(len(lines) // 4)]
        got = [rec for r in range(world) for rec in res[r][0][f]]
        assert len(got) == n == len(fixed)
        assert got == fixed
    if paired:
        for r in range(world):
            assert len(res[r][0][0]) == len(res[r][0][1])          # the same record indices of both mates on every rank
        assert any(sk != [0, 0] for _, sk in res[1:])             # ... and the files DO drift: some cut had to move


def test_a_refusal_is_every_ranks_refusal(tmp_path, monkeypatch):
    from ribodetector_amd.data_loader import gz_shard as gs
    monkeypatch.setenv("RD_GZ_SHARD_MIN", "65536")
    text = fastq_bytes(40000, seed=31)
    p = str(tmp_path / "stored.fq.gz")
    open(p, "wb").write(gzip.compress(text, 0))
    G = Ranks(3)
    res = G.run(lambda r: gs.prepare([p], r, 3, DEV, [False], G.all_gather(r), G.shift(r)))
    assert all(rr is None for rr, _ in res) and len({why for _, why in res}) == 1
    # one flipped bit in the middle of a good member: a decode error, a share that no longer starts where the one before ends, or - when
    # the damaged bits still decode - a CRC-32 that does not match: whichever it is, the same refusal on every rank (the one-decode path
    # then reports the damage with zlib's words)
    blob = bytearray(gzip.compress(text, 6))
    for at in (len(blob) // 2, len(blob) // 3 + 7, len(blob) * 4 // 5):
        bad = bytearray(blob)
        bad[at] ^= 0x10
        q = str(tmp_path / "flipped.fq.gz")
        open(q, "wb").write(bytes(bad))
        G = Ranks(3)
        res = G.run(lambda r: gs.prepare([q], r, 3, DEV, [False], G.all_gather(r), G.shift(r)))
        assert all(rr is None for rr, _ in res) and len({why for _, why in res}) == 1, res
    tiny = str(tmp_path / "tiny.fq.gz")
    open(tiny, "wb").write(gzip.compress(text[:100000], 6))
    G = Ranks(2)
    res = G.run(lambda r: gs.prepare([tiny], r, 2, DEV, [False], G.all_gather(r), G.shift(r)))
    assert all(rr is None and "per rank" in why for rr, why in res)
