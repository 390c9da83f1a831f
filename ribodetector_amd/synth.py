"""Seeded synthetic read generator (SURVEY.md §8d "Synthetic inputs").

Base distribution: 90 % i.i.d. uniform ACGT reads, 10 % windows of an embedded
rRNA-like template (E. coli 16S 5' fragment, the known-answer string of
SURVEY.md §8c) with 5 % substitutions; 0.1 % of all bases are turned into 'N'.

Two back ends with the same distribution:
  * numpy  (`reads_numpy`)  - small fixtures, FASTQ files, the CPU-baseline sample
  * torch  (`reads_torch`)  - 10^7..10^8 reads generated directly in HBM for bench.py

The numpy stream is what the golden fixtures are generated from, so it must not
change without regenerating tests/golden (tests/golden/make_golden.py).
"""
import numpy as np

RRNA_16S = (
    "AAATTGAAGAGTTTGATCATGGCTCAGATTGAACGCTGGCGGCAGGCCTAACACATGCAAGTCGAACGGTAACAGGAAGAAGCTTGCTTC"
    "TTTGCTGACGAGTGGCGGACGGGTGAGTAATGTCTGGGAAACTGCCTGATGGAGGGGGATAACTACTGGAAACGGTAGC"
)
assert len(RRNA_16S) == 169

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_TEMPLATE = np.frombuffer(RRNA_16S.encode(), dtype=np.uint8)


def reads_numpy(n, length, seed, rrna_frac=0.10, sub_rate=0.05, n_rate=0.001):
    """Return (arena uint8[sum(len)], offsets int64[n+1], lens int32[n]).

    `length` is an int (fixed length) or a (lo, hi) tuple (uniform integer
    lengths, inclusive)."""
    rng = np.random.default_rng(seed)
    if isinstance(length, (tuple, list)):
        lens = rng.integers(length[0], length[1] + 1, size=n, dtype=np.int64)
    else:
        lens = np.full(n, int(length), dtype=np.int64)
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    total = int(offsets[-1])
    arena = _ACGT[rng.integers(0, 4, size=total, dtype=np.int64)]
    is_rrna = rng.random(n) < rrna_frac
    tl = len(_TEMPLATE)
    starts = rng.integers(0, tl, size=n, dtype=np.int64)
    # position of every base inside its read
    read_of = np.repeat(np.arange(n, dtype=np.int64), lens)
    pos = np.arange(total, dtype=np.int64) - offsets[:-1][read_of]
    sel = is_rrna[read_of]
    tpl = _TEMPLATE[(starts[read_of] + pos) % tl]
    keep = rng.random(total) >= sub_rate           # 5 % substitutions stay random
    arena = np.where(sel & keep, tpl, arena)
    arena = np.where(rng.random(total) < n_rate, np.uint8(ord("N")), arena).astype(np.uint8)
    return arena, offsets, lens.astype(np.int32)


def as_strings(arena, offsets):
    b = arena.tobytes()
    return [b[offsets[i]:offsets[i + 1]].decode() for i in range(len(offsets) - 1)]


def write_fastq(path, arena, offsets, mate=None, prefix="syn"):
    """4-line FASTQ, header @syn.<idx>[/mate], quality 'I' * len."""
    import gzip
    op = gzip.open if str(path).endswith("gz") else open
    b = arena.tobytes()
    with op(path, "wt") as fh:
        for i in range(len(offsets) - 1):
            s = b[offsets[i]:offsets[i + 1]].decode()
            tag = "@%s.%d" % (prefix, i) + ("/%d" % mate if mate else "")
            fh.write("%s\n%s\n+\n%s\n" % (tag, s, "I" * len(s)))


def write_fastq_realistic(path, arena, offsets, mate=1, seed=0):
    """4-line FASTQ shaped like sequencer output for the host-side benchmarks: Illumina-style headers
    (@A00123:45:HXXXXXXX:lane:tile:x:y mate:N:0:index) and NovaSeq-like binned qualities (F 90 %, ':' 6 %, ',' 3 %, '#' 1 %).
    gzip compresses it ~4.8x (the constant-quality files of write_fastq: ~6x), which is what real FASTQ does.
    Reads of any lengths."""
    import gzip
    n = len(offsets) - 1
    rng = np.random.default_rng(seed)
    offsets = np.asarray(offsets, dtype=np.int64)
    total = int(offsets[-1] - offsets[0]) if n else 0
    qual = np.frombuffer(b"F:,#", dtype=np.uint8)[rng.choice(4, size=total, p=[0.90, 0.06, 0.03, 0.01])].tobytes()
    seq = np.asarray(arena, dtype=np.uint8).tobytes()
    base = int(offsets[0]) if n else 0
    op = gzip.open if str(path).endswith("gz") else open
    with op(path, "wb") as fh:
        step = 100000
        for s in range(0, n, step):
            parts = []
            for i in range(s, min(n, s + step)):
                a, b = int(offsets[i]), int(offsets[i + 1])
                parts.append(b"@A00123:45:HXXXXXXX:1:%d:%d:%d %d:N:0:ACGTACGT\n" % (1101 + i // 400000, 1000 + i % 30000, 1000 + i // 7, mate))
                parts.append(seq[a:b])
                parts.append(b"\n+\n")
                parts.append(qual[a - base:b - base])
                parts.append(b"\n")
            fh.write(b"".join(parts))


def reads_torch(n, length, seed, device, rrna_frac=0.10, sub_rate=0.05, n_rate=0.001):
    """Fixed-length reads generated on `device`: (arena uint8[n*length], offsets int64[n+1], lens int32[n]).

    Same distribution as reads_numpy (not the same stream)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    L = int(length)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    tpl = torch.tensor(list(RRNA_16S.encode()), dtype=torch.uint8, device=device)
    arena = torch.empty(n * L, dtype=torch.uint8, device=device)
    step = 1 << 20                                   # reads per generation block (bounds temp memory)
    for s in range(0, n, step):
        m = min(step, n - s)
        base = acgt[torch.randint(0, 4, (m, L), generator=g, device=device)]
        is_r = torch.rand(m, 1, generator=g, device=device) < rrna_frac
        st = torch.randint(0, tpl.numel(), (m, 1), generator=g, device=device)
        win = tpl[(st + torch.arange(L, device=device)[None, :]) % tpl.numel()]
        keep = torch.rand(m, L, generator=g, device=device) >= sub_rate
        base = torch.where(is_r & keep, win, base)
        isn = torch.rand(m, L, generator=g, device=device) < n_rate
        base = torch.where(isn, torch.full_like(base, ord("N")), base)
        arena[s * L:(s + m) * L] = base.reshape(-1)
    offsets = torch.arange(n + 1, dtype=torch.int64, device=device) * L
    lens = torch.full((n,), L, dtype=torch.int32, device=device)
    return arena, offsets, lens


SEQLIKE_HEADER = b"@A00123:45:HXXXXXXX:1:tttt:xxxxx:yyyyy m:N:0:ACGTACGT"      # (t / x / y / m: digits filled in per record)


def fastq_record_bytes(lens, style="const"):
    """bytes of record i in the image fastq_image_torch builds: header line + 2 len + 4"""
    return (18 if style == "const" else len(SEQLIKE_HEADER) + 5) + 2 * lens


def fastq_image_torch(arena, offsets, lens, mate=1, block=1 << 21, style="const", seed=0):
    """4-line FASTQ text of the reads as one uint8 tensor on the reads' device - 10^7-read files for the full-size tests are built
    in HBM in a fraction of a second instead of a Python loop. style "const" (SURVEY 8d): record i = '@s' + 9-digit index + '/' + mate + LF,
    bases, LF '+' LF, 'I' * len, LF (18 + 2 len bytes). style "seqlike": what a sequencer writes - an Illumina header (SEQLIKE_HEADER: tile,
    x, y from the index; fixed width so that the image stays one vectorised scatter) and NovaSeq-like binned qualities (F 90 %, ':' 6 %,
    ',' 3 %, '#' 1 %, as write_fastq_realistic) - the text a .gz benchmark should compress. offsets: int64[n+1] (or [n]); lens: int32[n]."""
    import torch
    dev = arena.device
    n = int(lens.numel())
    L = lens.to(torch.int64)
    seqlike = style == "seqlike"
    H = (len(SEQLIKE_HEADER) + 1) if seqlike else 14          # header line with its LF
    rec = H + 4 + 2 * L
    start = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(rec, 0, out=start[1:])
    out = torch.full((int(start[-1].item()),), ord("I"), dtype=torch.uint8, device=dev)
    p10 = torch.tensor([10 ** k for k in range(8, -1, -1)], dtype=torch.int64, device=dev)
    gen = None
    if seqlike:
        gen = torch.Generator(device=dev)
        gen.manual_seed(1000 + 10 * seed + mate)
        base_hdr = torch.tensor(list(SEQLIKE_HEADER + b"\n"), dtype=torch.uint8, device=dev)
        tpos, xpos, ypos, mpos = SEQLIKE_HEADER.index(b"tttt"), SEQLIKE_HEADER.index(b"xxxxx"), SEQLIKE_HEADER.index(b"yyyyy"), SEQLIKE_HEADER.index(b" m:") + 1
        qchars = torch.tensor(list(b"F:,#"), dtype=torch.uint8, device=dev)
    for s in range(0, n, block):
        e = min(n, s + block)
        idx = torch.arange(s, e, dtype=torch.int64, device=dev)
        st, ln, of = start[s:e], L[s:e], offsets[s:e].to(torch.int64)
        if seqlike:
            hdr = base_hdr[None, :].repeat(e - s, 1)
            hdr[:, tpos:tpos + 4] = (((1101 + idx // 400000)[:, None] // p10[None, 5:]) % 10 + 48).to(torch.uint8)
            hdr[:, xpos:xpos + 5] = (((10000 + idx % 20000)[:, None] // p10[None, 4:]) % 10 + 48).to(torch.uint8)
            hdr[:, ypos:ypos + 5] = (((10000 + (idx // 7) % 90000)[:, None] // p10[None, 4:]) % 10 + 48).to(torch.uint8)
            hdr[:, mpos] = 48 + int(mate)
        else:
            hdr = torch.empty((e - s, 14), dtype=torch.uint8, device=dev)
            hdr[:, 0] = ord("@")
            hdr[:, 1] = ord("s")
            hdr[:, 2:11] = ((idx[:, None] // p10[None, :]) % 10 + 48).to(torch.uint8)
            hdr[:, 11] = ord("/")
            hdr[:, 12] = 48 + int(mate)
            hdr[:, 13] = 10
        out[(st[:, None] + torch.arange(H, device=dev)[None, :]).reshape(-1)] = hdr.reshape(-1)
        tot = int(ln.sum().item())
        if tot:
            rid = torch.repeat_interleave(torch.arange(e - s, device=dev), ln, output_size=tot)
            cum = torch.cumsum(ln, 0) - ln
            k = torch.arange(tot, dtype=torch.int64, device=dev) - cum[rid]
            out[st[rid] + H + k] = arena[of[rid] + k]
            if seqlike:
                u = torch.rand(tot, device=dev, generator=gen)
                q = (u >= 0.90).to(torch.int64) + (u >= 0.96).to(torch.int64) + (u >= 0.99).to(torch.int64)
                out[st[rid] + H + ln[rid] + 3 + k] = qchars[q]
        sep = st + H + ln
        out[sep] = 10
        out[sep + 1] = ord("+")
        out[sep + 2] = 10
        out[st + rec[s:e] - 1] = 10
    return out


def bgzip_file(src, dst, level=6, threads=None, block=65280):
    """`src` as BGZF the way bgzip / htslib write it - one zlib-deflated member per 65,280 input bytes, 'BC' subfield, the empty
    end-of-file member - by several threads. (This build's own device writer frames BGZF too, but with its own encoder; a benchmark of
    what a user's bgzip-made file costs must not read that.)"""
    import os
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or max(1, len(os.sched_getaffinity(0)))

    def members(data):
        out = []
        for i in range(0, len(data), block):
            piece = data[i:i + block]
            co = zlib.compressobj(level, zlib.DEFLATED, -15)
            d = co.compress(piece) + co.flush()
            out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(d) + 25) + d
                       + struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece)))
        return b"".join(out)
    with open(src, "rb") as fi, open(dst, "wb") as fo, ThreadPoolExecutor(threads) as ex:
        window = []
        while True:
            data = fi.read(block * 128)
            if not data:
                break
            window.append(ex.submit(members, data))
            if len(window) >= 2 * threads:
                fo.write(window.pop(0).result())
        for w in window:
            fo.write(w.result())
        fo.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


def pgzip_file(src, dst, level=6, threads=None, chunk=8 << 20, repeat=1):
    """`src` (repeated `repeat` times) as ONE gzip member at `dst`, deflated by several threads the way pigz does it: every chunk is a
    raw DEFLATE piece primed with the 32 KiB before it and closed with a sync flush, the pieces concatenate to a single stream (the last
    one ends it). For the benchmarks' multi-GB inputs only: gzip.compress of 16 GB takes a quarter of an hour on one core."""
    import os
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or max(1, len(os.sched_getaffinity(0)))
    size = os.path.getsize(src)

    def pieces():
        for _ in range(repeat):
            with open(src, "rb") as fh:
                while True:
                    b = fh.read(chunk)
                    if not b:
                        break
                    yield b

    def deflate(job):
        data, prime, last = job
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY, prime) if prime else zlib.compressobj(level, zlib.DEFLATED, -15)
        return co.compress(data) + co.flush(zlib.Z_FINISH if last else zlib.Z_SYNC_FLUSH)

    def jobs():
        prev, it = b"", pieces()
        cur = next(it, None)
        while cur is not None:
            nxt = next(it, None)
            yield (cur, prev[-32768:], nxt is None)
            prev, cur = cur, nxt
    crc, total = 0, 0
    with open(dst, "wb") as fo, ThreadPoolExecutor(threads) as ex:
        fo.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff")
        if size == 0:
            fo.write(zlib.compressobj(level, zlib.DEFLATED, -15).flush())
        window = []
        for job in jobs():
            crc = zlib.crc32(job[0], crc)
            total += len(job[0])
            window.append(ex.submit(deflate, job))
            if len(window) >= 2 * threads:
                fo.write(window.pop(0).result())
        for w in window:
            fo.write(w.result())
        fo.write(struct.pack("<II", crc & 0xffffffff, total & 0xffffffff))
    return total
