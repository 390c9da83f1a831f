"""ONE single-stream .gz shared by the ranks of a node: every rank decodes its own compressed range on its own GPU.

The reference opens a `.gz` by extension and reads it through `gzip.open(path, 'rt')` (data_loader/seq_encoder.py:21-39,75-87): one
DEFLATE stream, decodable only from its first byte on, because every match may copy from the 32 KiB of text before it. Up to round 5
this build therefore decoded such a file ONCE per node under torchrun (rank 0's host threads into /dev/shm, labels gathered) - about
1.5 GPUs' worth of reads on an 8-GPU node. Here the round-5 device decoder (csrc/rd_inflate_stream.hpp: block-start search, sections
decoded with an unknown window) is given the RANK as its outer section:

    P1  rank r decodes the compressed bytes [r S / W, (r + 1) S / W) (cut at multiples of the section size) into 16-bit SYMBOLS whose
        markers point into the 32 KiB in front of the range, and the range's MAP (the 32 KiB behind it in the same form, 64 KiB) -
        gz.DeviceRangeGunzip; nothing waits for another rank;
    X1  all-gather of the maps and of where every range started / stopped: a range must start exactly where the one before says the
        stream goes on (nothing speculative is accepted), the windows follow by applying the maps in rank order (W - 1 gathers of 32 Ki);
    P2  symbols -> bytes with the rank's window, CRC-32 of the range's text; the first record boundary of the text is looked up
        (FASTQ: an '@' line whose second-next line starts with '+', the rule of fastx_parser.plan_ranges; FASTA: a '>' line);
    X2  all-gather of (CRC, length, the bytes in front of the boundary): the CRCs are combined and compared with the member's trailer,
        rank r appends the head of rank r + 1 - the rest of the record its text ends in;
    P3  every batch is framed on the device (rd_fastq_index / rd_fasta_index); mate files: the record counts are exchanged, the cut
        in front of rank r moves forward to the same record index K_r in both files and the records in front of it travel to rank r - 1.

What comes out is, per input file, a list of framed batches resident in HBM (ResidentRange) that device_reader.get_seq_chunks_device
turns into chunks like any other device batch: the run goes on in the sharded-parse mode of plain and BGZF inputs (every rank writes its
part, parts joined in rank order, no label gather, no /dev/shm). Anything the device decoder does not take - fixed-Huffman or stored
blocks at a range's start, a second member, a CRC that does not match, too little memory - is found by every rank from the same gathered
facts, and the run falls back to the one-decode path (whose decoders are the authority on damaged files and their messages).
"""
import os
import time

import numpy as np
import torch

from .. import _native as N
from .. import gz
from . import device_reader as dr

WIN = 32768
MIN_RANGE = 1 << 20            # compressed bytes per rank below which a file is not worth sharing (RD_GZ_SHARD_MIN)
HEAD_PROBE = 1 << 20           # bytes of a range's text looked at for its first record boundary (then x4 up to the carry pad)


def range_bounds(size, world, section=gz.DeviceStreamGunzip.SECTION):
    """W + 1 file offsets: rank r decodes [b[r], b[r + 1]) - multiples of the section size, so that the search of a range's last batch and
    the search of the next range's first batch look at the same sections"""
    return [min(size, ((size * r // world) // section) * section) for r in range(world)] + [size]


class ResidentRange:
    """a rank's share of one input file, framed and resident in HBM: what get_seq_chunks_device takes as `byte_range`"""

    def __init__(self, ix, batches, skip, comp_bytes, stream, stats):
        self.ix, self.batches, self.skip, self.comp_bytes, self.stream, self.stats = ix, batches, int(skip), int(comp_bytes), stream, stats

    def __iter__(self):        # (start, end) of the compressed bytes: detect.py logs e - b as "Rank r parses ..."
        return iter((0, self.comp_bytes))


class ResidentFeeder:
    """device_reader.DeviceFeeder's consumer interface over batches that are already framed"""

    def __init__(self, rr):
        self.rr, self.ix = rr, rr.ix
        self.left = list(rr.batches)
        rr.batches = None          # (the range is consumed once; its buffers are freed as the chunks are gathered)
        self.stage_s = dict(rr.stats)

    def next_batch(self):
        return self.left.pop(0) if self.left else None

    def close(self):
        self.left = []
        st, self.rr.stream = self.rr.stream, None
        if st is not None:
            st.synchronize()
            gz.release_stream(st, priority=-1)


def find_record_cut(head, prev_byte, fasta):
    """offset of the first record start in `head` (bytes of a range's text; prev_byte = the text byte in front of it, None = none):
    FASTQ - a line that starts with '@' and whose second-next line starts with '+' (a quality line may start with '@', but then the
    second-next line is a sequence); FASTA - a line that starts with '>'. None: not inside these bytes."""
    mark = b">" if fasta else b"@"
    line = 0 if (prev_byte is None or prev_byte == 10) else head.find(b"\n") + 1
    if line == 0 and not (prev_byte is None or prev_byte == 10):
        return None
    while line < len(head):
        if head[line:line + 1] == mark:
            if fasta:
                return line
            l1 = head.find(b"\n", line)
            l2 = head.find(b"\n", l1 + 1) if l1 >= 0 else -1
            if l2 < 0 or l2 + 1 >= len(head):
                return None
            if head[l2 + 1:l2 + 2] == b"+":
                return line
        nl = head.find(b"\n", line)
        if nl < 0:
            return None
        line = nl + 1
    return None


class _Phase1:
    """P1 of one file on one rank: the compressed bytes [lo, hi) as 16-bit symbols. A gzip member that ENDS inside the range (a lane-merged
    file: `cat L001.fq.gz L002.fq.gz`) closes a SEGMENT - its trailer is kept for the CRC check - and the next member is decoded from its
    own first block with a fresh decoder: nothing in front of a member's first byte can be referenced, so that segment needs no window.
    A member that ends within a block of the share's END hands over to the next rank only if that rank's search found exactly the next
    member's first block (an empty or fixed-Huffman member there is not found: the ranks see the disagreement and the one-decode path reads
    the file - about one lane-merged file in 10^4)."""

    def __init__(self, path, lo, hi, size, first_bit, device, stream):
        self.path, self.lo, self.hi, self.size, self.first_bit, self.device, self.stream = path, lo, hi, size, first_bit, device, stream
        self.segs = []             # [{"syms": [(int16 tensor, n)], "n_text", "final", "trailer"}]
        self.meta = {"status": 0, "why": None, "n_text": 0, "first_abs": None, "next_abs": None, "ended": False, "fresh_after": False, "map": None,
                     "segs": [], "lo": lo, "hi": hi, "t_decode": 0.0}

    def run(self):
        t0 = time.perf_counter()
        m = self.meta
        if self.hi <= self.lo:                           # (an empty share: nothing starts here, the range before decodes through)
            m["status"], m["why"] = -1, "a rank's share of the file is empty"
            return m
        D = gz.DeviceRangeGunzip
        dg = D(self.device, self.stream)
        span = dg.BATCH + dg.SLACK + 8192
        want = min(span, self.hi - self.lo + dg.SLACK + 8192)
        pinned = [torch.empty(want, dtype=torch.uint8, pin_memory=True) for _ in range(2)]     # (one batch in flight while the next is read)
        views = [t.numpy() for t in pinned]
        free = [0, 1]
        flight = []
        fd = os.open(self.path, os.O_RDONLY)
        pos, first, next_data = self.lo, self.first_bit, None
        seg = {"syms": [], "n_text": 0, "final": False, "trailer": None}
        member_end = None          # file offset behind the trailer of the member that just ended (the batches in flight behind it are dropped)

        def finish_one():
            nonlocal member_end
            tk, slot, at = flight.pop(0)
            r = dg.finish(tk)
            free.append(slot)
            if m["status"] or member_end is not None:
                return
            if r["status"]:
                m["status"], m["why"] = r["status"], "%s in the batch at byte %d" % (gz.GZS_ERRORS.get(r["status"], "error %d" % r["status"]), at)
                return
            n = r["n_text"]
            if m["first_abs"] is None:
                m["first_abs"] = at * 8 + r["first_start"] if r["first_start"] != 0xffffffff else None
            with torch.cuda.stream(self.stream):
                seg["syms"].append((tk["sym"][:n].clone(), n))
            tk["sym"] = None
            seg["n_text"] += n
            m["n_text"] += n
            if r["final"]:
                end = at + (r["end_bit"] + 7) // 8
                tr = os.pread(fd, 8, end)
                if len(tr) < 8:
                    m["status"], m["why"] = -3, "Compressed file ended before the end-of-stream marker was reached"
                    return
                seg["final"], seg["trailer"] = True, tr
                member_end = end + 8
                m["next_abs"] = None
            else:
                m["next_abs"] = at * 8 + r["next_start"]
        try:
            while not m["status"]:
                if member_end is not None:               # a member ended: what follows it?
                    while flight:
                        finish_one()
                    self.segs.append(seg)
                    e = member_end
                    while e < self.size:                 # zero padding between members is skipped (Python's gzip module does)
                        rest = os.pread(fd, 1 << 16, e)
                        k = len(rest) - len(rest.lstrip(b"\0"))
                        e += k
                        if k < len(rest):
                            break
                    seg = None
                    if e >= self.size:
                        m["ended"] = True
                        break
                    hl = gz.gzip_header_len(os.pread(fd, 1 << 16, e))      # (ValueError: not a gzip member - the host path reports it)
                    if hl is None:
                        raise ValueError("Compressed file ended before the end-of-stream marker was reached")
                    m["fresh_after"] = True
                    if e >= self.hi:                     # the next member starts in the next rank's share: exactly there, or the ranks disagree
                        m["next_abs"] = (e + hl) * 8
                        break
                    dg = D(self.device, self.stream)     # the next member: a fresh decoder, from its first block
                    seg = {"syms": [], "n_text": 0, "final": False, "trailer": None}
                    pos, first, member_end = e, hl * 8, None
                    m["fresh_after"] = False
                    rem = (self.hi - pos) % dg.SECTION   # (a first batch that brings the section grid back in line with the share's end)
                    next_data = rem if rem else None
                    continue
                if pos >= self.hi:
                    if flight:
                        finish_one()
                        continue
                    break
                data = next_data or min(dg.BATCH, self.hi - pos)
                next_data = None
                valid = min(data + dg.SLACK, self.size - pos)
                at_eof = pos + valid >= self.size
                while len(flight) >= 2 and member_end is None and not m["status"]:
                    finish_one()
                while not free and flight:               # (a batch's pinned bytes are free once its state has arrived)
                    finish_one()
                if m["status"] or member_end is not None:
                    continue
                slot = free.pop()
                have = 0
                while have < valid:
                    k = os.preadv(fd, [memoryview(views[slot])[have:valid]], pos + have)
                    if k <= 0:
                        raise ValueError("Compressed file ended before the end-of-stream marker was reached")
                    have += k
                tk = dg.submit(pinned[slot], valid, data, first, at_eof)
                flight.append((tk, slot, pos))
                pos += data
                first = 0xffffffff
            while flight:
                finish_one()
            if seg is not None:
                self.segs.append(seg)
        finally:
            os.close(fd)
        if not m["status"]:
            self.stream.synchronize()
            m["map"] = dg.map.cpu().numpy().view(np.uint16).copy() if dg.map is not None else (np.arange(WIN, dtype=np.uint16) | 0x8000)
            m["segs"] = [{"n_text": sg["n_text"], "final": sg["final"], "trailer": sg["trailer"]} for sg in self.segs]
        m["t_decode"] = time.perf_counter() - t0
        self.dg = dg
        return m


def _in_threads(fn, items):
    """fn(item) for every item, side by side when there are several (the work of one input file next to the other's)"""
    import threading
    if len(items) <= 1:
        for it in items:
            fn(it)
        return
    th = [threading.Thread(target=fn, args=(it,), name="rd-gzr") for it in items]
    for t in th:
        t.start()
    for t in th:
        t.join()


def _verdict_x1(metas, world, sizes):
    """None, or why the ranks' ranges do not add up to the file's members - the same answer on every rank"""
    for f in range(len(sizes)):
        for r in range(world):
            m = metas[r][f]
            if m["status"]:
                return "%s (rank %d)" % (m["why"], r)
            if m["first_abs"] is None:
                return "rank %d found no block start" % r
            if r + 1 < world:
                if m["ended"]:
                    return "only padding behind the share of rank %d" % r
                if m["next_abs"] != metas[r + 1][f]["first_abs"]:
                    return "the share of rank %d does not start where rank %d says the stream goes on" % (r + 1, r)
            elif not m["ended"]:
                return "Compressed file ended before the end-of-stream marker was reached"
    return None


def start_window(metas, f, rank):
    """(window uint8[32768], valid) in front of rank `rank`'s share of file f: the maps of the ranks before applied in order; a rank whose
    share holds the end of a member hands on only what it decoded behind it (nothing in front of a member's first byte is text of it)"""
    window, valid = np.zeros(WIN, dtype=np.uint8), 0
    for r in range(rank):
        m = metas[r][f]
        if m["fresh_after"]:
            window, valid = np.zeros(WIN, dtype=np.uint8), 0
        elif len(m["segs"]) > 1:
            window, valid = gz.apply_map(m["map"], np.zeros(WIN, dtype=np.uint8)), min(WIN, m["segs"][-1]["n_text"])
        else:
            window, valid = gz.apply_map(m["map"], window), min(WIN, valid + m["n_text"])
    return window, valid


def fits(paths, world, device):
    """is a share of every file small enough to keep its symbols AND its text in HBM (3 bytes per text byte, 8 text bytes per compressed
    byte assumed), and large enough to be worth sharing?"""
    lim = int(os.environ.get("RD_GZ_SHARD_MIN", MIN_RANGE))
    sizes = [os.path.getsize(p) for p in paths]
    if min(sizes) // world < lim:
        return False, "less than %d compressed bytes per rank" % lim
    free, _ = torch.cuda.mem_get_info(device)
    need = sum(s // world for s in sizes) * 8 * 3 + (6 << 30)
    if need > free // 2:
        return False, "a rank's share needs about %d MB of device memory, %d MB are free" % (need >> 20, free >> 20)
    return True, None


def prepare(paths, rank, world, device, fasta, all_gather, shift_to_prev, log=None):
    """-> ([ResidentRange per file], None) or (None, why not). Collective: every rank calls it with the same paths; all_gather(obj) ->
    list over ranks, shift_to_prev(uint8 tensor or None) -> the bytes rank + 1 sent (uint8 tensor, possibly empty)."""
    t_start = time.perf_counter()
    device = torch.device(device)
    torch.cuda.set_device(device)
    sizes = [os.path.getsize(p) for p in paths]
    ok, why = fits(paths, world, device)
    oks = all_gather((bool(ok), why))
    for o, w in oks:
        if not o:
            return None, w
    streams = [gz.acquire_stream(device, priority=-1) for _ in paths]
    tph = {}                    # seconds per phase of this call (stats of every ResidentRange)
    tmark = [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        tph[name] = round(tph.get(name, 0.0) + now - tmark[0], 4)
        tmark[0] = now

    def give_up(why):
        for st in streams:
            st.synchronize()
            gz.release_stream(st, priority=-1)
        return None, why
    # ---- P1: symbols and map of this rank's range of every file ------------------------------------------------------------------------
    ph = []
    for f, path in enumerate(paths):
        b = range_bounds(sizes[f], world)
        first = gz.GZS_SEARCH
        if rank == 0:
            with open(path, "rb") as fh:
                hl = gz.gzip_header_len(fh.read(1 << 16))
            if hl is None:
                raise ValueError("Compressed file ended before the end-of-stream marker was reached")
            first = hl * 8
        ph.append(_Phase1(path, b[rank], b[rank + 1], sizes[f], first, device, streams[f]))

    def run_p1(p1):
        try:
            torch.cuda.set_device(device)
            p1.run()
        except ValueError as e:                     # (a short read: the file changed or is damaged - the host path reports it)
            p1.meta["status"], p1.meta["why"] = -3, str(e)
        except BaseException as e:                  # noqa: BLE001 - raised on the caller's thread below
            p1.meta["status"], p1.meta["why"], p1.error = -4, repr(e), e
    _in_threads(run_p1, ph)                         # (the mates' ranges side by side: file reads and kernels of one hide the other's waits)
    for p1 in ph:
        if getattr(p1, "error", None) is not None:
            raise p1.error
    lap("p1_decode")
    metas = all_gather([p.meta for p in ph])                         # X1: [rank][file]
    lap("x1")
    why = _verdict_x1(metas, world, sizes)
    if why is not None:
        return give_up(why)
    # ---- P2: the window in front of this rank's range; symbols -> bytes; first record boundary ----------------------------------------------
    out, heads_mine = [], []
    for f, path in enumerate(paths):
        window, valid = start_window(metas, f, rank)
        p1, st = ph[f], streams[f]
        ix = (dr.FastaIndexer if fasta[f] else dr.FastqIndexer)(device, st)
        texts, seg_states = [], []
        for si, sg in enumerate(p1.segs):           # a segment behind a member's end starts a member: no window in front of it
            with torch.cuda.stream(st):
                win_dev = torch.from_numpy(window).to(device) if (si == 0 and valid) else None
                rstate = torch.zeros(8, dtype=torch.int64, device=device)
            for sym, n in sg["syms"]:
                text = ix.alloc_text(n)
                if n:
                    p1.dg.resolve(sym, n, win_dev, valid if si == 0 else 0, text[dr.PAD:], rstate)
                texts.append((text, n))
            sg["syms"] = None
            seg_states.append(rstate)
        st.synchronize()
        seg_res, status = [], 0
        for rstate in seg_states:
            h = rstate.cpu().numpy()
            seg_res.append({"crc": int(h.view(np.uint32)[4]), "len": int(h.view(np.uint64)[0])})
            status = status or int(h.view(np.uint32)[5])
        cut, head = 0, b""
        if rank > 0 and not status:
            prev = int(window[-1]) if valid else None      # (None: a member starts with the share - and so does a line)
            probe, cut = HEAD_PROBE, None
            flat_n = sum(n for _, n in texts)
            while cut is None:
                take = min(probe, flat_n)
                parts, left = [], take
                for text, n in texts:
                    k = min(left, n)
                    if k:
                        parts.append(text[dr.PAD:dr.PAD + k].cpu().numpy().tobytes())
                    left -= k
                    if not left:
                        break
                head = b"".join(parts)
                cut = find_record_cut(head, prev, fasta[f])
                if cut is None and (take >= flat_n or probe >= dr.PAD):      # no record starts in (the first 16 MiB of) this share
                    cut = -1
                    break
                probe *= 4
            head = head[:cut] if cut is not None and cut >= 0 else b""
        heads_mine.append({"segs": seg_res, "status": status, "cut": cut, "head": head})
        out.append((ix, texts))
    lap("p2_resolve")
    x2 = all_gather(heads_mine)                                      # X2: [rank][file]
    lap("x2")
    for f in range(len(paths)):                     # every member's CRC-32 and ISIZE: the ranks' pieces combined in order
        crc, total = 0, 0
        for r in range(world):
            e = x2[r][f]
            if e["status"]:
                return give_up("invalid distance too far back (rank %d)" % r)
            if e["cut"] is None or e["cut"] < 0:
                return give_up("no record starts in the first %d bytes of the share of rank %d" % (dr.PAD, r))
            if len(e["segs"]) != len(metas[r][f]["segs"]):
                return give_up("rank %d resolved %d of %d segments" % (r, len(e["segs"]), len(metas[r][f]["segs"])))
            for got, sg in zip(e["segs"], metas[r][f]["segs"]):
                if got["len"] != sg["n_text"]:
                    return give_up("rank %d resolved %d of %d bytes" % (r, got["len"], sg["n_text"]))
                crc = gz.crc32_combine(crc, got["crc"], got["len"])
                total += got["len"]
                if sg["final"]:
                    tr = sg["trailer"]
                    if int.from_bytes(tr[:4], "little") != crc:
                        return give_up("CRC check failed")
                    if int.from_bytes(tr[4:], "little") != (total & 0xffffffff):
                        return give_up("Incorrect length of data produced")
                    crc, total = 0, 0
        if total:
            return give_up("Compressed file ended before the end-of-stream marker was reached")
    # ---- P3: framing; the head of the next rank closes this rank's last record ----------------------------------------------------------------
    ranges, counts = [], []
    for f, path in enumerate(paths):
        ix, texts = out[f]
        st = streams[f]
        last_rank = rank == world - 1
        if fasta[f]:
            ix.keep_empty_tail = not last_rank
        cut = x2[rank][f]["cut"]
        batches = []
        left = cut                                   # bytes of this rank's text that belong to the rank before
        for text, n in texts:
            skip = min(left, n)
            left -= skip
            if n - skip > 0 or not batches:
                batches.append(ix.index(text, dr.PAD + skip, dr.PAD + n))
        head = b"" if last_rank else x2[rank + 1][f]["head"]
        text = ix.alloc_text(len(head))
        if head:
            src = torch.from_numpy(np.frombuffer(head, dtype=np.uint8).copy()).pin_memory()
            N.copy_bytes(text[dr.PAD:dr.PAD + len(head)], src, len(head), st)
            st.synchronize()
        batches.append(ix.index(text, dr.PAD, dr.PAD + len(head), final=True))
        for b in batches:
            ix.finish(b)
        n_rec = 0
        for b in batches:
            if b.status:
                break                                # (a malformed record: the reader reports it where the records before it end)
            n_rec += b.n
        counts.append(n_rec)
        ranges.append([ix, batches, 0, st])
    lap("p3_frame")
    skips = [0] * len(paths)
    if len(paths) > 1:
        x3 = all_gather(counts)                                      # X3: [rank][file]
        lap("x3")
        before = [[sum(x3[q][f] for q in range(r)) for f in range(len(paths))] for r in range(world + 1)]
        if len(set(before[world])) != 1:
            raise ValueError("paired-end files have different numbers of records")
        K = [max(before[r]) for r in range(world + 1)]
        for r in range(world):
            for f in range(len(paths)):
                if K[r] - before[r][f] > x3[r][f]:
                    return give_up("the mate files drift apart by more than a rank's share")
        skips = [K[rank] - before[rank][f] for f in range(len(paths))]
        for f in range(len(paths)):                                  # X4: the records in front of K_r travel to rank r - 1
            ix, batches, _, st = ranges[f]
            send = None
            if skips[f] > 0:
                pieces, left = [], skips[f]
                for b in batches:
                    k = min(left, b.n)
                    if k:
                        pieces.append((b, 0, k))
                    left -= k
                    if not left:
                        break
                ch = ix.gather(pieces)
                ch.ready.synchronize()
                nb = int(ch.total[0])
                if nb < 0:
                    raise RuntimeError("device chunk assembly failed (rd_fastq_gather)")
                send = ch.dev[0][:nb]
            got = shift_to_prev(send)
            if got is not None and got.numel():
                nb = int(got.numel())
                text = ix.alloc_text(nb)
                with torch.cuda.stream(st):
                    text[dr.PAD:dr.PAD + nb].copy_(got)
                st.synchronize()
                keep, ix.keep_empty_tail = getattr(ix, "keep_empty_tail", False), True
                b = ix.finish(ix.index(text, dr.PAD, dr.PAD + nb, final=True, chain=False))
                ix.keep_empty_tail = keep
                if b.status or b.n != K[rank + 1] - before[rank + 1][f]:
                    raise RuntimeError("gz ranges: rank %d sent %d records, %d expected (status %d)" % (rank + 1, b.n, K[rank + 1] - before[rank + 1][f], b.status))
                batches.append(b)
            ranges[f][2] = skips[f]
        lap("x4_mates")
    t_all = time.perf_counter() - t_start
    res = []
    for f in range(len(paths)):
        ix, batches, skip, st = ranges[f]
        m = metas[rank][f]
        stats = {"path": "device", "mode": "gz-range", "comp_bytes": m["hi"] - m["lo"], "text_bytes": m["n_text"], "decode_s": round(m["t_decode"], 4),
                 "prepare_s": round(t_all, 4), "skip_records": skip, "batches": len(batches), "phases_s": dict(tph)}
        res.append(ResidentRange(ix, batches, skip, m["hi"] - m["lo"], st, stats))
    return res, None
