"""Device-resident FASTQ ingest: the text of the input never comes back to the host.

The reference parses its input on the host - `gzip.open(path, 'rt')` / `open(path)` (data_loader/seq_encoder.py:21-39) feeding the
four-line state machine of data_loader/fastx_parser.py:15-37 - and so did this build up to round 4, even for BGZF inputs whose
members it inflated on the GPU (the text went HBM -> pinned host -> parser -> chunk buffer -> HBM again). Here

    BGZF file   : compressed bytes -> pinned -> H2D -> rd_gz_inflate_members (one wave per member) ┐
    plain file  : file bytes       -> pinned -> H2D ─────────────────────────────────────────────────┴-> batch text in HBM
    batch text  -> rd_fastq_index (newline scan, record framing, carry of the partial record chained ON THE DEVICE)
    batches     -> rd_fastq_gather -> chunks of exactly N records: text + rec_start / seq_off / seq_len, all in HBM
    chunk       -> rd_classify -> labels -> rd_gz_compress_selected (.gz outputs) | rd_select_pack (plain outputs) -> D2H -> file

so the host moves compressed bytes (or the file's bytes once, in and out) and a 64-byte summary per batch. Record semantics are
those of csrc/rd_host.cpp's reader (tests/test_gpu_device_reader.py holds the two to each other and to the reference's parser):
every line rstrip()-ed (a batch with trailing whitespace - CR LF files - is stripped on the device and indexed again), header must
start with '@', last line may lack its newline, fewer than four blank trailing lines tolerated.
RD_DEVICE_PARSE=0 keeps the host parser. FASTA: FastaIndexer - the batch is re-written as header / joined upper-case sequence
(fastx_parser.py:39-55) by rd_fasta_index and indexed there; RD_DEVICE_FASTA=0: the host parser.
"""
import ctypes as C
import os
import queue
import threading
import time
from collections import deque

import numpy as np
import torch

from .. import _native as N
from .. import gz

PAD = 16 << 20                 # bytes in front of a batch's new text: room for the carry (= the longest record the device path frames)
LINE_DIV = 4                   # the line table of a batch holds window / LINE_DIV lines (a FASTQ line of reads is >= 4 bytes; a batch of
                               # shorter lines overflows it, is framed again with a full-size table, and so are the batches chained to it)
EVERY = 4096                   # one record-offset sample per EVERY records travels to the host with the summary: bounds a chunk's bytes
FQ_ERRORS = {1: "FASTQ record does not start with '@'",
             2: "truncated FASTQ record at end of file (number of lines is not a multiple of 4)",
             3: "a record longer than a batch buffer may be (2 GiB)",
             4: "the batch before this one could not be framed", 5: "more lines than the line table holds"}


def device_ingest_kind(path, fmt=None):
    """how the device reader would take this input: "plain" | "bgzf" (members that carry their size: inflated one wave per member) |
    "stream" (any other gzip file - one DEFLATE stream, what sequencers write: the two-pass decoder of csrc/rd_inflate_stream.hpp;
    RD_DEVICE_INFLATE=members keeps such files with the host's decoders) | None (FASTA, no GPU, RD_DEVICE_PARSE=0,
    RD_DEVICE_INFLATE=0 for .gz)"""
    from . import fastx_parser as fx
    if fx.ingest_env("RD_DEVICE_PARSE", "1") == "0" or not torch.cuda.is_available():
        return None
    fmt = fmt or fx.get_seq_format(path)
    if not fmt.startswith("fq"):
        # FASTA (round 5: rd_fasta_index re-writes and indexes the batch on the device); RD_DEVICE_FASTA=0 keeps the host parser
        if fx.ingest_env("RD_DEVICE_FASTA", "1") == "0":
            return None
    if fmt.endswith("gz"):
        if fx.device_inflate_wanted(path):
            return "bgzf"
        if fx.ingest_env("RD_DEVICE_INFLATE", "auto") not in ("0", "members") and fx.file_info(path)[1] and gz.is_member_indexed(path) is None:
            return "stream"
        return None
    return None if fx.file_info(path)[1] else "plain"      # (a gzip file under a plain name goes to the host reader, which sniffs the magic)


def device_parse_wanted(path, fmt=None):
    """does the text of this input stay on the device? A GPU and RD_DEVICE_PARSE != 0: plain FASTQ / FASTA files, BGZF files, and
    single-stream .gz files (the default since round 5; RD_DEVICE_INFLATE=members keeps those with the host's decoders)"""
    return device_ingest_kind(path, fmt) is not None


class _StreamFallback(Exception):
    """the device stream decoder gives the file back before it has delivered anything"""


class DeviceChunk:
    """A chunk of n records whose text and index live in HBM. The attribute names follow fastx_parser.Chunk where the meaning is the
    same: seq_len (len() = n), verbatim, release; `dev` = (text, seq_off, seq_len, rec_start) device tensors, `ready` = the event
    behind the kernels that wrote them, `total` = pinned int64[1]: the chunk's text bytes (-1: assembly failed), valid after `ready`."""
    tensors = None
    records = None
    release = None
    shm = None
    verbatim = True

    def __init__(self, n, text, rec_start, seq_off, seq_len, ready, total):
        self.n, self.dev, self.ready, self.total = n, (text, seq_off, seq_len, rec_start), ready, total
        self.seq_len = seq_len
        self.buf = self.rec_start = self.seq_off = None

    @staticmethod
    def from_host(c, device, stream):
        """a chunk of the host reader (fastx_parser.Chunk with its pinned tensors) shipped to the device: what a stream the device
        decoder gave back (get_seq_chunks_device) is delivered as, so that a run sees ONE kind of chunk"""
        tbuf, toff, tlen, trs = c.tensors
        nb = int(tbuf.numel())
        with torch.cuda.device(device), torch.cuda.stream(stream):
            text = torch.empty(((nb + 255) // 256) * 256 + 256, dtype=torch.uint8, device=device)
            text[:nb].copy_(tbuf, non_blocking=True)
            so, sl, rs = (t.to(device, non_blocking=True) for t in (toff, tlen, trs))
            total = torch.tensor([nb], dtype=torch.int64).pin_memory()
            ready = torch.cuda.Event()
            ready.record(stream)
        dc = DeviceChunk(len(c.seq_len), text, rs, so, sl, ready, total)
        dc.shm = c                   # (the pinned source stays alive until the chunk is dropped)
        return dc

    def to_host(self):
        """(text bytes, rec_start, seq_off, seq_len) as numpy arrays - tests and small tools only (this is the copy the path avoids)"""
        self.ready.synchronize()
        nb = int(self.total[0])
        if nb < 0:
            raise RuntimeError("device chunk assembly failed")
        text, so, sl, rs = self.dev
        return text[:nb].cpu().numpy(), rs.cpu().numpy(), so.cpu().numpy(), sl.cpu().numpy()


class _Batch:
    __slots__ = ("text", "line_end", "summary", "host", "event", "slot", "gz_slot", "n", "begin", "end", "consumed", "status", "dirty",
                 "bad_record", "n_lines", "final", "orig", "samples", "samples_host", "args", "chain", "norm", "rec_tab", "hdr_tab", "norm_end")

    def bytes_bound(self, lo, hi):
        """an upper bound (tight within 2 * EVERY records) of the text bytes of records [lo, hi)"""
        s = self.samples_host
        k1 = -(-hi // EVERY)
        top = int(s[k1]) if k1 * EVERY <= self.n and k1 < len(s) else (self.consumed if getattr(self, "norm", None) is None else self.norm_end)
        return top - int(s[lo // EVERY])


class FastqIndexer:
    """rd_fastq_index / rd_fastq_gather on one stream: index() queues the framing of a batch behind whatever produced its text (and
    behind the batch before: the carry is read on the device), finish() sleeps until its 64-byte summary has arrived, gather() makes
    a chunk of records of finished batches."""

    def __init__(self, device, stream):
        self.device, self.stream = torch.device(device), stream
        self.lib = N.lib()
        self.prev = None               # producer side: (text, summary) of the batch queued last - what the next one chains to
        self.last_good = None          # consumer side: the same of the batch FINISHED last (differs after a repair)
        self.last_carry = 0            # ... and the bytes behind its last complete record: what the next batch's pad must hold
        self.full_tables = False       # a batch overflowed its line table: the batches queued from now on get full-size tables
        self._lock = threading.Lock()  # (producer and consumer both touch `prev` after a repair)
        self.stats = {"batches": 0, "stripped": 0, "reframed": 0, "regrown": 0, "index_wait_s": 0.0}

    def _sp(self):
        return C.c_void_p(self.stream.cuda_stream)

    def alloc_text(self, new_bytes):
        """a batch buffer: PAD + new_bytes + slack (the framing reads whole 64-byte pieces and may append one '\\n')"""
        with torch.cuda.stream(self.stream):
            return torch.empty(((PAD + int(new_bytes) + 64 + 127) // 64) * 64, dtype=torch.uint8, device=self.device)

    def index(self, text, start, end, final=False, chain=True, prev=None, full_table=False, pad=None):
        """frame text[start:end] (offsets in the batch buffer; start >= PAD unless the batch stands alone) behind the carry of the
        batch indexed before (chain) or of an explicit `prev` = (text, summary). pad: the room in front of `start` when it is not PAD (a
        batch framed again behind a record longer than PAD). Asynchronous: returns the batch, to be finish()-ed."""
        b = _Batch()
        if prev is None and chain:
            with self._lock:
                prev = self.prev      # (text, summary) as they were queued: a strip replaces the batch's own fields, not these
        full_table = full_table or (chain and self.full_tables)
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            window = (end - start) + ((PAD if pad is None else pad) if prev is not None else 0)
            if window >= 0x7fffffff - 4096:
                raise ValueError(FQ_ERRORS[3])
            b.text = text
            b.line_end = torch.empty((window + 2) if full_table else (window // LINE_DIV + 4096), dtype=torch.int32, device=self.device)
            # what travels to the host: the 64-byte summary and the record-offset samples, in ONE buffer, fetched by a kernel
            # (rd_copy_bytes: an SDMA queue is shared in order with copies that wait for kernels)
            ns = window // (4 * EVERY) + 3
            meta = torch.empty(64 + 4 * ns, dtype=torch.uint8, device=self.device)
            meta_host = torch.empty(64 + 4 * ns, dtype=torch.uint8, pin_memory=True)
            b.summary, b.samples = meta[:64].view(torch.int64), meta[64:].view(torch.int32)
            b.host, b.samples_host = meta_host[:64].view(torch.int64), meta_host[64:].view(torch.int32)
            ws = torch.empty(max(int(self.lib.rd_fastq_index_workspace_bytes(end)), 256), dtype=torch.uint8, device=self.device)
            N.check(self.lib.rd_fastq_index(N.ptr(text), int(start), int(end), N.ptr(prev[0]) if prev is not None else None,
                                            N.ptr(prev[1]) if prev is not None else None, 1 if final else 0, N.ptr(b.line_end),
                                            b.line_end.numel(), N.ptr(b.summary), N.ptr(ws), ws.numel(), self._sp()), "rd_fastq_index")
            N.check(self.lib.rd_fastq_sample(N.ptr(b.line_end), N.ptr(b.summary), EVERY, N.ptr(b.samples), ns, self._sp()), "rd_fastq_sample")
            N.copy_bytes(meta_host, meta, meta.numel(), self.stream, workgroups=4)
            b.event = N.new_event()
            b.event.record(self.stream)
        b.final, b.orig, b.slot, b.gz_slot, b.n = final, None, None, None, None
        b.args, b.chain = (start, end, final), (text, b.summary)
        if chain:
            with self._lock:
                self.prev = b.chain
        self.stats["batches"] += 1
        return b

    _REFRAMED = ("text", "line_end", "summary", "host", "samples", "samples_host", "begin", "end", "consumed", "n", "n_lines", "status", "dirty", "bad_record", "chain")

    def _reframe(self, b):
        """a batch that could not be framed as it was queued - more lines than its table holds (5: lines of < LINE_DIV bytes on average),
        chained to a batch that was repaired (4), or behind a record longer than the pad (3: a FASTA contig, an ultra-long read; the
        reference's parser joins lines without bound, fastx_parser.py:39-55) - is framed again on the consumer's side behind the batch
        FINISHED last: full-size tables, and a buffer whose pad holds that batch's carry, whatever its size. The producer's chain is
        moved to the repaired batch if it still ends in this one; the batches queued in between come through here too."""
        self.stats["reframed"] += 1
        start, end, final = b.args
        old_chain = b.chain
        text, pad = b.text, None
        if self.last_good is not None and self.last_carry > PAD:
            self.stats["regrown"] += 1
            pad = ((self.last_carry + (1 << 16) + 63) // 64) * 64
            if pad + (end - start) >= 0x7fffffff - 4096:
                raise ValueError(FQ_ERRORS[3])
            import logging
            logging.getLogger("predict").info("A record of more than %d bytes: the batch behind it is framed again with room for %d" % (PAD, pad))
            with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
                text = torch.empty(((pad + (end - start) + 64 + 127) // 64) * 64, dtype=torch.uint8, device=self.device)
                text[pad:pad + (end - start)].copy_(b.text[start:end])
            start, end = pad, pad + (end - start)
        if b.status == 5:
            self.full_tables = True
        nb = self.index(text, start, end, final=final, chain=False, prev=self.last_good, full_table=True, pad=pad)
        self.stats["batches"] -= 1
        self.wait(nb)
        self._read(nb)
        for k in self._REFRAMED:
            setattr(b, k, getattr(nb, k))
        b.args = nb.args
        with self._lock:
            if self.prev is old_chain:
                self.prev = b.chain

    def wait(self, b):
        t0 = time.perf_counter()
        N.wait_event(b.event)
        self.stats["index_wait_s"] += time.perf_counter() - t0

    @staticmethod
    def _read(b):
        h = b.host.numpy()
        b.begin, b.end, b.n_lines, b.n, b.consumed, b.bad_record, sd = (int(x) for x in h[:7])
        b.status, b.dirty = sd & 0xffffffff, (sd >> 32) & 0xffffffff
        b.samples_host = b.samples_host.numpy() if torch.is_tensor(b.samples_host) else b.samples_host

    def finish(self, b):
        """wait (sleeping) for the batch's summary; strips a dirty batch. Returns the batch with n / status filled in - a status != 0 is
        left to the caller (the records before the damage are delivered first)."""
        self.wait(b)
        self._read(b)
        if b.status in (3, 4, 5) and b.chain is not None:
            self._reframe(b)
        if b.chain is not None:
            self.last_good = b.chain
            self.last_carry = (b.end - b.consumed) if b.status == 0 else 0
        if b.dirty and b.status in (0, 2):       # (a truncated tail does not excuse the records in front of it from rstrip())
            self._strip(b)
        if b.status == 0 and b.bad_record != -1 and b.bad_record < b.n:
            b.status = 1
        return b

    def _strip(self, b):
        """a batch with trailing whitespace on some line: the bytes rstrip() removes are marked, the window is compacted and indexed
        again as a stream of its own (the complete records of the window: [begin, consumed); everything when the batch is final).
        The ORIGINAL buffer and summary stay what the next batch's carry is read from."""
        self.stats["stripped"] += 1
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            stop = b.end if b.final else b.consumed
            dele = torch.zeros(b.text.numel(), dtype=torch.uint8, device=self.device)
            N.check(self.lib.rd_fastq_strip_mark(N.ptr(b.text), N.ptr(b.line_end), N.ptr(b.summary), b.n_lines, N.ptr(dele), self._sp()),
                    "rd_fastq_strip_mark")
            kept = b.text[b.begin:stop][dele[b.begin:stop] == 0]
            text = torch.empty(((kept.numel() + 64 + 127) // 64) * 64, dtype=torch.uint8, device=self.device)
            text[:kept.numel()] = kept
            nb = self.index(text, 0, int(kept.numel()), final=True, chain=False, full_table=True)
        self.stats["batches"] -= 1
        nb.chain = None                       # (a stream of its own: nothing chains to it)
        nb = self.finish(nb)
        b.orig = (b.text, b.summary)          # (alive until the next batch's carry copy has been queued behind it)
        for k in ("text", "line_end", "summary", "samples", "samples_host", "begin", "end", "consumed", "n", "n_lines", "status", "bad_record"):
            setattr(b, k, getattr(nb, k))
        b.dirty = 0

    def gather(self, pieces):
        """pieces: [(batch, lo, hi)] -> DeviceChunk of sum(hi - lo) records, in that order"""
        n = sum(hi - lo for _, lo, hi in pieces)
        cap = sum(b.bytes_bound(lo, hi) for b, lo, hi in pieces)
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            text = torch.empty(((cap + 255) // 256) * 256 + 256, dtype=torch.uint8, device=self.device)
            rs = torch.empty(n + 1, dtype=torch.int64, device=self.device)
            so = torch.empty(n, dtype=torch.int64, device=self.device)
            sl = torch.empty(n, dtype=torch.int32, device=self.device)
            cursor = torch.zeros(len(pieces) + 1, dtype=torch.int64, device=self.device)
            at = 0
            for i, (b, lo, hi) in enumerate(pieces):
                N.check(self.lib.rd_fastq_gather(N.ptr(b.text), N.ptr(b.line_end), N.ptr(b.summary), lo, hi, b.bytes_bound(lo, hi), N.ptr(text),
                                                 text.numel(), C.c_void_p(cursor.data_ptr() + 8 * i), C.c_void_p(cursor.data_ptr() + 8 * (i + 1)),
                                                 C.c_void_p(rs.data_ptr() + 8 * at), C.c_void_p(so.data_ptr() + 8 * at),
                                                 C.c_void_p(sl.data_ptr() + 4 * at), self._sp()), "rd_fastq_gather")
                at += hi - lo
            total = torch.empty(1, dtype=torch.int64, pin_memory=True)
            N.copy_bytes(total.view(torch.uint8), cursor[len(pieces):].view(torch.uint8), 8, self.stream, workgroups=1)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return DeviceChunk(n, text, rs, so, sl, ready, total)


LINE_DIV_FA = 8               # FASTA: the line table of a batch holds window / 8 lines (shorter lines on average - reads under 14 bases: framed again, full size)


class FastaIndexer(FastqIndexer):
    """rd_fasta_index / rd_fasta_gather behind the interface of FastqIndexer: a batch of FASTA text is RE-WRITTEN on the device into
    `norm` (header, newline, the record's sequence lines joined and upper-cased, newline - fastx_parser.py:39-55 and the writer's join) and
    indexed there (rec_tab / hdr_tab); the carry of a batch is the raw text from its last header line on. No strip pass: strip() is
    part of the re-writing. keep_empty_tail: the stream is a rank's share of a file that goes on behind it (its last record counts even
    without a sequence; the end of the FILE drops such a record, like the reference)."""
    keep_empty_tail = False

    _REFRAMED = FastqIndexer._REFRAMED + ("norm", "rec_tab", "hdr_tab", "norm_end")

    def index(self, text, start, end, final=False, chain=True, prev=None, full_table=False, pad=None):
        b = _Batch()
        if prev is None and chain:
            with self._lock:
                prev = self.prev
        full_table = full_table or (chain and self.full_tables)
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            window = (end - start) + ((PAD if pad is None else pad) if prev is not None else 0)
            if window >= 0x7fffffff - 4096:
                raise ValueError(FQ_ERRORS[3])
            cap_lines = (window + 2) if full_table else (window // LINE_DIV_FA + 4096)
            cap_rec = cap_lines + 2
            norm_cap = ((window + cap_lines + 66 + 255) // 256) * 256
            b.text = text
            b.line_end = torch.empty(cap_lines, dtype=torch.int32, device=self.device)
            b.norm = torch.empty(norm_cap, dtype=torch.uint8, device=self.device)
            b.rec_tab = torch.empty(cap_rec, dtype=torch.int64, device=self.device)
            b.hdr_tab = torch.empty(cap_rec, dtype=torch.int32, device=self.device)
            ns = cap_rec // EVERY + 3
            meta = torch.empty(64 + 4 * ns, dtype=torch.uint8, device=self.device)
            meta_host = torch.empty(64 + 4 * ns, dtype=torch.uint8, pin_memory=True)
            b.summary, b.samples = meta[:64].view(torch.int64), meta[64:].view(torch.int32)
            b.host, b.samples_host = meta_host[:64].view(torch.int64), meta_host[64:].view(torch.int32)
            ws = torch.empty(max(int(self.lib.rd_fasta_index_workspace_bytes(end, cap_lines)), 256), dtype=torch.uint8, device=self.device)
            N.check(self.lib.rd_fasta_index(N.ptr(text), int(start), int(end), N.ptr(prev[0]) if prev is not None else None,
                                            N.ptr(prev[1]) if prev is not None else None, (2 if self.keep_empty_tail else 1) if final else 0,
                                            N.ptr(b.line_end), cap_lines, N.ptr(b.norm), norm_cap, N.ptr(b.rec_tab), N.ptr(b.hdr_tab), cap_rec, N.ptr(b.summary), N.ptr(ws),
                                            ws.numel(), self._sp()), "rd_fasta_index")
            N.check(self.lib.rd_fasta_sample(N.ptr(b.rec_tab), N.ptr(b.summary), EVERY, N.ptr(b.samples), ns, self._sp()), "rd_fasta_sample")
            N.copy_bytes(meta_host, meta, meta.numel(), self.stream, workgroups=4)
            b.event = N.new_event()
            b.event.record(self.stream)
        b.final, b.orig, b.slot, b.gz_slot, b.n = final, None, None, None, None
        b.args, b.chain = (start, end, final), (text, b.summary)
        if chain:
            with self._lock:
                self.prev = b.chain
        self.stats["batches"] += 1
        return b

    @staticmethod
    def _read(b):
        FastqIndexer._read(b)
        b.norm_end = int(b.host.numpy()[7])

    def finish(self, b):
        self.wait(b)
        self._read(b)
        if b.status in (3, 4, 5) and b.chain is not None:    # (as FastqIndexer.finish: tables or pad too small, or the batch chains to such a one)
            self._reframe(b)
        if b.chain is not None:
            self.last_good = b.chain
            self.last_carry = (b.end - b.consumed) if b.status == 0 else 0
        return b

    def gather(self, pieces):
        n = sum(hi - lo for _, lo, hi in pieces)
        cap = sum(b.bytes_bound(lo, hi) for b, lo, hi in pieces)
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            text = torch.empty(((cap + 255) // 256) * 256 + 256, dtype=torch.uint8, device=self.device)
            rs = torch.empty(n + 1, dtype=torch.int64, device=self.device)
            so = torch.empty(n, dtype=torch.int64, device=self.device)
            sl = torch.empty(n, dtype=torch.int32, device=self.device)
            cursor = torch.zeros(len(pieces) + 1, dtype=torch.int64, device=self.device)
            at = 0
            for i, (b, lo, hi) in enumerate(pieces):
                N.check(self.lib.rd_fasta_gather(N.ptr(b.norm), N.ptr(b.rec_tab), N.ptr(b.hdr_tab), N.ptr(b.summary), lo, hi, b.bytes_bound(lo, hi),
                                                 N.ptr(text), text.numel(), C.c_void_p(cursor.data_ptr() + 8 * i), C.c_void_p(cursor.data_ptr() + 8 * (i + 1)),
                                                 C.c_void_p(rs.data_ptr() + 8 * at), C.c_void_p(so.data_ptr() + 8 * at),
                                                 C.c_void_p(sl.data_ptr() + 4 * at), self._sp()), "rd_fasta_gather")
                at += hi - lo
            total = torch.empty(1, dtype=torch.int64, pin_memory=True)
            N.copy_bytes(total.view(torch.uint8), cursor[len(pieces):].view(torch.uint8), 8, self.stream, workgroups=1)
            ready = torch.cuda.Event()
            ready.record(self.stream)
        return DeviceChunk(n, text, rs, so, sl, ready, total)


class DeviceFeeder:
    """The producer thread of one input file: file bytes -> batches on the GPU (H2D, members inflated where the file is BGZF, records
    framed), two batches in flight; next_batch() hands them over in order, finished."""

    BATCH = 48 << 20          # compressed bytes per BGZF batch (~200 MB of text)
    FIRST = 6 << 20           # the first batch (the first chunk is small too); doubling up to BATCH
    FIRST_DEFAULT = 6 << 20   # (a test that patches FIRST also sets the stream reader's first batch)
    STREAM_FIRST = None       # ... of a single-stream .gz when it is to differ from FIRST (tests); the product: a whole batch at once - a section's decode is
                              # a latency chain, a batch of any size takes >= 12 ms (profiles/r06_gzs_first_ab.txt: 6 -> 64 MiB +1-2 % paired, +2-5 % single-end)
    MAX_MEMBERS = 4096        # members per batch: at most 256 MiB of text whatever the members claim
    PLAIN_BATCH = 96 << 20    # bytes of a plain file per batch
    PLAIN_FIRST = 12 << 20
    SLOTS = 2

    def __init__(self, path, device, compressed, span=None, byte_range=None, fasta=False, keep_empty_tail=False):
        """compressed: the file is BGZF (span = (first file byte, one past the last, text bytes to drop in front, text bytes to
        deliver) for a rank's share, fastx_parser.BgzfView.file_span); else plain text (byte_range = (start, end), record-aligned)"""
        self.path, self.device, self.compressed, self.span, self.byte_range = str(path), torch.device(device), compressed, span, byte_range
        self.fasta, self.keep_empty_tail = bool(fasta), bool(keep_empty_tail)
        self._stop = False
        self.out = queue.Queue(maxsize=self.SLOTS)
        self.slot_free = queue.Queue()
        for k in range(self.SLOTS):
            self.slot_free.put(k)
        self.dg = self.ix = None
        self.stage_s = {"read": 0.0, "index": 0.0, "wait_slot": 0.0, "submit": 0.0, "batches": 0, "bytes": 0}
        self._trace = [] if os.environ.get("RD_FEED_TRACE") else None
        self._ready = threading.Event()
        self._init_err = None
        self.th = threading.Thread(target=self._run, name="rd-feed", daemon=True)
        self.th.start()
        self._ready.wait()
        if self._init_err is not None:
            raise self._init_err

    def stop(self):
        self._stop = True
        while True:                      # unblock a producer waiting for a slot or for room in the queue
            try:
                self.out.get_nowait()
            except queue.Empty:
                break
        self.slot_free.put(0)

    def join(self):
        self.th.join()

    def close(self):
        """stop, join, and give the stream back once it has drained"""
        self.stop()
        self.join()
        for name in ("stream", "dstream"):
            st = getattr(self, name, None)
            setattr(self, name, None)
            if st is not None:
                st.synchronize()
                gz.release_stream(st, priority=getattr(self, "_dprio", -1) if name == "dstream" else -1)

    # ---- consumer side --------------------------------------------------------------------------------
    def next_batch(self):
        """the next batch, finished (its record count is known) - None at the end of the stream; raises what the producer raised"""
        item = self.out.get()
        if item is None:
            return None
        if isinstance(item, BaseException):
            raise item
        b = item
        self.ix.wait(b)
        try:
            if b.gz_slot is not None:
                self.dg.finish(b.gz_slot)              # (over already: the summary came later on the same stream) - the members' status words
        finally:
            if b.slot is not None:
                self.slot_free.put(b.slot)
        return self.ix.finish(b)

    # ---- producer thread ---------------------------------------------------------------------------------
    def _put(self, item):
        while not self._stop:
            try:
                self.out.put(item, timeout=0.05)
                return True
            except queue.Full:
                pass
        return False

    def _run(self):
        self._t0 = time.perf_counter()
        self.stage_s["thread_started_at"] = self._t0            # (perf_counter: against Predictor._t_run, tools/first_chunk_probe.py)
        try:
            torch.cuda.set_device(self.device)          # the current device is per thread (and defaults to 0)
            self.stream = gz.acquire_stream(self.device, priority=-1)
            self.dg = gz.DeviceGunzip(self.device, slots=self.SLOTS, stream=self.stream)
            self.ix = (FastaIndexer if self.fasta else FastqIndexer)(self.device, self.stream)
            if self.fasta:
                self.ix.keep_empty_tail = self.keep_empty_tail
        except BaseException as e:      # noqa: BLE001
            self._init_err = e
            self._ready.set()
            return
        self._ready.set()
        try:
            if self.compressed == "stream":
                self._run_stream()
            elif self.compressed:
                self._run_bgzf()
            else:
                self._run_plain()
            if not self._stop:           # the end of the stream: an empty final batch judges what the last batch left over
                text = self.ix.alloc_text(0)
                self._put(self.ix.index(text, PAD, PAD, final=True))
            self._put(None)
        except BaseException as e:      # noqa: BLE001 - raised by next_batch() on the consumer's thread, behind the batches before it
            self._put(e)
        if self._trace:
            self.stream.synchronize()
            t00 = self._trace[0][0]
            self.stage_s["gpu_trace"] = [(round(1e3 * (t - t00), 1), nb >> 20, round(e0.elapsed_time(e1), 2), round(e1.elapsed_time(e2), 2))
                                         for t, nb, e0, e1, e2 in self._trace][:60]      # (submit time ms, MB, H2D ms, index ms)
        if os.environ.get("RD_FEED_TRACE"):
            import sys
            sys.stderr.write("device feeder %s: %s %s\n" % (self.path, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in self.stage_s.items()},
                                                            self.ix.stats))

    def _submit_text(self, src, nbytes, slot, start_skip=0, limit=None):
        """pinned host bytes -> a batch buffer -> index"""
        text = self.ix.alloc_text(nbytes)
        trace = self._trace
        with torch.cuda.device(self.device), torch.cuda.stream(self.dg.stream):
            if trace is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self.dg.stream)
            N.copy_bytes(text[PAD:PAD + nbytes], src, nbytes, self.dg.stream)      # (a kernel: an SDMA queue is shared with copies that wait)
            if trace is not None:
                e1.record(self.dg.stream)
        end = PAD + nbytes if limit is None else PAD + min(nbytes, start_skip + limit)
        b = self.ix.index(text, PAD + start_skip, end)
        if trace is not None:
            e2 = torch.cuda.Event(enable_timing=True)
            e2.record(self.dg.stream)
            trace.append((time.perf_counter(), nbytes, e0, e1, e2))
        b.slot = slot
        return b

    def _run_plain(self):
        tm = self.stage_s
        pinned = [torch.empty(self.PLAIN_BATCH, dtype=torch.uint8, pin_memory=True) for _ in range(self.SLOTS)]
        views = [t.numpy() for t in pinned]
        batch = min(self.PLAIN_FIRST, self.PLAIN_BATCH)
        with open(self.path, "rb", buffering=0) as fh:
            left = None
            if self.byte_range is not None:
                fh.seek(int(self.byte_range[0]))
                left = int(self.byte_range[1]) - int(self.byte_range[0])
            while not self._stop and (left is None or left > 0):
                t0 = time.perf_counter()
                slot = self.slot_free.get()
                if self._stop:
                    break
                t1 = time.perf_counter()
                want, have = (batch if left is None else min(batch, left)), 0
                while have < want:
                    k = fh.readinto(memoryview(views[slot])[have:want])
                    if not k:
                        break
                    have += k
                t2 = time.perf_counter()
                if have == 0:
                    self.slot_free.put(slot)
                    break
                if left is not None:
                    left -= have
                b = self._submit_text(pinned[slot], have, slot)
                tm["wait_slot"] += t1 - t0
                tm["read"] += t2 - t1
                tm["submit"] += time.perf_counter() - t2
                tm["batches"] += 1
                tm["bytes"] += have
                if not self._put(b):
                    break
                if have < want:
                    break
                batch = min(2 * batch, self.PLAIN_BATCH)

    def _run_stream(self):
        """The gzip members of a file that are single DEFLATE streams, decoded on the device one after the other (gz.DeviceStreamGunzip,
        fresh state per member: `cat L001.fq.gz L002.fq.gz` is what lane-merged data looks like): batches of compressed bytes, two in
        flight; a batch's text length is known only when its 64-byte state has arrived, so its records are framed one batch behind the
        submission. A file whose FIRST batch the decoder gives back goes to the host reader as a whole (_StreamFallback reaches
        get_seq_chunks_device: nothing has been delivered yet); a later member it cannot take, to zlib on this thread."""
        tm = self.stage_s
        span = gz.DeviceStreamGunzip.BATCH + gz.DeviceStreamGunzip.SLACK + 8192
        if os.environ.get("RD_GZS_BATCH"):
            span = int(os.environ["RD_GZS_BATCH"]) + gz.DeviceStreamGunzip.SLACK + 8192
        pinned = [torch.empty(span, dtype=torch.uint8, pin_memory=True) for _ in range(self.SLOTS + 1)]
        views = [t.numpy() for t in pinned]
        fd = os.open(self.path, os.O_RDONLY)
        try:
            size, base = os.fstat(fd).st_size, 0
            tm["members"] = 0
            # the decoder's kernels on a stream of their own: a batch's decode is one long latency chain (>= 12 ms), and the framing of the batch
            # BEFORE it and the gathers of the consumer's chunks - microseconds of work - were queued behind it on the feeder's one stream:
            # the first chunk of a file left the reader when its LAST batch had been decoded (tools/first_chunk_probe.py, round 6)
            if getattr(self, "dstream", None) is None:
                self._dprio = int(os.environ.get("RD_GZS_PRIORITY", "-1"))      # (A/B: 0 = the decoder at the recurrence's priority, framing above it)
                self.dstream = gz.acquire_stream(self.device, priority=self._dprio)
            while not self._stop:
                dsg = gz.DeviceStreamGunzip(self.device, self.dstream)
                free = list(range(len(pinned)))
                try:
                    end = self._stream_batches(fd, dsg, pinned, views, free, tm, base)
                except _StreamFallback as e:
                    if base == 0:
                        raise
                    tm["fallback"] = "member at byte %d: %s" % (base, e)
                    self.dstream.synchronize()
                    self.stream.synchronize()
                    with open(self.path, "rb", buffering=0) as fh:
                        fh.seek(base)
                        self._host_tail(fh, b"")
                    return
                tm["members"] += 1
                if end is None or end >= size:
                    return
                # zero padding behind a member is skipped (Python's gzip module does); what follows must be another member
                while end < size:
                    rest = os.pread(fd, 1 << 16, end)
                    k = len(rest) - len(rest.lstrip(b"\0"))
                    end += k
                    if k < len(rest):
                        break
                if end >= size:
                    return
                base = end
        finally:
            os.close(fd)

    def _stream_batches(self, fd, dsg, pinned, views, free, tm, base=0):
        """one member that starts at file offset `base`; returns the offset behind its trailer (None: stopped)"""
        size = os.fstat(fd).st_size
        hl = gz.gzip_header_len(os.pread(fd, 1 << 16, base))
        if hl is None:
            raise ValueError("Compressed file ended before the end-of-stream marker was reached")
        pos, first, batch = base, hl * 8, min(int(os.environ.get("RD_GZS_FIRST", self.STREAM_FIRST or (self.FIRST if self.FIRST != DeviceFeeder.FIRST_DEFAULT else dsg.BATCH))) if base == 0 else dsg.BATCH, dsg.BATCH)
        flight = deque()                                  # (ticket, text, pinned slot, file offset of the batch)
        member_end = None
        good = {}                                         # what the last good batch left: where the stream goes on, window, CRC, length

        def finish_one():
            nonlocal member_end
            tk, text, slot, at = flight.popleft()
            t0 = time.perf_counter()
            r = dsg.finish(tk)
            tm["wait_slot"] += time.perf_counter() - t0
            free.append(slot)
            if member_end is not None:
                return True                               # (a batch submitted behind the member's last one: nothing of it is used)
            if r["status"]:
                if at == base:
                    # the member's FIRST batch: of the file's first member nothing has been delivered yet, so the whole file can still go to the host reader -
                    # what is not text (no block start passes the search: one wave would have to decode everything), what
                    # compresses 100:1, and damaged files, whose error messages are then zlib's
                    raise _StreamFallback(gz.GZS_ERRORS.get(r["status"], "error %d" % r["status"]))
                # a later batch: the text before it has been delivered, so the REST of the member is decoded by zlib on the host from the
                # bit this batch was to start at, with the window the batch before left as its dictionary (slow - one core - but the
                # same text, and zlib's messages for real damage); the batches in flight behind it are dropped
                tm["resumed_on_host"] = "%s at byte %d" % (gz.GZS_ERRORS.get(r["status"], "error %d" % r["status"]), at)
                while flight:
                    dsg.finish(flight.popleft()[0])
                member_end = self._resume_on_host(fd, size, good["abs_next"], good["win"], good["win_valid"], good["crc"], good["total_len"])
                return not self._stop
            if r["final"]:
                end = at + (r["end_bit"] + 7) // 8
                tr = os.pread(fd, 8, end)
                if len(tr) < 8:
                    raise ValueError("Compressed file ended before the end-of-stream marker was reached")
                if int.from_bytes(tr[:4], "little") != r["crc"]:
                    raise ValueError("CRC check failed")
                if int.from_bytes(tr[4:], "little") != (r["total_len"] & 0xffffffff):
                    raise ValueError("Incorrect length of data produced")
                member_end = end + 8
            good.update(abs_next=at * 8 + r["next_start"], win=tk["keep"][2], win_valid=r["win_valid"], crc=r["crc"], total_len=r["total_len"])
            b = self.ix.index(text, PAD, PAD + r["n_text"])
            tm.setdefault("first_batch_framed_at_s", round(time.perf_counter() - self._t0, 4))      # (since the feeder thread started)
            if len(tm.setdefault("batches_framed_at_s", [])) < 8:
                tm["batches_framed_at_s"].append((round(time.perf_counter() - self._t0, 4), int(r["n_text"])))
            tm["batches"] += 1
            tm["bytes"] += r["n_text"]
            return self._put(b)

        while not self._stop and member_end is None:
            data = min(batch, size - pos)
            if data <= 0:
                break
            valid = min(data + dsg.SLACK, size - pos)
            at_eof = pos + valid >= size
            t0 = time.perf_counter()
            while len(flight) >= self.SLOTS:
                if not finish_one():
                    return
            if member_end is not None:
                break
            slot = free.pop()
            t1 = time.perf_counter()
            have = 0
            while have < valid:
                k = os.preadv(fd, [memoryview(views[slot])[have:valid]], pos + have)
                if k <= 0:
                    raise ValueError("Compressed file ended before the end-of-stream marker was reached")
                have += k
            t2 = time.perf_counter()
            text = self.ix.alloc_text(dsg.text_cap(data))
            if dsg.stream is not self.ix.stream:          # (the buffer comes from the indexer's stream's pool: whatever last used it there is over first)
                ev = torch.cuda.Event()
                ev.record(self.ix.stream)
                dsg.stream.wait_event(ev)
            tk = dsg.submit(pinned[slot], valid, data, first, at_eof, text[PAD:])
            tm.setdefault("first_batch_submitted_at_s", round(time.perf_counter() - self._t0, 4))
            flight.append((tk, text, slot, pos))
            tm["read"] += t2 - t1
            tm["submit"] += time.perf_counter() - t2
            pos += data
            first = 0xffffffff                              # (from the second batch on: where the batch before said)
            batch = min(2 * batch, dsg.BATCH)
            if at_eof and pos >= size:
                break
        while flight:
            if not finish_one():
                return None
        if member_end is None:
            if not self._stop:
                raise ValueError("Compressed file ended before the end-of-stream marker was reached")
            return None
        return member_end

    def _resume_on_host(self, fd, size, start_bit, win_dev, win_valid, crc, total_len):
        """the rest of a gzip member from absolute bit `start_bit` of the file, by zlib: the compressed bytes are shifted to a byte
        boundary piece by piece (DEFLATE packs bits LSB first), the 32 KiB window the device decoder left is the preset dictionary.
        Returns the file offset behind the member's trailer (CRC-32 and ISIZE are checked against the carried values + this text)."""
        import zlib
        window = win_dev.cpu().numpy().tobytes()[32768 - min(int(win_valid), 32768):]
        d = zlib.decompressobj(-15, zdict=window) if window else zlib.decompressobj(-15)
        k, pos = start_bit & 7, start_bit >> 3
        fed = 0
        PIECE = 4 << 20
        while not d.eof:
            raw = os.pread(fd, PIECE + 1, pos)
            if len(raw) == 0 or self._stop:
                if self._stop:
                    return None
                raise ValueError("Compressed file ended before the end-of-stream marker was reached")
            arr = np.frombuffer(raw, dtype=np.uint8)
            n = len(arr) - 1 if len(arr) == PIECE + 1 else len(arr)
            if k:
                nxt = np.concatenate([arr[1:], np.zeros(1, np.uint8)])[:n] if len(arr) != PIECE + 1 else arr[1:n + 1]
                piece = ((arr[:n] >> k) | ((nxt.astype(np.uint16) << (8 - k)) & 0xff).astype(np.uint8)).tobytes()
            else:
                piece = arr[:n].tobytes()
            pos += n
            data = piece
            while data and not d.eof:
                try:
                    out = d.decompress(data, 16 << 20)
                except zlib.error as e:
                    raise ValueError(str(e))
                # (at the end of the stream CPython puts the bytes behind it into unused_data AND leaves them in unconsumed_tail)
                fed += len(data) - (len(d.unused_data) if d.eof else len(d.unconsumed_tail))
                data = d.unconsumed_tail
                if out:
                    crc = zlib.crc32(out, crc)
                    total_len += len(out)
                    src = torch.from_numpy(np.frombuffer(out, dtype=np.uint8).copy()).pin_memory()
                    b = self._submit_text(src, len(out), None)
                    b.orig = src
                    if not self._put(b):
                        return None
        # the member's trailer starts at the byte boundary behind its last bit: zlib says how many (shifted) bytes it consumed, the last
        # of them partly - two candidates unless the stream was byte aligned; CRC-32 and ISIZE say which
        base = (start_bit >> 3) + fed
        for end in ((base,) if k == 0 else (base, base + 1)):
            tr = os.pread(fd, 8, end)
            if len(tr) == 8 and int.from_bytes(tr[:4], "little") == (crc & 0xffffffff) and int.from_bytes(tr[4:], "little") == (total_len & 0xffffffff):
                return end + 8
        if os.environ.get("RD_FEED_TRACE"):
            import sys
            sys.stderr.write("resume: start_bit %d k %d fed %d crc %08x len %d; trailers %s\n" % (
                start_bit, k, fed, crc & 0xffffffff, total_len, [os.pread(fd, 8, e).hex() for e in (base - 1, base, base + 1)]))
        raise ValueError("CRC check failed")

    def _host_tail(self, fh, data):
        """the rest of a file whose members stop carrying their size (`cat a.bgzf.gz b.gz` is a legal .gz): zlib, member after member,
        the text shipped to the device in pieces and framed behind the batches in flight"""
        import zlib
        d, inside = zlib.decompressobj(31), False
        while not self._stop:
            if not data:
                data = fh.read(4 << 20)
                if not data:
                    if inside:
                        raise ValueError("Compressed file ended before the end-of-stream marker was reached")
                    return
            if not inside and not data.strip(b"\0"):       # zero padding behind the last member (Python's gzip module skips it too)
                data = b""
                continue
            try:
                out = d.decompress(data, 16 << 20)          # (at most 16 MB of text per call)
            except zlib.error as e:
                raise ValueError(str(e))
            inside = True
            if out:
                src = torch.from_numpy(np.frombuffer(out, dtype=np.uint8).copy()).pin_memory()
                b = self._submit_text(src, len(out), None)
                b.orig = src                                  # (keeps the pinned bytes alive until the batch is consumed)
                if not self._put(b):
                    return
            if d.eof:
                data, d, inside = d.unused_data, zlib.decompressobj(31), False
            else:
                data = d.unconsumed_tail

    def _run_bgzf(self):
        tm = self.stage_s
        dg = self.dg
        pinned = [torch.empty(self.BATCH + (1 << 20), dtype=torch.uint8, pin_memory=True) for _ in range(self.SLOTS)]
        bufs = [t.numpy() for t in pinned]
        carry = None                                    # bytes of an incomplete member, to go in front of the next batch
        batch = min(self.FIRST, self.BATCH)
        with open(self.path, "rb", buffering=0) as fh:
            eof = False
            file_left = text_left = None
            skip_text = 0
            if self.span is not None:           # a share of the file: whole members [c0, c1), text trimmed at both ends
                fh.seek(self.span[0])
                file_left, skip_text, text_left = self.span[1] - self.span[0], self.span[2], self.span[3]
                eof = file_left <= 0
            while not self._stop:
                t0 = time.perf_counter()
                slot = self.slot_free.get()                  # (its previous batch has left the GPU: next_batch() finished it)
                if self._stop:
                    break
                t1 = time.perf_counter()
                buf, have = bufs[slot], 0
                if carry is not None:
                    have = len(carry)
                    buf[:have] = carry
                    carry = None
                while have < batch and not eof:
                    cap = batch + (1 << 20) if file_left is None else min(batch + (1 << 20), have + file_left)
                    k = fh.readinto(memoryview(buf)[have:cap])
                    if not k:
                        eof = True
                    else:
                        have += k
                        if file_left is not None:
                            file_left -= k
                            eof = file_left <= 0
                if have == 0:
                    self.slot_free.put(slot)
                    break
                t2 = time.perf_counter()
                n, consumed, out_bytes, streaming = dg.index(buf, have, slot=slot, max_members=self.MAX_MEMBERS)
                t3 = time.perf_counter()
                tm["wait_slot"] += t1 - t0
                tm["read"] += t2 - t1
                tm["index"] += t3 - t2
                if streaming and n == 0:
                    self.slot_free.put(slot)
                    self._host_tail(fh, bytes(buf[consumed:have]))
                    break
                if n == 0 or consumed == 0:
                    if eof:
                        if consumed < have:
                            raise ValueError("Compressed file ended before the end-of-stream marker was reached")
                        self.slot_free.put(slot)
                        break
                    if consumed == 0 and have >= self.BATCH:       # a member that claims to be larger than the batch buffer: no progress possible
                        raise ValueError("gzip member larger than %d bytes: not a BGZF file (RD_DEVICE_INFLATE=0 reads it with the host's "
                                         "decoders)" % self.BATCH)
                if consumed < have:
                    carry = buf[consumed:have].copy()
                if n:
                    text = self.ix.alloc_text(out_bytes)
                    dg.submit(buf, consumed, n, out_bytes, slot=slot, text_out=text[PAD:PAD + out_bytes])
                    drop = min(skip_text, out_bytes)
                    skip_text -= drop
                    end = PAD + out_bytes
                    if text_left is not None:
                        take = min(out_bytes - drop, text_left)
                        text_left -= take
                        end = PAD + drop + take
                    b = self.ix.index(text, PAD + drop, end)
                    b.slot, b.gz_slot = slot, slot
                    tm["submit"] += time.perf_counter() - t3
                    tm["batches"] += 1
                    tm["bytes"] += out_bytes
                    if not self._put(b):
                        break
                else:
                    self.slot_free.put(slot)
                batch = min(2 * batch, self.BATCH)
                if text_left is not None and text_left <= 0:
                    break


def get_seq_chunks_device(seq_file, chunk_size=1048576, byte_range=None, first_chunk=None, schedule=None, device=None, stats=None):
    """fastx_parser.get_seq_chunks for FASTQ whose text stays on the device: DeviceChunk objects of exactly the scheduled number of
    records (fewer only at the end of the stream). byte_range: (start, end) of a plain file or a fastx_parser.BgzfRange. A damaged
    stream delivers the chunks before the damage, then raises ValueError like the host reader."""
    from . import fastx_parser as fx
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    torch.cuda.set_device(device)
    kind = device_ingest_kind(seq_file)
    compressed = "stream" if kind == "stream" else fx.get_seq_format(seq_file).endswith("gz")
    span = None
    keep_tail = False            # a share that ends before the file does (FASTA: its last record counts even without a sequence)
    from . import gz_shard
    resident = byte_range if isinstance(byte_range, gz_shard.ResidentRange) else None
    if resident is not None:     # a rank's share of a single-stream .gz, decoded and framed already (gz_shard.prepare)
        byte_range = None
    if isinstance(byte_range, fx.BgzfRange):
        a, b = byte_range
        c0, c1, drop = byte_range.view.file_span(a, b)
        keep_tail = b < byte_range.view.size
        span, byte_range = (c0, c1, drop, b - a), None
    elif byte_range is not None:
        keep_tail = int(byte_range[1]) < os.path.getsize(seq_file)
    if resident is not None:
        feeder = gz_shard.ResidentFeeder(resident)
    else:
        feeder = DeviceFeeder(seq_file, device, compressed, span=span, byte_range=byte_range, fasta=fx.get_seq_format(seq_file).startswith("fa"),
                              keep_empty_tail=keep_tail)
    skip_left = resident.skip if resident is not None else 0      # (mate files: the records in front of the ranks' common cut went to the rank before)
    want = chunk_size if not first_chunk else max(1, min(int(first_chunk), chunk_size))
    sched = list(schedule) if schedule else None
    pend = deque()                   # [batch, next record]
    avail, eof, err, framing = 0, False, None, False
    try:
        while True:
            if sched is not None:
                want = sched.pop(0) if len(sched) > 1 else sched[0]
            while avail < want and not eof:
                try:
                    b = feeder.next_batch()
                except ValueError as e:
                    err, eof = e, True
                    break
                except _StreamFallback as e:
                    # the device decoder gave the file back before it delivered anything (not text, 100:1 text, fixed-Huffman or stored
                    # blocks at its start, damage): the host reader - parallel decoders, zlib's messages - reads it, and its chunks are
                    # shipped to the device so that the run still sees one kind of chunk
                    import logging
                    logging.getLogger("predict").info("%s: %s - read by the host decoders" % (os.path.basename(str(seq_file)), e))
                    feeder.close()
                    if stats is not None:
                        stats.update({"path": "host", "fallback": str(e)})
                    stats = None
                    st = gz.acquire_stream(device)
                    try:
                        for c in fx.get_seq_chunks(seq_file, chunk_size=chunk_size, first_chunk=first_chunk, schedule=schedule, device=device):
                            yield DeviceChunk.from_host(c, device, st)
                    finally:
                        st.synchronize()
                        gz.release_stream(st)
                    return
                if b is None:
                    eof = True
                    break
                good = b.n
                if b.status:             # a malformed record: the chunks in front of it are delivered, then the error
                    err, eof, framing = ValueError(FQ_ERRORS.get(b.status, "FASTQ framing error %d" % b.status)), True, True
                    good = min(b.n, b.bad_record) if b.status == 1 and b.bad_record >= 0 else (b.n if b.status == 2 else 0)
                drop = min(skip_left, good)
                skip_left -= drop
                if good > drop:
                    pend.append([b, drop, good])
                    avail += good - drop
            if avail == 0 or (framing and avail < want):
                # (the host reader meets a malformed record while it fills a chunk and fails that call: the records it had gathered for
                # THAT chunk are not delivered, csrc/rd_host.cpp rd_reader_next - the same chunks come out of both readers)
                break
            take, pieces = min(want, avail), []
            left = take
            while left > 0:
                b, lo, hi = pend[0]
                k = min(left, hi - lo)
                pieces.append((b, lo, lo + k))
                left -= k
                if lo + k == hi:
                    pend.popleft()
                else:
                    pend[0][1] = lo + k
            avail -= take
            yield feeder.ix.gather(pieces)
            if sched is None:
                want = min(chunk_size, want * 2)
        if err is not None:
            raise err
    finally:
        feeder.close()
        if stats is not None:
            stats.update({"feeder": dict(feeder.stage_s), "indexer": dict(feeder.ix.stats) if feeder.ix else None})
