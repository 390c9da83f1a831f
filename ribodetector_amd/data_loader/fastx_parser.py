"""FASTQ / FASTA ingest - mirror of the reference's `data_loader/fastx_parser.py:15-55` (record semantics) and of the
chunked readers `seq_encoder.py:21-39,75-92`, re-designed so that a chunk is ONE byte arena plus offset arrays
(what the C ABI consumes) instead of a Python list of 4-tuples of str.

Record semantics kept from the reference:
  * FASTQ: 4-line state machine, every line `rstrip()`-ed, header / '+' line / quality preserved verbatim,
    bases NOT upper-cased (fastx_parser.py:18-37);
  * FASTA: multi-line sequences joined and upper-cased, blank lines skipped (fastx_parser.py:39-55);
  * gzip chosen by file extension, format by the extension before it (seq_encoder.py:21-39).
"""
import ctypes as C
import gzip
import io
import os
from collections import namedtuple
from mimetypes import guess_type
from pathlib import Path

import numpy as np

from .. import _native as N

FA_EXTS = [".fasta", ".fa", ".fna", ".fas"]
FQ_EXTS = [".fq", ".fastq"]


def get_seq_format(seq_file):
    """'fa' | 'fq' (+ 'gz') from the file name (reference seq_encoder.py:21-39)."""
    encoding = guess_type(str(seq_file))[1]
    if encoding is None:
        encoding = ""
    elif encoding == "gzip":
        encoding = "gz"
    else:
        raise ValueError('Unknown file encoding: "{}"'.format(encoding))
    name = Path(seq_file).stem if encoding == "gz" else Path(seq_file).name
    ext = Path(name).suffix
    if ext not in FA_EXTS + FQ_EXTS:
        raise ValueError('Unknown extension {}. Only fastq and fasta sequence formats are supported.\n'
                         'And the file must end with one of ".fasta", ".fa", ".fna", ".fas", ".fq", ".fastq"\n'
                         'and followed by ".gz" or ".gzip" if they are gzipped.'.format(ext))
    return ("fa" if ext in FA_EXTS else "fq") + encoding


def seq_parser(seq_fh, seq_type):
    """Record generator with the reference's semantics: FASTQ -> (header, seq, plus, qual), FASTA -> (header, seq)."""
    if seq_type == "fastq":
        state, rec = 0, []
        for line in seq_fh:
            line = line.rstrip()
            if state == 0:
                if line[:1] != "@":
                    # the reference would mis-frame silently here; malformed input is an error in this build
                    raise ValueError("FASTQ record does not start with '@': %r" % line[:50])
                rec = [line]
                state = 1
            else:
                rec.append(line)
                state += 1
                if state == 4:
                    yield tuple(rec)
                    state = 0
    else:
        header, parts = "", []
        for line in seq_fh:
            line = line.strip()
            if line == "":
                continue
            if line[0] == ">":
                if header != "":
                    yield header, "".join(parts)
                header, parts = line, []
            else:
                parts.append(line.upper())
        if parts:
            yield header, "".join(parts)


# A chunk of records as arrays.  buf: uint8 arena holding the record text; rec_start int64[n+1]: byte range of record i
# (verbatim text incl. its final newline, valid when `verbatim`); seq_off int64[n] / seq_len int32[n]: the bases.
# tensors: (buf, seq_off, seq_len, rec_start) as (pinned) torch tensors when the chunk came from the native reader, else None.
# release: called by the last consumer of a chunk whose buffers are a slot of a ShmArena; shm: where such a chunk lives.
Chunk = namedtuple("Chunk", "buf rec_start seq_off seq_len verbatim records tensors release shm", defaults=(None, None, None))

_WS = np.zeros(256, dtype=bool)
_WS[[9, 10, 11, 12, 13, 32]] = True


def _open_binary(path):
    fmt = get_seq_format(path)
    return (gzip.open(path, "rb") if fmt.endswith("gz") else open(path, "rb")), fmt


def _fastq_chunks(fh, chunk_reads, block_bytes=64 << 20):
    """Vectorised FASTQ framing: newline scan with numpy, no per-read Python."""
    carry = b""
    eof = False
    while not eof or carry:
        want_lines = 4 * chunk_reads
        data = carry
        nl = None
        while True:
            arr = np.frombuffer(data, dtype=np.uint8)
            nl = np.flatnonzero(arr == 10)
            if len(nl) >= want_lines or eof:
                break
            more = fh.read(block_bytes)
            if not more:
                eof = True
                if data and data[-1:] != b"\n":
                    data += b"\n"            # last line without newline: the reference's rstrip makes no difference
                continue
            data += more
        if len(nl) == 0:
            if data.strip():
                raise ValueError("truncated FASTQ record at end of file")
            return
        nlines = min(len(nl) - len(nl) % 4, want_lines) if not eof else len(nl)
        if eof and nlines > want_lines:
            nlines = want_lines
        if nlines % 4:
            raise ValueError("FASTQ: number of lines is not a multiple of 4 (truncated record)")
        if nlines == 0:
            carry = data
            if eof:
                return
            continue
        end_byte = int(nl[nlines - 1]) + 1
        arr = np.frombuffer(data, dtype=np.uint8)[:end_byte]
        carry = data[end_byte:]
        line_end = nl[:nlines]                                   # position of '\n'
        line_start = np.empty(nlines, dtype=np.int64)
        line_start[0] = 0
        line_start[1:] = line_end[:-1] + 1
        if not (arr[line_start[0::4]] == ord("@")).all():
            raise ValueError("FASTQ record does not start with '@'")
        # rstrip(): trailing whitespace of every line
        stripped_end = line_end.copy()
        verbatim = True
        while True:
            has = stripped_end > line_start
            prev = arr[np.maximum(stripped_end - 1, 0)]
            m = has & _WS[prev]
            if not m.any():
                break
            verbatim = False
            stripped_end[m] -= 1
        seq_off = line_start[1::4].copy()
        seq_len = (stripped_end[1::4] - seq_off).astype(np.int32)
        rec_start = np.empty(nlines // 4 + 1, dtype=np.int64)
        rec_start[:-1] = line_start[0::4]
        rec_start[-1] = end_byte
        records = None
        if not verbatim:                                         # rare: keep the stripped lines for the writer
            b = arr.tobytes()
            records = ["\n".join(b[line_start[4 * i + k]:stripped_end[4 * i + k]].decode("latin-1") for k in range(4))
                       for i in range(nlines // 4)]
        yield Chunk(arr, rec_start, seq_off, seq_len, verbatim, records)


def _fasta_chunks(fh, chunk_reads):
    text = io.TextIOWrapper(fh, encoding="latin-1")
    it = seq_parser(text, "fasta")
    while True:
        recs = []
        for r in it:
            recs.append(r)
            if len(recs) == chunk_reads:
                break
        if not recs:
            return
        out = [h + "\n" + s for h, s in recs]
        blob = ("\n".join(out) + "\n").encode("latin-1")
        arr = np.frombuffer(blob, dtype=np.uint8)
        lens_rec = np.array([len(o) + 1 for o in out], dtype=np.int64)
        rec_start = np.zeros(len(out) + 1, dtype=np.int64)
        np.cumsum(lens_rec, out=rec_start[1:])
        hl = np.array([len(h) + 1 for h, _ in recs], dtype=np.int64)
        seq_len = np.array([len(s) for _, s in recs], dtype=np.int32)
        yield Chunk(arr, rec_start, rec_start[:-1] + hl, seq_len, True, None)


def get_seq_chunks_numpy(seq_file, chunk_size=1048576):
    """numpy implementation of the chunk reader (kept as an independent cross-check of the native one in tests)."""
    fh, fmt = _open_binary(seq_file)
    with fh:
        if fmt.startswith("fq"):
            yield from _fastq_chunks(fh, chunk_size)
        else:
            yield from _fasta_chunks(fh, chunk_size)


def _shm_layout(cap_n, cap_b):
    """byte offsets of (rec_start int64[cap_n+1], seq_off int64[cap_n], seq_len int32[cap_n], buf uint8[cap_b]) in a slot file"""
    al = lambda x: (x + 4095) // 4096 * 4096    # noqa: E731
    o_rs = 0
    o_so = al(o_rs + 8 * (cap_n + 1))
    o_sl = al(o_so + 8 * cap_n)
    o_buf = al(o_sl + 4 * cap_n)
    return o_rs, o_so, o_sl, o_buf, o_buf + cap_b


def _shm_views(path, cap_n, cap_b, create):
    import torch
    o_rs, o_so, o_sl, o_buf, total = _shm_layout(cap_n, cap_b)
    if create:
        # reserve the pages NOW: a sparse file in a full /dev/shm (Docker's default is 64 MB) would kill the process with SIGBUS at
        # the first store past the capacity - no Python exception, no message to the other ranks. posix_fallocate fails with
        # ENOSPC instead, which travels the parser-error path like any other exception.
        import os
        fd = os.open(path, os.O_CREAT | os.O_RDWR | os.O_TRUNC, 0o600)
        try:
            os.posix_fallocate(fd, 0, total)
        except OSError as e:
            os.close(fd)
            try:
                os.remove(path)
            except OSError:
                pass
            raise OSError(e.errno, "shared-memory chunk slot %s (%d bytes): %s - set RD_SHARED_DECODE=0 (every rank decodes for itself) "
                                   "or enlarge /dev/shm" % (path, total, os.strerror(e.errno)))
        os.close(fd)
    mm = np.memmap(path, dtype=np.uint8, mode="r+", shape=(total,))
    t = lambda a: torch.from_numpy(a)           # noqa: E731
    return (t(mm[o_buf:o_buf + cap_b]), t(mm[o_rs:o_rs + 8 * (cap_n + 1)].view(np.int64)), t(mm[o_so:o_so + 8 * cap_n].view(np.int64)),
            t(mm[o_sl:o_sl + 4 * cap_n].view(np.int32)))


class ShmArena:
    """Chunk buffers in /dev/shm, for the multi-rank CLI on gzip input (one DEFLATE stream: not splittable): the node's rank 0
    inflates and parses the stream ONCE, straight into a slot of this arena; the other ranks map the slot and take the bases of
    their share of the records (round 2: every rank inflated and parsed the whole stream, W times the work on the same cores).
    A slot is one file (offset arrays + record text) and is reused once its chunk has been written out, so the pages are faulted
    in once. describe()/attach() are the two ends of the per-chunk message."""

    def __init__(self, tag):
        import os
        import threading
        self.dir = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
        self.tag, self.slots, self.lock = tag, [], threading.Lock()

    def alloc(self, nbytes, nrec):
        import os
        with self.lock:
            for sl in self.slots:
                if not sl["busy"] and sl["cap_b"] >= nbytes and sl["cap_n"] >= nrec:
                    sl["busy"] = True
                    return sl["views"], sl
            cap_n, cap_b = int(nrec * 1.05) + 16, int(nbytes * 1.15) + 4096
            path = os.path.join(self.dir, "%s.%d" % (self.tag, len(self.slots)))
            sl = {"path": path, "cap_n": cap_n, "cap_b": cap_b, "busy": True, "views": _shm_views(path, cap_n, cap_b, True)}
            self.slots.append(sl)
            return sl["views"], sl

    def free(self, sl):
        with self.lock:
            sl["busy"] = False

    def close(self):
        import os
        for sl in self.slots:
            try:
                os.remove(sl["path"])
            except OSError:
                pass
        self.slots = []

    SLOTS_PER_FILE = 6      # chunks in flight per input file: reader queue + GPU pipeline + writer queue

    @classmethod
    def fits(cls, nfiles, chunk_records, record_bytes):
        """enough free space under the arena's directory for the slots a run will hold (with a margin)? Checked before the shared
        decode is switched on; the slots themselves are fallocate'd, so a wrong estimate is an exception, not a SIGBUS."""
        import os
        import shutil
        d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
        need = int(1.3 * cls.SLOTS_PER_FILE * nfiles * chunk_records * (record_bytes * 1.15 + 20))
        try:
            return shutil.disk_usage(d).free >= need, need
        except OSError:
            return False, need

    @staticmethod
    def sweep_stale(prefix="rd_"):
        """remove slot files whose creating process is gone (tag = rd_<port>_<pid>_f<i>.<slot>): a run that was killed (SIGKILL, a
        node reset) cannot clean up after itself, and the files are RAM"""
        import os
        import re
        d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
        n = 0
        for name in os.listdir(d):
            m = re.match(r"rd_\d+_(\d+)_f\d+\.\d+$", name)
            if m and not os.path.exists("/proc/%s" % m.group(1)):
                try:
                    os.remove(os.path.join(d, name))
                    n += 1
                except OSError:
                    pass
        return n

    _attached = {}

    @classmethod
    def attach(cls, shm):
        """the Chunk another rank described (Chunk.shm), mapped read-write-shared; mappings are cached per slot file"""
        path, cap_n, cap_b, n, nb = shm
        key = (path, cap_n, cap_b)
        if key not in cls._attached:
            cls._attached[key] = _shm_views(path, cap_n, cap_b, False)
        buf, rs, so, sl = cls._attached[key]
        return Chunk(buf[:nb].numpy(), rs[:n + 1].numpy(), so[:n].numpy(), sl[:n].numpy(), True, None, (buf[:nb], so[:n], sl[:n], rs[:n + 1]))


def ingest_env(name, default):
    """one of the three older ingest switches - RD_DEVICE_PARSE, RD_DEVICE_FASTA, RD_DEVICE_INFLATE - as set, or as RD_INGEST implies:
        RD_INGEST=device   (default) text stays on the GPU: plain, BGZF and single-stream .gz, FASTQ and FASTA
        RD_INGEST=members  records framed on the GPU, BGZF members inflated there, a single-stream .gz by the host's decoders (round 4)
        RD_INGEST=host     the host reader for everything (parallel decoders, pinned chunks, H2D per chunk: rounds 1-3)
    The old names keep working and win where both are set."""
    v = os.environ.get(name)
    if v is not None:
        return v
    mode = os.environ.get("RD_INGEST", "device").lower()
    if mode == "host":
        return {"RD_DEVICE_PARSE": "0", "RD_DEVICE_FASTA": "0", "RD_DEVICE_INFLATE": "0"}.get(name, default)
    if mode == "members":
        return {"RD_DEVICE_INFLATE": "members"}.get(name, default)
    if mode != "device":
        raise RuntimeError("RD_INGEST must be device, members or host; got %r" % mode)
    return default


def device_inflate_wanted(path):
    """a GPU, and a .gz that starts with BGZF blocks (bgzip / htslib output, and every .gz the CLI writes with its device deflate):
    its members are inflated on the device. RD_DEVICE_INFLATE=0 keeps the host's decoders; =1 also takes files of this build's host
    writer (4 MiB members: a wave per member is a long time per member - the host's parallel member decoder suits them better)."""
    import os
    mode = ingest_env("RD_DEVICE_INFLATE", "auto")
    if mode == "0":
        return False
    import torch
    if not torch.cuda.is_available():
        return False
    from .. import gz
    kind = gz.is_member_indexed(path)
    return kind == "BC" or (kind == "RD" and mode == "1")


def bgzf_all_the_way(path):
    """does the file consist of BGZF blocks up to its last byte (the last one being BGZF's empty end-of-file block or any complete
    block)? Checked at both ends - a plain gzip member in the middle is found by the index walk of the sharded reader, which then
    refuses loudly"""
    from .. import gz
    if gz.is_member_indexed(path) != "BC":
        return False
    try:
        size = os.path.getsize(path)
        with open(path, "rb") as fh:
            fh.seek(max(0, size - 28))
            tail = fh.read(28)
    except OSError:
        return False
    return tail == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


class _DeviceInflateFeeder:
    """threads: compressed file -> batches of whole members -> GPU (one wave per member) -> pinned host text -> rd_reader_feed.
    Members without a size subfield (a plain gzip member concatenated behind BGZF blocks) cannot be handed to the device: that is an
    error of this path (the caller chose it for a file that starts as BGZF), reported through the reader like a damaged file."""

    BATCH = 48 << 20        # compressed bytes per launch (~200 MB of text, ~3,500 BGZF blocks)
    FIRST = 6 << 20         # the first launch (the reader's first chunk is small too: get_seq_chunks first_chunk); doubling up to BATCH

    def __init__(self, path, handle, span=None, device=None):
        """span = (first file byte, one past the last, text bytes to drop in front, text bytes to deliver): a rank's share of the file
        (BgzfView.file_span); None = the whole file. device: where the members are inflated - captured HERE, on the caller's thread (the
        current device is per thread and defaults to 0 in a new one: under torchrun every rank would inflate on GPU 0)"""
        import threading
        import torch
        self.path, self.h, self.span = path, handle, span
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._stop = False
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def stop(self):
        self._stop = True

    def join(self):
        self.th.join()

    def _host_tail(self, fh, data, full):
        """the rest of a file whose members stop carrying their size: zlib, member after member, text queued behind the GPU batches"""
        import zlib
        d, inside = zlib.decompressobj(31), False
        while not self._stop:
            if not data:
                data = fh.read(4 << 20)
                if not data:
                    if inside:
                        raise ValueError("Compressed file ended before the end-of-stream marker was reached")
                    return
            try:
                out = d.decompress(data, 16 << 20)          # (at most 16 MB of text per call)
            except zlib.error as e:
                raise ValueError(str(e))
            inside = True
            if out:
                full.put((None, np.frombuffer(out, dtype=np.uint8), len(out)))
            if d.eof:
                data, d, inside = d.unused_data, zlib.decompressobj(31), False
            else:
                data = d.unconsumed_tail

    def _run(self):
        """Two threads per file. This one reads the file, walks the member headers and QUEUES the batch on the GPU (H2D of the members,
        one wave per member, D2H of the text into a pinned buffer: gz.DeviceGunzip.submit) - without waiting for it: the recurrence
        kernels own every CU for ~30 ms at a time, a batch runs in the gap behind the launch that was on the GPU when it was queued,
        and the next batch is read and indexed meanwhile. The second thread waits for a batch (sleeping), checks its status words and
        hands the text to the parser (rd_reader_feed). Two batches in flight, three pinned text buffers."""
        import queue
        import threading
        import time
        import torch
        from .. import gz
        L = N.host_lib()
        SLOTS, TEXTS = 2, 3
        full, free, slot_free = queue.Queue(), queue.Queue(), queue.Queue()
        texts = [None] * TEXTS
        for i in range(TEXTS):
            free.put(i)
        for k in range(SLOTS):
            slot_free.put(k)
        state = {"err": b"", "closed": False}
        tm = self.stage_s = {"read": 0.0, "index": 0.0, "wait_slot": 0.0, "wait_buffer": 0.0, "submit": 0.0, "wait_gpu": 0.0, "feed": 0.0, "batches": 0}
        dg = None

        def feed():
            torch.cuda.set_device(self.device)
            while True:
                item = full.get()
                if item is None:
                    break
                slot, i, nbytes = item[:3]
                skip = item[3] if len(item) > 3 else 0
                if slot is None:                    # text inflated on the host (the tail of a mixed file): i is a numpy array
                    if not state["closed"] and L.rd_reader_feed(self.h, i.ctypes.data, nbytes) != 0:
                        state["closed"] = True
                        self._stop = True
                    continue
                t0 = time.perf_counter()
                try:
                    dg.finish(slot)
                except BaseException as e:      # a damaged member: the records of the batches before it are delivered, then the error
                    if not state["err"]:
                        state["err"] = (str(e) or repr(e)).encode()[:400]
                    state["closed"] = True
                    self._stop = True
                slot_free.put(slot)
                t1 = time.perf_counter()
                if not state["closed"] and nbytes > 0 and L.rd_reader_feed(self.h, texts[i].data_ptr() + skip, nbytes) != 0:
                    state["closed"] = True          # the reader was closed: drain the queue, stop the producer
                    self._stop = True
                free.put(i)
                tm["wait_gpu"] += t1 - t0
                tm["feed"] += time.perf_counter() - t1
            L.rd_reader_feed_end(self.h, state["err"])
        ft = threading.Thread(target=feed, daemon=True)
        try:
            torch.cuda.set_device(self.device)
            dg = gz.DeviceGunzip(self.device, slots=SLOTS)
            ft.start()
            pinned = [torch.empty(self.BATCH + (1 << 20), dtype=torch.uint8, pin_memory=True) for _ in range(SLOTS)]
            bufs = [t.numpy() for t in pinned]
            carry = None                                    # bytes of an incomplete member, to go in front of the next batch
            batch = min(self.FIRST, self.BATCH)
            with open(self.path, "rb", buffering=0) as fh:
                eof = False
                file_left = skip_text = text_left = None
                if self.span is not None:           # a share of the file: whole members [c0, c1), text trimmed at both ends
                    fh.seek(self.span[0])
                    file_left, skip_text, text_left = self.span[1] - self.span[0], self.span[2], self.span[3]
                    eof = file_left <= 0
                while not self._stop:
                    t0 = time.perf_counter()
                    slot = slot_free.get()                  # (its previous batch has left the GPU: finish() returned)
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
                        slot_free.put(slot)
                        break
                    t2 = time.perf_counter()
                    n, consumed, out_bytes, streaming = dg.index(buf, have, slot=slot)
                    t3 = time.perf_counter()
                    tm["wait_slot"] += t1 - t0
                    tm["read"] += t2 - t1
                    tm["index"] += t3 - t2
                    if streaming and n == 0:
                        # a member without a size subfield behind the BGZF blocks (`cat a.bgzf.gz b.gz` is a legal .gz): the rest of the
                        # file is inflated here, on the host, one zlib stream after the other, and fed behind the batches in flight
                        slot_free.put(slot)
                        self._host_tail(fh, bytes(buf[consumed:have]), full)
                        break
                    if (n == 0 or consumed == 0) and eof:
                        if consumed < have:
                            raise ValueError("Compressed file ended before the end-of-stream marker was reached")
                        slot_free.put(slot)
                        break
                    if consumed == 0 and have >= self.BATCH:      # (nothing more can be read into the buffer: the loop would spin)
                        raise ValueError("gzip member larger than %d bytes: not a BGZF file (RD_DEVICE_INFLATE=0 reads it with the host's "
                                         "decoders)" % self.BATCH)
                    if consumed < have:
                        carry = buf[consumed:have].copy()
                    if n:
                        i = free.get()
                        t4 = time.perf_counter()
                        if texts[i] is None or texts[i].numel() < out_bytes:
                            texts[i] = None
                            texts[i] = torch.empty(int(out_bytes * 1.25) + 4096, dtype=torch.uint8, pin_memory=True)
                        dg.submit(buf, consumed, n, out_bytes, slot=slot, host_text=texts[i])
                        if text_left is None:
                            full.put((slot, i, out_bytes))
                        else:
                            drop = min(skip_text, out_bytes)
                            take = min(out_bytes - drop, text_left)
                            skip_text -= drop
                            text_left -= take
                            full.put((slot, i, take, drop))
                        tm["wait_buffer"] += t4 - t3
                        tm["submit"] += time.perf_counter() - t4
                        tm["batches"] += 1
                    else:
                        slot_free.put(slot)
                    batch = min(2 * batch, self.BATCH)
                    if text_left is not None and text_left <= 0:
                        break
        except BaseException as e:      # reported by rd_reader_next on the consumer's thread, after the records before the damage
            if not state["err"]:
                state["err"] = (str(e) or repr(e)).encode()[:400]
        full.put(None)
        if ft.ident is not None:
            ft.join()
        else:
            L.rd_reader_feed_end(self.h, state["err"])
        if os.environ.get("RD_FEED_TRACE"):
            import sys
            sys.stderr.write("feeder %s: %s\n" % (self.path, {k: round(v, 3) for k, v in tm.items()}))


class NativeReader:
    """librd_host.so reader: records are parsed in C++ straight into (pinned) buffers that go to the GPU as they are."""

    h = None

    def __init__(self, path, est_record_bytes=320, byte_range=None, arena=None, device=None):
        """byte_range = (start, end): parse only those bytes of a plain file; both must be record boundaries (plan_ranges).
        device: the GPU that inflates the members of a BGZF input (the caller's rank's device; default: the calling thread's current one)"""
        import torch
        self._torch = torch
        self._pin = torch.cuda.is_available()
        fmt = get_seq_format(path)                  # raises ValueError like the reference for unknown extensions
        self.h = C.c_void_p()
        self._feeder = None
        f = 1 if fmt.startswith("fa") else 0
        if isinstance(byte_range, BgzfRange):
            # a rank's share of a BGZF file (plan_ranges with BgzfView): its members are inflated on this rank's GPU, the text before
            # the first and behind the last record of the share is dropped
            a, b = byte_range
            c0, c1, drop = byte_range.view.file_span(a, b)
            N.host_check(N.host_lib().rd_reader_open_feed(f, C.byref(self.h)), "rd_reader_open_feed")
            if f == 1 and b < byte_range.view.size:      # FASTA share that ends before the stream does: its last record counts even if empty
                N.host_lib().rd_reader_set_flush_empty_tail(self.h, 1)
            self._feeder = _DeviceInflateFeeder(path, self.h, span=(c0, c1, drop, b - a), device=device)
        elif byte_range is None and fmt.endswith("gz") and device_inflate_wanted(path):
            # a .gz whose members say how long they are (BGZF; this build's own outputs): the members are inflated on the GPU and the
            # text is FED to the parser (ribodetector_amd/gz.py:DeviceGunzip, csrc/rd_inflate_dev.hpp) - no host inflate thread at all
            N.host_check(N.host_lib().rd_reader_open_feed(f, C.byref(self.h)), "rd_reader_open_feed")
            self._feeder = _DeviceInflateFeeder(path, self.h, device=device)
        elif byte_range is None:
            N.host_check(N.host_lib().rd_reader_open(str(path).encode(), f, C.byref(self.h)), "rd_reader_open")
        else:
            N.host_check(N.host_lib().rd_reader_open_range(str(path).encode(), f, int(byte_range[0]), int(byte_range[1]), C.byref(self.h)),
                         "rd_reader_open_range")
        self.est = est_record_bytes
        self.eof = False
        self.arena = arena                          # ShmArena: chunk buffers in shared memory instead of pinned memory

    def close(self):
        if self.h:
            if self._feeder is not None:         # abort (wakes a feeder inside rd_reader_feed) -> join -> free: no thread is inside a
                self._feeder.stop()              # feed call when the reader goes away
                N.host_lib().rd_reader_feed_abort(self.h)
                self._feeder.join()
                self._feeder = None
            N.host_lib().rd_reader_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _alloc(self, nbytes, nrec):
        if self.arena is not None:
            views, self._slot = self.arena.alloc(nbytes, nrec)
            return views
        t = self._torch
        kw = dict(pin_memory=True) if self._pin else {}
        return (t.empty(nbytes, dtype=t.uint8, **kw), t.empty(nrec + 1, dtype=t.int64, **kw), t.empty(nrec, dtype=t.int64, **kw),
                t.empty(nrec, dtype=t.int32, **kw))

    def read(self, want):
        """Exactly `want` records unless the file ends first; buffers grow as needed. Returns a Chunk or None at EOF."""
        if self.eof:
            return None
        L = N.host_lib()
        buf, rs, so, sl = self._alloc(max(1 << 16, want * self.est), want)
        n_tot, b_tot = 0, 0
        n, nb = C.c_int64(0), C.c_int64(0)
        while n_tot < want:
            rc = L.rd_reader_next(self.h, want - n_tot, buf.data_ptr() + b_tot, buf.numel() - b_tot, rs.data_ptr() + 8 * n_tot,
                                  so.data_ptr() + 8 * n_tot, sl.data_ptr() + 4 * n_tot, C.byref(n), C.byref(nb))
            if rc < 0:
                N.host_check(rc, "rd_reader_next")
            hint = 0 if n.value else int(nb.value)        # nothing delivered: the bytes the next record needs
            if n.value:
                if b_tot:                               # offsets of this call are relative to its own buffer start
                    rs[n_tot:n_tot + n.value + 1] += b_tot
                    so[n_tot:n_tot + n.value] += b_tot
                n_tot += n.value
                b_tot += nb.value
                self.est = int(1.08 * b_tot / n_tot) + 16          # bytes per record seen so far: sizes the next pinned buffer
            if rc == 1:
                self.eof = True
                break
            if n_tot < want:                            # buffer full before `want` records: grow and continue
                old_slot = getattr(self, "_slot", None)
                grown = self._alloc(max(2 * buf.numel(), b_tot + (want - n_tot) * self.est + (1 << 16), b_tot + hint + (1 << 16)), want)
                grown[0][:b_tot] = buf[:b_tot]
                grown[1][:n_tot + 1] = rs[:n_tot + 1]
                grown[2][:n_tot] = so[:n_tot]
                grown[3][:n_tot] = sl[:n_tot]
                buf, rs, so, sl = grown
                if self.arena is not None:
                    self.arena.free(old_slot)
        if n_tot == 0:
            if self.arena is not None:
                self.arena.free(self._slot)
            return None
        release = shm = None
        if self.arena is not None:
            slot, arena = self._slot, self.arena
            release = lambda: arena.free(slot)      # noqa: E731
            shm = (slot["path"], slot["cap_n"], slot["cap_b"], n_tot, b_tot)
        return Chunk(buf[:b_tot].numpy(), rs[:n_tot + 1].numpy(), so[:n_tot].numpy(), sl[:n_tot].numpy(), True, None,
                     (buf[:b_tot], so[:n_tot], sl[:n_tot], rs[:n_tot + 1]), release, shm)


def _fmt_id(path):
    return 1 if get_seq_format(path).startswith("fa") else 0


def file_info(path):
    """(size in bytes, starts with the gzip magic)"""
    size, gz = C.c_int64(0), C.c_int32(0)
    N.host_check(N.host_lib().rd_host_file_info(str(path).encode(), C.byref(size), C.byref(gz)), "rd_host_file_info")
    return int(size.value), bool(gz.value)


def find_record_start(path, pos):
    out = C.c_int64(0)
    N.host_check(N.host_lib().rd_host_find_record_start(str(path).encode(), _fmt_id(path), int(pos), C.byref(out)), "rd_host_find_record_start")
    return int(out.value)


def count_records(path, start, end):
    out = C.c_int64(0)
    N.host_check(N.host_lib().rd_host_count_records(str(path).encode(), _fmt_id(path), int(start), int(end), C.byref(out)), "rd_host_count_records")
    return int(out.value)


def skip_records(path, start, k):
    out = C.c_int64(0)
    N.host_check(N.host_lib().rd_host_skip_records(str(path).encode(), _fmt_id(path), int(start), int(k), C.byref(out)), "rd_host_skip_records")
    return int(out.value)


class _PlainView:
    """what plan_ranges needs of an input: its size and the three record-boundary helpers (librd_host.so, on the file's own bytes)"""

    def __init__(self, path):
        self.path = path
        self.size = file_info(path)[0]

    def find_record_start(self, pos):
        return find_record_start(self.path, pos)

    def count_records(self, start, end):
        return count_records(self.path, start, end)

    def skip_records(self, start, k):
        return skip_records(self.path, start, k)

    def make_range(self, start, end):
        return (start, end)


class BgzfRange(tuple):
    """(start, end) in the DECOMPRESSED stream of a BGZF file, with the member index that maps them to file bytes (`.view`)"""
    view = None


class BgzfView:
    """The same four answers for the decompressed stream of a BGZF FASTQ file (members that say how long they are: rd_host_gz_index
    walks the headers, nothing is decoded for the index). Positions are offsets into the text. find_record_start and skip_records
    look at small windows of text (a few members, inflated by zlib here); count_records covers a rank's whole share: on the GPU when
    there is one (gz.DeviceGunzip: the members are inflated in batches and the newlines counted where they land), by zlib otherwise.
    The rules are those of csrc/rd_host.cpp's rd_host_find_record_start / rd_host_count_records / rd_host_skip_records for FASTQ."""

    WINDOW = 1 << 18

    def __init__(self, path, index=None, fasta=None):
        self.path = str(path)
        self.fasta = get_seq_format(path).startswith("fa") if fasta is None else bool(fasta)      # records start with '>' lines
        if index is None:
            index = self.build_index(self.path)
        self.comp_off, self.comp_len, self.out_off = index      # per non-empty member: file offset / size; text offsets (n + 1 entries)
        self.size = int(self.out_off[-1])

    @staticmethod
    def build_index(path):
        """(comp_off int64[n], comp_len int64[n], out_off int64[n + 1]) of the non-empty members; ValueError if a member without a
        size subfield is met (the file is not BGZF all the way: not for the sharded reader)"""
        L = N.host_lib()
        size = os.path.getsize(path)
        mm = np.memmap(path, dtype=np.uint8, mode="r") if size else np.zeros(0, dtype=np.uint8)
        offs, lens, outs = [], [], []
        pos, PIECE = 0, 64 << 20
        ent = np.zeros(((PIECE // 28) + 16, 3), dtype=np.int64)       # 24-byte entries: in_off, out_off, (in_len, out_len)
        while pos < size:
            piece = mm[pos:min(size, pos + PIECE + (1 << 17))]
            n, consumed, ob = C.c_int64(0), C.c_int64(0), C.c_int64(0)
            rc = L.rd_host_gz_index(piece.ctypes.data, len(piece), 0, 0, ent.ctypes.data, len(ent), C.byref(n), C.byref(consumed), C.byref(ob))
            if rc < 0:
                raise ValueError(L.rd_host_last_error().decode())
            if rc == 1 and n.value == 0 or consumed.value == 0:
                raise ValueError("%s: a gzip member without a size subfield at byte %d" % (path, pos + consumed.value) if rc == 1 else
                                 "Compressed file ended before the end-of-stream marker was reached")
            e = ent[: n.value]
            il = (e[:, 2] & 0xffffffff).astype(np.int64)
            ol = (e[:, 2] >> 32).astype(np.int64)
            # a member's bytes in the file: header (in_off is behind it) .. trailer; its start = the previous member's end
            offs.append(pos + e[:, 0])            # (start of the DEFLATE data; the gzip header lies before it)
            lens.append(il)
            outs.append(ol)
            pos += consumed.value
        if not offs:
            return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(1, np.int64)
        data_off, il, ol = np.concatenate(offs), np.concatenate(lens), np.concatenate(outs)
        return data_off, il, np.concatenate([[0], np.cumsum(ol)]).astype(np.int64)

    # ---- text of a range, by zlib (small windows) ---------------------------------------------------------------------------
    def _member_of(self, u):
        return int(np.searchsorted(self.out_off, u, side="right") - 1)

    def text(self, a, b):
        import zlib
        a, b = max(0, int(a)), min(self.size, int(b))
        if b <= a:
            return b""
        m0, m1 = self._member_of(a), self._member_of(b - 1)
        out = []
        with open(self.path, "rb") as fh:
            for m in range(m0, m1 + 1):
                fh.seek(int(self.comp_off[m]))
                out.append(zlib.decompress(fh.read(int(self.comp_len[m])), -15))
        t = b"".join(out)
        base = int(self.out_off[m0])
        return t[a - base:b - base]

    def _next_line(self, u):
        """position behind the next newline at or after u (the size if there is none)"""
        while u < self.size:
            w = self.text(u, u + self.WINDOW)
            k = w.find(b"\n")
            if k >= 0:
                return u + k + 1
            u += len(w)
        return self.size

    def find_record_start(self, pos):
        pos = int(pos)
        if pos <= 0:
            return 0
        if pos >= self.size:
            return self.size
        line = pos if self.text(pos - 1, pos) == b"\n" else self._next_line(pos)
        while line < self.size:
            if self.fasta:
                if self.text(line, line + 1) == b">":
                    return line
            elif self.text(line, line + 1) == b"@":
                l2 = self._next_line(self._next_line(line))
                if l2 < self.size and self.text(l2, l2 + 1) == b"+":
                    return line
            line = self._next_line(line)
        return self.size

    def skip_records(self, start, k):
        if self.fasta:           # from one header line to the next, k times (rd_host_skip_records, format 1)
            at = int(start)
            for _ in range(int(k)):
                if at >= self.size:
                    break
                at = self._next_line(at)
                while at < self.size and self.text(at, at + 1) != b">":
                    at = self._next_line(at)
            return min(at, self.size)
        at, left = int(start), 4 * int(k)
        while left > 0 and at < self.size:
            w = np.frombuffer(self.text(at, at + (4 << 20)), dtype=np.uint8)
            nl = np.flatnonzero(w == 10)
            if len(nl) >= left:
                return at + int(nl[left - 1]) + 1
            left -= len(nl)
            at += len(w)
        return min(at, self.size)

    def count_records(self, start, end):
        start, end = max(0, int(start)), min(self.size, int(end))
        if end <= start:
            return 0
        if self.fasta:                      # lines that start with '>' (start is a line start: a record boundary or 0)
            return self._count_newlines(start, end, headers=True)
        lines = self._count_newlines(start, end)
        if self.text(end - 1, end) != b"\n":
            lines += 1                      # last line without terminator
        return lines // 4

    def _count_newlines(self, a, b, headers=False):
        """newlines in [a, b) - or (headers) the lines of it that start with '>' (a is a line start)"""
        m0, m1 = self._member_of(a), self._member_of(b - 1)
        import torch
        if not torch.cuda.is_available():
            total, prev_nl = 0, True
            for m in range(m0, m1 + 1, 64):
                lo, hi = int(self.out_off[m]), int(self.out_off[min(m + 64, m1 + 1)])
                t = self.text(max(a, lo), min(b, hi))
                if headers:
                    total += t.count(b"\n>") + (1 if t[:1] == b">" and (max(a, lo) == a or prev_nl) else 0)
                    prev_nl = t[-1:] == b"\n" if t else prev_nl
                else:
                    total += t.count(b"\n")
            return total
        from .. import gz
        dg = gz.DeviceGunzip(torch.device("cuda", torch.cuda.current_device()))
        total, m, prev_nl = 0, m0, True
        hdr = 64                                               # bytes read in front of a member's DEFLATE data: its gzip header
        with open(self.path, "rb") as fh:
            while m <= m1:
                e = min(m1 + 1, max(m + 1, int(np.searchsorted(self.comp_off, self.comp_off[m] + (32 << 20)))))   # ~32 MB of members
                c0 = max(0, int(self.comp_off[m]) - hdr)
                c1 = int(self.comp_off[e - 1]) + int(self.comp_len[e - 1]) + 8
                fh.seek(c0)
                buf = np.frombuffer(fh.read(c1 - c0), dtype=np.uint8).copy()
                # the member table of this batch, built from the index (no second walk over the headers)
                tab = np.zeros((e - m, 3), dtype=np.int64)
                tab[:, 0] = self.comp_off[m:e] - c0
                tab[:, 1] = self.out_off[m:e] - self.out_off[m]
                tab[:, 2] = self.comp_len[m:e] | ((self.out_off[m + 1:e + 1] - self.out_off[m:e]) << 32)
                dg.set_members(tab)
                ob = int(self.out_off[e] - self.out_off[m])
                text = dg.inflate(buf, len(buf), e - m, ob)
                lo = max(a, int(self.out_off[m])) - int(self.out_off[m])
                hi = min(b, int(self.out_off[e])) - int(self.out_off[m])
                if headers:         # '>' behind a '\n'; the window's first byte is a line start at `a`, else iff the batch before ended a line
                    w = text[lo:hi]
                    first = True if max(a, int(self.out_off[m])) == a else prev_nl
                    total += int(((w[1:] == 62) & (w[:-1] == 10)).sum()) + (1 if hi > lo and first and bool(w[0] == 62) else 0)
                    prev_nl = bool(w[-1] == 10) if hi > lo else prev_nl
                else:
                    total += int((text[lo:hi] == 10).sum())
                m = e
        return total

    def make_range(self, start, end):
        r = BgzfRange((int(start), int(end)))
        r.view = self
        return r

    def file_span(self, a, b):
        """(first file byte, one past the last, text bytes to drop in front) of the members that hold the text [a, b)"""
        if b <= a:
            return 0, 0, 0
        m0, m1 = self._member_of(a), self._member_of(b - 1)
        prev_end = int(self.comp_off[m0 - 1] + self.comp_len[m0 - 1] + 8) if m0 > 0 else 0
        return prev_end, int(self.comp_off[m1] + self.comp_len[m1] + 8), int(a - self.out_off[m0])


def plan_ranges(paths, rank, world, all_gather=None, views=None):
    """Ranges [(start, end)] - one per input file - that rank `rank` of `world` parses: bytes of a plain file, or (views = BgzfView
    objects) positions in the decompressed stream of a BGZF file.

    One file: the file is cut at the record boundaries next to size*r/world. Two mate files: record i of R1 and record i of R2 must
    land on the same rank although their byte positions differ. Every rank counts the records of its own first-cut range of both
    files (a newline count), the counts are exchanged (`all_gather(obj) -> list over ranks`, two tiny collectives), and the cut
    before rank r moves forward, in each file, to record index K_r = max over the files of the records before its first cut:
    only forward scans, each rank touches only its own bytes plus a few records past its end. The ranges concatenate to the
    whole file and hold the same record indices in every file, so the ranks' outputs concatenate to the single-rank output."""
    paths = list(paths)
    views = list(views) if views is not None else [_PlainView(p) for p in paths]
    sizes = [v.size for v in views]
    cuts = [[0] + [v.find_record_start((sz * r) // world) for r in range(1, world)] + [sz] for v, sz in zip(views, sizes)]
    for c in cuts:
        for r in range(1, world + 1):
            c[r] = max(c[r], c[r - 1])
    if len(paths) == 1 or world == 1:
        return [v.make_range(c[rank], c[rank + 1]) for v, c in zip(views, cuts)]
    if all_gather is None:
        raise ValueError("plan_ranges: several files need an all_gather callable")
    mine = [v.count_records(c[rank], c[rank + 1]) for v, c in zip(views, cuts)]
    counts = all_gather(mine)                                   # [world][files]
    before = [[sum(counts[q][f] for q in range(r)) for f in range(len(paths))] for r in range(world + 1)]
    if len(set(before[world])) != 1:
        raise ValueError("paired-end files have different numbers of records")
    K = [max(before[r]) for r in range(world)]
    start = [v.skip_records(c[rank], K[rank] - before[rank][f]) for f, (v, c) in enumerate(zip(views, cuts))]
    starts = all_gather(start)                                  # [world][files]
    return [views[f].make_range(starts[rank][f], starts[rank + 1][f] if rank + 1 < world else sizes[f]) for f in range(len(paths))]


def chunk_schedule(seq_file, chunk_size, byte_range=None, first_chunk=1 << 17):
    """Record counts of the successive chunks of a plain file for readers that cannot use byte segments (mate files: chunk i of
    R1 and chunk i of R2 must hold the same records, so both follow ONE schedule, computed from the first file): small first
    chunks, doubling up to chunk_size, and a last chunk_size worth of records cut into halves down to first_chunk - what runs
    after the last byte was parsed (the last chunk's kernels and its write) is hidden behind nothing. The number of records is
    an estimate from the head of the file; the readers keep reading small chunks if the schedule ends before the file does."""
    size = file_info(seq_file)[0]
    b0, b1 = (0, size) if byte_range is None else (int(byte_range[0]), int(byte_range[1]))
    if b1 <= b0:
        return []
    head = min(max(find_record_start(seq_file, min(b1, b0 + (1 << 18))), b0), b1)
    nrec = count_records(seq_file, b0, head) if head > b0 else 0
    if not nrec:
        return []
    total = int((b1 - b0) / ((head - b0) / nrec))
    first = max(1, min(first_chunk, chunk_size))
    top = chunk_size                                   # largest chunk: chunk_size, or less when the file is too small for the two ramps
    while True:
        up, t = [], first
        while t < top:
            up.append(t)
            t *= 2
        down, t = [], top // 2
        while t >= first:
            down.append(t)
            t //= 2
        down.append(first)
        if sum(up) + top + sum(down) <= total or top <= first:
            break
        top //= 2
    body = max(0, total - sum(up) - sum(down))
    rest = body % top
    return up + [top] * (body // top) + ([rest] if rest >= first else []) + down


def get_seq_chunks(seq_file, chunk_size=1048576, byte_range=None, first_chunk=None, arena=None, schedule=None, device=None):
    """Chunks of at most `chunk_size` records (reference seq_encoder.py:75-87), as `Chunk` arrays, parsed by librd_host.so.
    byte_range: parse only that part of a plain file (multi-rank CLI, plan_ranges). first_chunk: the first chunk holds that many
    records, the following ones twice as many each up to chunk_size (the kernels start earlier; mate files given the same
    schedule still pair up chunk by chunk)."""
    r = NativeReader(seq_file, byte_range=byte_range, arena=arena, device=device)
    want = chunk_size if not first_chunk else max(1, min(int(first_chunk), chunk_size))
    sched = list(schedule) if schedule else None    # explicit record counts (chunk_schedule); afterwards: chunks of its last entry
    try:
        while True:
            if sched is not None:
                want = sched.pop(0) if len(sched) > 1 else sched[0]
            c = r.read(want)
            if sched is None:
                want = min(chunk_size, want * 2)
            if c is None:
                return
            yield c
    finally:
        r.close()


def plan_segments(seq_file, chunk_size, byte_range=None, first_chunk=1 << 18):
    """Record-aligned byte segments of a plain file holding about `chunk_size` records each (the first ones fewer: `first_chunk`,
    doubling, so that the first kernels start while the rest of the file is still being parsed). The record size is estimated
    from the head of the range; the cuts are found with rd_host_find_record_start (the same resynchronisation the multi-rank byte
    ranges use). Segments concatenate to the range, so their records concatenate to the range's records."""
    size = file_info(seq_file)[0]
    b0, b1 = (0, size) if byte_range is None else (int(byte_range[0]), int(byte_range[1]))
    if b1 <= b0:
        return []
    head = find_record_start(seq_file, min(b1, b0 + (1 << 18)))
    head = min(max(head, b0), b1)
    nrec = count_records(seq_file, b0, head) if head > b0 else 0
    est = (head - b0) / nrec if nrec else float(b1 - b0)
    segs, pos, want = [], b0, max(1, min(first_chunk, chunk_size))
    while pos < b1:
        target = pos + max(1, int(want * est))
        nxt = b1 if target >= b1 else min(max(find_record_start(seq_file, target), pos), b1)
        if nxt <= pos:                               # a record longer than the step: take everything up to the next boundary
            nxt = min(find_record_start(seq_file, pos + 1), b1)
            if nxt <= pos:
                nxt = b1
        segs.append((pos, nxt))
        pos = nxt
        want = min(chunk_size, want * 2)
    # ... and small last chunks: what runs after the last byte was parsed (the last chunk's kernels and its write) is not hidden
    # behind anything, so the last segment is halved until it is down to the size of the first
    small = max(1, int(min(first_chunk, chunk_size) * est))
    while len(segs) > 1 and segs[-1][1] - segs[-1][0] > 2 * small:
        a, b = segs.pop()
        mid = min(max(find_record_start(seq_file, (a + b) // 2), a), b)
        if mid <= a or mid >= b:
            segs.append((a, b))
            break
        segs += [(a, mid), (mid, b)]
    return segs


def get_seq_chunks_parallel(seq_file, chunk_size=1048576, byte_range=None, workers=2, first_chunk=1 << 18):
    """The chunks of a PLAIN file parsed by `workers` reader threads over consecutive byte segments (plan_segments), delivered in
    file order. One librd_host.so reader delivers ~34 M reads/s of 100 bp FASTQ - the GPU path takes as much - so a single-end run
    was bound by its one parser thread; chunks hold about chunk_size records instead of exactly that many
    (the reference's --chunk_size bounds memory, it is not a framing promise: seq_encoder.py:75-87)."""
    import threading
    segs = plan_segments(seq_file, chunk_size, byte_range, first_chunk)
    if len(segs) <= 1 or workers <= 1:
        yield from get_seq_chunks(seq_file, chunk_size, byte_range)
        return
    results = [None] * len(segs)                      # per segment: list of chunks | exception
    done = [threading.Event() for _ in segs]
    ahead = threading.Semaphore(workers + 1)          # segments parsed but not yet consumed: bounds the pinned memory in flight
    nxt = [0]
    lock = threading.Lock()
    stop = threading.Event()

    def work():
        while not stop.is_set():
            ahead.acquire()
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= len(segs) or stop.is_set():
                ahead.release()
                return
            try:
                r = NativeReader(seq_file, byte_range=segs[i])
                out = []
                try:
                    est_n = int((segs[i][1] - segs[i][0]) / max(r.est, 1)) + 1024
                    while True:
                        c = r.read(max(est_n, 1024))
                        if c is None:
                            break
                        out.append(c)
                finally:
                    r.close()
                results[i] = out
            except BaseException as e:               # noqa: BLE001 - surfaced by the consumer
                results[i] = e
            done[i].set()
    th = [threading.Thread(target=work, daemon=True) for _ in range(workers)]
    [t.start() for t in th]
    try:
        for i in range(len(segs)):
            done[i].wait()
            res, results[i] = results[i], None
            if isinstance(res, BaseException):
                raise res
            for c in res:
                yield c
            ahead.release()
    finally:
        stop.set()
        for _ in th:
            ahead.release()


def get_pairedread_chunks(r1_seq_file, r2_seq_file, chunk_size=1048576):
    """zip of the two mates' chunk streams (reference seq_encoder.py:90-92)."""
    for c1, c2 in zip(get_seq_chunks(r1_seq_file, chunk_size), get_seq_chunks(r2_seq_file, chunk_size)):
        if len(c1.seq_len) != len(c2.seq_len):
            raise ValueError("paired-end files have different numbers of records")
        yield c1, c2


def select_records(chunk, mask):
    """Bytes of the records where mask is True, in input order, each terminated by '\\n'
    (reference: fh.write('\\n'.join(selected) + '\\n'), detect.py:489-492)."""
    mask = np.asarray(mask, dtype=bool)
    if not mask.any():
        return b""
    if not chunk.verbatim:
        return ("\n".join(r for r, m in zip(chunk.records, mask) if m) + "\n").encode("latin-1")
    if mask.all():
        return chunk.buf[chunk.rec_start[0]:chunk.rec_start[-1]].tobytes()
    delta = np.zeros(len(chunk.buf) + 1, dtype=np.int8)
    np.add.at(delta, chunk.rec_start[:-1][mask], 1)
    np.add.at(delta, chunk.rec_start[1:][mask], -1)
    keep = np.cumsum(delta[:-1], dtype=np.int8).astype(bool)
    return chunk.buf[keep].tobytes()


class NativeWriter:
    """librd_host.so writer: gzip level 5 when the name ends with 'gz', else plain (reference detect.py:729-741)."""

    def __init__(self, path):
        self.h = C.c_void_p()
        N.host_check(N.host_lib().rd_writer_open(str(path).encode(), C.byref(self.h)), "rd_writer_open")

    @property
    def threads(self):
        """compressor threads of this writer (the -t/--threads value in force when it was opened)"""
        return int(N.host_lib().rd_writer_threads(self.h))

    def write_selected(self, chunk, labels, want):
        """append the records of `chunk` whose label == want, in input order (reference detect.py:485-492)"""
        if not chunk.verbatim:
            raise ValueError("write_selected needs a chunk with normalised record text")
        labels = np.ascontiguousarray(labels, dtype=np.int8)
        buf = np.ascontiguousarray(chunk.buf)
        rs = np.ascontiguousarray(chunk.rec_start, dtype=np.int64)
        N.host_check(N.host_lib().rd_writer_write_selected(self.h, buf.ctypes.data, rs.ctypes.data, len(labels), labels.ctypes.data,
                                                           int(want)), "rd_writer_write_selected")

    def write_text(self, ptr, nbytes):
        """append text that already is the selected records in input order (packed on the GPU: gz.DeviceSelect); ptr: host address"""
        N.host_check(N.host_lib().rd_writer_write_text(self.h, ptr, int(nbytes)), "rd_writer_write_text")

    def set_eof_marker(self, on):
        """off: a part of a file that is joined with others afterwards gets no BGZF end-of-file block of its own"""
        N.host_check(N.host_lib().rd_writer_set_eof_marker(self.h, 1 if on else 0), "rd_writer_set_eof_marker")

    def write_members(self, ptr, nbytes):
        """append complete gzip members made on the GPU (ribodetector_amd/gz.py); ptr: host address of the bytes"""
        N.host_check(N.host_lib().rd_writer_write_members(self.h, ptr, int(nbytes)), "rd_writer_write_members")

    def close(self):
        if self.h:
            N.host_check(N.host_lib().rd_writer_close(self.h), "rd_writer_close")
            self.h = None


def open_for_write(read_file):
    return NativeWriter(read_file)


def concatenate_parts(final_path, part_paths):
    """final = part0 + part1 + ... (then the parts are removed). Plain text concatenates trivially; for .gz outputs every part is a
    sequence of complete gzip members, and a concatenation of members is a valid gzip file (RFC 1952; the single-rank writer
    produces several members per file as well). The copy runs inside the kernel (sendfile)."""
    import os
    with open(final_path, "wb") as out:
        for p in part_paths:
            with open(p, "rb") as src:
                left = os.fstat(src.fileno()).st_size
                off = 0
                while left > 0:
                    k = os.sendfile(out.fileno(), src.fileno(), off, min(left, 1 << 30))
                    if k <= 0:
                        raise OSError("sendfile stalled while joining %s" % p)
                    off += k
                    left -= k
            os.remove(p)


def place_part(final_path, part_path, offset):
    """copy a part file into `final_path` at byte `offset` (the file exists at its full size already) and remove the part: the
    multi-rank CLI's ranks join their parts concurrently, each at the offset that the sizes of the lower ranks' parts give.
    copy_file_range keeps the bytes inside the kernel; a plain pread/pwrite loop is the fallback where it is not supported."""
    import os
    src = os.open(part_path, os.O_RDONLY)
    dst = os.open(final_path, os.O_WRONLY)
    try:
        left = os.fstat(src).st_size
        so, do = 0, int(offset)
        while left > 0:
            try:
                k = os.copy_file_range(src, dst, min(left, 1 << 30), so, do)
            except (OSError, AttributeError):
                buf = os.pread(src, min(left, 8 << 20), so)
                k = os.pwrite(dst, buf, do)
            if k <= 0:
                raise OSError("copy stalled while placing %s" % part_path)
            so += k
            do += k
            left -= k
    finally:
        os.close(src)
        os.close(dst)
    os.remove(part_path)
