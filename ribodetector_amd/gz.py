"""Device-side gzip of the output files (C ABI rd_gz_*, csrc/rd_deflate.hpp).

The reference writes its outputs through `gzip.open(path, 'wt', compresslevel=5)` when the name ends in 'gz' (reference
detect.py:729-741). Here the records of a chunk that carry one label are compressed where they already are - in HBM - into complete
gzip members (BGZF framing), and only the compressed bytes travel to the host, which appends them to the file.
"""
import ctypes as C

import torch

from . import _native as N

MEMBER = 65280      # input bytes per gzip member (BGZF's block size)

# Side streams are taken from a process-wide pool and given back: torch's caching allocator keeps freed blocks PER STREAM, so a run
# that creates its own streams leaves its gigabytes of batch / chunk buffers cached under streams nobody will use again - a process
# that calls detect.main() repeatedly (a service, the benchmarks) grew by ~20 GB per call until the runtime itself ran out of memory.
_stream_pool = {}
_stream_lock = __import__("threading").Lock()


def acquire_stream(device, priority=0):
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), priority)
    with _stream_lock:
        free = _stream_pool.setdefault(key, [])
        if free:
            return free.pop()
    with torch.cuda.device(device):
        return torch.cuda.Stream(device, priority=priority)


def release_stream(stream, priority=0):
    """give a stream back - once everything queued on it has finished"""
    if stream is None:
        return
    with _stream_lock:
        _stream_pool.setdefault((stream.device.index, priority), []).append(stream)


def eof_block():
    """BGZF's 28-byte end-of-file marker (an empty gzip member); appended once when a device-written file is closed"""
    buf = (C.c_uint8 * 28)()
    n = N.lib().rd_gz_eof_block(buf, 28)
    if n != 28:
        raise RuntimeError("rd_gz_eof_block failed: %s" % N.lib().rd_last_error().decode())
    return bytes(buf)


class DeviceGzip:
    """compress_selected(text, rec_start, labels, label): the records i of the chunk with labels[i] == label, in input order, as
    gzip members. text uint8[*], rec_start int64[n+1], labels int8[n] (or uint8) - all on the device. Returns (out, info): out is a
    device uint8 buffer owned by this object until the next call with the same `slot`, info a device int64[4] tensor = (compressed
    bytes, uncompressed bytes, members, 0). Asynchronous on the current stream."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._ws = None
        self._out = {}

    def compress_selected(self, text, rec_start, labels, label, slot=0, out_frac=1.0):
        lib = N.lib()
        n = int(labels.numel())
        tb = int(text.numel())
        if rec_start.dtype != torch.int64 or rec_start.numel() < n + 1 or not rec_start.is_contiguous():
            raise TypeError("compress_selected: rec_start must be a contiguous int64 tensor of n + 1 entries")
        if labels.dtype not in (torch.int8, torch.uint8) or text.dtype != torch.uint8:
            raise TypeError("compress_selected: labels must be int8 / uint8 and text uint8")
        need = int(lib.rd_gz_workspace_bytes(n, tb))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=self.device)
        # out_frac < 1: a smaller output buffer than the worst case (every member stored). FASTQ shrinks 4-6x, so half the bound is
        # plenty; if a chunk does not fit (info[0] > out.numel()) the stream in `out` is incomplete and the caller falls back
        cap = max(int(int(lib.rd_gz_out_bound(tb)) * float(out_frac)) + (1 << 20), 256) if out_frac < 1.0 else max(int(lib.rd_gz_out_bound(tb)), 256)
        out = self._out.get(slot)
        if out is None or out.numel() < cap:
            self._out[slot] = None
            out = self._out[slot] = torch.empty(cap, dtype=torch.uint8, device=self.device)
        info = torch.empty(4, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            N.check(lib.rd_gz_compress_selected(N.ptr(text), tb, N.ptr(rec_start), N.ptr(labels), n, int(label), N.ptr(out), out.numel(),
                                                N.ptr(info), N.ptr(self._ws), self._ws.numel(), N.stream_ptr(self.device)),
                    "rd_gz_compress_selected")
        return out, info


class DeviceSelect:
    """pack_selected(text, rec_start, labels, label): the records with labels[i] == label as ONE contiguous text on the device, in input
    order (C ABI rd_select_pack) - the plain-output counterpart of DeviceGzip for chunks whose text lives in HBM
    (data_loader/device_reader.py). Returns (out, info): info int64[4] on the device, info[1] = bytes, info[3] != 0 = bad record table."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._ws = None
        self._out = {}

    def pack_selected(self, text, rec_start, labels, label, slot=0):
        lib = N.lib()
        n, tb = int(labels.numel()), int(text.numel())
        if rec_start.dtype != torch.int64 or rec_start.numel() < n + 1 or not rec_start.is_contiguous():
            raise TypeError("pack_selected: rec_start must be a contiguous int64 tensor of n + 1 entries")
        if labels.dtype not in (torch.int8, torch.uint8) or text.dtype != torch.uint8:
            raise TypeError("pack_selected: labels must be int8 / uint8 and text uint8")
        need = int(lib.rd_select_workspace_bytes(n))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=self.device)
        out = self._out.get(slot)
        if out is None or out.numel() < tb + 16:
            self._out[slot] = None
            out = self._out[slot] = torch.empty(tb + 256, dtype=torch.uint8, device=self.device)
        info = torch.empty(4, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            N.check(lib.rd_select_pack(N.ptr(text), tb, N.ptr(rec_start), N.ptr(labels), n, int(label), N.ptr(out), out.numel(), N.ptr(info),
                                       N.ptr(self._ws), self._ws.numel(), N.stream_ptr(self.device)), "rd_select_pack")
        return out, info


# ---- the input side: gzip members inflated on the device (csrc/rd_inflate_dev.hpp) ---------------------------------------------------------

GZI_ERRORS = {1: "invalid block type", 2: "invalid Huffman code", 3: "invalid code lengths set", 4: "more data than the member's ISIZE says",
              5: "invalid distance too far back", 6: "Compressed file ended before the end-of-stream marker was reached",
              7: "Incorrect length of data produced", 8: "CRC check failed", 9: "invalid stored block lengths", 10: "member table entry outside the buffers"}


def is_member_indexed(path):
    """does the file start with a gzip member that says how long it is? "BC" (BGZF: bgzip, htslib, this build's device writer),
    "RD" (this build's host writer: 4 MiB members) or None"""
    try:
        with open(path, "rb") as fh:
            h = fh.read(64)
    except OSError:
        return None
    if len(h) < 18 or h[:3] != b"\x1f\x8b\x08" or h[3] != 4:
        return None
    xlen = h[10] | (h[11] << 8)
    q = 12
    while q + 4 <= min(12 + xlen, len(h)):
        sl = h[q + 2] | (h[q + 3] << 8)
        if h[q:q + 2] == b"BC" and sl == 2:
            return "BC"
        if h[q:q + 2] == b"RD" and sl == 4:
            return "RD"
        q += 4 + sl
    return None


class _GunzipSlot:
    """the buffers of one batch in flight: member table (pinned + device), compressed bytes, text, status"""

    def __init__(self):
        self.cap_members = 0
        self.mem_host = self.mem_dev = self.status = self.status_host = None
        self.comp_dev = self.text_dev = None
        self.event = None
        self.n = 0


class DeviceGunzip:
    """index(buf, nbytes) walks the member headers of a host buffer; inflate(...) ships the members to the GPU, decodes them (one wave
    per member) and returns the text as a device tensor after synchronising. submit(...) / finish(...) are the same in two halves, for
    a caller that keeps several batches in flight (data_loader/fastx_parser.py): everything of a batch - H2D of the compressed bytes
    and the member table, the kernel, D2H of the text into the caller's pinned buffer and of the status words - is queued on this
    object's stream (high priority: it runs in the gaps between the recurrence launches), and finish() sleeps until it is done."""

    def __init__(self, device, slots=1, stream=None):
        import numpy as np
        self.device = torch.device(device)
        if stream is not None:
            self.stream = stream
        else:
            with torch.cuda.device(self.device):
                self.stream = torch.cuda.Stream(self.device, priority=-1)
        self._np = np
        self._slots = [_GunzipSlot() for _ in range(max(1, slots))]

    # (slot 0's buffers under their old names: tools and tests time the kernel on them)
    _status = property(lambda self: self._slots[0].status)
    _mem_dev = property(lambda self: self._slots[0].mem_dev)
    _mem_host = property(lambda self: self._slots[0].mem_host)
    _comp_dev = property(lambda self: self._slots[0].comp_dev)
    _text_dev = property(lambda self: self._slots[0].text_dev)

    def index(self, buf, nbytes, slot=0, max_members=None):
        """walk the members in buf[:nbytes] (host numpy uint8): (n, consumed, out_bytes, streaming_needed). max_members bounds the walk
        (the rest of the bytes stays unconsumed): a batch of m BGZF blocks inflates to at most m * 65,536 bytes, whatever it claims"""
        sl = self._slots[slot]
        cap = max(1024, nbytes // 64 + 16)
        if sl.mem_host is None or sl.cap_members < cap:
            sl.cap_members = cap
            sl.mem_host = torch.empty(cap * 24, dtype=torch.uint8, pin_memory=True)
        if max_members is not None:
            cap = max(1, min(cap, int(max_members)))
        n, consumed, ob = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        rc = N.host_lib().rd_host_gz_index(buf.ctypes.data, int(nbytes), 0, 0, sl.mem_host.data_ptr(), cap, C.byref(n), C.byref(consumed), C.byref(ob))
        if rc < 0:
            raise ValueError(N.host_lib().rd_host_last_error().decode())
        return int(n.value), int(consumed.value), int(ob.value), rc == 1

    def set_members(self, table, slot=0):
        """a member table made by the caller instead of index(): int64[n, 3] rows of (offset of the DEFLATE data in the buffer, offset of
        the text, in_len | out_len << 32) - the 24-byte layout of rd_gz_member. Returns n."""
        sl = self._slots[slot]
        rows = self._np.ascontiguousarray(table, dtype=self._np.int64).reshape(-1, 3)
        need = rows.shape[0] * 24
        if sl.mem_host is None or sl.mem_host.numel() < need:
            sl.mem_host = torch.empty(max(need, 1 << 16), dtype=torch.uint8, pin_memory=True)
            sl.cap_members = sl.mem_host.numel() // 24
        sl.mem_host[:need].copy_(torch.from_numpy(rows.view(self._np.uint8).reshape(-1)))
        return rows.shape[0]

    def submit(self, buf, nbytes, n, out_bytes, slot=0, host_text=None, text_out=None):
        """queue the batch indexed by the last index(..., slot) call over buf[:nbytes] (buf: pinned host memory if the copy is to be
        asynchronous; it may be reused once finish() has returned). host_text: pinned uint8 tensor that receives the text.
        text_out: device uint8 tensor of >= out_bytes bytes that receives the text instead of this object's own buffer (the caller's
        batch buffer: data_loader/device_reader.py indexes the records where the members land)."""
        lib = N.lib()
        sl = self._slots[slot]
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            if sl.comp_dev is None or sl.comp_dev.numel() < nbytes + 16:
                sl.comp_dev = None
                sl.comp_dev = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=self.device)
            if text_out is not None:
                if text_out.numel() < out_bytes or text_out.dtype != torch.uint8 or not text_out.is_cuda:
                    raise ValueError("DeviceGunzip.submit: text_out must be a device uint8 tensor of at least out_bytes bytes")
                text_dev = text_out
            elif sl.text_dev is None or sl.text_dev.numel() < out_bytes:
                sl.text_dev = None
                sl.text_dev = torch.empty(int(out_bytes * 1.25) + 4096, dtype=torch.uint8, device=self.device)
            if text_out is None:
                text_dev = sl.text_dev
            if sl.status is None or sl.status.numel() < n:
                sl.status = torch.empty(max(n, 1024) * 2, dtype=torch.int32, device=self.device)
                sl.status_host = torch.empty(max(n, 1024) * 2, dtype=torch.int32, pin_memory=True)
            if sl.mem_dev is None or sl.mem_dev.numel() < n * 24:
                sl.mem_dev = torch.empty(max(n, 1024) * 2 * 24, dtype=torch.uint8, device=self.device)
            src = torch.from_numpy(buf[:nbytes])
            if src.is_pinned():
                N.copy_bytes(sl.comp_dev, src, nbytes, self.stream)                # (a kernel: see rd_copy_bytes)
            else:
                sl.comp_dev[:nbytes].copy_(src, non_blocking=True)
            N.copy_bytes(sl.mem_dev, sl.mem_host, n * 24, self.stream, workgroups=4)
            N.check(lib.rd_gz_inflate_members(N.ptr(sl.comp_dev), int(nbytes), N.ptr(sl.mem_dev), n, N.ptr(text_dev), int(out_bytes),
                                              N.ptr(sl.status), C.c_void_p(self.stream.cuda_stream)), "rd_gz_inflate_members")
            if host_text is not None:
                host_text[:out_bytes].copy_(text_dev[:out_bytes], non_blocking=True)
            N.copy_bytes(sl.status_host.view(torch.uint8), sl.status.view(torch.uint8), n * 4, self.stream, workgroups=4)
            sl.event = N.new_event()
            sl.event.record(self.stream)
        sl.n = n
        return slot

    def finish(self, slot=0):
        """wait for the batch of `slot` (sleeping, not spinning: the host cores belong to readers and writers); ValueError names the
        first member that did not decode"""
        import time
        sl = self._slots[slot]
        N.wait_event(sl.event)
        st = sl.status_host[: sl.n].numpy()
        bad = self._np.flatnonzero(st)
        if bad.size:
            i = int(bad[0])
            raise ValueError("gzip member %d: %s" % (i, GZI_ERRORS.get(int(st[i]), "error %d" % int(st[i]))))

    def inflate(self, buf, nbytes, n, out_bytes):
        """the n members indexed by the last index() call over buf[:nbytes] -> device tensor of out_bytes bytes"""
        self.submit(buf, nbytes, n, out_bytes, slot=0)
        self.finish(0)
        return self._slots[0].text_dev[:out_bytes]


# ---- ONE DEFLATE stream inflated on the device (csrc/rd_inflate_stream.hpp; RD_DEVICE_INFLATE=members: off) ------------------------------

GZS_ERRORS = {1: "invalid deflate data", 2: "a section did not end where the next one starts", 3: "a section produced more symbols than its slot holds",
              4: "Compressed file ended before the end-of-stream marker was reached", 5: "invalid distance too far back",
              6: "the batch produced more text than its buffer holds", 7: "no block start found where the stream goes on"}


def gzip_header_len(buf):
    """length of the RFC 1952 member header at the start of `buf` (bytes-like); None if more bytes are needed; ValueError if it is none"""
    b = bytes(buf[:10])
    if len(b) < 10:
        return None
    if b[0] != 0x1f or b[1] != 0x8b:
        raise ValueError("Not a gzipped file (%r)" % b[:2])
    if b[2] != 8:
        raise ValueError("Unknown compression method")
    flg, p = b[3], 10
    if flg & 4:
        if len(buf) < p + 2:
            return None
        p += 2 + (buf[p] | (buf[p + 1] << 8))
    for bit in (8, 16):
        if flg & bit:
            while True:
                if p >= len(buf):
                    return None
                p += 1
                if buf[p - 1] == 0:
                    break
    if flg & 2:
        p += 2
    return p if p <= len(buf) else None


class DeviceStreamGunzip:
    """The batches of ONE gzip member through C ABI rd_gz_stream_inflate: submit() queues a batch - its compressed bytes (pinned host
    memory) to HBM by a copy kernel, block-start search, section decode, window chain, resolve, CRC - and returns a ticket; finish()
    sleeps until the batch's 64-byte state has arrived and returns it. The stream's state (window, CRC, length, where the next batch
    starts) is carried on the device from one submit() to the next."""

    SECTION = 16 << 10          # compressed bytes per section = per wave
    BATCH = 64 << 20            # compressed bytes per batch (its sections); the bytes behind them are searched for the next batch's start
    SLACK = 256 << 10           # ... this many (a block start every 10-60 KB in zlib / pigz output)
    CAP_RATIO = 24              # symbols a section may produce per compressed byte (FASTQ: 4-5; a section also takes over its
                                # successors that hold no block start)
    TEXT_RATIO = 12             # text bytes a batch may produce per compressed byte

    def __init__(self, device, stream):
        import os
        self.device, self.stream = torch.device(device), stream
        self.carry = self.win = None
        self._ws = None
        if os.environ.get("RD_GZS_BATCH"):            # (experiments: compressed bytes per batch)
            self.BATCH = int(os.environ["RD_GZS_BATCH"])

    def text_cap(self, data_bytes):
        return int(data_bytes) * self.TEXT_RATIO + (1 << 20)

    def submit(self, src, valid_bytes, data_bytes, first_start_bit, at_eof, text_out):
        """src: pinned uint8 tensor holding valid_bytes bytes of the file from the batch's first byte; the sections cover
        [0, data_bytes); text_out: device uint8 tensor that receives the text"""
        lib = N.lib()
        cap_syms = self.SECTION * self.CAP_RATIO
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            comp = torch.empty(((valid_bytes + 4096 + 255) // 256) * 256, dtype=torch.uint8, device=self.device)
            N.copy_bytes(comp, src, valid_bytes, self.stream)
            comp[valid_bytes:valid_bytes + 4096].zero_()
            need = int(lib.rd_gz_stream_workspace_bytes(data_bytes, self.SECTION, cap_syms, text_out.numel()))
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            state = torch.zeros(8, dtype=torch.int64, device=self.device)
            win_out = torch.empty(32768, dtype=torch.uint8, device=self.device)
            N.check(lib.rd_gz_stream_inflate(N.ptr(comp), comp.numel(), int(data_bytes), int(valid_bytes), self.SECTION, cap_syms,
                                             int(first_start_bit) & 0xffffffff, N.ptr(self.carry), int(self._delta_bits) if self.carry is not None else 0,
                                             1 if at_eof else 0, N.ptr(self.win), N.ptr(win_out), N.ptr(text_out), text_out.numel(), N.ptr(state),
                                             N.ptr(self._ws), self._ws.numel(), C.c_void_p(self.stream.cuda_stream)), "rd_gz_stream_inflate")
            host = torch.empty(64, dtype=torch.uint8, pin_memory=True)
            N.copy_bytes(host, state.view(torch.uint8), 64, self.stream, workgroups=1)
            ev = N.new_event()
            ev.record(self.stream)
        self.carry, self.win, self._delta_bits = state, win_out, int(data_bytes) * 8
        return {"host": host, "event": ev, "keep": (comp, state, win_out)}

    @staticmethod
    def finish(ticket):
        """-> dict(total_len, n_text, crc, status, final, end_bit, next_start, bad_section, n_sections)"""
        import numpy as np
        N.wait_event(ticket["event"])
        h = ticket["host"].numpy()
        q = h.view(np.uint64)
        d = h.view(np.uint32)
        return {"total_len": int(q[0]), "n_text": int(h.view(np.int64)[1]), "crc": int(d[4]), "status": int(d[5]), "final": int(d[6]), "end_bit": int(d[7]),
                "next_start": int(d[8]), "win_valid": int(d[9]), "bad_section": int(d[10]), "n_sections": int(d[11])}


# ---- a RANGE of one DEFLATE stream: every rank of a node decodes its own part of ONE .gz (csrc/rd_inflate_stream.hpp, round 6) ---------

GZS_SEARCH = 0xfffffffe        # first_start_bit of a range that begins inside the stream (include/ribodetector_amd.h RD_GZS_SEARCH)


class DeviceRangeGunzip(DeviceStreamGunzip):
    """The batches of a RANGE of a gzip member through C ABI rd_gz_range_decode: as DeviceStreamGunzip, but the 32 KiB in front of the
    range need not be known - a batch's text comes back as 16-bit symbols (a byte, or 0x8000 | i = byte i of the window in front of the
    RANGE) and `self.map` is the window behind the batches so far in the same form. resolve() turns a batch's symbols into bytes once
    the window is known (data_loader/gz_shard.py: the ranks exchange their maps)."""

    def __init__(self, device, stream):
        super().__init__(device, stream)
        self.map = None
        self._rws = None

    def submit(self, src, valid_bytes, data_bytes, first_start_bit, at_eof):
        lib = N.lib()
        cap_syms = self.SECTION * self.CAP_RATIO
        text_cap = self.text_cap(data_bytes)
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            comp = torch.empty(((valid_bytes + 4096 + 255) // 256) * 256, dtype=torch.uint8, device=self.device)
            N.copy_bytes(comp, src, valid_bytes, self.stream)
            comp[valid_bytes:valid_bytes + 4096].zero_()
            sym = torch.empty(text_cap, dtype=torch.int16, device=self.device)
            need = int(lib.rd_gz_range_workspace_bytes(data_bytes, self.SECTION, cap_syms, text_cap))
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            state = torch.zeros(8, dtype=torch.int64, device=self.device)
            map_out = torch.empty(32768, dtype=torch.int16, device=self.device)
            N.check(lib.rd_gz_range_decode(N.ptr(comp), comp.numel(), int(data_bytes), int(valid_bytes), self.SECTION, cap_syms,
                                           int(first_start_bit) & 0xffffffff, N.ptr(self.carry), int(self._delta_bits) if self.carry is not None else 0,
                                           1 if at_eof else 0, N.ptr(self.map), N.ptr(map_out), N.ptr(sym), text_cap, N.ptr(state),
                                           N.ptr(self._ws), self._ws.numel(), C.c_void_p(self.stream.cuda_stream)), "rd_gz_range_decode")
            host = torch.empty(64, dtype=torch.uint8, pin_memory=True)
            N.copy_bytes(host, state.view(torch.uint8), 64, self.stream, workgroups=1)
            ev = N.new_event()
            ev.record(self.stream)
        self.carry, self.map, self._delta_bits = state, map_out, int(data_bytes) * 8
        return {"host": host, "event": ev, "keep": (comp, state, map_out), "sym": sym}

    @staticmethod
    def finish(ticket):
        import numpy as np
        r = DeviceStreamGunzip.finish(ticket)
        r["first_start"] = int(ticket["host"].numpy().view(np.uint64)[6])       # the bit the batch's text starts at (0xffffffff: none)
        return r

    def resolve(self, sym, n, window, win_valid, text_out, rstate):
        """text_out[:n] = the n symbols of `sym` as bytes; window: device uint8[32768] (None when win_valid == 0: the member starts with
        the range); rstate: device int64[8], zeroed before the range's first batch - accumulates the CRC-32 and length of the range's
        text; a marker in front of the window's valid bytes sets its status word"""
        lib = N.lib()
        with torch.cuda.device(self.device), torch.cuda.stream(self.stream):
            need = int(lib.rd_gz_range_resolve_workspace_bytes(n))
            if self._rws is None or self._rws.numel() < need:
                self._rws = None
                self._rws = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=self.device)
            N.check(lib.rd_gz_range_resolve(N.ptr(sym), int(n), N.ptr(window), int(win_valid), N.ptr(text_out), N.ptr(rstate), N.ptr(self._rws),
                                            self._rws.numel(), C.c_void_p(self.stream.cuda_stream)), "rd_gz_range_resolve")


def apply_map(m, window):
    """the window behind a range = its map applied to the window in front of it (numpy: m uint16[32768], window uint8[32768])"""
    import numpy as np
    m = np.asarray(m).view(np.uint16)
    return np.where(m & 0x8000, window[m & 0x7fff], (m & 0xff).astype(np.uint8)).astype(np.uint8)


def crc32_combine(crc1, crc2, len2):
    """CRC-32 of A || B from crc32(A), crc32(B), len(B) - zlib's crc32_combine (it is not exposed by Python's zlib module)"""
    if len2 <= 0:
        return crc1 & 0xffffffff

    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[n]) for n in range(32)]
    odd = [0xedb88320] + [1 << n for n in range(31)]       # the operator for one zero bit
    even = square(odd)                                      # two
    odd = square(even)                                      # four
    while True:
        even = square(odd)                                  # first pass: one zero byte
        if len2 & 1:
            crc1 = times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            crc1 = times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return (crc1 ^ crc2) & 0xffffffff
