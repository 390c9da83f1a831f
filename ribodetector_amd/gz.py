"""Device-side gzip of the output files (C ABI rd_gz_*, csrc/rd_deflate.hpp).

The reference writes its outputs through `gzip.open(path, 'wt', compresslevel=5)` when the name ends in 'gz' (reference
detect.py:729-741). Here the records of a chunk that carry one label are compressed where they already are - in HBM - into complete
gzip members (BGZF framing), and only the compressed bytes travel to the host, which appends them to the file.
"""
import ctypes as C

import torch

from . import _native as N

MEMBER = 65280      # input bytes per gzip member (BGZF's block size)


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
