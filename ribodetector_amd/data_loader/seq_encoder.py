"""Device-side nucleotide encoder - mirror of the reference's `ribodetector.data_loader.seq_encoder`
(seq_encoder.py:11-18,126-145) and of the collate functions that call it (detect.py:666-726).

The reference builds Python lists of 4-tuples per read on the host (its real bottleneck); here the
raw ASCII bytes go to the GPU and the HIP encoder kernels produce the same tensors there:
    encode_read(read)                     -> FloatTensor[len,4]          (torch.FloatTensor(encode_read(...)))
    encode_variable_len_read(read, L)     -> FloatTensor[L,4] zero padded (CPU-product layout)
    encode_codes(batch)                   -> uint8 codes [n,stride] (0 A,1 C,2 G,3 T/U,4 other)
    pack_reads(batch, max_len)            -> PackedSequence identical to pack_sequence(enforce_sorted=False)
All functions raise if the HIP extension is missing (no host fallback).
"""
from collections import namedtuple

import numpy as np
import torch
from torch.nn.utils.rnn import PackedSequence

from .. import _native as N

# kept for API familiarity (reference seq_encoder.py:11-18); the kernels implement exactly this table.
BASE_DICT = {"A": (1, 0, 0, 0), "C": (0, 1, 0, 0), "G": (0, 0, 1, 0), "T": (0, 0, 0, 1), "U": (0, 0, 0, 1)}
ZERO_LIST = (0, 0, 0, 0)

ReadBatch = namedtuple("ReadBatch", "arena offsets lens")   # uint8[*], int64[n], int32[n] (device tensors)


def batch_from_strings(seqs, device="cuda"):
    """Host helper: list of str/bytes -> ReadBatch on `device` (arena = concatenated bytes)."""
    bs = [s.encode("latin-1") if isinstance(s, str) else bytes(s) for s in seqs]
    lens = np.array([len(b) for b in bs], dtype=np.int32)
    offs = np.zeros(len(bs), dtype=np.int64)
    if len(bs) > 1:
        np.cumsum(lens[:-1], out=offs[1:])
    arena = np.frombuffer(b"".join(bs) or b"\0", dtype=np.uint8).copy()
    return ReadBatch(torch.from_numpy(arena).to(device), torch.from_numpy(offs).to(device), torch.from_numpy(lens).to(device))


def batch_from_numpy(arena, offsets, lens, device="cuda"):
    arena = np.array(arena, dtype=np.uint8)      # private writable copy
    if arena.size == 0:
        arena = np.zeros(1, dtype=np.uint8)
    n = len(lens)
    return ReadBatch(torch.from_numpy(arena).to(device),
                     torch.from_numpy(np.array(offsets[:n], dtype=np.int64)).to(device),
                     torch.from_numpy(np.array(lens, dtype=np.int32)).to(device))


def encode_codes(batch, max_len, stride=None):
    n = int(batch.lens.numel())
    stride = int(stride or max_len)
    out = torch.empty((n, stride), dtype=torch.uint8, device=batch.arena.device)
    N.check(N.lib().rd_encode_codes(N.ptr(batch.arena), N.ptr(batch.offsets), N.ptr(batch.lens), n, int(max_len), stride,
                                    N.ptr(out), N.stream_ptr(batch.arena.device)), "rd_encode_codes")
    return out


def encode_padded(batch, max_len):
    """[n, max_len, 4] fp32, zero rows past the read (np.array([encode_variable_len_read(...)]), detect_cpu.py:699-700)"""
    n = int(batch.lens.numel())
    out = torch.empty((n, int(max_len), 4), dtype=torch.float32, device=batch.arena.device)
    N.check(N.lib().rd_encode_onehot_padded(N.ptr(batch.arena), N.ptr(batch.offsets), N.ptr(batch.lens), n, int(max_len),
                                            N.ptr(out), N.stream_ptr(batch.arena.device)), "rd_encode_onehot_padded")
    return out


def encode_read(read, device="cuda"):
    """FloatTensor[len,4] of one read (reference: torch.FloatTensor(SeqEncoder.encode_read(read)))."""
    if len(read) == 0:
        return torch.zeros((0, 4), dtype=torch.float32, device=device)
    return encode_padded(batch_from_strings([read], device), len(read))[0]


def encode_variable_len_read(read, max_len=100, device="cuda"):
    return encode_padded(batch_from_strings([read], device), max_len)[0]


def pack_reads(batch, max_len):
    """PackedSequence of the truncated one-hot reads, equal to the reference collate's
    pack_sequence([FloatTensor(encode_read(r[:max_len])) ...], enforce_sorted=False) (detect.py:681-685).
    (torch leaves the order of equal-length reads unspecified; here ties keep input order.)"""
    n = int(batch.lens.numel())
    dev = batch.arena.device
    L = N.lib()
    ws = torch.empty(int(L.rd_classify_workspace_bytes(n, int(max_len))), dtype=torch.uint8, device=dev)
    sorted_idx = torch.empty(n, dtype=torch.int64, device=dev)
    unsorted_idx = torch.empty(n, dtype=torch.int64, device=dev)
    batch_sizes = torch.empty(int(max_len), dtype=torch.int64, device=dev)
    total = torch.empty(1, dtype=torch.int64, device=dev)
    st = N.stream_ptr(dev)
    N.check(L.rd_pack_plan(N.ptr(batch.lens), n, int(max_len), N.ptr(sorted_idx), N.ptr(unsorted_idx), N.ptr(batch_sizes),
                           N.ptr(total), N.ptr(ws), ws.numel(), st), "rd_pack_plan")
    tot = int(total.item())
    data = torch.empty((tot, 4), dtype=torch.float32, device=dev)
    if tot:
        N.check(L.rd_pack_onehot(N.ptr(batch.arena), N.ptr(batch.offsets), N.ptr(batch.lens), n, int(max_len), N.ptr(sorted_idx),
                                 N.ptr(batch_sizes), N.ptr(data), st), "rd_pack_onehot")
    bs = batch_sizes.cpu()
    bs = bs[: int((bs > 0).sum())]
    return PackedSequence(data, bs, sorted_idx, unsorted_idx)
