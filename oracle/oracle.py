"""ctypes front end of the CPU oracle (oracle/rd_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package (ribodetector_amd)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librd_oracle.so")
MODES = {"none": 0, "rrna": 1, "norrna": 2, "both": 3}
WEIGHT_KEYS = ["rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0",
               "rnn.weight_ih_l0_reverse", "rnn.weight_hh_l0_reverse", "rnn.bias_ih_l0_reverse",
               "rnn.bias_hh_l0_reverse", "out.weight", "out.bias"]


def build(force=False):
    src = os.path.join(_HERE, "rd_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "librd_oracle.so"])
    return _SO


class _W(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in
                ("w_ih", "w_hh", "b_ih", "b_hh", "w_ih_r", "w_hh_r", "b_ih_r", "b_hh_r", "w_out", "b_out")]


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Oracle:
    def __init__(self, state_dict):
        """state_dict: mapping of the 10 reference tensor names -> numpy fp32 arrays."""
        if not os.path.exists(_SO):
            build()
        self.lib = C.CDLL(_SO)
        self._keep = [np.ascontiguousarray(np.asarray(state_dict[k], dtype=np.float32)) for k in WEIGHT_KEYS]
        shapes = [(512, 4), (512, 128), (512,), (512,), (512, 4), (512, 128), (512,), (512,), (2, 256), (2,)]
        for a, s, k in zip(self._keep, shapes, WEIGHT_KEYS):
            assert a.shape == s, (k, a.shape)
        self.w = _W(*[_fp(a) for a in self._keep])

    @staticmethod
    def _args(arena, offsets, lens):
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        if arena.size == 0:
            arena = np.zeros(1, dtype=np.uint8)
        return arena, offsets, lens

    def encode_codes(self, seq_bytes):
        a = np.frombuffer(bytes(seq_bytes), dtype=np.uint8)
        out = np.empty(len(a), dtype=np.uint8)
        if len(a):
            self.lib.rdo_encode_codes(_p(a, C.c_uint8), C.c_int64(len(a)), _p(out, C.c_uint8))
        return out

    def encode_onehot(self, seq_bytes):
        a = np.frombuffer(bytes(seq_bytes), dtype=np.uint8)
        out = np.empty((len(a), 4), dtype=np.float32)
        if len(a):
            self.lib.rdo_encode_onehot(_p(a, C.c_uint8), C.c_int64(len(a)), _fp(out))
        return out

    def encode_padded(self, seq_bytes, max_len):
        a = np.frombuffer(bytes(seq_bytes) or b"\0", dtype=np.uint8)
        out = np.empty((max_len, 4), dtype=np.float32)
        self.lib.rdo_encode_padded(_p(a, C.c_uint8), C.c_int64(len(bytes(seq_bytes))), C.c_int(max_len), _fp(out))
        return out

    def encode_padded_batch(self, arena, offsets, max_len):
        """[n, max_len, 4] fp32 for the reads offsets[i]..offsets[i+1] (offsets has n+1 entries)"""
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        out = np.empty((n, max_len, 4), dtype=np.float32)
        if n > 0:
            self.lib.rdo_encode_padded_batch(_p(arena, C.c_uint8), _p(offsets, C.c_int64), C.c_int64(n), C.c_int(max_len), _fp(out))
        return out

    def pack_sequence(self, arena, offsets, lens, max_len):
        arena, offsets, lens = self._args(arena, offsets, lens)
        n = len(lens)
        T = np.minimum(lens, max_len)
        tot, tmax = int(T.sum()), int(T.max()) if n else 0
        data = np.zeros((tot, 4), dtype=np.float32)
        bs = np.zeros(tmax, dtype=np.int64)
        si = np.zeros(n, dtype=np.int64)
        ui = np.zeros(n, dtype=np.int64)
        self.lib.rdo_pack_sequence.restype = C.c_int64
        r = self.lib.rdo_pack_sequence(_p(arena, C.c_uint8), _p(offsets, C.c_int64), _p(lens, C.c_int32), C.c_int64(n),
                                       C.c_int(max_len), _fp(data), _p(bs, C.c_int64), _p(si, C.c_int64), _p(ui, C.c_int64))
        assert r == tot
        return data, bs, si, ui

    def sorted_last_indices(self, batch_sizes, n):
        bs = np.ascontiguousarray(batch_sizes, dtype=np.int64)
        out = np.zeros(n, dtype=np.int64)
        self.lib.rdo_sorted_last_indices(_p(bs, C.c_int64), C.c_int64(len(bs)), C.c_int64(n), _p(out, C.c_int64))
        return out

    def forward_packed(self, arena, offsets, lens, max_len):
        """reference GPU-path semantics (model.py forward1) -> logits [n,2] fp32"""
        arena, offsets, lens = self._args(arena, offsets, lens)
        out = np.zeros((len(lens), 2), dtype=np.float32)
        self.lib.rdo_forward_packed(C.byref(self.w), _p(arena, C.c_uint8), _p(offsets, C.c_int64), _p(lens, C.c_int32),
                                    C.c_int64(len(lens)), C.c_int(max_len), _fp(out))
        return out

    def forward_padded(self, arena, offsets, lens, max_len, batched=False, batch=1024, nthreads=0):
        """ribodetector_cpu semantics (model_cpu.py forward_last) -> logits [n,2] fp32"""
        arena, offsets, lens = self._args(arena, offsets, lens)
        out = np.zeros((len(lens), 2), dtype=np.float32)
        if batched:
            self.lib.rdo_forward_padded_batched(C.byref(self.w), _p(arena, C.c_uint8), _p(offsets, C.c_int64),
                                                _p(lens, C.c_int32), C.c_int64(len(lens)), C.c_int(max_len),
                                                C.c_int(batch), C.c_int(nthreads), _fp(out))
        else:
            self.lib.rdo_forward_padded(C.byref(self.w), _p(arena, C.c_uint8), _p(offsets, C.c_int64),
                                        _p(lens, C.c_int32), C.c_int64(len(lens)), C.c_int(max_len), _fp(out))
        return out

    def argmax(self, logits):
        lg = np.ascontiguousarray(logits, dtype=np.float32)
        out = np.zeros(len(lg), dtype=np.uint8)
        self.lib.rdo_argmax(_fp(lg), C.c_int64(len(lg)), _p(out, C.c_uint8))
        return out

    def pair_fuse(self, l1, l2, ensure):
        l1 = np.ascontiguousarray(l1, dtype=np.float32)
        l2 = np.ascontiguousarray(l2, dtype=np.float32)
        out = np.zeros(len(l1), dtype=np.int8)
        self.lib.rdo_pair_fuse(_fp(l1), _fp(l2), C.c_int64(len(l1)), C.c_int(MODES[ensure]), _p(out, C.c_int8))
        return out

    def count_labels(self, labels):
        lab = np.ascontiguousarray(labels, dtype=np.int8)
        cnt = (C.c_uint64 * 3)()
        self.lib.rdo_count_labels(_p(lab, C.c_int8), C.c_int64(len(lab)), cnt)
        return [int(cnt[0]), int(cnt[1]), int(cnt[2])]

    def num_threads(self):
        return int(self.lib.rdo_num_threads())


def load_default():
    """Oracle with the shipped weights (ribodetector_amd/data/*.safetensors)."""
    from safetensors.numpy import load_file
    p = os.path.join(_HERE, "..", "ribodetector_amd", "data", "ribodetector_600k_variable_len70_101_epoch47.safetensors")
    return Oracle(load_file(p))
