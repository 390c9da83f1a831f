"""SeqModel - drop-in for the reference's `ribodetector.model.model.SeqModel` (model/model.py:10-37) on MI355X.

Same constructor arguments (`config.json: arch.args`), same `load_state_dict` tensor names/shapes, same call:
`model.to('cuda'); model.eval(); logits = model(packed_sequence)` -> fp32 Tensor[B,2] in input order.
The arithmetic runs in hand-written HIP kernels behind the C ABI (include/ribodetector_amd.h); there is no
torch.nn.LSTM inside and no CPU fallback.

Entries:
  * `forward(x: PackedSequence)`  - API parity with the reference (the one-hot tensor is turned back into bases);
  * `forward(x: Tensor[B,L,4])` when built with pack_seq=false - the reference's forward2 (padded input);
  * `classify_bytes(arena, offsets, lens, max_len)` - the native fast entry: raw ASCII reads resident in HBM,
    encoder + recurrence + FC + argmax fused on the device.
"""
import contextlib
import ctypes as C
import logging
import os

import numpy as np
import torch
from torch.nn.utils.rnn import PackedSequence

from .. import _native as N

log = logging.getLogger("ribodetector_amd")


@contextlib.contextmanager
def _device_lock(device):
    """One process at a time sizes and allocates a prefix-state table on a device: ranks that share a GPU would otherwise all read
    the same "free" figure and allocate against it together. An flock on a file named after the device's UUID (advisory, released
    with the process)."""
    import fcntl
    import tempfile
    try:
        uid = str(torch.cuda.get_device_properties(device).uuid)
    except Exception:
        uid = "dev%d" % (device.index if device.index is not None else torch.cuda.current_device())
    path = os.path.join(tempfile.gettempdir(), "rd_prefix_%s.lock" % "".join(ch for ch in uid if ch.isalnum() or ch in "-_"))
    fd = os.open(path, os.O_CREAT | os.O_RDWR, 0o666)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        yield
    finally:
        try:
            fcntl.flock(fd, fcntl.LOCK_UN)
        finally:
            os.close(fd)


STATE_KEYS = ["rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0",
              "rnn.weight_ih_l0_reverse", "rnn.weight_hh_l0_reverse", "rnn.bias_ih_l0_reverse",
              "rnn.bias_hh_l0_reverse", "out.weight", "out.bias"]


class SeqModel:
    def __init__(self, input_size, hidden_size, num_layers, num_classes, batch_first=True, bidirectional=True,
                 pack_seq=True):
        # the kernels cover exactly the shipped architecture (reference config.json:4-15); anything else is an error,
        # never a silent fallback.
        if num_layers != 1:
            raise NotImplementedError("SeqModel: num_layers=%r is not covered by the HIP kernels (only 1)" % (num_layers,))
        if not bidirectional:
            raise NotImplementedError("SeqModel: bidirectional=False is not covered by the HIP kernels")
        if not pack_seq and not batch_first:
            raise NotImplementedError("SeqModel: pack_seq=False needs batch_first=True (the reference's forward2 indexes the "
                                      "batch on dim 0, model/model.py:67-72)")
        if input_size != 4 or hidden_size != 128 or num_classes != 2:
            raise NotImplementedError("SeqModel: kernels are built for input_size=4, hidden_size=128, num_classes=2")
        self.input_size, self.hidden_size, self.num_layers, self.num_classes = input_size, hidden_size, num_layers, num_classes
        self.batch_first, self.bidirectional, self.pack_seq = batch_first, bidirectional, pack_seq
        H4, H = 4 * hidden_size, hidden_size
        self._shapes = dict(zip(STATE_KEYS, [(H4, input_size), (H4, H), (H4,), (H4,), (H4, input_size), (H4, H), (H4,), (H4,),
                                             (num_classes, 2 * H), (num_classes,)]))
        self._state = None
        self._handle = None
        self.device = None
        self._variant = "auto"
        self._semantics = "packed"
        self._refine = None
        self._refine_async = 0
        # prefix-state table (set_prefix_table): none unless asked for - RD_PREFIX_K, config.json kernel.prefix_k (the CLI) or an
        # explicit call; the reference's .to(device) has no side allocations (reference detect.py:93,115-119) and neither has this one
        self._prefix = os.environ.get("RD_PREFIX_K", 0)
        self._prefix_cap = None
        self._ptab = None
        self._ptab_variant = None
        self._ws = None
        self.training = True

    # ---- weights ---------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        sd = {}
        missing = [k for k in STATE_KEYS if k not in state_dict]
        unexpected = [k for k in state_dict if k not in STATE_KEYS]
        if missing or (strict and unexpected):
            raise RuntimeError("Error(s) in loading state_dict for SeqModel: Missing key(s): %s. Unexpected key(s): %s."
                               % (missing, unexpected))
        for k in STATE_KEYS:
            v = state_dict[k]
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.shape != self._shapes[k]:
                raise RuntimeError("size mismatch for %s: got %s, expected %s" % (k, a.shape, self._shapes[k]))
            sd[k] = a
        self._state = sd
        if self._handle is not None:
            self._create()
        return self

    def state_dict(self):
        if self._state is None:
            raise RuntimeError("SeqModel: no weights loaded")
        return {k: torch.from_numpy(v.copy()) for k, v in self._state.items()}

    def _destroy(self):
        if self._handle is not None:
            N.lib().rd_model_destroy(self._handle)
            self._handle = None
        self._ptab = None
        self._ptab_variant = None

    def _create(self):
        self._destroy()
        if self._state is None:
            raise RuntimeError("SeqModel.to(): call load_state_dict() first")
        w = N.RdWeights(*[self._state[k].ctypes.data_as(C.POINTER(C.c_float)) for k in STATE_KEYS],
                        self.input_size, self.hidden_size, self.num_classes)
        h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        N.check(N.lib().rd_model_create(C.byref(w), idx, C.byref(h)), "rd_model_create")
        self._handle = h
        self.set_variant(self._variant)
        self.set_semantics(self._semantics)
        if self._refine is not None:
            self.set_refine(self._refine)
        self.set_prefix_table(self._prefix)
        if self._refine_async:
            self.set_refine_async(self._refine_async)

    def to(self, device, non_blocking=False):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("SeqModel runs on the GPU only (HIP kernels); got device %r. "
                               "The reference's CPU product is `ribodetector_cpu`." % (device,))
        if not torch.cuda.is_available():
            raise RuntimeError("No visible ROCm/HIP device")
        self.device = device
        self._create()
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else "cuda:%d" % device)

    def eval(self):
        self.training = False
        return self

    def set_variant(self, name):
        if name not in N.VARIANTS:
            raise RuntimeError("SeqModel.set_variant: unknown kernel variant %r (this build has: %s)" % (name, ", ".join(sorted(N.VARIANTS))))
        self._variant = name
        if self._handle is not None:
            N.check(N.lib().rd_set_variant(self._handle, N.VARIANTS[name]), "rd_set_variant")
            if self._ptab is not None and self._ptab_variant != self._kernel():
                self.set_prefix_table(self._prefix, self._prefix_cap)   # the rows are the state of the kernel that built them
        return self

    def _kernel(self):
        return "mfma_f16x3_t32" if self._variant == "auto" else self._variant

    def _built_k(self):
        """k of the table in self._ptab ((4^k + 1) KiB), whichever kernel's rows it holds"""
        return 0 if self._ptab is None else (int(self._ptab.numel()) // 1024 - 1).bit_length() // 2

    def set_semantics(self, name):
        """'packed' (default): the reference GPU product, forward1 over min(len, max_len) real timesteps.
        'padded': the reference CPU product `ribodetector_cpu` (model_cpu.forward_last): zero-padded input, gather at the
        last non-zero row. The two differ only for reads shorter than max_len or ending in non-ACGT bases."""
        self._semantics = name
        if self._handle is not None:
            N.check(N.lib().rd_set_semantics(self._handle, N.SEMANTICS[name]), "rd_set_semantics")
        return self

    def set_refine(self, thresh):
        """margin below which a read is re-evaluated in float64 (C ABI rd_set_refine; default 2.5e-4, 0 = off)"""
        self._refine = float(thresh)
        if self._handle is not None:
            N.check(N.lib().rd_set_refine(self._handle, C.c_float(self._refine)), "rd_set_refine")
        return self

    REFINE_DEFAULT = 2.5e-4

    def set_refine_async(self, calls=1):
        """C ABI rd_set_refine_async: classify_bytes only records the reads inside the noise band; the candidates of `calls`
        consecutive calls (1..16; True = 1; 0 / False = the inline pass) are re-evaluated in float64 together on a stream the model
        owns, beside the recurrence of the next call. A call's results are final once `calls` further calls have been issued or
        after sync_results() - see include/ribodetector_amd.h for the buffer contract. forward() (the reference-compatible call)
        always returns final logits."""
        k = 1 if calls is True else 0 if calls is False else int(calls)
        if self._handle is not None:
            self.sync_results()
            N.check(N.lib().rd_set_refine_async(self._handle, k), "rd_set_refine_async")
        self._refine_async = k
        return self

    def sync_results(self):
        """everything issued later on the current stream sees the final results of all earlier classify_bytes calls"""
        if self._handle is not None:
            with torch.cuda.device(self.device):
                N.check(N.lib().rd_sync_results(self._handle, N.stream_ptr(self.device)), "rd_sync_results")
        return self

    def set_prefix_table(self, k="auto", cap=None):
        """Prefix-state table (C ABI rd_set_prefix_table): the recurrence state after every possible sequence of k
        bases, (4^k + 1) KiB of HBM, built in milliseconds by the kernel that will use it (mfma_f16x3_t32 and mfma_f32 each have
        their own rows; `simple` has none); a read then starts k steps in, with bit-identical logits. Opt-in: a model builds no
        table unless this is called (or RD_PREFIX_K / config.json kernel.prefix_k is set; the CLI and bench.py do ask).
        k = 0: none; 4..13: that; "auto": 12 (16 GiB) when table + build scratch (a quarter more) fit a quarter of the free device
        memory, else the largest k that does, else none. `cap` bounds what "auto" picks (the CLI passes the k that pays off for the
        size of its input: building level k costs 4^k steps). If the allocation fails anyway (another process took the memory),
        the next smaller k is tried, down to none - never an exception for want of memory. Sizing and allocation run under a
        per-device lock, so that ranks sharing a GPU do not size against the same free bytes at once."""
        if isinstance(k, str) and k != "auto":
            k = int(k)
        self._prefix = k
        if cap is not None or k != "auto":
            self._prefix_cap = cap
        if self._handle is None:
            return self
        lib = N.lib()
        if self._kernel() not in ("mfma_f16x3_t32", "mfma_f32") and k != 0:
            log.info("prefix-state table: kernel %s has none (k = 0)", self._kernel())
            k = 0
        with torch.cuda.device(self.device), _device_lock(self.device):
            if k == "auto":
                free, _ = torch.cuda.mem_get_info(self.device)
                free += torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)   # cached blocks are reusable
                k = N.PREFIX_K_AUTO if self._prefix_cap is None else max(0, min(N.PREFIX_K_AUTO, int(self._prefix_cap)))
                while k >= N.PREFIX_K_MIN and int(lib.rd_prefix_table_bytes(k)) + int(lib.rd_prefix_scratch_bytes(k)) > free // 4:
                    k -= 1
                if k < N.PREFIX_K_MIN:
                    k = 0
            k = int(k)
            if k != 0 and not (N.PREFIX_K_MIN <= k <= N.PREFIX_K_MAX):
                raise RuntimeError("SeqModel.set_prefix_table: k must be 0, 'auto' or in [%d, %d]; got %r" % (N.PREFIX_K_MIN, N.PREFIX_K_MAX, k))
            if k == int(lib.rd_prefix_k(self._handle)) and (k == 0 or (self._ptab is not None and self._ptab_variant == self._kernel())):
                return self
            self.sync_results()
            asked = k
            while True:
                tab = scr = None
                if k:
                    try:
                        if self._ptab is not None and self._built_k() == k:
                            tab = self._ptab                        # same size, another kernel's rows: rebuilt in place
                        else:
                            self._ptab = None
                            N.check(lib.rd_set_prefix_table(self._handle, 0, None, 0, None, 0, N.stream_ptr(self.device)), "rd_set_prefix_table")
                            tab = torch.empty(int(lib.rd_prefix_table_bytes(k)), dtype=torch.uint8, device=self.device)
                        scr = torch.empty(int(lib.rd_prefix_scratch_bytes(k)), dtype=torch.uint8, device=self.device)
                    except torch.OutOfMemoryError:
                        tab = scr = None
                        torch.cuda.empty_cache()
                        k = k - 1 if k > N.PREFIX_K_MIN else 0
                        continue
                N.check(lib.rd_set_prefix_table(self._handle, k, N.ptr(tab), 0 if tab is None else tab.numel(), N.ptr(scr),
                                                0 if scr is None else scr.numel(), N.stream_ptr(self.device)), "rd_set_prefix_table")
                self._ptab = tab                         # (the scratch - the level below, a quarter of the table - is released here)
                self._ptab_variant = self._kernel() if k else None
                break
            if k:
                log.info("prefix-state table: k = %d, %d bytes of HBM (+ %d scratch during the build), kernel %s%s", k, tab.numel(),
                         scr.numel(), self._variant, "" if k == asked else " - k = %d did not fit" % asked)
            elif asked:
                log.warning("prefix-state table: not enough device memory for k = %d or any smaller table - none attached", asked)
        return self

    @property
    def prefix_k(self):
        return 0 if self._handle is None else int(N.lib().rd_prefix_k(self._handle))

    def refine(self, arena, offsets, lens, max_len, logits, labels=None, mate_logits=None, thresh=None):
        """float64 re-evaluation (C ABI rd_refine) of the reads whose own margin - or, with `mate_logits`, whose PAIR margin
        (logits + mate_logits, what decides the pair label under --ensure none, reference detect.py:657) - is inside the noise
        band; `logits` (and `labels`) are updated in place on the current stream. thresh None = the model's band (if that is
        0 because the inline pass is switched off: the library default)."""
        n = int(lens.numel())
        if thresh is None:
            thresh = self._refine if self._refine else (0.0 if self._refine is None else self.REFINE_DEFAULT)
        with torch.cuda.device(self.device):
            N.check(N.lib().rd_refine(self._handle, N.ptr(arena), N.ptr(offsets), N.ptr(lens), n, int(max_len), N.ptr(logits),
                                      N.ptr(labels), N.ptr(mate_logits), C.c_float(float(thresh)), N.stream_ptr(self.device)), "rd_refine")
        return logits

    def refine_pairs(self, arena, offsets, lens, max_len, logits, mate_logits, labels=None, thresh=None):
        """one mate of a pair batch: call once per mate, then pair_fuse()"""
        return self.refine(arena, offsets, lens, max_len, logits, labels, mate_logits, thresh)

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    # ---- native fast entry --------------------------------------------------------------------------
    def _workspace(self, n, max_len):
        need = int(N.lib().rd_classify_workspace_bytes(n, max_len))
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.device:
            self._ws = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=self.device)
        return self._ws

    def classify_bytes(self, arena, offsets, lens, max_len, want_labels=True, logits=None, labels=None):
        """arena uint8[*] (ASCII), offsets int64[n] (start of read i), lens int32[n]; all on self.device.
        Returns (logits fp32[n,2], labels uint8[n] | None), row i <-> read i. Asynchronous on the current stream."""
        if self._handle is None:
            raise RuntimeError("SeqModel: call .to('cuda') before inference")
        n = int(lens.numel())
        for t, dt, nm in ((arena, torch.uint8, "arena"), (offsets, torch.int64, "offsets"), (lens, torch.int32, "lens")):
            if t.dtype != dt or not t.is_cuda or not t.is_contiguous():
                raise TypeError("classify_bytes: %s must be a contiguous %s CUDA tensor" % (nm, dt))
        if offsets.numel() < n:
            raise ValueError("classify_bytes: offsets shorter than lens")
        if logits is None:
            logits = torch.empty((n, 2), dtype=torch.float32, device=self.device)
        if want_labels and labels is None:
            labels = torch.empty((n,), dtype=torch.uint8, device=self.device)
        ws = self._workspace(n, max_len)
        with torch.cuda.device(self.device):        # the kernels launch on the calling thread's current HIP device
            return self._classify_launch(arena, offsets, lens, n, max_len, logits, labels, want_labels, ws)

    def _classify_launch(self, arena, offsets, lens, n, max_len, logits, labels, want_labels, ws):
        N.check(N.lib().rd_classify(self._handle, N.ptr(arena), N.ptr(offsets), N.ptr(lens), n, int(max_len), N.ptr(logits),
                                    N.ptr(labels if want_labels else None), N.ptr(ws), ws.numel(), N.stream_ptr(self.device)),
                "rd_classify")
        return logits, (labels if want_labels else None)

    # ---- reference-compatible call --------------------------------------------------------------------
    def forward2(self, x):
        """pack_seq=false entry of the reference (model/model.py:40-50 forward2 + last_pad_out_items :67-72): x is a padded
        Tensor [B, L, 4] of one-hot / all-zero rows; the BiLSTM runs over all L rows and the output row is the last non-zero
        one (row L-1 for an all-zero read). That is the 'padded' semantics of the kernels (the reference's CPU product uses the
        same function), so the rows are turned back into bases and classified with RD_SEM_PADDED and max_len = L."""
        if not torch.is_tensor(x) or x.dim() != 3 or x.shape[2] != 4:
            raise TypeError("SeqModel.forward2 expects a padded Tensor [B, L, 4]; got %s" % (tuple(x.shape) if torch.is_tensor(x) else type(x).__name__))
        if self._handle is None:
            raise RuntimeError("SeqModel: call .to('cuda') before inference")
        dev = self.device
        data = x.to(dev, non_blocking=True)
        B, L = int(data.shape[0]), int(data.shape[1])
        if B == 0:
            return torch.empty((0, 2), dtype=torch.float32, device=dev)
        rs, mx = data.sum(2), data.max(2)
        ok = ((rs == 1) & (mx.values == 1)) | ((rs == 0) & (data.abs().sum(2) == 0))
        if not bool(ok.all()):
            raise ValueError("SeqModel.forward2: input rows must be one-hot (A,C,G,T) or all-zero")
        code = torch.where(rs == 1, mx.indices, torch.full_like(mx.indices, 4))
        arena = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=dev)[code].reshape(-1)
        offsets = torch.arange(B, dtype=torch.int64, device=dev) * L
        lens = torch.full((B,), L, dtype=torch.int32, device=dev)
        prev = self._semantics
        self.set_semantics("padded")
        try:
            logits, _ = self.classify_bytes(arena, offsets, lens, L, want_labels=False)
            self.sync_results()
        finally:
            self.set_semantics(prev)
        return logits

    def forward(self, x):
        """pack_seq=true (config.json): x is a PackedSequence of one-hot fp32 rows [sum T, 4] (what the reference collate
        builds, detect.py:681-685); returns logits fp32[B,2] in the original (unsorted) batch order, like forward1 +
        last_items(unsort=True). pack_seq=false: x is a padded Tensor -> forward2."""
        if not self.pack_seq:
            return self.forward2(x)
        if not isinstance(x, PackedSequence):
            raise TypeError("SeqModel.forward expects a PackedSequence (config pack_seq=true; build the model with pack_seq=false "
                            "for padded Tensor input, the reference's forward2); got %s" % type(x).__name__)
        if self._handle is None:
            raise RuntimeError("SeqModel: call .to('cuda') before inference")
        dev = self.device
        data = x.data.to(dev, non_blocking=True)
        if data.dim() != 2 or data.shape[1] != 4:
            raise ValueError("SeqModel.forward: PackedSequence.data must be [sum T, 4]")
        bs = x.batch_sizes.to(torch.int64).cpu()
        tmax, n, total = int(bs.numel()), int(bs[0]), int(data.shape[0])
        # rows must be one-hot or all-zero: that is the only input the reference encoder can produce
        rs, mx = data.sum(1), data.max(1)
        ok = ((rs == 1) & (mx.values == 1)) | ((rs == 0) & (data.abs().sum(1) == 0))
        if not bool(ok.all()):
            raise ValueError("SeqModel.forward: input rows must be one-hot (A,C,G,T) or all-zero")
        code = torch.where(rs == 1, mx.indices, torch.full_like(mx.indices, 4))
        ascii_ = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=dev)[code]
        bs_d = bs.to(dev)
        t_of_row = torch.repeat_interleave(torch.arange(tmax, device=dev), bs_d, output_size=total)
        cum = torch.cumsum(bs_d, 0) - bs_d
        j_of_row = torch.arange(total, device=dev) - cum[t_of_row]
        sorted_idx = x.sorted_indices.to(dev) if x.sorted_indices is not None else torch.arange(n, device=dev)
        orig = sorted_idx[j_of_row]
        arena = torch.full((n, tmax), ord("N"), dtype=torch.uint8, device=dev)
        arena[orig, t_of_row] = ascii_
        len_sorted = (bs_d[None, :] > torch.arange(n, device=dev)[:, None]).sum(1)
        lens = torch.empty(n, dtype=torch.int32, device=dev)
        lens[sorted_idx] = len_sorted.to(torch.int32)
        offsets = torch.arange(n, dtype=torch.int64, device=dev) * tmax
        logits, _ = self.classify_bytes(arena.reshape(-1), offsets, lens, tmax, want_labels=False)
        self.sync_results()
        return logits

    __call__ = forward

    # ---- profiling hooks for bench.py ---------------------------------------------------------------
    def profile_enable(self, on=True):
        N.check(N.lib().rd_profile_enable(self._handle, 1 if on else 0), "rd_profile_enable")

    def profile_read(self):
        n, ms = C.c_int64(0), C.c_double(0)
        N.check(N.lib().rd_profile_read(self._handle, C.byref(n), C.byref(ms)), "rd_profile_read")
        return int(n.value), float(ms.value)

    def __str__(self):
        return "SeqModel(BiLSTM 4->128 x2, Linear 256->2) [HIP/gfx950]\nTrainable parameters: 137730"


def pair_fuse(logits1, logits2, ensure, counts=None):
    """Pair label logic of reference detect.py:616-663 on the device. Returns int8[n] in {0,1,-1};
    `counts` (uint64... stored as int64[3] tensor) is added to when given."""
    n = int(logits1.shape[0])
    out = torch.empty((n,), dtype=torch.int8, device=logits1.device)
    with torch.cuda.device(logits1.device):
        return _pair_fuse_launch(logits1, logits2, n, ensure, out, counts)


def _pair_fuse_launch(logits1, logits2, n, ensure, out, counts):
    N.check(N.lib().rd_pair_fuse(N.ptr(logits1), N.ptr(logits2), n, N.ENSURE_MODES[ensure], N.ptr(out), N.ptr(counts),
                                 N.stream_ptr(logits1.device)), "rd_pair_fuse")
    return out


def count_labels(labels, counts):
    """counts int64[3] device tensor (non-rRNA, rRNA, unclassified), added to (reference detect.py:485-486)."""
    with torch.cuda.device(labels.device):
        N.check(N.lib().rd_count_labels(N.ptr(labels), int(labels.numel()), N.ptr(counts), N.stream_ptr(labels.device)),
                "rd_count_labels")
    return counts
