"""Prefix-state table of the default kernel (include/ribodetector_amd.h rd_set_prefix_table).

The table is built by the classifying kernel itself, so starting a read from its row must give the SAME BITS as stepping over
the bases: every test here compares logits with torch.equal against the model without a table, whose parity with the
reference's golden vectors and the oracle is what tests/test_gpu_parity.py establishes (reference model/model.py:32-37)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def model():
    from ribodetector_amd.model import model as module_arch
    from ribodetector_amd.parse_config import ConfigParser
    cfg = ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json"))
    m = cfg.init_obj("arch", module_arch)
    m.load_state_dict(cfg.load_state_dict("mcc"))
    m.set_prefix_table(0)
    return m.to("cuda:0").eval()


def _batch(reads):
    arena = np.frombuffer(b"".join(reads), dtype=np.uint8).copy() if reads else np.zeros(0, np.uint8)
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    off = np.zeros(len(reads) + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    dev = "cuda:0"
    a = torch.from_numpy(arena).to(dev) if len(arena) else torch.zeros(1, dtype=torch.uint8, device=dev)
    return a, torch.from_numpy(off[:-1].copy()).to(dev), torch.from_numpy(lens).to(dev)


def _edge_reads(seed=5):
    rng = np.random.default_rng(seed)
    reads = []
    for L in list(range(0, 40)) + [63, 64, 65, 99, 100, 101, 127, 128, 129, 150, 191, 192, 193, 210, 300]:
        for _ in range(6):
            reads.append(bytes(rng.choice(list(b"ACGT"), L).astype(np.uint8)))
    base = bytes(rng.choice(list(b"ACGT"), 120).astype(np.uint8))
    for pos in list(range(0, 16)) + [50, 99, 119]:                       # one foreign letter at every prefix position
        for ch in b"NacgtURY-":
            r = bytearray(base)
            r[pos] = ch
            reads.append(bytes(r))
    reads += [b"N" * 100, b"A" * 100, b"T" * 13, b"T" * 12, b"G" * 5, b"ACGTACGTACGTA" + b"N" * 87, b"ACGTACGTACGTAC" + b"N" * 86,
              b"ACGTUUUUACGTACGTAAAA" * 5, b"acgt" * 25]
    for _ in range(200):                                                 # trailing non-ACGT runs (padded semantics: pos moves)
        L = int(rng.integers(1, 140))
        r = bytearray(rng.choice(list(b"ACGT"), L).astype(np.uint8))
        k = int(rng.integers(0, L + 1))
        r[L - k:] = b"N" * k
        reads.append(bytes(r))
    return reads


@pytest.mark.parametrize("sem", ["packed", "padded"])
@pytest.mark.parametrize("max_len", [100, 37, 300, 13, 12])
def test_table_start_is_bit_identical(model, sem, max_len):
    a, o, l = _batch(_edge_reads())
    model.set_semantics(sem)
    try:
        model.set_prefix_table(0)
        assert model.prefix_k == 0
        ref_logits, ref_labels = model.classify_bytes(a, o, l, max_len)
        ref_logits, ref_labels = ref_logits.clone(), ref_labels.clone()
        for k in (4, 7, 12):
            model.set_prefix_table(k)
            assert model.prefix_k == k
            lg, lb = model.classify_bytes(a, o, l, max_len)
            assert torch.equal(lg, ref_logits), (sem, max_len, k, float((lg - ref_logits).abs().max()))
            assert torch.equal(lb, ref_labels)
    finally:
        model.set_semantics("packed")
        model.set_prefix_table(0)


def test_table_start_is_bit_identical_at_scale(model, report):
    from ribodetector_amd import synth
    n = 1 << 20
    a, off, lens = synth.reads_torch(n, 100, seed=77, device="cuda:0")
    o = off[:-1].contiguous()
    model.set_prefix_table(0)
    ref = model.classify_bytes(a, o, lens, 100)[0].clone()
    g = torch.Generator(device="cuda:0")
    g.manual_seed(3)
    vl = torch.randint(40, 101, (n,), generator=g, device="cuda:0", dtype=torch.int32)   # variable length inside the 100-byte rows
    refv = model.classify_bytes(a, o, vl, 100)[0].clone()
    try:
        for k in (12, 13):
            model.set_prefix_table(k)
            assert torch.equal(model.classify_bytes(a, o, lens, 100)[0], ref)
            assert torch.equal(model.classify_bytes(a, o, vl, 100)[0], refv)
        # how many reads of the synthetic stream can use a row (0.1 % N per base: ~1.2 % cannot at k = 12)
        codes = torch.zeros(256, dtype=torch.bool, device="cuda:0")
        codes[[65, 67, 71, 84, 85]] = True
        elig = codes[a.view(n, 100)[:, :12].long()].all(1).float().mean().item()
        report["prefix_table"] = {"eligible_frac_k12_synth": elig}
        assert elig > 0.97
    finally:
        model.set_prefix_table(0)


def test_other_kernels_ignore_the_table(model):
    a, o, l = _batch(_edge_reads(9)[:400])
    try:
        for v in ("mfma_f32", "simple"):
            model.set_variant(v)
            model.set_prefix_table(0)
            ref = model.classify_bytes(a, o, l, 100)[0].clone()
            model.set_prefix_table(6)
            assert torch.equal(model.classify_bytes(a, o, l, 100)[0], ref)
    finally:
        model.set_variant("auto")
        model.set_prefix_table(0)


def test_auto_and_argument_checks(model):
    from ribodetector_amd import _native as N
    lib = N.lib()
    assert lib.rd_prefix_table_bytes(3) == 0 and lib.rd_prefix_table_bytes(14) == 0 and lib.rd_prefix_table_bytes(0) == 0
    assert lib.rd_prefix_table_bytes(4) == (4 ** 4 + 1) * 1024 and lib.rd_prefix_table_bytes(13) == (4 ** 13 + 1) * 1024
    assert lib.rd_prefix_scratch_bytes(4) == 4 ** 3 * 1024 and lib.rd_prefix_scratch_bytes(3) == 0
    buf = torch.empty((4 ** 4 + 1) * 1024 + 256, dtype=torch.uint8, device="cuda:0")
    scr = torch.empty(4 ** 3 * 1024, dtype=torch.uint8, device="cuda:0")
    st = N.stream_ptr(model.device)
    assert lib.rd_set_prefix_table(model._handle, 3, N.ptr(buf), buf.numel(), N.ptr(scr), scr.numel(), st) != 0 and b"out of range" in lib.rd_last_error()
    assert lib.rd_set_prefix_table(model._handle, 4, N.ptr(buf), 1024, N.ptr(scr), scr.numel(), st) != 0 and b"table too small" in lib.rd_last_error()
    assert lib.rd_set_prefix_table(model._handle, 4, N.ptr(buf), buf.numel(), N.ptr(scr), 1024, st) != 0 and b"scratch too small" in lib.rd_last_error()
    assert lib.rd_set_prefix_table(model._handle, 4, C.c_void_p(buf.data_ptr() + 8), buf.numel() - 8, N.ptr(scr), scr.numel(), st) != 0 and b"aligned" in lib.rd_last_error()
    assert lib.rd_set_prefix_table(model._handle, 4, None, 0, None, 0, st) != 0
    assert model.prefix_k == 0                                           # a refused call leaves the model as it was
    with pytest.raises(RuntimeError):
        model.set_prefix_table(3)
    model.set_prefix_table("auto")
    try:
        free, total = torch.cuda.mem_get_info(model.device)
        if total > 200e9:                                                # an MI355X with nothing else on it: 16 GiB is < 1/4 of free
            assert model.prefix_k == 12
    finally:
        model.set_prefix_table(0)
