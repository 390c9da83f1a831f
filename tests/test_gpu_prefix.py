"""Prefix-state table (include/ribodetector_amd.h rd_set_prefix_table): the default kernel's and, since round 4,
the exact-fp32 kernel's own (each kernel's rows are its own state representation).

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


@pytest.mark.parametrize("variant", ["auto", "mfma_f32"])
@pytest.mark.parametrize("sem", ["packed", "padded"])
@pytest.mark.parametrize("max_len", [100, 37, 300, 13, 12])
def test_table_start_is_bit_identical(model, sem, max_len, variant):
    a, o, l = _batch(_edge_reads())
    model.set_variant(variant)
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
        model.set_variant("auto")


def test_table_start_is_bit_identical_at_scale_fp32(model):
    """the exact-fp32 kernel from ITS table (k = 12), 2^18 reads of fixed and of variable length"""
    from ribodetector_amd import synth
    n = 1 << 18
    a, off, lens = synth.reads_torch(n, 100, seed=78, device="cuda:0")
    o = off[:-1].contiguous()
    g = torch.Generator(device="cuda:0")
    g.manual_seed(4)
    vl = torch.randint(1, 101, (n,), generator=g, device="cuda:0", dtype=torch.int32)
    try:
        model.set_variant("mfma_f32")
        model.set_prefix_table(0)
        ref, refv = model.classify_bytes(a, o, lens, 100)[0].clone(), model.classify_bytes(a, o, vl, 100)[0].clone()
        model.set_prefix_table(12)
        assert model.prefix_k == 12
        assert torch.equal(model.classify_bytes(a, o, lens, 100)[0], ref)
        assert torch.equal(model.classify_bytes(a, o, vl, 100)[0], refv)
    finally:
        model.set_prefix_table(0)
        model.set_variant("auto")


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


def test_each_kernel_its_own_rows(model):
    """The rows are the state of the kernel that built them. SeqModel.set_variant rebuilds an attached table for the new kernel;
    at the C ABI a table built for another kernel is ignored (the reads run all their steps) rather than misread; the plain-FMA
    cross-check kernel has no table at all."""
    from ribodetector_amd import _native as N
    lib = N.lib()
    a, o, l = _batch(_edge_reads(9)[:400])
    try:
        refs = {}
        for v in ("auto", "mfma_f32", "simple"):
            model.set_variant(v)
            model.set_prefix_table(0)
            refs[v] = model.classify_bytes(a, o, l, 100)[0].clone()
        model.set_variant("auto")
        model.set_prefix_table(6)
        tab = model._ptab
        assert model._ptab_variant == "mfma_f16x3_t32"
        model.set_variant("mfma_f32")                                       # rebuilt in place, for the fp32 kernel
        assert model.prefix_k == 6 and model._ptab_variant == "mfma_f32" and model._ptab is tab
        assert torch.equal(model.classify_bytes(a, o, l, 100)[0], refs["mfma_f32"])
        # C ABI only: switch the kernel under the table - the fp32 rows must NOT be read as the default kernel's state
        N.check(lib.rd_set_variant(model._handle, N.VARIANTS["auto"]), "rd_set_variant")
        ws = model._workspace(len(l), 100)
        lg = torch.empty((len(l), 2), dtype=torch.float32, device="cuda:0")
        N.check(lib.rd_classify(model._handle, N.ptr(a), N.ptr(o), N.ptr(l), len(l), 100, N.ptr(lg), None, N.ptr(ws), ws.numel(),
                                N.stream_ptr(model.device)), "rd_classify")
        assert torch.equal(lg, refs["auto"])
        model.set_variant("simple")                                         # no table for this kernel: detached, same results
        model.set_prefix_table(6)
        assert model.prefix_k == 0
        assert torch.equal(model.classify_bytes(a, o, l, 100)[0], refs["simple"])
        st = N.stream_ptr(model.device)
        buf = torch.empty(int(lib.rd_prefix_table_bytes(4)), dtype=torch.uint8, device="cuda:0")
        scr = torch.empty(int(lib.rd_prefix_scratch_bytes(4)), dtype=torch.uint8, device="cuda:0")
        assert lib.rd_set_prefix_table(model._handle, 4, N.ptr(buf), buf.numel(), N.ptr(scr), scr.numel(), st) == -3   # RD_E_UNSUPPORTED
        assert b"no prefix-state table" in lib.rd_last_error()
    finally:
        model.set_variant("auto")
        model.set_prefix_table(0)


def test_no_table_unless_asked_and_small_memory_falls_back(caplog):
    """SeqModel.to('cuda') allocates nothing by itself (the reference's .to(device) has no side allocations, detect.py:93,115-119);
    'auto' sizes table + scratch against a quarter of the free memory; when the allocation fails anyway the next smaller k is
    tried; one log line says what was built."""
    import logging
    from ribodetector_amd.model import model as module_arch
    from ribodetector_amd.parse_config import ConfigParser
    cfg = ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json"))
    m = cfg.init_obj("arch", module_arch)
    m.load_state_dict(cfg.load_state_dict("mcc"))
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    before = torch.cuda.mem_get_info()[0]
    m.to("cuda:0").eval()
    assert m.prefix_k == 0 and m._ptab is None and before - torch.cuda.mem_get_info()[0] < (64 << 20)   # weights + tables: ~2 MB
    a, o, l = _batch(_edge_reads(3)[:300])
    ref = m.classify_bytes(a, o, l, 100)[0].clone()
    free, total = torch.cuda.mem_get_info()
    hog = []
    try:
        # leave less than 8 GiB free: 'auto' must come out below 12 (16 GiB), and still work
        left = 6 << 30
        while free - left > (1 << 30):
            chunk = min(free - left, 64 << 30)
            hog.append(torch.empty(chunk, dtype=torch.uint8, device="cuda:0"))
            free, _ = torch.cuda.mem_get_info()
        with caplog.at_level(logging.INFO, logger="ribodetector_amd"):
            m.set_prefix_table("auto")
        k_auto = m.prefix_k
        assert 4 <= k_auto < 12 and (4 ** k_auto + 1) * 1024 * 1.25 <= (8 << 30) / 4 + 1024
        assert any("prefix-state table: k = %d" % k_auto in r.getMessage() for r in caplog.records)
        assert torch.equal(m.classify_bytes(a, o, l, 100)[0], ref)
        # an explicit k that cannot be allocated falls back instead of raising
        m.set_prefix_table(0)
        caplog.clear()
        with caplog.at_level(logging.INFO, logger="ribodetector_amd"):
            m.set_prefix_table(12)
        assert 0 <= m.prefix_k < 12
        assert any("did not fit" in r.getMessage() or "not enough device memory" in r.getMessage() for r in caplog.records)
        assert torch.equal(m.classify_bytes(a, o, l, 100)[0], ref)
    finally:
        m.set_prefix_table(0)
        del hog
        torch.cuda.empty_cache()


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
