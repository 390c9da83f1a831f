"""Parity of the HIP path (through the C ABI) with the reference's golden outputs and with the CPU oracle.

Tolerances: logits 1e-4 absolute (BASELINE.json north_star: "per-read logits within 1e-4 fp32"); labels identical,
except that a label may differ only where the oracle's own margin |logit1-logit0| < 2e-4 (SURVEY §7 "label identity");
integer / byte outputs (codes, one-hot, indices, pair labels from given logits, counts) are bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4
M_REFINE = 2.5e-4          # the library default band of the float64 refine pass (RD_REFINE_DEFAULT)
VARIANTS = ["mfma_f32", "simple", "mfma_f16x3_t32"]


def _run(model, arena, offsets, lens, max_len):
    from ribodetector_amd.data_loader import seq_encoder as E
    b = E.batch_from_numpy(arena, offsets, lens, model.device)
    lg, lab = model.classify_bytes(b.arena, b.offsets, b.lens, max_len)
    torch.cuda.synchronize()
    return lg.cpu().numpy(), lab.cpu().numpy()


def _check(lg, lab, ref_logits, what):
    err = np.abs(lg - ref_logits).max()
    assert err < TOL, "%s: max logit error %.3g" % (what, err)
    ref_lab = (ref_logits[:, 1] > ref_logits[:, 0]).astype(np.uint8)
    bad = np.flatnonzero(lab != ref_lab)
    margin = np.abs(ref_logits[:, 1] - ref_logits[:, 0])
    assert (margin[bad] < 2e-4).all(), "%s: label mismatch at margins %s" % (what, margin[bad])
    assert ((lg[:, 1] > lg[:, 0]).astype(np.uint8) == lab).all(), what   # labels consistent with own logits
    return err


@pytest.mark.parametrize("variant", VARIANTS)
def test_kat(gpu_model, golden, variant):
    gpu_model.set_variant(variant)
    kat = golden.json("kat")
    reads = [c["read"].encode() for c in kat["cases"].values()]
    ref = np.array([c["logits"] for c in kat["cases"].values()], dtype=np.float32)
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    lg, lab = _run(gpu_model, np.frombuffer(b"".join(reads), dtype=np.uint8), off, lens, kat["max_len"])
    _check(lg, lab, ref, "kat/" + variant)
    assert [int(x) for x in lab] == [c["label"] for c in kat["cases"].values()]
    gpu_model.set_variant("auto")


@pytest.mark.parametrize("variant", VARIANTS)
def test_golden_se_edge_varlen(gpu_model, golden, variant, report):
    gpu_model.set_variant(variant)
    d = golden.npz("se100")
    lg, lab = _run(gpu_model, d["arena"], d["offsets"], d["lens"], 100)
    e1 = _check(lg, lab, d["logits"], "se100/" + variant)
    assert (lab == d["labels"]).all()
    d = golden.npz("edge")
    lg, lab = _run(gpu_model, d["arena"], d["offsets"], d["lens"], 100)
    e2 = _check(lg, lab, d["logits"], "edge/" + variant)
    assert (lab == d["labels"]).all()
    d = golden.npz("varlen")
    e3 = 0
    for L in (300, 170):
        lg, lab = _run(gpu_model, d["arena"], d["offsets"], d["lens"], L)
        e3 = max(e3, _check(lg, lab, d["logits_l%d" % L], "varlen%d/%s" % (L, variant)))
    print("max logit error vs reference golden [%s]: se100 %.3g edge %.3g varlen %.3g" % (variant, e1, e2, e3))
    report["golden_max_logit_err/" + variant] = {"se100": float(e1), "edge": float(e2), "varlen": float(e3)}
    gpu_model.set_variant("auto")


def test_vs_oracle_ragged_and_empty(gpu_model, oracle):
    """seeded inputs against the oracle: ragged lengths incl. 0 and 1, reads longer than max_len, tile-boundary counts"""
    from ribodetector_amd import synth
    for n, length, L, seed in ((1, 100, 100, 1), (63, (1, 130), 100, 2), (64, 100, 100, 3), (65, (0, 40), 50, 4),
                               (257, (90, 110), 100, 5), (1000, (0, 260), 128, 6), (300, (120, 140), 129, 7)):
        arena, off, lens = synth.reads_numpy(n, length, seed)
        ref = oracle.forward_packed(arena, off, lens, L)
        lg, lab = _run(gpu_model, arena, off, lens, L)
        _check(lg, lab, ref, "oracle n=%d" % n)


def test_every_length_through_the_code_staging(gpu_model, oracle):
    """the default kernel stages the bases in 16-byte pieces of 64-step chunks; the piece that holds a read's end is loaded as the 16
    bytes that END there and shifted (reads shorter than 16 bases take a byte path): every length 0..210, three reads each, at -l
    values on, below and above the chunk size, the last reads of the arena flush with its end"""
    rng = np.random.default_rng(7)
    lens = np.repeat(np.arange(0, 211, dtype=np.int32), 3)
    rng.shuffle(lens)
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    arena = np.frombuffer(b"ACGTUNacgt", dtype=np.uint8)[rng.choice(10, int(off[-1]), p=[.22, .22, .22, .2, .04, .04, .015, .015, .015, .015])]
    for L in (15, 16, 17, 63, 64, 65, 100, 150, 200):
        ref = oracle.forward_packed(arena, off, lens, L)
        lg, lab = _run(gpu_model, arena, off, lens, L)
        _check_tail(lg, lab, ref, L, "all lengths, -l %d" % L)


def test_long_reads_chunked_codes(gpu_model, oracle):
    """max_len beyond one staged code chunk (TC=128) exercises the double-buffered code staging"""
    from ribodetector_amd import synth
    arena, off, lens = synth.reads_numpy(96, (380, 700), seed=9)
    ref = oracle.forward_packed(arena, off, lens, 600)
    lg, lab = _run(gpu_model, arena, off, lens, 600)
    _check(lg, lab, ref, "long")


def test_reference_call_packed_sequence(gpu_model, golden):
    """model(PackedSequence) - the reference's own call signature (detect.py:185-188) incl. un-permutation"""
    from torch.nn.utils.rnn import pack_sequence
    d = golden.npz("varlen")
    idx = np.arange(0, 512)
    lut = np.zeros((256, 4), dtype=np.float32)
    for k, ch in enumerate(b"ACGT"):
        lut[ch, k] = 1
    lut[ord("U"), 3] = 1
    seqs = [torch.from_numpy(lut[d["arena"][d["offsets"][i]:d["offsets"][i + 1]][:170]]) for i in idx]
    x = pack_sequence(seqs, enforce_sorted=False)
    out = gpu_model(x.to("cuda"))
    assert out.shape == (512, 2) and out.dtype == torch.float32
    assert np.abs(out.cpu().numpy() - d["logits_l170"][idx]).max() < TOL
    with pytest.raises(ValueError):
        bad = x.data.clone()
        bad[3, 1] = 0.5
        gpu_model(torch.nn.utils.rnn.PackedSequence(bad, x.batch_sizes, x.sorted_indices, x.unsorted_indices))


def test_reference_call_padded_tensor_forward2(golden, oracle):
    """SeqModel(pack_seq=False)(padded Tensor [B, L, 4]) - the reference's forward2 (model/model.py:40-50): golden logits come
    from the reference's own forward2 on the same padded one-hot tensors (tests/golden/make_golden.py F7)."""
    import os
    from ribodetector_amd.data_loader import seq_encoder as E
    from ribodetector_amd.model import model as module_arch
    from ribodetector_amd.parse_config import ConfigParser
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cfg = ConfigParser.from_json(os.path.join(root, "ribodetector_amd", "config.json"))
    m2 = module_arch.SeqModel(**dict(cfg["arch"]["args"], pack_seq=False))
    m2.load_state_dict(cfg.load_state_dict("mcc"))
    m2.to("cuda:0").eval()
    d = golden.npz("forward2")
    b = E.batch_from_numpy(d["arena"], d["offsets"][:-1], d["lens"], "cuda")
    for L in (100, 64):
        x = E.encode_padded(b, L)                                  # [B, L, 4] one-hot, zero rows past the read
        out = m2(x)
        assert out.shape == (len(d["lens"]), 2) and out.dtype == torch.float32
        err = np.abs(out.cpu().numpy() - d["logits_l%d" % L]).max()
        assert err < TOL, (L, err)
        assert m2._semantics == "packed"                           # the per-call switch to the padded semantics is undone
    with pytest.raises(TypeError):
        m2(torch.zeros((4, 10), device="cuda"))
    with pytest.raises(ValueError):
        bad = E.encode_padded(b, 64)
        bad[0, 0, :] = 0.5
        m2(bad)
    from torch.nn.utils.rnn import pack_sequence
    with pytest.raises(TypeError):                                 # a pack_seq=true model refuses a plain Tensor, like forward1 does
        cfg.init_obj("arch", module_arch).load_state_dict(cfg.load_state_dict("mcc")).to("cuda:0")(x)


def test_encoders_bit_exact(gpu_model, golden, oracle):
    from ribodetector_amd.data_loader import seq_encoder as E
    d = golden.npz("collate")
    b = E.batch_from_numpy(d["arena"], d["offsets"], d["lens"], "cuda")
    L = int(d["max_len"])
    assert (E.encode_padded(b, L).cpu().numpy() == d["padded"]).all()
    p = E.pack_reads(b, L)
    assert (p.data.cpu().numpy() == d["data"]).all()
    assert (p.batch_sizes.numpy() == d["batch_sizes"]).all()
    T = np.minimum(d["lens"], L)
    si, ui = p.sorted_indices.cpu().numpy(), p.unsorted_indices.cpu().numpy()
    assert (T[si] == np.sort(T)[::-1]).all() and (si[ui] == np.arange(len(T))).all()
    # larger ragged batch against the oracle
    e = golden.npz("edge")
    b = E.batch_from_numpy(e["arena"], e["offsets"], e["lens"], "cuda")
    for max_len, stride in ((100, 112), (100, 100), (100, 101), (37, 40), (300, 304), (16, 16), (15, 18)):
        codes = E.encode_codes(b, max_len, stride=stride).cpu().numpy()       # every piece path: inside / read end / padding / short reads
        for i in range(len(e["lens"])):
            s = bytes(e["arena"][e["offsets"][i]:e["offsets"][i + 1]])[:max_len]
            want = np.full(stride, 4, dtype=np.uint8)
            want[:len(s)] = oracle.encode_codes(s)
            assert (codes[i] == want).all(), (max_len, stride, i)
    from ribodetector_amd import synth
    a3, o3, l3 = synth.reads_numpy(3000, (0, 140), seed=77, n_rate=0.05)       # ragged batch, odd block count, unaligned views
    b3 = E.batch_from_numpy(a3, o3[:-1], l3, "cuda")
    full = torch.empty((3000 * 100 + 1,), dtype=torch.uint8, device="cuda")
    for view in (full[:-1], full[1:]):                                        # aligned / byte-shifted output
        out = view.view(3000, 100)
        from ribodetector_amd import _native as N
        N.check(N.lib().rd_encode_codes(N.ptr(b3.arena), N.ptr(b3.offsets), N.ptr(b3.lens), 3000, 100, 100, N.ptr(out), N.stream_ptr("cuda")), "rd_encode_codes")
        got = out.cpu().numpy()
        for i in range(0, 3000, 7):
            s = bytes(a3[o3[i]:o3[i + 1]])[:100]
            want = np.full(100, 4, dtype=np.uint8)
            want[:len(s)] = oracle.encode_codes(s)
            assert (got[i] == want).all(), i
    data, bs, osi, oui = oracle.pack_sequence(e["arena"], e["offsets"], e["lens"], 100)
    p = E.pack_reads(b, 100)
    assert (p.batch_sizes.numpy() == bs).all()
    assert (p.sorted_indices.cpu().numpy() == osi).all() and (p.unsorted_indices.cpu().numpy() == oui).all()
    assert (p.data.cpu().numpy() == data).all()
    assert (E.encode_read("ACGTUNacgt").cpu().numpy() == oracle.encode_onehot(b"ACGTUNacgt")).all()


def test_pair_fusion_and_counts(gpu_model, golden, oracle):
    from ribodetector_amd.model import model as M
    d = golden.npz("pe")
    l1, l2 = torch.from_numpy(d["r1_logits"]).cuda(), torch.from_numpy(d["r2_logits"]).cuda()
    for mode in ("none", "rrna", "norrna", "both"):
        counts = torch.zeros(3, dtype=torch.int64, device="cuda")
        lab = M.pair_fuse(l1, l2, mode, counts)
        lab2 = M.pair_fuse(l1, l2, mode, counts)     # counters accumulate like detect.py:388-389
        torch.cuda.synchronize()
        assert (lab.cpu().numpy() == d["labels_" + mode]).all() and (lab2 == lab).all()
        want = oracle.count_labels(d["labels_" + mode])
        assert counts.cpu().tolist() == [2 * w for w in want]
    # end to end on the pair fixture: HIP logits -> HIP fuse
    g1, _ = _run(gpu_model, d["r1_arena"], d["r1_offsets"], d["r1_lens"], 100)
    g2, _ = _run(gpu_model, d["r2_arena"], d["r2_offsets"], d["r2_lens"], 100)
    assert np.abs(g1 - d["r1_logits"]).max() < TOL and np.abs(g2 - d["r2_logits"]).max() < TOL
    for mode in ("none", "rrna", "norrna", "both"):
        lab = M.pair_fuse(torch.from_numpy(g1).cuda(), torch.from_numpy(g2).cuda(), mode).cpu().numpy()
        assert (lab == oracle.pair_fuse(g1, g2, mode)).all()
        assert (lab != d["labels_" + mode]).mean() < 0.005
    lab8 = torch.from_numpy((g1[:, 1] > g1[:, 0]).astype(np.uint8)).cuda()
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    M.count_labels(lab8, counts)
    assert counts.cpu().tolist() == [int((lab8 == 0).sum()), int((lab8 == 1).sum()), 0]


def test_properties_order_and_batching(gpu_model):
    """size-independent properties: permutation equivariance, batch-split invariance, determinism (bit-exact:
    every read is computed independently of its tile mates)"""
    from ribodetector_amd import synth
    n = 5000
    arena, off, lens = synth.reads_numpy(n, (40, 200), seed=12)
    lg, lab = _run(gpu_model, arena, off, lens, 150)
    lg2, lab2 = _run(gpu_model, arena, off, lens, 150)
    assert (lg == lg2).all() and (lab == lab2).all()
    perm = np.random.default_rng(0).permutation(n)
    lgp, labp = _run(gpu_model, arena, off[:-1][perm], lens[perm], 150)
    assert (lgp == lg[perm]).all() and (labp == lab[perm]).all()
    parts = [_run(gpu_model, arena, off[:-1][s:e], lens[s:e], 150)[0] for s, e in ((0, 1), (1, 777), (777, n))]
    assert (np.concatenate(parts) == lg).all()


def test_full_size_batch_properties(gpu_model, oracle):
    """BASELINE config A batch (32,768 reads x 100 bp, reference batch size at -m 32) + a 1M-read chunk:
    label counts consistent, spot-checked against the oracle, checksum stable across runs."""
    from ribodetector_amd import synth
    from ribodetector_amd.model import model as M
    arena, off, lens = synth.reads_torch(1 << 20, 100, seed=1, device="cuda")
    lg, lab = gpu_model.classify_bytes(arena, off[:-1].contiguous(), lens, 100)
    lg_b, lab_b = gpu_model.classify_bytes(arena, off[:-1].contiguous(), lens, 100)
    torch.cuda.synchronize()
    assert torch.equal(lg, lg_b) and torch.equal(lab, lab_b)
    counts = torch.zeros(3, dtype=torch.int64, device="cuda")
    M.count_labels(lab, counts)
    c = counts.cpu().tolist()
    assert c[0] + c[1] == 1 << 20 and c[2] == 0 and 0.03 < c[1] / (1 << 20) < 0.3
    assert torch.equal(((lg[:, 1] > lg[:, 0]).to(torch.uint8)), lab)
    idx = torch.arange(0, 1 << 20, 4099, device="cuda")[:256]
    sub_arena = arena.view(-1, 100)[idx].cpu().numpy().reshape(-1)
    ref = oracle.forward_packed(sub_arena, np.arange(257, dtype=np.int64) * 100, np.full(256, 100, dtype=np.int32), 100)
    _check(lg[idx].cpu().numpy(), lab[idx].cpu().numpy(), ref, "1M spot check")
    # the reference batch (32,768) as one call gives the same rows
    lg32, _ = gpu_model.classify_bytes(arena, off[:32768].contiguous(), lens[:32768].contiguous(), 100)
    assert torch.equal(lg32, lg[:32768])


@pytest.mark.parametrize("variant", ["auto", "simple", "mfma_f32"])
def test_padded_cpu_product_semantics(gpu_model, golden, oracle, variant):
    """RD_SEM_PADDED reproduces the reference's CPU product (model_cpu.forward_last: zero-padded input, gather at the last
    non-zero row) - golden `cpu_logits` come from the reference's model_cpu.SeqModel itself."""
    gpu_model.set_variant(variant)
    gpu_model.set_semantics("padded")
    try:
        d = golden.npz("se100")
        lg, lab = _run(gpu_model, d["arena"], d["offsets"], d["lens"], 100)
        _check(lg, lab, d["cpu_logits"], "padded se100")
        d = golden.npz("varlen")
        lg, lab = _run(gpu_model, d["arena"], d["offsets"], d["lens"], 170)
        _check(lg, lab, d["cpu_logits_l170"], "padded varlen170")
        # short reads, trailing / all N, lowercase, empty: against the oracle's restatement of the CPU product
        e = golden.npz("edge")
        for L in (100, 64, 301):
            ref = oracle.forward_padded(e["arena"], e["offsets"], e["lens"], L)
            lg, lab = _run(gpu_model, e["arena"], e["offsets"], e["lens"], L)
            _check(lg, lab, ref, "padded edge L=%d" % L)
        # and it really differs from the packed semantics on short reads, so the switch is live
        gpu_model.set_semantics("packed")
        lgp, _ = _run(gpu_model, e["arena"], e["offsets"], e["lens"], 100)
        assert np.abs(lgp - oracle.forward_padded(e["arena"], e["offsets"], e["lens"], 100)).max() > 1e-2
    finally:
        gpu_model.set_semantics("packed")
        gpu_model.set_variant("auto")


def test_config_c_and_d_shapes_properties(gpu_model, oracle):
    """BASELINE configs[3]/[4] shapes on one GPU's shard: 150 bp pairs with --ensure both, and 40-300 bp reads with -l 300:
    determinism, order equivariance under length bucketing, label consistency, spot checks against the oracle."""
    from ribodetector_amd import synth
    from ribodetector_amd.data_loader import seq_encoder as E
    from ribodetector_amd.model import model as M
    # configs[4]: variable length, the bucketing permutation is non-trivial
    n = 200000
    arena, off, lens = synth.reads_numpy(n, (40, 300), seed=44)
    b = E.batch_from_numpy(arena, off, lens, "cuda")
    lg, lab = gpu_model.classify_bytes(b.arena, b.offsets, b.lens, 300)
    lg2, lab2 = gpu_model.classify_bytes(b.arena, b.offsets, b.lens, 300)
    torch.cuda.synchronize()
    assert torch.equal(lg, lg2) and torch.equal(lab, lab2)
    perm = torch.randperm(n, device="cuda")
    lgp, labp = gpu_model.classify_bytes(b.arena, b.offsets[perm].contiguous(), b.lens[perm].contiguous(), 300)
    assert torch.equal(lgp, lg[perm]) and torch.equal(labp, lab[perm])
    idx = np.arange(0, n, n // 96)[:96]
    ref = oracle.forward_packed(arena, off[idx], lens[idx], 300)
    _check(lg[idx].cpu().numpy(), lab[idx].cpu().numpy(), ref, "configs[4] spot check")
    # truncation: -l 170 on the same reads equals classifying the 170-prefixes
    lgt, _ = gpu_model.classify_bytes(b.arena, b.offsets, b.lens, 170)
    lgc, _ = gpu_model.classify_bytes(b.arena, b.offsets, torch.clamp(b.lens, max=170), 300)
    assert torch.equal(lgt, lgc)
    # configs[3]: 150 bp pairs, all four modes consistent with the per-mate labels
    n = 100000
    a1, o1, l1 = synth.reads_numpy(n, 150, seed=45, rrna_frac=0.3)
    a2, o2, l2 = synth.reads_numpy(n, 150, seed=46, rrna_frac=0.3)
    b1, b2 = E.batch_from_numpy(a1, o1, l1, "cuda"), E.batch_from_numpy(a2, o2, l2, "cuda")
    g1, lab1 = gpu_model.classify_bytes(b1.arena, b1.offsets, b1.lens, 150)
    g2, lab2 = gpu_model.classify_bytes(b2.arena, b2.offsets, b2.lens, 150)
    both = M.pair_fuse(g1, g2, "both")
    assert torch.equal(both == 1, (lab1 == 1) & (lab2 == 1)) and torch.equal(both == 0, (lab1 == 0) & (lab2 == 0))
    assert torch.equal(M.pair_fuse(g1, g2, "rrna") == 1, both == 1) and torch.equal(M.pair_fuse(g1, g2, "norrna") == 0, both == 0)
    none = M.pair_fuse(g1, g2, "none")
    conc = both >= 0
    assert torch.equal(none[conc], both[conc])          # concordant pairs: the logit sum agrees with both mates
    assert (both == -1).any() and (none[~conc] == 0).any() and (none[~conc] == 1).any()
    idx = np.arange(0, n, n // 64)[:64]
    ref = oracle.forward_packed(a1, o1[idx], l1[idx], 150)
    _check(g1[idx].cpu().numpy(), lab1[idx].cpu().numpy(), ref, "configs[3] spot check")


def test_abi_error_paths_on_device(gpu_model):
    """status codes + rd_last_error through the C ABI with a live model (no exception crosses the boundary)"""
    from ribodetector_amd import _native as N
    L = N.lib()
    h = gpu_model._handle
    arena = torch.zeros(400, dtype=torch.uint8, device="cuda")
    off = torch.arange(4, dtype=torch.int64, device="cuda") * 100
    ln = torch.full((4,), 100, dtype=torch.int32, device="cuda")
    lg = torch.empty((4, 2), dtype=torch.float32, device="cuda")
    ws = torch.empty(1 << 16, dtype=torch.uint8, device="cuda")
    st = N.stream_ptr()
    args = lambda n, L_, wsb: (h, N.ptr(arena), N.ptr(off), N.ptr(ln), n, L_, N.ptr(lg), None, N.ptr(ws), wsb, st)
    assert L.rd_classify(*args(4, 100, ws.numel())) == 0
    assert L.rd_classify(*args(0, 100, ws.numel())) == 0                       # empty batch is fine
    assert L.rd_classify(*args(4, 100, 16)) == -4 and b"workspace" in L.rd_last_error()
    assert L.rd_classify(*args(4, 0, ws.numel())) == -1 and b"max_len" in L.rd_last_error()
    assert L.rd_classify(*args(4, 20000, ws.numel())) == -1
    assert L.rd_classify(*args(-1, 100, ws.numel())) == -1
    assert L.rd_set_variant(h, 99) == -3 and L.rd_set_semantics(h, 7) == -1
    for diag_id in (41, 42, 22, 10, 3, 5, 14):        # diagnostic / removed / never-defined ids: refused by the product build
        assert L.rd_set_variant(h, diag_id) == -3 and b"not available" in L.rd_last_error()
    with pytest.raises(RuntimeError):
        gpu_model.set_variant("nope")
    with pytest.raises(TypeError):
        gpu_model.classify_bytes(arena.cpu(), off, ln, 100)
    torch.cuda.synchronize()
    # the model still works after the failed calls
    lg2, _ = gpu_model.classify_bytes(arena + ord("A"), off, ln, 100)
    assert torch.isfinite(lg2).all()


def test_random_bytes_and_lengths_stress(gpu_model, oracle):
    """arbitrary byte values (0x00..0xFF), ragged lengths around tile and chunk boundaries, overlapping / unordered offsets:
    the HIP path must agree with the oracle and never read outside [off, off+len)"""
    rng = np.random.default_rng(99)
    for trial, (n, maxlen) in enumerate([(129, 67), (200, 130), (77, 300), (513, 100)]):
        lens = rng.integers(0, maxlen + 40, n).astype(np.int32)
        total = int(lens.sum())
        arena = rng.integers(0, 256, total + 8, dtype=np.uint8)
        # mostly valid bases so that the recurrence is exercised, with arbitrary bytes sprinkled in
        valid = np.frombuffer(b"ACGTU", dtype=np.uint8)[rng.integers(0, 5, total + 8)]
        arena = np.where(rng.random(total + 8) < 0.9, valid, arena).astype(np.uint8)
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        perm = rng.permutation(n)                       # reads handed over in a different order than they lie in the arena
        o2, l2 = off[:-1][perm], lens[perm]
        ref = oracle.forward_packed(arena, np.concatenate([o2, [0]]), l2, maxlen)
        from ribodetector_amd.data_loader import seq_encoder as E
        b = E.batch_from_numpy(arena, o2, l2, "cuda")
        lg, lab = gpu_model.classify_bytes(b.arena, b.offsets, b.lens, maxlen)
        torch.cuda.synchronize()
        _check(lg.cpu().numpy(), lab.cpu().numpy(), ref, "stress %d" % trial)


def _ref_tail():
    """the reference's own tail, from reference-made fixtures (tests/golden/scale_*.npz via tests/scale_sets.py)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from scale_sets import reference_tail
    t = reference_tail()
    # the citations below are live: if the fixtures stop saying this, the relaxed bars lose their justification and the tests fail
    assert t["scale_se100"]["max"] > 1.4e-4 and t["self_gap"] > 3.5e-4 and t["extra_max_vs_f64"] > 2e-4
    return t


def _check_tail(lg, lab, ref_logits, maxlen, what):
    """Like _check (1e-4 on every read) for reads of up to 100 steps on small sets. For longer reads and 10^5+ reads the bar is
    1e-4 for 99.9 % of the reads, 5e-5 for 99 %, 1e-3 for all, labels equal wherever the checker's own margin exceeds the observed
    error. Justification, from data the REFERENCE produced (tests/golden/scale_*.npz, asserted in _ref_tail): over 100,005 reads
    of 100 bp the reference itself is up to 1.5e-4 from the float64 value of its own function, and on a rounding-sensitive read it
    differs from ITSELF by 3.7e-4 when called with a batch of 2,048 instead of 1 - two faithful fp32 evaluations cannot be held to
    1e-4 on every read of a large set; tests/test_gpu_scale.py holds the kernels to 1e-4 against the reference on 140,000 reads."""
    if maxlen <= 100:
        return _check(lg, lab, ref_logits, what)
    _ref_tail()
    e = np.abs(lg - ref_logits).max(axis=1)
    assert np.quantile(e, 0.999) < TOL and np.quantile(e, 0.99) < 5e-5 and e.max() < 1e-3, \
        "%s: logit error max %.3g p99.9 %.3g p99 %.3g" % (what, e.max(), np.quantile(e, 0.999), np.quantile(e, 0.99))
    ref_lab = (ref_logits[:, 1] > ref_logits[:, 0]).astype(np.uint8)
    bad = np.flatnonzero(lab != ref_lab)
    margin = np.abs(ref_logits[:, 1] - ref_logits[:, 0])
    assert (margin[bad] < 2 * e[bad] + 2e-4).all(), "%s: label mismatch at margins %s" % (what, margin[bad])
    assert ((lg[:, 1] > lg[:, 0]).astype(np.uint8) == lab).all(), what
    return float(e.max())


def test_error_against_float64_truth(gpu_model, oracle, report):
    """The HIP kernels are as close to the exact (float64) value of the recurrence as the reference's fp32 arithmetic is:
    same median and 99th-percentile error as the fp32 oracle, for every kernel variant, at 100 and at 300 steps."""
    import os
    import sys
    from ribodetector_amd import synth
    from ribodetector_amd.data_loader import seq_encoder as E
    from ribodetector_amd.parse_config import ConfigParser
    sys.path.insert(0, os.path.dirname(__file__))
    from f64_truth import f64_forward
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sd = ConfigParser.from_json(os.path.join(root, "ribodetector_amd", "config.json")).load_state_dict("mcc")
    out = {}
    try:
        for maxlen, n in ((100, 6000), (300, 3000)):
            arena, off, lens = synth.reads_numpy(n, (maxlen - 60, maxlen + 20), seed=77 + maxlen, rrna_frac=0.3, n_rate=0.01)
            truth = f64_forward(sd, arena, off, lens, maxlen)
            eo = np.abs(oracle.forward_packed(arena, off, lens, maxlen) - truth).max(axis=1)
            b = E.batch_from_numpy(arena, off[:-1], lens, "cuda")
            for v in ("auto", "mfma_f32", "simple"):
                gpu_model.set_variant(v)
                lg, _ = gpu_model.classify_bytes(b.arena, b.offsets, b.lens, maxlen)
                ek = np.abs(lg.cpu().numpy().astype(np.float64) - truth).max(axis=1)
                out["%s@%d" % (v, maxlen)] = {"median": float(np.median(ek)), "p99": float(np.quantile(ek, 0.99)), "max": float(ek.max())}
                assert np.median(ek) < 1.5 * np.median(eo) + 2e-7, (v, maxlen, np.median(ek), np.median(eo))
                assert np.quantile(ek, 0.99) < 2.0 * np.quantile(eo, 0.99) + 1e-6, (v, maxlen)
                assert ek.max() < 4 * eo.max() + 2e-5, (v, maxlen, ek.max(), eo.max())
            out["oracle_fp32@%d" % maxlen] = {"median": float(np.median(eo)), "p99": float(np.quantile(eo, 0.99)), "max": float(eo.max())}
    finally:
        gpu_model.set_variant("auto")
    report["error_vs_float64"] = out


@pytest.mark.parametrize("sem", ["packed", "padded"])
def test_soak_random_reads(gpu_model, oracle, report, sem):
    """differential soak against the oracle: RD_SOAK_READS reads per case (default 4,000; the round-1 evidence run used
    200,000), lengths 0..max_len+40, rRNA-like / random / non-ACGT content mixed, several max_len values"""
    import os
    from ribodetector_amd import synth
    from ribodetector_amd.data_loader import seq_encoder as E
    n = int(os.environ.get("RD_SOAK_READS", "4000"))
    gpu_model.set_variant("auto")
    gpu_model.set_semantics(sem)
    worst = 0.0
    try:
        for k, maxlen in enumerate((100, 1, 37, 150, 300)):
            arena, off, lens = synth.reads_numpy(n, (0, maxlen + 40), seed=1000 + k, rrna_frac=0.3, n_rate=0.02)
            ref = (oracle.forward_packed if sem == "packed" else oracle.forward_padded)(arena, off, lens, maxlen)
            b = E.batch_from_numpy(arena, off[:-1], lens, "cuda")
            lg, lab = gpu_model.classify_bytes(b.arena, b.offsets, b.lens, maxlen)
            torch.cuda.synchronize()
            worst = max(worst, _check_tail(lg.cpu().numpy(), lab.cpu().numpy(), ref, maxlen, "soak %s -l %d" % (sem, maxlen)))
    finally:
        gpu_model.set_semantics("packed")
    report["soak_%s" % sem] = {"reads_per_case": n, "cases": 5, "max_abs_logit_err": float(worst)}


def test_integration_md_binding_runs(gpu_model, oracle):
    """The ctypes stub INTEGRATION.md shows a reference maintainer (`ribodetector/hip_backend.py`) is executed as written
    (only the library path is made absolute) and must classify like the package's own wrapper."""
    import os
    import re
    from ribodetector_amd import _native as N
    from ribodetector_amd import synth
    from ribodetector_amd.data_loader import seq_encoder as E
    from ribodetector_amd.parse_config import ConfigParser
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# ribodetector/hip_backend\.py.*?)```", md, re.S).group(1)
    block = block.replace('C.CDLL("librd_hip.so")', "C.CDLL(%r)" % N.LIB_PATH)
    ns = {}
    exec(compile(block, "INTEGRATION.md:hip_backend", "exec"), ns)
    sd = ConfigParser.from_json(os.path.join(root, "ribodetector_amd", "config.json")).load_state_dict("mcc")
    sd = {k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}
    hm = ns["HipModel"](sd, device=0)
    arena, off, lens = synth.reads_numpy(3000, (30, 130), seed=12, rrna_frac=0.3)
    b = E.batch_from_numpy(arena, off[:-1], lens, "cuda")
    logits, labels = hm.classify(b.arena, b.offsets, b.lens, 100)
    want, wlab = gpu_model.classify_bytes(b.arena, b.offsets, b.lens, 100)
    torch.cuda.synchronize()
    assert torch.equal(logits, want) and torch.equal(labels, wlab)
    _check(logits.cpu().numpy(), labels.cpu().numpy(), oracle.forward_packed(arena, off, lens, 100), "INTEGRATION.md stub")
    fused = hm.pair_labels(logits, logits.flip(0).contiguous(), "both")
    assert fused.dtype == torch.int8 and set(fused.unique().tolist()) <= {-1, 0, 1}


# ---- float64 refinement of the reads inside the fp32 noise band (rd_refine.hpp) -------------------------------------------

def _sd():
    import os
    from ribodetector_amd.parse_config import ConfigParser
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    return ConfigParser.from_json(os.path.join(root, "ribodetector_amd", "config.json")).load_state_dict("mcc")


def test_refine_matches_float64_and_leaves_the_rest_alone(gpu_model, oracle):
    """with a wide band (0.5: a few % of the reads) the refined reads carry the float64 value of the function (1e-6), the
    others are bit-identical to the un-refined run; both semantics; every kernel variant feeds the same pass"""
    import os
    import sys
    from ribodetector_amd import synth
    sys.path.insert(0, os.path.dirname(__file__))
    from f64_truth import f64_forward_torch
    n, L = 60000, 100
    arena, off, lens = synth.reads_torch(n, L, seed=321, device="cuda", rrna_frac=0.3, n_rate=0.002)
    offs = off[:-1].contiguous()
    truth = f64_forward_torch(_sd(), arena, L, "cuda")
    try:
        for v in ("auto", "mfma_f32"):
            gpu_model.set_variant(v)
            gpu_model.set_refine(0.0)
            lg0, lab0 = gpu_model.classify_bytes(arena, offs, lens, L)
            gpu_model.set_refine(0.5)
            lg1, lab1 = gpu_model.classify_bytes(arena, offs, lens, L)
            torch.cuda.synchronize()
            band = (lg0[:, 1] - lg0[:, 0]).abs() < 0.5
            assert 0.005 < float(band.double().mean()) < 0.2
            assert torch.equal(lg1[~band], lg0[~band]) and torch.equal(lab1[~band], lab0[~band])
            assert float((lg1[band].double() - truth[band]).abs().max()) < 1e-6
            assert float((lg0[band].double() - truth[band]).abs().max()) > 1e-6          # the pass did something
            assert torch.equal(lab1[band], (truth[band][:, 1] > truth[band][:, 0]).to(torch.uint8))
        # padded (ribodetector_cpu) semantics: refined reads against the fp32 oracle of that product
        gpu_model.set_variant("auto")
        gpu_model.set_semantics("padded")
        a2, o2, l2 = synth.reads_numpy(4000, (1, 140), seed=5, rrna_frac=0.3, n_rate=0.05)
        from ribodetector_amd.data_loader import seq_encoder as E
        b = E.batch_from_numpy(a2, o2[:-1], l2, "cuda")
        gpu_model.set_refine(1.0)
        lgp, labp = gpu_model.classify_bytes(b.arena, b.offsets, b.lens, 100)
        ref = oracle.forward_padded(a2, o2, l2, 100)
        assert np.abs(lgp.cpu().numpy() - ref).max() < 5e-5
        gpu_model.set_semantics("packed")
        lgk, _ = gpu_model.classify_bytes(b.arena, b.offsets, b.lens, 100)      # ragged lengths incl. 1-base reads, packed
        assert np.abs(lgk.cpu().numpy() - oracle.forward_packed(a2, o2, l2, 100)).max() < 5e-5
    finally:
        gpu_model.set_semantics("packed")
        gpu_model.set_variant("auto")
        gpu_model.set_refine(M_REFINE)


def test_refine_pair_margin(gpu_model):
    """rd_refine with mate logits: reads whose PAIR margin is inside twice the band are re-evaluated too (--ensure none)"""
    import os
    import sys
    from ribodetector_amd import synth
    sys.path.insert(0, os.path.dirname(__file__))
    from f64_truth import f64_forward_torch
    n, L = 40000, 100
    a1, off, lens = synth.reads_torch(n, L, seed=11, device="cuda", rrna_frac=0.3)
    a2, _, _ = synth.reads_torch(n, L, seed=12, device="cuda", rrna_frac=0.3)
    offs = off[:-1].contiguous()
    try:
        gpu_model.set_refine(0.0)
        g1, _ = gpu_model.classify_bytes(a1, offs, lens, L)
        g2, _ = gpu_model.classify_bytes(a2, offs, lens, L)
        r1 = g1.clone()
        gpu_model.set_refine(0.25)
        gpu_model.refine_pairs(a1, offs, lens, L, r1, g2)
        torch.cuda.synchronize()
        own = (g1[:, 1] - g1[:, 0]).abs() < 0.25
        pair = ((g1[:, 1] + g2[:, 1]) - (g1[:, 0] + g2[:, 0])).abs() < 0.5
        sel = own | pair
        assert bool((pair & ~own).any())
        assert torch.equal(r1[~sel], g1[~sel])
        truth = f64_forward_torch(_sd(), a1, L, "cuda")
        assert float((r1[sel].double() - truth[sel]).abs().max()) < 1e-6
    finally:
        gpu_model.set_refine(M_REFINE)


def test_labels_equal_float64_labels_at_scale(gpu_model, report):
    """2^21 reads x 100 bp against a float64 evaluation of the same function: with the default refine band every label equals the
    float64 label (reads whose exact margin is below 1e-6 excepted - there the yardstick's own rounding decides). The logits carry
    fp32 rounding noise: 3e-6 rms, 99.99 % of the reads within 5e-5; a few reads per million are rounding-sensitive far beyond
    that for EVERY fp32 evaluation - read 1,169,376 of this very set is 1.5e-4 (batch of 2,048) resp. 2.2e-4 (alone) from the exact
    value under THE REFERENCE (tests/golden/scale_se100.npz, extra rows: reference-made, asserted in _ref_tail and in
    tests/test_oracle.py), 6.7e-4 under the CPU oracle, 5.0e-4 under the default kernel - so the bound on the tail is a count, not
    zero: <= 4 reads per 2^21 beyond 1e-4, the reference's own rate being 1 per 100,005 (same fixture)."""
    _ref_tail()
    import os
    import sys
    from ribodetector_amd import synth
    sys.path.insert(0, os.path.dirname(__file__))
    from f64_truth import f64_forward_torch
    n, L = 1 << 21, 100
    arena, off, lens = synth.reads_torch(n, L, seed=2026, device="cuda")
    offs = off[:-1].contiguous()
    truth = f64_forward_torch(_sd(), arena, L, "cuda")
    tm = (truth[:, 1] - truth[:, 0])
    tl = (tm > 0).to(torch.uint8)
    try:
        gpu_model.set_refine(0.0)
        lg0, lab0 = gpu_model.classify_bytes(arena, offs, lens, L)
        gpu_model.set_refine(M_REFINE)
        lg, lab = gpu_model.classify_bytes(arena, offs, lens, L)
        torch.cuda.synchronize()
    finally:
        gpu_model.set_refine(M_REFINE)
    e = (lg.double() - truth).abs().max(dim=1).values
    bad = (lab != tl) & (tm.abs() > 1e-6)
    bad0 = (lab0 != tl)
    refined = int(((lg0[:, 1] - lg0[:, 0]).abs() < M_REFINE).sum())
    q = torch.quantile(e, torch.tensor([0.5, 0.9999], dtype=torch.float64, device=e.device))
    report["labels_vs_float64_2M"] = {"reads": n, "refined": refined, "mismatches_refined": int(bad.sum()), "mismatches_unrefined": int(bad0.sum()),
                                      "rms_logit_err_vs_f64": float((e ** 2).mean().sqrt()), "median": float(q[0]), "p9999": float(q[1]),
                                      "max_abs_logit_err_vs_f64": float(e.max()), "n_over_5e-5": int((e > 5e-5).sum()), "n_over_1e-4": int((e > 1e-4).sum()),
                                      "min_abs_truth_margin": float(tm.abs().min())}
    assert int(bad.sum()) == 0
    assert float((e ** 2).mean().sqrt()) < 5e-6 and float(q[1]) < 5e-5
    assert int((e > 1e-4).sum()) <= 4 and int((e > 5e-5).sum()) <= 40
    assert 3 <= refined <= 400


def test_logit_tail_against_the_oracle_at_one_million_reads(gpu_model, oracle, report):
    """VERDICT r1 asked for tests/diag_error_tail.py as a test: 2^20 reads x 100 bp, default kernel against the fp32 CPU oracle
    (= the reference's arithmetic). Both sides carry ~3e-6 rms of independent rounding noise, so their difference has a tail: the
    bound is 1e-4 for all but at most 3 reads per million (observed 0-2), 5e-5 at the 99.99th percentile, labels equal wherever
    the oracle's own margin exceeds 2e-4. The allowance of 3 per million is the reference's own rate: 1 of the 100,005 reads of
    tests/golden/scale_se100.npz is beyond 1e-4 from float64 for the REFERENCE (asserted in _ref_tail)."""
    _ref_tail()
    from conftest import FULL
    from ribodetector_amd import synth
    n, L = (1 << 20) if FULL else (1 << 18), 100          # (RD_TEST_FULL=1: the million reads of the name; the CPU oracle takes 80 s on them)
    arena, off, lens = synth.reads_torch(n, L, seed=4242, device="cuda", rrna_frac=0.3, n_rate=0.002)
    lg, lab = gpu_model.classify_bytes(arena, off[:-1].contiguous(), lens, L)
    ref = oracle.forward_packed(arena.cpu().numpy(), off.cpu().numpy(), lens.cpu().numpy(), L)
    lg, lab = lg.cpu().numpy(), lab.cpu().numpy()
    e = np.abs(lg - ref).max(axis=1)
    margin = np.abs(ref[:, 1] - ref[:, 0])
    bad = np.flatnonzero(lab != (ref[:, 1] > ref[:, 0]))
    report["auto_vs_oracle_1M"] = {"reads": n, "rms": float(np.sqrt((e.astype(np.float64) ** 2).mean())), "p9999": float(np.quantile(e, 0.9999)),
                                   "max": float(e.max()), "n_over_5e-5": int((e > 5e-5).sum()), "n_over_1e-4": int((e > 1e-4).sum()),
                                   "label_mismatches": int(len(bad)), "largest_oracle_margin_among_mismatches": float(margin[bad].max()) if len(bad) else None}
    assert np.quantile(e, 0.9999) < 5e-5 and int((e > 1e-4).sum()) <= 3 and e.max() < 1e-3
    assert (margin[bad] < 2e-4).all()


def test_refine_async_equals_inline(gpu_model):
    """rd_set_refine_async: the recurrence kernel's epilogue records the candidates, the candidates of K calls are evaluated
    together on the model's own stream and joined by the K-th following call or rd_sync_results. Same bits as the inline pass -
    with rotating buffer sets (K = 1 and K = 3), with ONE set reused by every call (the library sees the clash and synchronises
    first), through forward() (always final), inside a hipGraph capture (inline by design), and with more candidates than the queue
    holds (second tier). The band is widened to 0.5 so that every batch holds hundreds of candidates."""
    from ribodetector_amd import synth
    from torch.nn.utils.rnn import pack_sequence
    n, L, dev = 8192, 100, "cuda"
    batches = [synth.reads_torch(n, L, seed=900 + i, device=dev, rrna_frac=0.3) for i in range(6)]
    offs, lens = batches[0][1][:-1].contiguous(), batches[0][2]
    try:
        gpu_model.set_refine(0.5)
        want = [tuple(t.clone() for t in gpu_model.classify_bytes(b[0], offs, lens, L)) for b in batches]
        raw = gpu_model.set_refine(0.0).classify_bytes(batches[0][0], offs, lens, L)[0].clone()
        gpu_model.set_refine(0.5)
        assert int((raw != want[0][0]).any(dim=1).sum()) > 50                       # the pass really changes rows at this band
        gpu_model.set_refine_async(True)
        # (1) two alternating sets of output buffers; results of call i read after call i+1 was issued
        lg = [torch.empty((n, 2), dtype=torch.float32, device=dev) for _ in range(2)]
        lb = [torch.empty((n,), dtype=torch.uint8, device=dev) for _ in range(2)]
        of2, ln2 = [offs.clone(), offs.clone()], [lens.clone(), lens.clone()]      # inputs alternate too (a shared pointer = a clash)
        got = []
        for i, b in enumerate(batches):
            gpu_model.classify_bytes(b[0], of2[i & 1], ln2[i & 1], L, logits=lg[i & 1], labels=lb[i & 1])
            if i:
                got.append((lg[(i - 1) & 1].clone(), lb[(i - 1) & 1].clone()))      # joined by the call just issued
        gpu_model.sync_results()
        got.append((lg[(len(batches) - 1) & 1].clone(), lb[(len(batches) - 1) & 1].clone()))
        torch.cuda.synchronize()
        for i, (w, g) in enumerate(zip(want, got)):
            assert torch.equal(w[0], g[0]) and torch.equal(w[1], g[1]), i
        # (1b) groups of 3 calls, 4 sets of buffers: nothing is final before the sync / the third following call; everything after
        gpu_model.set_refine_async(3)
        lg4 = [torch.empty((n, 2), dtype=torch.float32, device=dev) for _ in range(6)]
        lb4 = [torch.empty((n,), dtype=torch.uint8, device=dev) for _ in range(6)]
        of6, ln6 = [offs.clone() for _ in range(6)], [lens.clone() for _ in range(6)]
        for i, b in enumerate(batches):
            gpu_model.classify_bytes(b[0], of6[i], ln6[i], L, logits=lg4[i], labels=lb4[i])
        first_group = [(lg4[i].clone(), lb4[i].clone()) for i in range(3)]      # calls 0-2: evaluated beside call 3, joined by it
        gpu_model.sync_results()
        torch.cuda.synchronize()
        for i in range(3):
            assert torch.equal(first_group[i][0], want[i][0]) and torch.equal(first_group[i][1], want[i][1]), i
        for i in range(6):
            assert torch.equal(lg4[i], want[i][0]) and torch.equal(lb4[i], want[i][1]), i
        gpu_model.set_refine_async(1)
        # (2) one set of buffers for every call: the clash is detected, the pending pass is joined before the next launch
        for i, b in enumerate(batches[:3]):
            gpu_model.classify_bytes(b[0], offs, lens, L, logits=lg[0], labels=lb[0])
            gpu_model.sync_results()
            assert torch.equal(lg[0], want[i][0]) and torch.equal(lb[0], want[i][1]), i
        for i, b in enumerate(batches[:3]):                                          # ... also without the explicit sync in between
            gpu_model.classify_bytes(b[0], offs, lens, L, logits=lg[0], labels=lb[0])
        gpu_model.sync_results()
        assert torch.equal(lg[0], want[2][0]) and torch.equal(lb[0], want[2][1])
        # (3) the reference-compatible call returns final logits
        a = batches[1][0].view(n, L)[:512]
        code = torch.full((256,), 4, dtype=torch.int64, device=dev)
        for k, ch in enumerate(b"ACGT"):
            code[ch] = k
        oh = torch.nn.functional.one_hot(code[a.long()], 5)[:, :, :4].float()
        out = gpu_model(pack_sequence([oh[i] for i in range(512)], enforce_sorted=False))
        assert torch.equal(out, want[1][0][:512])
        # (4) captured in a graph the pass stays inside the call
        buf = batches[2][0].clone()
        gpu_model.classify_bytes(buf, offs, lens, L, logits=lg[1], labels=lb[1])
        gpu_model.sync_results()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            gpu_model.classify_bytes(buf, offs, lens, L, logits=lg[1], labels=lb[1])
        buf.copy_(batches[3][0])
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(lg[1], want[3][0]) and torch.equal(lb[1], want[3][1])
        # (5) more candidates than the queue holds (8,192): the second tier re-evaluates the calls of the group, same bits again
        nb = 1 << 18
        big = [synth.reads_torch(nb, L, seed=950 + i, device=dev, rrna_frac=0.3) for i in range(2)]
        ob, lnb = big[0][1][:-1].contiguous(), big[0][2]
        gpu_model.set_refine_async(0)
        gpu_model.set_refine(1.0)
        wantb = [tuple(t.clone() for t in gpu_model.classify_bytes(b[0], ob, lnb, L)) for b in big]
        rawb = gpu_model.set_refine(0.0).classify_bytes(big[0][0], ob, lnb, L)[0].clone()
        gpu_model.set_refine(1.0)
        assert int(((rawb[:, 1] - rawb[:, 0]).abs() < 1.0).sum()) > 8192          # one call alone overflows the queue
        gpu_model.set_refine_async(2)
        outs = [gpu_model.classify_bytes(b[0], ob.clone(), lnb.clone(), L) for b in big]
        gpu_model.sync_results()
        torch.cuda.synchronize()
        for i in range(2):
            assert torch.equal(outs[i][0], wantb[i][0]) and torch.equal(outs[i][1], wantb[i][1]), i
        # (6) the pipelined caller's form (bench.py, the CLI): calls on the main stream, rd_sync_results on a SIDE stream that is ordered
        # behind them by an event; the side stream consumes the results while the main stream goes on with the next batch (which
        # records into the model's other queue). Two alternating buffer sets, reuse ordered by the consumer's event.
        gpu_model.set_refine_async(0)
        gpu_model.set_refine(0.5)
        gpu_model.set_refine_async(16)
        side, cur = torch.cuda.Stream(), torch.cuda.current_stream()
        consumed, got6 = [None, None], []
        for rep in range(3):
            for i, b in enumerate(batches):
                k = i & 1
                if consumed[k] is not None:
                    cur.wait_event(consumed[k])
                gpu_model.classify_bytes(b[0], of2[k], ln2[k], L, logits=lg[k], labels=lb[k])
                ev = torch.cuda.Event()
                ev.record(cur)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    gpu_model.sync_results()
                    got6.append((lg[k].clone(), lb[k].clone()))
                    consumed[k] = torch.cuda.Event()
                    consumed[k].record(side)
        torch.cuda.synchronize()
        for i, g in enumerate(got6):
            w = want[i % len(batches)]
            assert torch.equal(w[0], g[0]) and torch.equal(w[1], g[1]), i
    finally:
        gpu_model.sync_results()
        gpu_model.set_refine_async(0)
        gpu_model.set_refine(gpu_model.REFINE_DEFAULT)


def test_classify_is_capturable_in_a_hip_graph(gpu_model):
    """every launch of rd_classify (memset, steps, bucketing, recurrence, refine) is asynchronous on the caller's stream and nothing
    synchronises, so a caller with small batches (the reference's 16,384-read batch) can capture the call in a hipGraph; the replay
    gives bit-identical results on new bytes in the same buffers"""
    from ribodetector_amd import synth
    n, L = 16384, 100
    a1, off, lens = synth.reads_torch(n, L, seed=5, device="cuda")
    a2, _, _ = synth.reads_torch(n, L, seed=6, device="cuda")
    offs = off[:-1].contiguous()
    want1, wl1 = gpu_model.classify_bytes(a1, offs, lens, L)
    want2, wl2 = gpu_model.classify_bytes(a2, offs, lens, L)
    buf = a1.clone()
    lg = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    lab = torch.empty((n,), dtype=torch.uint8, device="cuda")
    gpu_model.classify_bytes(buf, offs, lens, L, logits=lg, labels=lab)       # workspace allocated before the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gpu_model.classify_bytes(buf, offs, lens, L, logits=lg, labels=lab)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(lg, want1) and torch.equal(lab, wl1)
    buf.copy_(a2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(lg, want2) and torch.equal(lab, wl2)
