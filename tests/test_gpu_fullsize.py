"""BASELINE.json configs at their real per-GPU sizes, end to end through the `ribodetector` CLI (reference flow detect.py:464-499):
synthetic FASTQ files built in tmpfs, the CLI's output files compared - read by read, in order - with ONE in-HBM classification of
the same reads (bit-for-bit labels: every read is computed independently of batch and chunk boundaries), the counters compared,
and a sample of the reads around the multiples of the nominal chunk size checked against the CPU oracle. (Since round 3 the
CLI's chunks hold ABOUT --chunk_size x batch records: small first and last chunks, byte segments for a single plain file,
data_loader/fastx_parser.py; the comparison with one in-HBM classification covers every read whatever the cuts are.)

    configs[1]  10 M single-end 100 bp, --chunk_size 256 -m 32  (nominal chunks of 8,388,608 reads)
    configs[2]  10 M pairs 100 bp, --ensure rrna                (nominal chunks of 4,194,304 pairs)
    configs[3]  per-GPU shard of 50 M pairs 150 bp on 8 GPUs: 6.25 M pairs, -l 150
    configs[4]  per-GPU shard of 100 M reads 40-300 bp on 8 GPUs: 12.5 M reads, -l 300 (length-bucketed)
RD_FULLSIZE_SCALE (default 1.0) scales the read counts for quick runs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SCALE = float(os.environ.get("RD_FULLSIZE_SCALE", "1.0"))
TMP = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"


def _write(path, image):
    with open(path, "wb") as fh:
        fh.write(memoryview(image.cpu().numpy()))


def _indices_in(path):
    """record indices (the 9 digits of the '@s' headers) of a FASTQ file written by the CLI, in file order"""
    from ribodetector_amd.data_loader import fastx_parser as fx
    out = []
    p10 = 10 ** np.arange(8, -1, -1, dtype=np.int64)
    for c in fx.get_seq_chunks(path, chunk_size=1 << 21):
        st = np.asarray(c.rec_start[:-1], dtype=np.int64)
        dig = c.buf[(st[:, None] + np.arange(2, 11)[None, :]).reshape(-1)].reshape(-1, 9).astype(np.int64) - 48
        out.append(dig @ p10)
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)


def _oracle_sample(oracle, arena, off, lens, idx, max_len, logits, labels, what):
    a, o, l = arena.cpu().numpy(), off.cpu().numpy(), lens.cpu().numpy()
    ref = oracle.forward_packed(a, np.concatenate([o[idx], [0]]), l[idx], max_len)
    got = logits[torch.as_tensor(idx, device=logits.device)].cpu().numpy()
    e = np.abs(got - ref).max(axis=1)
    margin = np.abs(ref[:, 1] - ref[:, 0])
    bad = np.flatnonzero(labels[torch.as_tensor(idx, device=labels.device)].cpu().numpy() != (ref[:, 1] > ref[:, 0]))
    if max_len <= 100:
        assert e.max() < 1e-4, (what, e.max())
    else:       # beyond 100 steps fp32 noise reaches 1e-4 on a few reads per 10^5 for any implementation - the reference included:
                # tests/golden/scale_*.npz (reference-made; tests/test_oracle.py, tests/test_gpu_scale.py) record its own tail
        assert np.quantile(e, 0.999) < 1e-4 and e.max() < 1e-3, (what, e.max())
    assert (margin[bad] < 2e-4 + 2 * e[bad]).all(), (what, margin[bad])
    return float(e.max())


def _boundary_sample(n, chunk, k, seed):
    """indices straddling every chunk boundary plus a random rest, sorted, k in total"""
    rng = np.random.default_rng(seed)
    parts = [np.arange(max(0, b - 2048), min(n, b + 2048)) for b in range(chunk, n, chunk)]
    parts.append(rng.integers(0, n, size=k))
    return np.unique(np.concatenate(parts))[:k]


def _run_se(tmp, n, read_len, max_len, cli_extra, chunk, seed, oracle, gpu_model, report, name, var=False):
    from ribodetector_amd import detect, synth
    dev = "cuda"
    if var:
        arena, off, lens = synth.reads_torch(n, read_len, seed=seed, device=dev)
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        lens = torch.randint(40, read_len + 1, (n,), generator=g, device=dev, dtype=torch.int32)   # rows keep their stride
    else:
        arena, off, lens = synth.reads_torch(n, read_len, seed=seed, device=dev)
    offs = off[:-1].contiguous()
    lg, lab = gpu_model.classify_bytes(arena, offs, lens, max_len)
    torch.cuda.synchronize()
    inp, out, rr = (os.path.join(tmp, x) for x in ("in.fq", "nonrrna.fq", "rrna.fq"))
    _write(inp, synth.fastq_image_torch(arena, offs, lens, 1))
    p = detect.main(["-l", str(max_len), "-i", inp, "-o", out, "-r", rr] + cli_extra)
    labn = lab.cpu().numpy()
    assert p.num_read == n and p.num_rrna == int(labn.sum()) and p.num_nonrrna == int((labn == 0).sum())
    assert np.array_equal(_indices_in(rr), np.flatnonzero(labn == 1))
    assert np.array_equal(_indices_in(out), np.flatnonzero(labn == 0))
    assert os.path.getsize(out) + os.path.getsize(rr) == os.path.getsize(inp)
    idx = _boundary_sample(n, chunk, 65536 if max_len <= 100 else 16384, seed)
    emax = _oracle_sample(oracle, arena, offs, lens, idx, max_len, lg, lab, name)
    report["fullsize_" + name] = {"reads": n, "rrna": p.num_rrna, "non_rrna": p.num_nonrrna, "oracle_sample": int(len(idx)),
                                  "oracle_sample_max_abs_logit_err": emax, "chunks": -(-n // chunk)}
    for f in (inp, out, rr):
        os.remove(f)


def test_config1_10M_single_end_100bp(tmp_path, oracle, gpu_model, report):
    import tempfile
    n = int(10_000_000 * SCALE)
    with tempfile.TemporaryDirectory(dir=TMP) as tmp:
        _run_se(tmp, n, 100, 100, ["--chunk_size", "256", "-m", "32"], 32768 * 256, 1, oracle, gpu_model, report, "config1_se100")


def test_config4_shard_variable_length_40_300(tmp_path, oracle, gpu_model, report):
    import tempfile
    n = int(12_500_000 * SCALE)
    with tempfile.TemporaryDirectory(dir=TMP) as tmp:
        # -l 300, -m 32 -> batch 16,384; chunk_size 256 -> chunks of 4,194,304 reads
        _run_se(tmp, n, 300, 300, ["--chunk_size", "256", "-m", "32"], 16384 * 256, 4, oracle, gpu_model, report, "config4_var300", var=True)


@pytest.mark.parametrize("name,pairs,read_len,ensure,seed", [("config2_pe100_rrna", 10_000_000, 100, "rrna", 2),
                                                              ("config3_shard_pe150", 6_250_000, 150, "none", 3)])
def test_paired_configs(tmp_path, oracle, gpu_model, report, name, pairs, read_len, ensure, seed):
    import tempfile
    from ribodetector_amd import detect, synth
    from ribodetector_amd.model import model as M
    n = int(pairs * SCALE)
    dev = "cuda"
    a1, off, lens = synth.reads_torch(n, read_len, seed=seed, device=dev)
    a2, _, _ = synth.reads_torch(n, read_len, seed=seed + 100, device=dev)
    offs = off[:-1].contiguous()
    g1, l1 = gpu_model.classify_bytes(a1, offs, lens, read_len)
    g2, l2 = gpu_model.classify_bytes(a2, offs, lens, read_len)
    if ensure == "none":
        gpu_model.refine_pairs(a1, offs, lens, read_len, g1, g2)
        gpu_model.refine_pairs(a2, offs, lens, read_len, g2, g1)
    counts = torch.zeros(3, dtype=torch.int64, device=dev)
    lab = M.pair_fuse(g1, g2, ensure, counts)
    torch.cuda.synchronize()
    labn = lab.cpu().numpy()
    batch = 2 ** int(np.floor(np.log2((32 - 2) * 1024 * 1024 / (2 * read_len * 6.4))))
    chunk = batch * 256
    with tempfile.TemporaryDirectory(dir=TMP) as tmp:
        f = [os.path.join(tmp, x) for x in ("r_1.fq", "r_2.fq", "n_1.fq", "n_2.fq", "rr_1.fq", "rr_2.fq")]
        _write(f[0], synth.fastq_image_torch(a1, offs, lens, 1))
        _write(f[1], synth.fastq_image_torch(a2, offs, lens, 2))
        p = detect.main(["-l", str(read_len), "-i", f[0], f[1], "-o", f[2], f[3], "-r", f[4], f[5], "-e", ensure, "--chunk_size", "256", "-m", "32"])
        c = counts.cpu().tolist()
        assert p.num_read == n and [p.num_nonrrna, p.num_rrna, p.num_unknown] == c and c[2] == 0
        for mate in (0, 1):
            assert np.array_equal(_indices_in(f[4 + mate]), np.flatnonzero(labn == 1))
            assert np.array_equal(_indices_in(f[2 + mate]), np.flatnonzero(labn == 0))
            assert os.path.getsize(f[2 + mate]) + os.path.getsize(f[4 + mate]) == os.path.getsize(f[mate])
    idx = _boundary_sample(n, chunk, 32768 if read_len <= 100 else 12288, seed)
    e1 = _oracle_sample(oracle, a1, offs, lens, idx, read_len, g1, l1, name + "/R1")
    e2 = _oracle_sample(oracle, a2, offs, lens, idx, read_len, g2, l2, name + "/R2")
    report["fullsize_" + name] = {"pairs": n, "non_rrna": c[0], "rrna": c[1], "chunks": -(-n // chunk), "oracle_sample_pairs": int(len(idx)),
                                  "oracle_sample_max_abs_logit_err": max(e1, e2)}
