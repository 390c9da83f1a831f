#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference; biopython is absent, so a
`Bio.Seq` stub is injected - `Seq` is used only by the reference's training helpers).
Only *data* is written: inputs (read bytes) and the reference's outputs for them.

Reference entry points used (the oracle definition of SURVEY.md §8c):
  ribodetector.parse_config.ConfigParser.from_json / init_obj   (parse_config.py:29-57)
  ribodetector.model.model.SeqModel  (forward1, model.py:32-37)  - GPU-path semantics
  ribodetector.model.model.SeqModel(pack_seq=False)  (forward2 + last_pad_out_items, model.py:40-50,67-72) - padded Tensor input
  ribodetector.model.model_cpu.SeqModel (forward_last, model_cpu.py:29-37) - CPU-product semantics
  ribodetector.detect.unlabeled_read_collate_fn / unlabeled_paired_read_collate_fn (detect.py:666-726)
  ribodetector.detect.Predictor.separate_paired_reads (detect.py:616-663)
  ribodetector.data_loader.seq_encoder.encode_read / encode_variable_len_read (seq_encoder.py:126-145)
  ribodetector.data_loader.fastx_parser.seq_parser (fastx_parser.py:15-55)

Usage:  python tests/golden/make_golden.py
"""
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

bio = types.ModuleType("Bio")
bseq = types.ModuleType("Bio.Seq")
bseq.Seq = object
bio.Seq = bseq
sys.modules["Bio"] = bio
sys.modules["Bio.Seq"] = bseq
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402

from ribodetector import detect as R  # noqa: E402
from ribodetector.data_loader import seq_encoder as RE  # noqa: E402
from ribodetector.data_loader.fastx_parser import seq_parser  # noqa: E402
from ribodetector.model import model as RM  # noqa: E402
from ribodetector.model import model_cpu as RMC  # noqa: E402
from ribodetector.parse_config import ConfigParser  # noqa: E402

from ribodetector_amd import synth  # noqa: E402

torch.set_num_threads(8)
REFPKG = "/root/reference/ribodetector"


def load_models():
    cfg = ConfigParser.from_json(os.path.join(REFPKG, "config.json"))
    m = cfg.init_obj("arch", RM)
    sd = torch.load(os.path.join(REFPKG, cfg["state_file"]["mcc"]), map_location="cpu")["state_dict"]
    m.load_state_dict(sd)
    m.eval()
    args = dict(cfg["arch"]["args"])
    args.pop("pack_seq")
    mc = RMC.SeqModel(**args)
    mc.load_state_dict(sd)
    mc.eval()
    return m, mc


def recs(seqs):
    return [("@r%d" % i, s, "+", "I" * len(s)) for i, s in enumerate(seqs)]


def ref_logits(m, seqs, max_len, bs=1024):
    out = []
    with torch.no_grad():
        for i in range(0, len(seqs), bs):
            _, x = R.unlabeled_read_collate_fn(recs(seqs[i:i + bs]), max_len=max_len, pack_seq=True)
            out.append(m(x))
    return torch.cat(out).numpy().astype(np.float32)


def cpu_logits(mc, seqs, max_len, bs=1024):
    out = []
    with torch.no_grad():
        for i in range(0, len(seqs), bs):
            x = np.array([RE.encode_variable_len_read(s, max_len=max_len) for s in seqs[i:i + bs]], dtype=np.float32)
            out.append(mc(torch.from_numpy(x)))
    return torch.cat(out).numpy().astype(np.float32)


def pack(seqs):
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    arena = np.frombuffer("".join(seqs).encode("latin-1"), dtype=np.uint8).copy()
    return arena, off, lens


class _Args:
    def __init__(self, ensure):
        self.ensure = ensure


def pair_labels(r1_logits, r2_logits, ensure):
    """Labels via the reference's own Predictor.separate_paired_reads."""
    p = R.Predictor.__new__(R.Predictor)
    p.args = _Args(ensure)
    n = len(r1_logits)
    ids = [str(i) for i in range(n)]
    d1, d2 = p.separate_paired_reads(ids, torch.from_numpy(r1_logits), ids, torch.from_numpy(r2_logits))
    lab = np.full(n, -9, dtype=np.int8)
    for k, v in d1.items():
        for s in v:
            lab[int(s)] = k
    assert (lab != -9).all() and {k: v for k, v in d1.items()} == {k: v for k, v in d2.items()}
    return lab


def make_forward2():
    """F7: the reference's padded-Tensor entry (pack_seq=false -> forward2, model.py:40-50): input [B, L, 4] zero-padded one-hot,
    BiLSTM over all L rows, gather at the last non-zero row (last_pad_out_items, model.py:67-72)."""
    cfg = ConfigParser.from_json(os.path.join(REFPKG, "config.json"))
    args = dict(cfg["arch"]["args"])
    args["pack_seq"] = False
    m2 = RM.SeqModel(**args)
    m2.load_state_dict(torch.load(os.path.join(REFPKG, cfg["state_file"]["mcc"]), map_location="cpu")["state_dict"])
    m2.eval()
    e = np.load(os.path.join(HERE, "edge.npz"))
    seqs = synth.as_strings(e["arena"], e["offsets"])
    av, ov, lv = synth.reads_numpy(256, (1, 140), seed=71, rrna_frac=0.3, n_rate=0.02)
    seqs += synth.as_strings(av, ov)
    out = {}
    for L in (100, 64):
        x = np.array([RE.encode_variable_len_read(s, max_len=L) for s in seqs], dtype=np.float32)
        with torch.no_grad():
            out["logits_l%d" % L] = m2(torch.from_numpy(x)).numpy().astype(np.float32)
    a, o, l = pack(seqs)
    np.savez_compressed(os.path.join(HERE, "forward2.npz"), arena=a, offsets=o, lens=l, **out)
    print("forward2:", len(seqs), "reads")


def main():
    if "--only-forward2" in sys.argv:
        return make_forward2()
    m, mc = load_models()
    S = synth.RRNA_16S
    # ---- F1 known-answer table ---------------------------------------------
    kat_reads = {
        "A100": "A" * 100, "C100": "C" * 100, "G100": "G" * 100, "T100": "T" * 100, "U100": "U" * 100,
        "ACGT25": "ACGT" * 25, "N100": "N" * 100, "acgt25_lower": "acgt" * 25, "ACGT10": "ACGT" * 10,
        "16S_0_100": S[0:100], "16S_0_150_trunc": S[0:150], "16S_50_150": S[50:150], "16S_0_60": S[0:60],
    }
    names = list(kat_reads)
    lg = ref_logits(m, [kat_reads[k] for k in names], 100)
    kat = {k: {"read": kat_reads[k], "logits": [float(lg[i, 0]), float(lg[i, 1])],
               "label": int(np.argmax(lg[i]))} for i, k in enumerate(names)}
    # reverse-direction LUT: logits contribution of h_rev(last base)
    with open(os.path.join(HERE, "kat.json"), "w") as fh:
        json.dump({"max_len": 100, "cases": kat}, fh, indent=1)

    # ---- F2 SE-100 -------------------------------------------------------------------------
    arena, off, lens = synth.reads_numpy(2304, 100, seed=11)
    seqs = synth.as_strings(arena, off)
    lg = ref_logits(m, seqs, 100)
    np.savez_compressed(os.path.join(HERE, "se100.npz"), arena=arena, offsets=off, lens=lens, max_len=100,
                        logits=lg, labels=np.argmax(lg, 1).astype(np.uint8),
                        cpu_logits=cpu_logits(mc, seqs, 100))
    print("se100: label1 frac", float(np.argmax(lg, 1).mean()), "min margin", float(np.abs(lg[:, 0] - lg[:, 1]).min()))

    # ---- F3 edge cases (-l 100) ----------------------------------------------------------
    rng = np.random.default_rng(5)

    def rnd(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    edge = []
    for L in (1, 2, 3, 15, 16, 17, 39, 40, 63, 64, 65, 99, 100, 101, 150, 300):
        edge += [rnd(L), S[:L] if L <= len(S) else (S * 2)[:L]]
    b = rnd(100)
    edge += ["N" * 5 + b[5:], b[:95] + "N" * 5, b[:40] + "NNNN" + b[44:], b[:99] + "N", "N" + b[1:],
             b.lower(), b[:50] + b[50:].lower(), b.replace("T", "U"), "RYKMSWBDHVN" * 9 + "A",
             "N" * 100, "N" * 37, "X" * 10 + b[10:], b[:60] + "-" * 3 + b[63:], "*" * 100, b[:80] + "acgtn" * 4,
             "U" * 100, "u" * 100, S[:99] + "N", S[:100].replace("T", "U"), "A", "N", "T" * 300]
    lg = ref_logits(m, edge, 100)
    a2, o2, l2 = pack(edge)
    np.savez_compressed(os.path.join(HERE, "edge.npz"), arena=a2, offsets=o2, lens=l2, max_len=100,
                        logits=lg, labels=np.argmax(lg, 1).astype(np.uint8))

    # ---- F4 paired-end, 4 ensure modes ------------------------------------------------
    a1, o1, l1 = synth.reads_numpy(1024, 100, seed=21, rrna_frac=0.3)
    a2, o2, l2 = synth.reads_numpy(1024, 100, seed=22, rrna_frac=0.3)
    s1, s2 = synth.as_strings(a1, o1), synth.as_strings(a2, o2)
    # engineered discordant pairs: one clearly rRNA mate + one clearly non-rRNA mate, and weak/strong mixes
    for i in range(0, 64):
        s1[i] = S[(i % 60):(i % 60) + 100] if i % 2 == 0 else rnd(100)
        s2[i] = rnd(100) if i % 2 == 0 else S[(i % 50):(i % 50) + 100]
    for i in range(64, 96):
        s1[i] = S[:40 + i - 64] + rnd(60 - (i - 64))          # partial-rRNA mates: small margins
        s2[i] = rnd(100)
    a1, o1, l1 = pack(s1)
    a2, o2, l2 = pack(s2)
    with torch.no_grad():
        _, x1, _, x2 = R.unlabeled_paired_read_collate_fn(list(zip(recs(s1), recs(s2))), max_len=100, pack_seq=True)
        g1, g2 = m(x1).numpy().astype(np.float32), m(x2).numpy().astype(np.float32)
    modes = {e: pair_labels(g1, g2, e) for e in ("rrna", "norrna", "both", "none")}
    maj = (np.argmax(g1, 1) + np.argmax(g2, 1))
    print("pe: discordant", int((maj == 1).sum()), "none-label1", int(modes["none"].sum()),
          "both-unclassified", int((modes["both"] == -1).sum()))
    np.savez_compressed(os.path.join(HERE, "pe.npz"), r1_arena=a1, r1_offsets=o1, r1_lens=l1,
                        r2_arena=a2, r2_offsets=o2, r2_lens=l2, max_len=100, r1_logits=g1, r2_logits=g2,
                        **{"labels_" + k: v for k, v in modes.items()})

    # ---- F5 variable length 40..300 ------------------------------------------------------
    av, ov, lv = synth.reads_numpy(2048, (40, 300), seed=31)
    sv = synth.as_strings(av, ov)
    np.savez_compressed(os.path.join(HERE, "varlen.npz"), arena=av, offsets=ov, lens=lv,
                        logits_l300=ref_logits(m, sv, 300), logits_l170=ref_logits(m, sv, 170),
                        cpu_logits_l170=cpu_logits(mc, sv, 170))

    # ---- F6 collate / PackedSequence layout of one tiny batch -----------------------
    tiny = ["ACGTN", "AC", "GGGTTTA", "T", "ACG", "NNAC", "ACGTACGTAC"]
    _, x = R.unlabeled_read_collate_fn(recs(tiny), max_len=8, pack_seq=True)
    onehot = [np.array(RE.encode_read(s[:8]), dtype=np.float32).reshape(-1, 4) for s in tiny]
    padded = np.array([RE.encode_variable_len_read(s, max_len=8) for s in tiny], dtype=np.float32)
    at, ot, lt = pack(tiny)
    np.savez_compressed(os.path.join(HERE, "collate.npz"), arena=at, offsets=ot, lens=lt, max_len=8,
                        data=x.data.numpy(), batch_sizes=x.batch_sizes.numpy(),
                        sorted_indices=x.sorted_indices.numpy(), unsorted_indices=x.unsorted_indices.numpy(),
                        sorted_last_indices=RM.sorted_last_indices(x).numpy(),
                        onehot_concat=np.concatenate(onehot), padded=padded)

    # ---- F8 FASTQ / FASTA parser fixture -------------------------------------------------
    fq = "@h1 desc\nACGTNacgt\n+\nIIIIIIIII\n@h2\nGGGG\n+h2\n@@@@\n@h3\nTTTTTTTTTTTT  \n+\nIIIIIIIIIIII\n"
    fa = ">s1 d\nACGT\nacgtnn\n\n>s2\nGGGG\n>s3\nTT\nTT\n"
    with open(os.path.join(HERE, "parser.json"), "w") as fh:
        json.dump({"fastq_text": fq, "fastq_records": [list(r) for r in seq_parser(io.StringIO(fq), "fastq")],
                   "fasta_text": fa, "fasta_records": [list(r) for r in seq_parser(io.StringIO(fa), "fasta")]},
                  fh, indent=1)
    make_forward2()
    tot = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE) if f.endswith((".npz", ".json")))
    print("golden fixtures written,", tot, "bytes total")


if __name__ == "__main__":
    main()
