#!/usr/bin/env python3
"""Reference logits AT SCALE (10^5 reads) - the fixtures that pin the tail bars of the parity tests to the reference itself.

Like make_golden.py this IMPORTS the reference (build container only; `Bio.Seq` stubbed) and writes data only:
  tests/golden/scale_se100.npz   100,000 seeded 100 bp reads + the five rounding-sensitive reads of profiles/r02_outlier.json
                                 (among them read 1,169,376 of synth.reads_torch(2^21, 100, seed=2026)), -l 100
  tests/golden/scale_pe150.npz   20,000 seeded 150 bp reads, -l 150         (BASELINE configs[3] geometry)
  tests/golden/scale_var300.npz  20,000 seeded 40-300 bp reads, -l 300      (BASELINE configs[4] geometry)
Each holds: `ref` float32[n,2] = the reference's own forward1 (ribodetector.model.model.SeqModel, model/model.py:32-37, through
its collate detect.py:666-689) on torch CPU; `extra` = the reads that do not come from the seed; `sha256` of the read bytes (the
reads themselves regenerate from ribodetector_amd.synth with the recorded seed); and `stats` (JSON): the distance of the
reference from a float64 evaluation of the same function (tests/f64_truth.py) - rms, quantiles, max, the number of reads beyond
1e-4 - plus the float64 logits of the 32 reads where the reference is furthest off, so that CPU tests can assert the tail
without re-running 10^5 reads in float64.

Usage:  python tests/golden/make_golden_scale.py
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

bio = types.ModuleType("Bio")
bseq = types.ModuleType("Bio.Seq")
bseq.Seq = object
bio.Seq = bseq
sys.modules["Bio"] = bio
sys.modules["Bio.Seq"] = bseq
sys.path.insert(0, "/root/reference")

import torch  # noqa: E402

from ribodetector import detect as R  # noqa: E402
from ribodetector.model import model as RM  # noqa: E402
from ribodetector.parse_config import ConfigParser as RefConfig  # noqa: E402

from f64_truth import f64_forward  # noqa: E402
from ribodetector_amd import synth  # noqa: E402
from ribodetector_amd.parse_config import ConfigParser  # noqa: E402

REFPKG = "/root/reference/ribodetector"
SETS = {   # name -> (reads, length spec, max_len, seed)
    "scale_se100": (100000, 100, 100, 31000),
    "scale_pe150": (20000, 150, 150, 31001),
    "scale_var300": (20000, (40, 300), 300, 31002),
}
SYNTH_KW = dict(rrna_frac=0.3, n_rate=0.002)


def stream(name):
    """(arena, off, lens, extra reads) of a set - also what the tests call to regenerate the reads"""
    n, length, max_len, seed = SETS[name]
    arena, off, lens = synth.reads_numpy(n, length, seed=seed, **SYNTH_KW)
    return arena, off, lens


def with_extra(arena, off, lens, extra):
    if not len(extra):
        return arena, off, lens
    ea = np.frombuffer(b"".join(extra), dtype=np.uint8)
    el = np.array([len(e) for e in extra], dtype=np.int32)
    a = np.concatenate([arena, ea])
    ln = np.concatenate([lens, el])
    o = np.zeros(len(ln) + 1, dtype=np.int64)
    np.cumsum(ln, out=o[1:])
    return a, o, ln


def ref_model():
    cfg = RefConfig.from_json(os.path.join(REFPKG, "config.json"))
    m = cfg.init_obj("arch", RM)
    m.load_state_dict(torch.load(os.path.join(REFPKG, cfg["state_file"]["mcc"]), map_location="cpu")["state_dict"])
    return m.eval()


def ref_logits(m, seqs, max_len, bs=2048):
    out = []
    with torch.no_grad():
        for i in range(0, len(seqs), bs):
            recs = [("@r%d" % k, s, "+", "I" * len(s)) for k, s in enumerate(seqs[i:i + bs])]
            _, x = R.unlabeled_read_collate_fn(recs, max_len=max_len, pack_seq=True)
            out.append(m(x))
    return torch.cat(out).numpy().astype(np.float32)


def stats(ref, truth):
    e = np.abs(ref.astype(np.float64) - truth).max(axis=1)
    worst = np.argsort(-e)[:32]
    return {"rms": float(np.sqrt((e ** 2).mean())), "median": float(np.median(e)), "p999": float(np.quantile(e, 0.999)),
            "p9999": float(np.quantile(e, 0.9999)), "max": float(e.max()), "n_over_5e-5": int((e > 5e-5).sum()),
            "n_over_1e-4": int((e > 1e-4).sum()), "min_margin_f64": float(np.abs(truth[:, 1] - truth[:, 0]).min()),
            "label_mismatches_vs_f64": int(((ref[:, 1] > ref[:, 0]) != (truth[:, 1] > truth[:, 0])).sum()),
            "worst": {"index": worst.tolist(), "f64": truth[worst].tolist(), "err": e[worst].tolist()}}


def main():
    torch.set_num_threads(8)
    m = ref_model()
    if "--extra-only" in sys.argv:     # add st["extra"]["ref_alone"] to an existing scale_se100.npz without re-running the 10^5 reads
        path = os.path.join(HERE, "scale_se100.npz")
        z = dict(np.load(path))
        st = json.loads(z["stats"].tobytes().decode())
        st["extra"]["ref_alone"] = [ref_logits(m, [bytes(r).decode()], 100)[0].tolist() for r in z["extra"]]
        z["stats"] = np.frombuffer(json.dumps(st).encode(), dtype=np.uint8)
        np.savez_compressed(path, **z)
        print(json.dumps(st["extra"], indent=1))
        return
    sd = ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json")).load_state_dict("mcc")
    for name, (n, length, max_len, seed) in SETS.items():
        arena, off, lens = stream(name)
        extra = []
        if name == "scale_se100":
            o = json.load(open(os.path.join(ROOT, "profiles", "r02_outlier.json")))
            extra = [o[k]["read"].encode() for k in sorted(o, key=int)]
            extra_idx = [int(k) for k in sorted(o, key=int)]
        a, of, ln = with_extra(arena, off, lens, extra)
        seqs = synth.as_strings(a, of)
        ref = ref_logits(m, seqs, max_len)
        truth = f64_forward(sd, a, of, ln, max_len)
        st = stats(ref, truth)
        st.update({"reads": int(len(ln)), "seeded_reads": n, "length": length, "max_len": max_len, "seed": seed, "synth_kwargs": SYNTH_KW,
                   "torch": torch.__version__, "what": "reference forward1 (torch CPU nn.LSTM) vs float64 evaluation of the same function"})
        if extra:
            st["extra"] = {"source": "profiles/r02_outlier.json: rounding-sensitive reads of synth.reads_torch(2**21, 100, seed=2026)",
                           "stream_index": extra_idx, "rows": list(range(n, n + len(extra))),
                           "ref": ref[n:].tolist(), "f64": truth[n:].tolist(),
                           # the same reads through the same reference call, each as a batch of ONE: another reduction order inside
                           # torch's LSTM / GEMM, i.e. the reference against itself
                           "ref_alone": [ref_logits(m, [e.decode()], max_len)[0].tolist() for e in extra]}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), ref=ref,
                            extra=np.frombuffer(b"".join(extra), dtype=np.uint8).reshape(len(extra), -1) if extra else np.zeros((0, 0), np.uint8),
                            sha256=np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8),
                            stats=np.frombuffer(json.dumps(st).encode(), dtype=np.uint8))
        print(name, {k: v for k, v in st.items() if k not in ("worst", "extra")}, flush=True)
        if extra:
            for i, r in enumerate(st["extra"]["rows"]):
                print("  extra read %d: reference %s  float64 %s  |diff| %.3g" % (extra_idx[i], ref[r].tolist(), truth[r].tolist(),
                                                                                 float(np.abs(ref[r] - truth[r]).max())))


if __name__ == "__main__":
    main()
