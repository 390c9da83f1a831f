#!/usr/bin/env python3
"""Rounding noise of the REFERENCE ITSELF (build container only: imports /root/reference like tests/golden/make_golden.py).

The reference's forward1 (model/model.py:32-37, torch CPU nn.LSTM) is compared with a float64 evaluation of the same function
(tests/f64_truth.py) on seeded synthetic 100 bp reads, next to the CPU oracle (oracle/rd_oracle.c) on the same reads - to show
that the oracle's noise IS the reference's noise - and on the rounding-sensitive read recorded in profiles/r02_outlier.json.
Writes profiles/r02_reference_noise.json.   python tools/reference_noise.py [reads]"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))
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
from oracle import oracle as O  # noqa: E402
from ribodetector_amd import synth  # noqa: E402
from ribodetector_amd.parse_config import ConfigParser  # noqa: E402

REFPKG = "/root/reference/ribodetector"


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
    return torch.cat(out).numpy().astype(np.float64)


def stats(e):
    e = np.abs(e).max(axis=1)
    return {"rms": float(np.sqrt((e ** 2).mean())), "median": float(np.median(e)), "p9999": float(np.quantile(e, 0.9999)), "max": float(e.max()),
            "n_over_5e-5": int((e > 5e-5).sum()), "n_over_1e-4": int((e > 1e-4).sum())}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    torch.set_num_threads(8)
    m = ref_model()
    sd = ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json")).load_state_dict("mcc")
    ora = O.load_default()
    arena, off, lens = synth.reads_numpy(n, 100, seed=4242, rrna_frac=0.3, n_rate=0.002)
    seqs = synth.as_strings(arena, off)
    ref = ref_logits(m, seqs, 100)
    orc = ora.forward_packed(arena, off, lens, 100).astype(np.float64)
    truth = f64_forward(sd, arena, off, lens, 100)
    out = {"reads": n, "len": 100, "torch": torch.__version__,
           "reference_torch_cpu_vs_f64": stats(ref - truth), "oracle_vs_f64": stats(orc - truth), "oracle_vs_reference": stats(orc - ref)}
    pj = os.path.join(ROOT, "profiles", "r02_outlier.json")
    if os.path.exists(pj):
        o = json.load(open(pj))
        k = sorted(o, key=lambda x: -abs(o[x]["truth_gpu_f64"][0] - o[x]["oracle_fp32"][0]))[0]
        read = o[k]["read"]
        a = np.frombuffer(read.encode(), dtype=np.uint8)
        out["outlier_read"] = {"index": int(k), "read": read, "reference_torch_cpu": ref_logits(m, [read], 100)[0].tolist(),
                               "float64": f64_forward(sd, a, np.array([0, 100]), np.array([100], dtype=np.int32), 100)[0].tolist(),
                               "oracle_fp32": o[k]["oracle_fp32"], "default_kernel": o[k].get("auto/refine=0.00025") or o[k].get("auto/refine=0")}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r02_reference_noise.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
