#!/usr/bin/env python
"""Where do the reads with a large fp32 logit error sit? Relates |kernel - float64| to quantities the kernel has at hand when it
finishes a read (its own logit margin, |logit|): is there a cheap detector of rounding-sensitive reads?
python tools/sens_probe.py [--reads N] [--len L] [--seed S]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(__file__))
from acc_experiment import f64_truth_gpu                              # noqa: E402
from ribodetector_amd import synth                                    # noqa: E402
from ribodetector_amd.model import model as M                         # noqa: E402
from ribodetector_amd.parse_config import ConfigParser                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1 << 22)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--seed", type=int, default=99)
    ap.add_argument("--rrna-frac", type=float, default=0.1)
    ap.add_argument("--variant", default="auto")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json"))
    sd = cfg.load_state_dict("mcc")
    model = cfg.init_obj("arch", M)
    model.load_state_dict(sd)
    model.to(dev).eval()
    model.set_variant(a.variant)
    n, L = a.reads, a.len
    arena, off, lens = synth.reads_torch(n, L, seed=a.seed, device=dev, rrna_frac=a.rrna_frac, n_rate=0.001)
    truth = f64_truth_gpu(sd, arena, L, dev)
    lg, _ = model.classify_bytes(arena, off[:-1].contiguous(), lens, L)
    e = (lg.double() - truth).abs().max(dim=1).values
    marg = (lg[:, 1] - lg[:, 0]).abs().double()
    tmarg = (truth[:, 1] - truth[:, 0]).abs()
    out = {"reads": n, "len": L, "variant": a.variant, "rrna_frac": a.rrna_frac, "err": {"rms": float((e ** 2).mean().sqrt()), "max": float(e.max()),
           "n_over_2e-5": int((e > 2e-5).sum()), "n_over_3e-5": int((e > 3e-5).sum()), "n_over_5e-5": int((e > 5e-5).sum()), "n_over_1e-4": int((e > 1e-4).sum())},
           "by_margin_window": [], "worst": []}
    for m in (0.1, 0.25, 0.5, 1.0, 2.0, 3.0, 4.0, 6.0, 8.0, 100.0):
        inside = marg < m
        eo = e[~inside]
        out["by_margin_window"].append({"window": m, "frac_inside": float(inside.double().mean()),
                                        "outside_max_err": float(eo.max()) if eo.numel() else 0.0,
                                        "outside_n_over_2e-5": int((eo > 2e-5).sum()), "outside_n_over_3e-5": int((eo > 3e-5).sum()),
                                        "outside_n_over_5e-5": int((eo > 5e-5).sum()),
                                        "inside_rms": float((e[inside] ** 2).mean().sqrt()) if inside.any() else 0.0})
    top = torch.topk(e, 30).indices
    for i in top.tolist():
        out["worst"].append({"err": float(e[i]), "kernel_margin": float(lg[i, 1] - lg[i, 0]), "truth_margin": float(truth[i, 1] - truth[i, 0]),
                             "logit0": float(lg[i, 0])})
    # decade histogram of the error among all reads vs among |margin| < 2
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
