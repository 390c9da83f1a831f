#!/usr/bin/env python
"""Which rounding source sets the logit-error tail of the split-precision kernel?

Runs kernel variants of librd_hip_diag.so (ACC bits of rd_lstm_t32.hpp) and the product kernels on N synthetic 100 bp reads
and compares each with a float64 evaluation of the same recurrence (torch float64 on the GPU) and with the fp32 CPU oracle.
  RD_HIP_LIB=ribodetector_amd/csrc/librd_hip_diag.so python tools/acc_experiment.py [--reads 1000000] [--len 100] [--oracle-reads 200000]
Prints one JSON document."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from ribodetector_amd import _native as N                             # noqa: E402
from ribodetector_amd import synth                                    # noqa: E402
from ribodetector_amd.model import model as M                         # noqa: E402
from ribodetector_amd.parse_config import ConfigParser                # noqa: E402


sys.path.insert(0, os.path.join(ROOT, "tests"))
from f64_truth import f64_forward_torch as f64_truth_gpu              # noqa: E402


def stats(e):
    e = e.abs().max(dim=1).values if e.dim() == 2 else e.abs()
    q = torch.quantile(e[: min(len(e), 1 << 24)].double(), torch.tensor([0.5, 0.99, 0.9999], dtype=torch.float64, device=e.device))
    return {"rms": float((e.double() ** 2).mean().sqrt()), "median": float(q[0]), "p99": float(q[1]), "p9999": float(q[2]), "max": float(e.max()),
            "n_over_5e-5": int((e > 5e-5).sum()), "n_over_1e-4": int((e > 1e-4).sum())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1 << 20)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--oracle-reads", type=int, default=200000)
    ap.add_argument("--variants", default="")
    ap.add_argument("--seed", type=int, default=4242)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json"))
    sd = cfg.load_state_dict("mcc")
    model = cfg.init_obj("arch", M)
    model.load_state_dict(sd)
    model.to(dev).eval()
    n, L = a.reads, a.len
    arena, off, lens = synth.reads_torch(n, L, seed=a.seed, device=dev, rrna_frac=0.3, n_rate=0.002)
    offs = off[:-1].contiguous()
    t0 = time.time()
    truth = f64_truth_gpu(sd, arena, L, dev)
    torch.cuda.synchronize()
    out = {"reads": n, "len": L, "lib": os.path.basename(N.LIB_PATH), "truth_s": time.time() - t0, "variants": {}}
    no = min(n, a.oracle_reads)
    ref = None
    if no:
        from oracle import oracle as O
        ora = O.load_default()
        t0 = time.time()
        ref = torch.from_numpy(ora.forward_packed(arena[: no * L].cpu().numpy(), off[: no + 1].cpu().numpy(), lens[:no].cpu().numpy(), L)).to(dev)
        out["oracle_s"] = time.time() - t0
        out["oracle_fp32_vs_f64"] = stats(ref.double() - truth[:no])
    names = [v for v in a.variants.split(",") if v] or [v for v in ("auto", "mfma_f32") if v in N.VARIANTS]
    for v in names:
        model.set_variant(v)
        model.classify_bytes(arena, offs, lens, L)
        model.profile_enable(True)
        lg, lab = model.classify_bytes(arena, offs, lens, L)
        torch.cuda.synchronize()
        _, ms = model.profile_read()
        model.profile_enable(False)
        rec = {"ms": ms, "vs_f64": stats(lg.double() - truth)}
        tl = (truth[:, 1] > truth[:, 0]).to(torch.uint8)
        bad = (lab != tl)
        rec["labels_vs_f64"] = {"mismatches": int(bad.sum()), "max_truth_margin_among_mismatches": float((truth[:, 1] - truth[:, 0]).abs()[bad].max()) if bad.any() else None}
        if ref is not None:
            rec["vs_oracle"] = stats(lg[:no] - ref)
        out["variants"][v] = rec
    model.set_variant("auto")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
