#!/usr/bin/env python
"""How many reads sit close enough to the decision boundary for fp32 rounding noise (~5e-5 on the logits at 100 bp) to flip
their label: distribution of |logit1 - logit0| over N synthetic reads. python tools/margin_stats.py [--reads 10000000]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from ribodetector_amd import synth                                    # noqa: E402
from ribodetector_amd.model import model as M                         # noqa: E402
from ribodetector_amd.parse_config import ConfigParser                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10000000)
    a = ap.parse_args()
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cfg = ConfigParser.from_json(os.path.join(root, "ribodetector_amd", "config.json"))
    model = cfg.init_obj("arch", M)
    model.load_state_dict(cfg.load_state_dict("mcc"))
    model.to("cuda:0").eval()
    dev = torch.device("cuda", 0)
    counts = {t: 0 for t in (1e-4, 2e-4, 1e-3, 1e-2, 1e-1)}
    flips = 0
    done = 0
    step = 1 << 20
    while done < a.reads:
        n = min(step, a.reads - done)
        arena, off, lens = synth.reads_torch(n, 100, seed=900 + done // step, device=dev)
        offs = off[:-1].contiguous()
        out = {}
        for v in ("auto", "mfma_f32"):
            model.set_variant(v)
            lg, lab = model.classify_bytes(arena, offs, lens, 100)
            out[v] = (lg.clone(), lab.clone())
        m = (out["mfma_f32"][0][:, 1] - out["mfma_f32"][0][:, 0]).abs()
        for t in counts:
            counts[t] += int((m < t).sum())
        flips += int((out["auto"][1] != out["mfma_f32"][1]).sum())
        done += n
    print(json.dumps({"reads": done, "reads_with_margin_below": {str(k): v for k, v in counts.items()},
                      "label_differences_default_vs_exact_fp32_kernel": flips}))


if __name__ == "__main__":
    main()
