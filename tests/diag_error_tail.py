"""Not a test (not collected): tail of the logit error of the HIP kernels and of the fp32 oracle itself against a float64
evaluation of the same recurrence. Run on the GPU box: python tests/diag_error_tail.py [reads] [max_len]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from oracle import oracle as O                                       # noqa: E402
from ribodetector_amd import synth                                   # noqa: E402
from ribodetector_amd.data_loader import seq_encoder as E            # noqa: E402
from ribodetector_amd.model import model as M                        # noqa: E402
from ribodetector_amd.parse_config import ConfigParser               # noqa: E402
sys.path.insert(0, os.path.dirname(__file__))
from f64_truth import f64_forward                                     # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    cfg = ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json"))
    sd = cfg.load_state_dict("mcc")
    model = cfg.init_obj("arch", M)
    model.load_state_dict(sd)
    model.to("cuda:0").eval()
    ora = O.load_default()
    arena, off, lens = synth.reads_numpy(n, (max(1, L - 60), L + 20), seed=4242, rrna_frac=0.3, n_rate=0.01)
    truth = f64_forward(sd, arena, off, lens, L)
    ref = ora.forward_packed(arena, off, lens, L).astype(np.float64)
    out = {"reads": n, "max_len": L}

    def stats(x):
        e = np.abs(x - truth).max(axis=1)
        return {"max": float(e.max()), "p9999": float(np.quantile(e, 0.9999)), "p99": float(np.quantile(e, 0.99)),
                "median": float(np.median(e)), "n_over_1e-4": int((e > 1e-4).sum()), "n_over_5e-5": int((e > 5e-5).sum())}
    out["oracle_fp32_vs_f64"] = stats(ref)
    b = E.batch_from_numpy(arena, off[:-1], lens, "cuda")
    for v in ("auto", "mfma_f32", "simple"):
        model.set_variant(v)
        lg, _ = model.classify_bytes(b.arena, b.offsets, b.lens, L)
        lg = lg.cpu().numpy().astype(np.float64)
        out[v + "_vs_f64"] = stats(lg)
        e = np.abs(lg - ref).max(axis=1)
        out[v + "_vs_oracle"] = {"max": float(e.max()), "p9999": float(np.quantile(e, 0.9999)), "n_over_1e-4": int((e > 1e-4).sum())}
        if v == "auto":
            w = int(np.argmax(e))
            out["worst_read"] = {"index": w, "len": int(lens[w]), "truth": truth[w].tolist(), "oracle": ref[w].tolist(), "kernel": lg[w].tolist()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
