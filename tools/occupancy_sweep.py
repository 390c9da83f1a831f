#!/usr/bin/env python
"""Duration of the recurrence kernel alone (HIP events of rd_profile_*) as a function of the number of workgroups, per kernel
variant: separates what the STRUCTURE of a phase costs (few workgroups: the chip is far from its power cap and runs at its full
clock) from what the power cap costs (all 256 CUs busy for tens of milliseconds). Prints microseconds per phase of one
workgroup = duration / rounds / (2 L + 1), rounds = ceil(workgroups / 256).
    RD_HIP_LIB=ribodetector_amd/csrc/librd_hip_diag.so python tools/occupancy_sweep.py --variants mfma_f16x3_t32 mfma_f16x3_t32_diag_mfmaonly"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from ribodetector_amd import synth                                    # noqa: E402
from ribodetector_amd.model import model as M                         # noqa: E402
from ribodetector_amd.parse_config import ConfigParser                # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--wgs", type=int, nargs="*", default=[1, 8, 32, 64, 128, 256, 512, 2048, 16384])
    ap.add_argument("--variants", nargs="*", default=["mfma_f16x3_t32"])
    ap.add_argument("--reads-per-wg", type=int, default=64)
    a = ap.parse_args()
    L = a.len
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cfg = ConfigParser.from_json(os.path.join(root, "ribodetector_amd", "config.json"))
    model = cfg.init_obj("arch", M)
    model.load_state_dict(cfg.load_state_dict("mcc"))
    model.to("cuda:0").eval()
    model.set_refine(0.0)
    out = {}
    for var in a.variants:
        model.set_variant(var)
        out[var] = {}
        for w in a.wgs:
            n = w * a.reads_per_wg
            arena, off, lens = synth.reads_torch(n, L, seed=3, device=torch.device("cuda", 0))
            offs = off[:-1].contiguous()
            logits = torch.empty((n, 2), dtype=torch.float32, device="cuda:0")
            labels = torch.empty((n,), dtype=torch.uint8, device="cuda:0")
            reps = max(3, min(50, 4096 // w))
            for _ in range(2):
                model.classify_bytes(arena, offs, lens, L, logits=logits, labels=labels)
            torch.cuda.synchronize()
            model.profile_enable(True)
            for _ in range(reps):
                model.classify_bytes(arena, offs, lens, L, logits=logits, labels=labels)
            torch.cuda.synchronize()
            launches, ms = model.profile_read()
            model.profile_enable(False)
            us = ms * 1e3 / launches
            rounds = -(-w // 256)
            out[var][w] = {"kernel_us": round(us, 1), "us_per_phase": round(us / rounds / (2 * L + 1), 4)}
            print(var, w, out[var][w], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
