#!/usr/bin/env python
"""reads/s of SeqModel.classify_bytes as a function of the reads per call (100 bp, inputs in HBM): where launch overhead
and the 256-workgroup fill of the chip start to matter. python tools/batch_sweep.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from ribodetector_amd import synth                                    # noqa: E402
from ribodetector_amd.model import model as M                         # noqa: E402
from ribodetector_amd.parse_config import ConfigParser                # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--sizes", type=int, nargs="*", default=[1024, 4096, 16384, 32768, 65536, 262144, 1048576])
    a = ap.parse_args()
    L = a.len
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    cfg = ConfigParser.from_json(os.path.join(root, "ribodetector_amd", "config.json"))
    model = cfg.init_obj("arch", M)
    model.load_state_dict(cfg.load_state_dict("mcc"))
    model.to("cuda:0").eval()
    out = {}
    for n in a.sizes:
        arena, off, lens = synth.reads_torch(n, L, seed=3, device=torch.device("cuda", 0))
        offs = off[:-1].contiguous()
        logits = torch.empty((n, 2), dtype=torch.float32, device="cuda:0")
        labels = torch.empty((n,), dtype=torch.uint8, device="cuda:0")
        reps = max(3, min(200, (1 << 22) // n))
        for _ in range(3):
            model.classify_bytes(arena, offs, lens, L, logits=logits, labels=labels)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            model.classify_bytes(arena, offs, lens, L, logits=logits, labels=labels)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        out[n] = {"us_per_call": dt * 1e6, "reads_per_s": n / dt, "read_steps_per_s": n * L / dt}
        # the float64 pass: inline (above), switched off, and deferred (rd_set_refine_async: the candidates of 1 / 8 consecutive calls
        # recorded in the model's queue and evaluated together on its own stream; ten sets of buffers in rotation, as the contract
        # of the mode asks; the two batches alternate, so about every call of 65,536 reads holds a candidate)
        a2, _, _ = synth.reads_torch(n, L, seed=4, device=torch.device("cuda", 0))
        NSETS = 18                                        # deferred mode with groups of up to 16 calls: results consumed 16 calls later
        sets = [((arena if i % 2 == 0 else a2).clone(), offs.clone(), lens.clone(), torch.empty_like(logits), torch.empty_like(labels))
                for i in range(NSETS)]

        def timed_calls(k=reps):
            for i in range(3):
                s = sets[i % NSETS]
                model.classify_bytes(s[0], s[1], s[2], L, logits=s[3], labels=s[4])
            model.sync_results()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(k):
                s = sets[i % NSETS]
                model.classify_bytes(s[0], s[1], s[2], L, logits=s[3], labels=s[4])
            model.sync_results()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k
        d_inline = timed_calls()
        model.set_refine(0.0)
        d_off = timed_calls()
        model.set_refine(M.SeqModel.REFINE_DEFAULT)
        model.set_refine_async(1)
        d_async1 = timed_calls()
        model.set_refine_async(8)
        d_async8 = timed_calls()
        model.set_refine_async(16)
        d_async16 = timed_calls()
        model.set_refine_async(0)
        out[n].update({"us_inline": d_inline * 1e6, "us_refine_off": d_off * 1e6, "us_deferred_groups_of_1": d_async1 * 1e6,
                       "us_deferred_groups_of_8": d_async8 * 1e6, "inline_over_off": d_inline / d_off,
                       "us_deferred_groups_of_16": d_async16 * 1e6, "deferred1_over_off": d_async1 / d_off,
                       "deferred8_over_off": d_async8 / d_off, "deferred16_over_off": d_async16 / d_off})
        # the same call captured in a hipGraph (every launch of rd_classify is asynchronous on the caller's stream, so it can be
        # captured as is): what the seven launches per call cost at small batches
        for refine in (True, False):
            model.set_refine(M.SeqModel.REFINE_DEFAULT if refine else 0.0)
            model.classify_bytes(arena, offs, lens, L, logits=logits, labels=labels)      # workspace allocated outside the capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                model.classify_bytes(arena, offs, lens, L, logits=logits, labels=labels)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                g.replay()
            torch.cuda.synchronize()
            dg = (time.perf_counter() - t) / reps
            out[n]["graph_us_per_call" + ("" if refine else "_no_refine")] = dg * 1e6
            out[n]["graph_reads_per_s" + ("" if refine else "_no_refine")] = n / dg
        model.set_refine(M.SeqModel.REFINE_DEFAULT)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
