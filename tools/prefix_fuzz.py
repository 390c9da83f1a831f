#!/usr/bin/env python
"""Differential fuzz of the prefix-state table on the GPU: random reads (lengths 0..220, A/C/G/T with N, lower case, U and IUPAC
letters sprinkled in, runs of N at either end), random -l, random k in 4..13, both semantics, random batch sizes - the logits and
labels with the table must equal those without it, bit for bit (tests/test_gpu_prefix.py holds the fixed cases).
python tools/prefix_fuzz.py [rounds]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from ribodetector_amd.model import model as M                         # noqa: E402
from ribodetector_amd.parse_config import ConfigParser                # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    cfg = ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json"))
    m = cfg.init_obj("arch", M)
    m.load_state_dict(cfg.load_state_dict("mcc"))
    m.set_prefix_table(0)
    m.to("cuda:0").eval()
    rng = np.random.default_rng(2027)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    odd = np.frombuffer(b"NacgtURYKMn-*", dtype=np.uint8)
    t0, total = time.time(), 0
    for r in range(rounds):
        n = int(rng.choice([1, 63, 64, 65, 500, 4097, 20000]))
        lens = rng.integers(0, 221, n).astype(np.int32)
        if r % 5 == 0:
            lens[:] = int(rng.integers(1, 221))                   # a fixed-length batch
        off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        arena = alpha[rng.integers(0, 4, int(off[-1]))].copy()
        bad = rng.random(len(arena)) < rng.choice([0.0, 0.002, 0.05])
        arena[bad] = odd[rng.integers(0, len(odd), int(bad.sum()))]
        for i in rng.integers(0, n, max(1, n // 20)):               # runs of N at the start / the end of some reads
            L = int(lens[i])
            if L:
                k = int(rng.integers(0, L + 1))
                if rng.random() < 0.5:
                    arena[off[i]:off[i] + k] = ord("N")
                else:
                    arena[off[i + 1] - k:off[i + 1]] = ord("N")
        a = torch.from_numpy(arena).cuda() if len(arena) else torch.zeros(1, dtype=torch.uint8, device="cuda")
        o = torch.from_numpy(off[:-1].copy()).cuda()
        ln = torch.from_numpy(lens).cuda()
        max_len = int(rng.choice([1, 12, 13, 14, 37, 64, 65, 100, 128, 129, 150, 220, 300]))
        sem = "padded" if r % 2 else "packed"
        m.set_semantics(sem)
        m.set_prefix_table(0)
        ref, rlab = (t.clone() for t in m.classify_bytes(a, o, ln, max_len))
        for k in sorted(set(int(x) for x in rng.integers(4, 14, 2))):
            m.set_prefix_table(k)
            lg, lab = m.classify_bytes(a, o, ln, max_len)
            if not (torch.equal(lg, ref) and torch.equal(lab, rlab)):
                bad_rows = (lg != ref).any(1).nonzero().flatten()[:5].tolist()
                print("MISMATCH round %d n %d max_len %d sem %s k %d rows %s" % (r, n, max_len, sem, k, bad_rows))
                return 1
        total += n
    print("prefix fuzz: %d rounds, %d reads, every table start bit-identical (%.1f s)" % (rounds, total, time.time() - t0))
    return 0


if __name__ == "__main__":
    sys.exit(main())
