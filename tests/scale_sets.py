"""The reference-at-scale fixtures (tests/golden/scale_*.npz, written by tests/golden/make_golden_scale.py from an import of the
reference): loader shared by tests/test_oracle.py and tests/test_gpu_scale.py. The reads regenerate from the recorded seed; the
fixture's sha256 of the read bytes guards the stream against drifting away from the logits."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ("scale_se100", "scale_pe150", "scale_var300")


class ScaleSet:
    def __init__(self, name):
        from ribodetector_amd import synth
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.name = name
        self.ref = z["ref"]                                  # the reference's own logits, float32 [n, 2]
        self.stats = json.loads(z["stats"].tobytes().decode())
        st = self.stats
        length = st["length"] if isinstance(st["length"], int) else tuple(st["length"])
        arena, off, lens = synth.reads_numpy(st["seeded_reads"], length, seed=st["seed"], **st["synth_kwargs"])
        extra = z["extra"]
        if extra.size:
            arena = np.concatenate([arena, extra.reshape(-1)])
            lens = np.concatenate([lens, np.full(extra.shape[0], extra.shape[1], dtype=np.int32)])
            off = np.zeros(len(lens) + 1, dtype=np.int64)
            np.cumsum(lens, out=off[1:])
        assert hashlib.sha256(arena.tobytes()).digest() == z["sha256"].tobytes(), "synthetic stream of %s changed: regenerate the fixture" % name
        self.arena, self.off, self.lens = arena, off, lens
        self.max_len = st["max_len"]
        self.n = len(lens)
        self.n_seeded = st["seeded_reads"]
        assert self.ref.shape == (self.n, 2)

    def subset(self, rows):
        """(arena, off, lens) of the given rows, repacked"""
        rows = np.asarray(rows, dtype=np.int64)
        lens = self.lens[rows]
        off = np.zeros(len(rows) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        arena = np.concatenate([self.arena[self.off[r]:self.off[r + 1]] for r in rows]) if len(rows) else np.zeros(0, np.uint8)
        return arena, off, lens


def err_stats(e):
    e = np.asarray(e, dtype=np.float64)
    return {"rms": float(np.sqrt((e ** 2).mean())), "p999": float(np.quantile(e, 0.999)), "p9999": float(np.quantile(e, 0.9999)),
            "max": float(e.max()), "n_over_5e-5": int((e > 5e-5).sum()), "n_over_1e-4": int((e > 1e-4).sum())}


def reference_tail():
    """What the relaxed bars of the large parity tests cite (reference-made data only, no reads regenerated):
    worst distance of the reference from float64 per fixture, and the largest distance of the reference from ITSELF (batch of
    2,048 vs batch of 1) on the rounding-sensitive reads of scale_se100."""
    out = {}
    for name in NAMES:
        st = json.loads(np.load(os.path.join(GOLDEN, name + ".npz"))["stats"].tobytes().decode())
        out[name] = {k: st[k] for k in ("rms", "p999", "p9999", "max", "n_over_1e-4", "reads")}
        if "extra" in st:
            ex = st["extra"]
            out["self_gap"] = float(np.abs(np.array(ex["ref"]) - np.array(ex["ref_alone"])).max())
            out["extra_max_vs_f64"] = float(max(np.abs(np.array(ex["ref"]) - np.array(ex["f64"])).max(),
                                                np.abs(np.array(ex["ref_alone"]) - np.array(ex["f64"])).max()))
    return out
