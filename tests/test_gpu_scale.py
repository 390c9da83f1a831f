"""The three recurrence kernels against THE REFERENCE at 10^5 reads (tests/golden/scale_*.npz: the reference's own forward1,
model/model.py:32-37, run in the build container by tests/golden/make_golden_scale.py) - not against the oracle, not against
the builder's float64 yardstick alone.

How the bars are set (every relaxation must cite reference-made data):
  * `stats` inside each fixture is the distance of the REFERENCE from a float64 evaluation of the same function: at 100 bp
    2.4e-6 rms, 1.0e-5 at p99.9, 1.6e-5 at p99.99, worst read 1.5e-4 (tests/test_oracle.py asserts these from the fixture).
  * a kernel is held (a) to the north star's 1e-4 against the reference on every seeded read where that is reachable for two fp32
    evaluations (it is, on all 140,000 seeded reads here), (b) to the reference's OWN noise level against float64 - rms and high
    quantiles no worse than the reference's by more than a stated factor - and (c) to identical labels;
  * the five rounding-sensitive reads appended to the 100 bp set (error amplification ~100x: the reference differs from itself
    by 3.7e-4 on one of them when the batch size changes) are held to 1e-3 and to identical labels."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from f64_truth import f64_forward_torch_varlen   # noqa: E402
from scale_sets import NAMES, ScaleSet, err_stats   # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
VARIANTS = ("auto", "mfma_f32", "simple")


@pytest.fixture(scope="module")
def sets():
    from ribodetector_amd.parse_config import ConfigParser
    sd = ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json")).load_state_dict("mcc")
    out = {}
    for name in NAMES:
        s = ScaleSet(name)
        s.f64 = f64_forward_torch_varlen(sd, s.arena, s.off, s.lens, s.max_len, "cuda:0")
        out[name] = s
    return out


def _classify(model, s, variant, refine):
    dev = "cuda:0"
    a = torch.from_numpy(s.arena).to(dev)
    o = torch.from_numpy(s.off[:-1].copy()).to(dev)
    l = torch.from_numpy(s.lens).to(dev)
    model.set_variant(variant)
    model.set_refine(refine)
    try:
        lg, lab = model.classify_bytes(a, o, l, s.max_len)
        return lg.cpu().numpy(), lab.cpu().numpy()
    finally:
        model.set_variant("auto")
        model.set_refine(model.REFINE_DEFAULT)


def test_float64_yardstick_matches_the_fixture(sets):
    """the float64 values this box computes on its GPU are the ones the fixture recorded in the build container"""
    for name, s in sets.items():
        w = s.stats["worst"]
        assert np.abs(s.f64[w["index"]] - np.array(w["f64"])).max() < 1e-9, name
        e_ref = np.abs(s.ref.astype(np.float64) - s.f64).max(axis=1)
        st = err_stats(e_ref)
        for k in ("rms", "p999", "p9999", "max"):
            assert abs(st[k] - s.stats[k]) < 1e-9 + 1e-6 * s.stats[k], (name, k, st[k], s.stats[k])


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name", NAMES)
def test_kernel_against_the_reference_at_scale(gpu_model, sets, report, name, variant):
    s = sets[name]
    ns = s.n_seeded
    lg, lab = _classify(gpu_model, s, variant, 0.0)                  # the raw kernel (no float64 pass): this is a logit test
    e_ref = np.abs(lg.astype(np.float64) - s.ref).max(axis=1)        # kernel vs the reference's own logits
    e_f64 = np.abs(lg.astype(np.float64) - s.f64).max(axis=1)        # kernel vs the exact value
    r_f64 = np.abs(s.ref.astype(np.float64) - s.f64).max(axis=1)     # reference vs the exact value (= the fixture's stats)
    a, b, c = err_stats(e_ref[:ns]), err_stats(e_f64[:ns]), err_stats(r_f64[:ns])
    ref_lab = (s.ref[:, 1] > s.ref[:, 0]).astype(np.uint8)
    mism = np.flatnonzero(lab != ref_lab)
    report["scale/%s/%s" % (name, variant)] = {"vs_reference": a, "vs_float64": b, "reference_vs_float64": c, "label_mismatches": int(len(mism)),
                                              "extra_vs_reference": e_ref[ns:].tolist(), "extra_vs_float64": e_f64[ns:].tolist()}
    # (a) the north star's bar against the reference, on every seeded read
    assert a["max"] < 1e-4, (name, variant, a)
    assert a["rms"] < 6e-6 and a["p999"] < 5e-5, (name, variant, a)   # observed: rms 2.8e-6 ... 4.8e-6, p99.9 1.4e-5 ... 4.1e-5
    # (b) no further from the exact function than the reference itself is (its own tail: fixture stats)
    loose = 2.0 if variant == "simple" else 1.35                      # the plain-FMA cross-check accumulates k-ascending: a longer chain
    assert b["rms"] < loose * c["rms"] and b["p999"] < loose * (1.5 if variant == "simple" else 1.2) * c["p999"], (name, variant, b, c)
    assert b["n_over_1e-4"] <= c["n_over_1e-4"] + (0 if s.max_len <= 100 else 2), (name, variant, b, c)
    # (c) labels: identical to the reference's on every read (the smallest float64 margin of the sets is 1.5e-5; a mismatch would
    # have to sit inside the two noises, and then it is reported, not hidden)
    assert len(mism) == 0 or np.abs(s.f64[mism, 1] - s.f64[mism, 0]).max() < 2e-5, (name, variant, mism[:5])
    # the rounding-sensitive reads
    if s.n > ns:
        assert e_ref[ns:].max() < 1e-3 and (lab[ns:] == ref_lab[ns:]).all()


def test_refined_labels_equal_reference_labels(gpu_model, sets):
    """the product path (default kernel + float64 re-evaluation inside the noise band): labels = the reference's on all 140,005 reads"""
    for name, s in sets.items():
        lg, lab = _classify(gpu_model, s, "auto", gpu_model.REFINE_DEFAULT)
        assert (lab == (s.ref[:, 1] > s.ref[:, 0])).all(), name
        assert (lab == (s.f64[:, 1] > s.f64[:, 0])).all(), name
