"""The product recurrence kernel's instruction stream is pinned (tools/isa_pin.py): the numbers under profiles/ were measured on ONE
hand-scheduled ISA, so a source change that alters it must come with refreshed profiles and a refreshed pin - it fails here
otherwise. Cross-compiles with hipcc (no GPU needed)."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_default_kernel_isa_matches_the_pin():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_pin
    pinned = json.load(open(isa_pin.PIN))
    now = isa_pin.fingerprint()
    assert now == pinned, ("the ISA of rd_lstm_mfma_f16x3_t32_kernel changed: re-measure (tools/profile_round.sh), commit the new profiles and run "
                           "`python tools/isa_pin.py --update`", now, pinned)
    assert pinned["t32_classify"]["mfma"] >= 2 * 96      # two phases of 96 MFMAs in the loop body
