#!/usr/bin/env python3
"""isa_pin.py - the product recurrence kernel's ISA as a fingerprint.

The default kernel (csrc/rd_lstm_t32.hpp, rd_lstm_mfma_f16x3_t32_kernel<false>) is hand-scheduled and frozen: its measured numbers
under profiles/ belong to ONE instruction stream. This tool compiles the translation unit to ISA with the build's own flags
(cross-compiles on a box without a GPU, ~15 s), keeps the kernel's body, normalises it like tools/isa_diff.sh (comments dropped,
basic-block labels renumbered) and hashes it.

    python tools/isa_pin.py            print the fingerprint of the working tree and compare it with profiles/t32_isa_pin.json
    python tools/isa_pin.py --update   write profiles/t32_isa_pin.json (do this together with refreshed profiles of the new kernel)

tests/test_isa_pin.py fails when the fingerprint differs from the committed one: a change of the kernel's instructions has to come
with new measurements, not ride along unnoticed."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PIN = os.path.join(ROOT, "profiles", "t32_isa_pin.json")
KERNELS = {"t32_classify": r"rd_lstm_mfma_f16x3_t32_kernelILb0E", "t32_table_build": r"rd_lstm_mfma_f16x3_t32_kernelILb1E"}


def compile_isa():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    flags = [f for f in G.HIP_FLAGS if f not in ("-fPIC", "-shared")]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
                               os.path.join(G.CSRC, "rd_kernels.hip"), "-o", out], stderr=subprocess.DEVNULL)
        return open(out).read()


def body(isa, name_re):
    out, on, labels = [], False, {}
    for line in isa.splitlines():
        if not on:
            if re.match(r"^_Z\w*%s\w*:" % name_re, line):
                on = True
            continue
        if line.startswith("\t.amdhsa_kernel") or re.match(r"^\.Lfunc_end", line):
            break
        line = line.split(";")[0].rstrip()
        if not line.strip() or line.lstrip().startswith((".p2align", ".section", ".type", ".size", ".globl", ".protected")):
            continue
        line = re.sub(r"\.str(\.\d+)?@", ".str@", line)       # (the index of a string constant depends on the OTHER kernels of the translation unit)
        out.append(re.sub(r"\.LBB\d+_(\d+)", lambda m: ".L%d" % labels.setdefault(m.group(0), len(labels)), line))
    return out


def fingerprint():
    isa = compile_isa()
    rec = {}
    for key, name in KERNELS.items():
        b = body(isa, name)
        if not b:
            raise SystemExit("kernel %s not found in the ISA" % name)
        rec[key] = {"instruction_lines": len(b), "mfma": sum("v_mfma" in l for l in b), "sha256": hashlib.sha256("\n".join(b).encode()).hexdigest()}
    return rec


def main():
    now = fingerprint()
    if "--update" in sys.argv:
        json.dump(now, open(PIN, "w"), indent=1, sort_keys=True)
        print("wrote", PIN)
    print(json.dumps(now, indent=1, sort_keys=True))
    if os.path.exists(PIN):
        same = json.load(open(PIN)) == now
        print("matches profiles/t32_isa_pin.json" if same else "DIFFERS from profiles/t32_isa_pin.json")
        return 0 if same else 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
