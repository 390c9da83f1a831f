#!/usr/bin/env python
"""Energy model of one phase of the default kernel: instruction counts from the ISA (`hipcc -S`) x the per-instruction
prices measured at the package power cap (tools/ubench/op_energy.hip, profiles/r02_ubench.txt), in units of one bare
v_mfma_f32_32x32x16_f16. At the cap time is proportional to energy, so the model predicts the ratio of the product kernel to its
MFMA-only variant (measured: profiles/r02_variants_ab2.txt) and what removing a group of instructions can be worth.

    python tools/energy_model.py            (runs hipcc -S on the kernels: needs no GPU)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
# price per instruction in units of one bare 32x32x16 f16 MFMA (two boxes, profiles/r02_ubench.txt); instructions not listed
# (scalar ALU, branches, waits; the integer / select instructions of the rarely taken code-staging branch that sits between the two
# phases, and a dozen address computations) are taken as free
PRICE = {
    "v_exp_f32": 0.041, "v_rcp_f32": 0.033, "v_fma_f32": 0.038, "v_fmac_f32": 0.038, "v_fmamk_f32": 0.038, "v_fmaak_f32": 0.038,
    "v_add_f32": 0.036, "v_sub_f32": 0.036, "v_mul_f32": 0.037, "v_min_f32": 0.036, "v_max_f32": 0.036, "v_fma_mix_f32": 0.038,
    "v_mov_b32": 0.026, "v_cvt_pk_f16_f32": 0.025,
    "v_pk_fma_f32": 0.19, "ds_read_b128": 0.105, "ds_read_b64": 0.08, "ds_read_b32": 0.06, "ds_read_u8": 0.06, "ds_write_b128": 0.10,
    "ds_write_b64": 0.085, "ds_write2st64_b64": 0.12, "ds_write2_b32": 0.085, "ds_write_b32": 0.06,
}


def main():
    src = os.path.join(ROOT, "ribodetector_amd", "csrc", "rd_kernels.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-mllvm",
                               "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only",
                               src, "-o", out])
        text = open(out).read()
    lines = text.split("\n")
    start = next((i for i, l in enumerate(lines) if re.match(r"^_ZN\S*rd_lstm_mfma_f16x3_t32_kernelILi6ELi240EE\S*:", l)), None)
    if start is None:
        sys.exit("product kernel not found in the ISA")
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start + 1:end]
    bars = [i for i, l in enumerate(body) if l.strip().startswith("s_barrier")]
    # the phase loop = the two longest barrier-to-barrier stretches that hold 96 MFMAs each
    phases = []
    for a, b in zip(bars, bars[1:]):
        seg = body[a + 1:b]
        if sum("v_mfma_f32_32x32x16_f16" in l for l in seg) == 96:
            phases.append(seg)
    if not phases:
        sys.exit("no 96-MFMA phase found")
    # (the first stretch also holds the loop's pre-header - 190 v_mov_b32 that zero the accumulators of a tile; the other one holds the
    #  rarely taken code-staging branch, a dozen instructions)
    seg = min(phases, key=lambda g: sum("v_mov_b32" in l for l in g))
    cnt = collections.Counter()
    for l in seg:
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
            continue
        op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", t[0])
        cnt[op] += 1
    mfma = cnt["v_mfma_f32_32x32x16_f16"]
    groups = collections.OrderedDict()
    groups["MFMA"] = float(mfma)
    groups["transcendentals"] = sum(PRICE[k] * cnt[k] for k in ("v_exp_f32", "v_rcp_f32"))
    groups["plain VALU"] = sum(PRICE[k] * v for k, v in cnt.items() if k.startswith("v_") and k in PRICE and k not in ("v_exp_f32", "v_rcp_f32"))
    groups["LDS"] = sum(PRICE[k] * v for k, v in cnt.items() if k.startswith("ds_") and k in PRICE)
    total = sum(groups.values())
    print("instructions of one phase (per wave):")
    for k in sorted(cnt, key=lambda k: -cnt[k]):
        if k in PRICE or k.startswith("v_mfma"):
            print("  %-26s %4d   x %.3f" % (k, cnt[k], 1.0 if k.startswith("v_mfma") else PRICE[k]))
    unpriced = {k: v for k, v in cnt.items() if k not in PRICE and not k.startswith("v_mfma") and (k.startswith("v_") or k.startswith("ds_"))}
    if unpriced:
        print("  (vector / LDS instructions without a measured price, taken as free: %s)" % ", ".join("%s x%d" % kv for kv in sorted(unpriced.items())))
    print("energy of a phase in MFMA-equivalents:")
    for k, v in groups.items():
        print("  %-18s %6.1f" % (k, v))
    lds_b = PRICE["ds_read_b128"] * 16   # the B fragments stay in the MFMA-only variant
    print("  total %.1f = %.2f x the MFMAs alone; against the MFMA-only variant (MFMAs + the 16 B-fragment reads): %.2f (measured 1.23-1.24)"
          % (total, total / mfma, total / (mfma + lds_b)))
    print("  worth of removing: the 16 table-row reads %.1f %%, the 64 pre-activation FMAs %.1f %%, one VALU op per cell (16) %.2f %%"
          % (100 * PRICE["ds_read_b128"] * 16 / total, 100 * PRICE["v_fmamk_f32"] * 64 / total, 100 * 0.036 * 16 / total))


if __name__ == "__main__":
    main()
