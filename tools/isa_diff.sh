#!/bin/bash
# tools/isa_diff.sh - is the product recurrence kernel instruction-identical between two states of the source?
#
#   tools/isa_diff.sh <git-rev>            compare <git-rev> with the working tree
#   tools/isa_diff.sh <a.s> <b.s>          compare two `hipcc -S --cuda-device-only` outputs
#
# Both sides are compiled with the flags of __graft_entry__.HIP_FLAGS; of each ISA file only the body of the kernel whose
# name contains rd_lstm_mfma_f16x3_t32_kernel is kept (first instantiation), comments are dropped and basic-block labels are
# renumbered in order of appearance (function order and template arguments may differ). Exit code 0 = identical.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
KERNEL="${KERNEL:-rd_lstm_mfma_f16x3_t32_kernel}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -S --cuda-device-only"
compile() {   # <source root> <out.s>
    /opt/rocm/bin/hipcc $FLAGS -I "$1/include" "$1/ribodetector_amd/csrc/rd_kernels.hip" -o "$2" 2>/dev/null
}
body() {      # <file.s> : instructions of the kernel, normalised
    python3 - "$1" "$KERNEL" <<'PY'
import re, sys
path, kern = sys.argv[1], sys.argv[2]
out, on, labels = [], False, {}
for line in open(path):
    if not on:
        if re.match(r"^_Z\w*%s\w*:" % re.escape(kern), line):
            on = True
        continue
    if line.startswith("\t.amdhsa_kernel") or re.match(r"^\.Lfunc_end", line):   # the kernel descriptor carries the (mangled) name
        break
    line = line.split(";")[0].rstrip()
    if not line.strip() or line.lstrip().startswith((".p2align", ".section", ".type", ".size", ".globl", ".protected")):
        continue
    line = re.sub(r"\.LBB\d+_(\d+)", lambda m: ".L%d" % labels.setdefault(m.group(0), len(labels)), line)
    out.append(line)
sys.stdout.write("\n".join(out) + "\n")
PY
}
TMP="$(mktemp -d /tmp/isa_diff.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
if [ $# -eq 1 ]; then
    mkdir -p "$TMP/old"
    git -C "$ROOT" archive "$1" include ribodetector_amd/csrc | tar -x -C "$TMP/old"
    compile "$TMP/old" "$TMP/a.s"
    compile "$ROOT" "$TMP/b.s"
    A="$TMP/a.s"; B="$TMP/b.s"
else
    A="$1"; B="$2"
fi
body "$A" > "$TMP/a.body"
body "$B" > "$TMP/b.body"
na=$(wc -l < "$TMP/a.body"); nb=$(wc -l < "$TMP/b.body")
if [ "$na" -eq 0 ] || [ "$nb" -eq 0 ]; then echo "kernel $KERNEL not found ($na / $nb lines)"; exit 2; fi
if diff -q "$TMP/a.body" "$TMP/b.body" > /dev/null; then
    echo "IDENTICAL: $KERNEL, $na instruction lines on both sides"
else
    echo "DIFFERENT: $KERNEL, $na vs $nb lines; first differences:"
    diff "$TMP/a.body" "$TMP/b.body" | head -40
    exit 1
fi
