#!/usr/bin/env python3
"""Re-serialise the reference checkpoint's `state_dict` (10 fp32 tensors, 137,730 params)
into ribodetector_amd/data/*.safetensors.

Runs ONLY in the build container (needs /root/reference). The reference loads the same
tensors with torch.load(...)['state_dict'] (reference detect.py:101,115-116); optimizer
state / metrics in the .pth are training artefacts and are dropped.
The weights are data, not code; tensor names and shapes are kept verbatim so that
`SeqModel.load_state_dict` accepts either file.
"""
import hashlib
import json
import os
import sys

import torch
from safetensors.torch import save_file

SRC = "/root/reference/ribodetector/data/ribodetector_600k_variable_len70_101_epoch47.pth"
DST_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ribodetector_amd", "data")
DST = os.path.join(DST_DIR, "ribodetector_600k_variable_len70_101_epoch47.safetensors")


def main():
    raw = open(SRC, "rb").read()
    sha = hashlib.sha256(raw).hexdigest()
    state = torch.load(SRC, map_location="cpu")
    sd = {k: v.contiguous().float() for k, v in state["state_dict"].items()}
    meta = {
        "source": os.path.basename(SRC),
        "source_sha256": sha,
        "epoch": str(state.get("epoch")),
        "arch": str(state.get("arch")),
    }
    os.makedirs(DST_DIR, exist_ok=True)
    save_file(sd, DST, metadata=meta)
    digests = {k: hashlib.sha256(v.numpy().tobytes()).hexdigest() for k, v in sd.items()}
    with open(os.path.join(DST_DIR, "weights_digest.json"), "w") as fh:
        json.dump({"source_sha256": sha, "tensors": {k: {"shape": list(v.shape), "sha256": digests[k]}
                                                     for k, v in sd.items()}}, fh, indent=1)
    print("wrote", DST, os.path.getsize(DST), "bytes; source sha256", sha)


if __name__ == "__main__":
    sys.exit(main())
