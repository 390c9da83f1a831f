#!/usr/bin/env python
"""Achieved HBM bandwidth of the standalone encoder / label kernels: algorithmic bytes / kernel time.
python tools/encoder_bench.py [--reads 1048576] [--len 100]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from ribodetector_amd import synth                                    # noqa: E402
from ribodetector_amd import _native as N                             # noqa: E402
from ribodetector_amd.data_loader import seq_encoder as E             # noqa: E402
from ribodetector_amd.model import model as M                         # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1 << 20)
    ap.add_argument("--len", type=int, default=100)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n, L = a.reads, a.len
    arena, off, lens = synth.reads_torch(n, L, seed=1, device=dev)
    batch = E.ReadBatch(arena, off[:-1].contiguous(), lens)
    lib, st = N.lib(), N.stream_ptr(dev)
    out = {"reads": n, "len": L, "peak_GBps": 8000}

    codes = torch.empty((n, L), dtype=torch.uint8, device=dev)
    t = timed(lambda: N.check(lib.rd_encode_codes(N.ptr(batch.arena), N.ptr(batch.offsets), N.ptr(batch.lens), n, L, L, N.ptr(codes), st), "codes"))
    out["encode_codes"] = {"ms": t * 1e3, "GBps": n * (2 * L + 12) / t / 1e9}

    oh = torch.empty((n, L, 4), dtype=torch.float32, device=dev)
    t = timed(lambda: N.check(lib.rd_encode_onehot_padded(N.ptr(batch.arena), N.ptr(batch.offsets), N.ptr(batch.lens), n, L, N.ptr(oh), st), "onehot"))
    out["encode_onehot_padded"] = {"ms": t * 1e3, "GBps": n * (17 * L + 12) / t / 1e9}

    ws = torch.empty(int(lib.rd_classify_workspace_bytes(n, L)), dtype=torch.uint8, device=dev)
    si = torch.empty(n, dtype=torch.int64, device=dev)
    ui = torch.empty(n, dtype=torch.int64, device=dev)
    bs = torch.empty(L, dtype=torch.int64, device=dev)
    tot = torch.empty(1, dtype=torch.int64, device=dev)
    t = timed(lambda: N.check(lib.rd_pack_plan(N.ptr(batch.lens), n, L, N.ptr(si), N.ptr(ui), N.ptr(bs), N.ptr(tot), N.ptr(ws), ws.numel(), st), "plan"))
    out["pack_plan"] = {"ms": t * 1e3, "GBps": n * (4 + 16) / t / 1e9}
    data = torch.empty((int(tot.item()), 4), dtype=torch.float32, device=dev)
    t = timed(lambda: N.check(lib.rd_pack_onehot(N.ptr(batch.arena), N.ptr(batch.offsets), N.ptr(batch.lens), n, L, N.ptr(si), N.ptr(bs), N.ptr(data), st), "pack"))
    out["pack_onehot"] = {"ms": t * 1e3, "GBps": n * (17 * L + 16) / t / 1e9}

    l1 = torch.randn((n, 2), device=dev)
    l2 = torch.randn((n, 2), device=dev)
    cnt = torch.zeros(3, dtype=torch.int64, device=dev)
    lab = torch.empty(n, dtype=torch.int8, device=dev)
    t = timed(lambda: N.check(lib.rd_pair_fuse(N.ptr(l1), N.ptr(l2), n, 1, N.ptr(lab), N.ptr(cnt), st), "fuse"))
    out["pair_fuse"] = {"ms": t * 1e3, "GBps": n * 17 / t / 1e9}
    t = timed(lambda: N.check(lib.rd_count_labels(N.ptr(lab), n, N.ptr(cnt), st), "count"))
    out["count_labels"] = {"ms": t * 1e3, "GBps": n / t / 1e9}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
