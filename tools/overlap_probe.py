#!/usr/bin/env python
"""Do the latency-bound side kernels run BESIDE the recurrence, or behind it? One thread keeps the recurrence kernel busy (back-to-back
rd_classify launches of 2^20 reads); another submits a side operation over and over - the FASTQ index of a 64 MB batch, the inflate of
~900 BGZF members, the deflate of a 2^18-record chunk - on a stream of a given kind, and measures submit -> complete per operation.
Kinds: plain stream, high-priority stream, stream restricted to the first S compute units (C ABI rd_stream_create) with the recurrence
unrestricted or restricted to the other 256 - S. Prints one JSON: per kind the side operations' latency and the recurrence's ms per launch.
    python tools/overlap_probe.py [--seconds 1.5] [--cus 16,32,64]"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np      # noqa: E402
import torch            # noqa: E402
from ribodetector_amd import _native as N, synth      # noqa: E402
from ribodetector_amd.gz import DeviceGunzip, DeviceGzip      # noqa: E402
from ribodetector_amd.model import model as module_arch      # noqa: E402
from ribodetector_amd.parse_config import ConfigParser      # noqa: E402

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
DEV = torch.device("cuda", 0)


def masked_stream(lo, hi):
    """a stream on compute units [lo, hi) of 256"""
    mask = (C.c_uint32 * 8)()
    for i in range(lo, hi):
        mask[i // 32] |= 1 << (i % 32)
    h = C.c_void_p()
    N.check(N.lib().rd_stream_create(0, mask, 8, 0, C.byref(h)), "rd_stream_create")
    return torch.cuda.ExternalStream(h.value, device=DEV), h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=1.5)
    ap.add_argument("--cus", default="16,32,64")
    a = ap.parse_args()
    torch.cuda.set_device(DEV)
    cfg = ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json"))
    model = cfg.init_obj("arch", module_arch)
    model.load_state_dict(cfg.load_state_dict("mcc"))
    model.to(DEV).eval()
    model.set_prefix_table(11)
    P = 1 << 20
    arena, off, lens = synth.reads_torch(P, 100, seed=1, device=DEV)
    offs = off[:-1].contiguous()
    logits = torch.empty((P, 2), dtype=torch.float32, device=DEV)
    # side operations' inputs
    nq = 1 << 18
    text = synth.fastq_image_torch(arena[: nq * 100], off[: nq + 1], lens[:nq], mate=1)
    rs = torch.zeros(nq + 1, dtype=torch.int64, device=DEV)
    torch.cumsum(18 + 2 * lens[:nq].to(torch.int64), 0, out=rs[1:])
    lab = torch.zeros(nq, dtype=torch.int8, device=DEV)
    dgz = DeviceGzip(DEV)
    out, info = dgz.compress_selected(text, rs, lab, 0)
    torch.cuda.synchronize()
    comp = out[: int(info[0])].cpu().numpy()
    tb = int(text.numel())
    L = N.lib()

    def op_index(st, state):
        if "t" not in state:
            state["t"] = torch.empty(tb + 4096, dtype=torch.uint8, device=DEV)
            state["t"][:tb] = text
            state["le"] = torch.empty(tb // 4 + 4096, dtype=torch.int32, device=DEV)
            state["s"] = torch.empty(8, dtype=torch.int64, device=DEV)
            state["ws"] = torch.empty(max(int(L.rd_fastq_index_workspace_bytes(tb)), 256), dtype=torch.uint8, device=DEV)
        N.check(L.rd_fastq_index(N.ptr(state["t"]), 0, tb, None, None, 1, N.ptr(state["le"]), state["le"].numel(), N.ptr(state["s"]), N.ptr(state["ws"]),
                                 state["ws"].numel(), C.c_void_p(st.cuda_stream)), "rd_fastq_index")

    def op_inflate(st, state):
        if "du" not in state:
            du = state["du"] = DeviceGunzip(DEV, stream=st)
            state["nm"], state["consumed"], state["ob"], _ = du.index(comp, len(comp))
            du.inflate(comp, state["consumed"], state["nm"], state["ob"])
        du = state["du"]
        N.check(L.rd_gz_inflate_members(N.ptr(du._comp_dev), state["consumed"], N.ptr(du._mem_dev), state["nm"], N.ptr(du._text_dev), state["ob"],
                                        N.ptr(du._status), C.c_void_p(st.cuda_stream)), "rd_gz_inflate_members")

    def op_deflate(st, state):
        g = state.setdefault("g", DeviceGzip(DEV))
        g.compress_selected(text, rs, lab, 0)      # (on the current stream = st: the caller sets it)

    def run(kind, side_stream, main_stream, op):
        stop = threading.Event()
        lat = []

        def side():
            torch.cuda.set_device(DEV)
            state = {}
            with torch.cuda.stream(side_stream):
                op(side_stream, state)
                side_stream.synchronize()
                while not stop.is_set():
                    t0 = time.perf_counter()
                    op(side_stream, state)
                    ev = torch.cuda.Event()
                    ev.record(side_stream)
                    while not ev.query():
                        time.sleep(1e-4)
                    lat.append(time.perf_counter() - t0)
                    time.sleep(2e-3)
        th = threading.Thread(target=side)
        n = 0
        with torch.cuda.stream(main_stream):
            for _ in range(3):
                model.classify_bytes(arena, offs, lens, 100, want_labels=False, logits=logits)
            main_stream.synchronize()
            th.start()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < a.seconds:
                model.classify_bytes(arena, offs, lens, 100, want_labels=False, logits=logits)
                n += 1
                if n % 4 == 0:
                    main_stream.synchronize()
            main_stream.synchronize()
            dt = time.perf_counter() - t0
        stop.set()
        th.join()
        # the side operation alone
        alone = []
        state = {}
        with torch.cuda.stream(side_stream):
            op(side_stream, state)
            side_stream.synchronize()
            for _ in range(5):
                t1 = time.perf_counter()
                op(side_stream, state)
                side_stream.synchronize()
                alone.append(time.perf_counter() - t1)
        return {"kind": kind, "recurrence_ms_per_launch": 1e3 * dt / n, "side_ops": len(lat), "side_latency_ms_median": 1e3 * float(np.median(lat)) if lat else None,
                "side_latency_ms_p90": 1e3 * float(np.quantile(lat, 0.9)) if lat else None, "side_alone_ms": 1e3 * float(np.median(alone))}

    cur = torch.cuda.current_stream(DEV)
    # the recurrence alone
    with torch.cuda.stream(cur):
        for _ in range(3):
            model.classify_bytes(arena, offs, lens, 100, want_labels=False, logits=logits)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model.classify_bytes(arena, offs, lens, 100, want_labels=False, logits=logits)
        torch.cuda.synchronize()
        base = 1e3 * (time.perf_counter() - t0) / 20
    res = {"recurrence_alone_ms_per_launch": base, "runs": []}
    ops = {"fastq_index_57MB": op_index, "inflate_%d_members" % (len(comp) // 14000): op_inflate, "deflate_2^18_records": op_deflate}
    handles = []
    for name, op in ops.items():
        kinds = [("plain stream", torch.cuda.Stream(DEV), cur), ("high-priority stream", torch.cuda.Stream(DEV, priority=-1), cur)]
        for s in [int(x) for x in a.cus.split(",")]:
            side, h1 = masked_stream(0, s)
            main, h2 = masked_stream(s, 256)
            handles += [h1, h2]
            kinds.append(("side on %d CUs, recurrence on all" % s, side, cur))
            kinds.append(("side on %d CUs, recurrence on the other %d" % (s, 256 - s), side, main))
        for kind, side, main in kinds:
            r = run(kind, side, main, op)
            r["op"] = name
            res["runs"].append(r)
            sys.stderr.write(json.dumps(r) + "\n")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
