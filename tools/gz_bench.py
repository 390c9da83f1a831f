#!/usr/bin/env python3
"""gz_bench.py - the device gzip path (csrc/rd_deflate.hpp) on one chunk of FASTQ resident in HBM: the records of a 2^20-record chunk
are split by label into two files' worth of gzip members. Reports, per FASTQ profile: compressed size against zlib level 5 (the
reference's writer, detect.py:729-741), milliseconds and GB/s of uncompressed text per call, reads/s, and the time of each kernel
(run it under `rocprofv3 --kernel-trace --stats` for the per-kernel table; tools/profile_round.sh does).

    python tools/gz_bench.py [--records 1048576] [--out gpurun_out/gz_bench.json]"""
import argparse
import json
import os
import sys
import zlib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=1 << 20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--inflate-sweep", action="store_true", help="also time the inflate kernel with 1/4 ... 4 times the members in one launch")
    ap.add_argument("--zlib-members", action="store_true", help="also time the inflate kernel on BGZF members made by zlib level 6 (what bgzip writes: many short matches)")
    ap.add_argument("--stages", action="store_true", help="with RD_HIP_LIB=.../librd_hip_diag.so: cycles per stage of the deflate kernel")
    a = ap.parse_args()
    import numpy as np
    import torch
    from ribodetector_amd import synth
    from ribodetector_amd.gz import DeviceGzip
    dev = torch.device("cuda", 0)
    dg = DeviceGzip(dev)
    n = a.records
    prof = None
    if a.stages:
        import ctypes as C
        from ribodetector_amd import _native as N
        prof = torch.zeros(32, dtype=torch.int64, device=dev)
        N.check(N.lib().rd_gz_diag_set_profile(C.c_void_p(prof.data_ptr())), "rd_gz_diag_set_profile")
    arena, off, lens = synth.reads_torch(n, 100, seed=2000, device=dev)
    rec = {"records": n}
    # (1) the bench's own FASTQ (constant quality); (2) sequencer-like: Illumina headers + binned qualities, built on the host for 2^18
    # records and tiled (the device path sees the same bytes a file would hold)
    texts = {"bench_fastq": synth.fastq_image_torch(arena, off, lens, mate=1)}
    m = min(n, 1 << 18)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        a_np, o_np, _ = synth.reads_numpy(m, 100, seed=11)
        p = os.path.join(d, "r.fq")
        synth.write_fastq_realistic(p, a_np, o_np, mate=1, seed=1)
        t = torch.from_numpy(np.fromfile(p, dtype=np.uint8)).to(dev)
    texts["sequencer_like"] = t.repeat(max(1, n // m))
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for name, text in texts.items():
        nl = torch.nonzero(text == 10).flatten()
        rs = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), nl[3::4] + 1])
        nr = int(rs.numel()) - 1
        labels = (torch.rand(nr, generator=g, device=dev) < 0.1).to(torch.int8)      # 10 % "rRNA"
        outs = {}
        for lab in (0, 1):
            dg.compress_selected(text, rs, labels, lab, slot=lab)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            for lab in (0, 1):
                outs[lab] = dg.compress_selected(text, rs, labels, lab, slot=lab)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if prof is not None:
            st = prof.cpu().tolist()
            prof.zero_()
            tot = float(sum(st)) or 1.0
            names = ("load", "crc_thread0", "parse_rest", "wait_slowest_wave", "codes_rest", "emit", "copy", "-", "strip_candidates", "strip_walk", "strip_tokens",
                     "huff_rank_sort", "huff_merge", "huff_node_depths", "huff_lengths_codes")
            rec.setdefault("stages", {})[name] = {k: round(x / tot, 4) for k, x in zip(names, st[:15]) if k != "-"}
            rec.setdefault("stage_cycles_per_member", {})[name] = {k: round(x / (reps + 1) / max(1, sum(int(-(-int(o[1][1]) // 65280)) for o in outs.values())))
                                                                   for k, x in zip(names, st[:15]) if k != "-"}
        comp = sum(int(outs[lab][1][0]) for lab in (0, 1))
        plain = sum(int(outs[lab][1][1]) for lab in (0, 1))
        assert plain == int(text.numel())
        # zlib level 5 on a 64 MB sample of the same text (one stream), scaled
        sample = text[: min(int(text.numel()), 64 << 20)].cpu().numpy().tobytes()
        z5 = len(zlib.compress(sample, 5)) * (int(text.numel()) / len(sample))
        # the way back: the members just made (label-0 file), inflated on the device (one wave per member)
        inf = {}
        try:
            from ribodetector_amd.gz import DeviceGunzip
            du = DeviceGunzip(dev)
            nb0 = int(outs[0][1][0])
            cbuf = outs[0][0][:nb0].cpu().numpy()
            nm, consumed, ob, _ = du.index(cbuf, nb0)
            du.inflate(cbuf, consumed, nm, ob)
            lib = __import__("ribodetector_amd._native", fromlist=["lib"]).lib()
            import ctypes as C
            from ribodetector_amd import _native as N
            st = torch.cuda.current_stream(dev)
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            for _ in range(5):
                N.check(lib.rd_gz_inflate_members(N.ptr(du._comp_dev), consumed, N.ptr(du._mem_dev), nm, N.ptr(du._text_dev), ob, N.ptr(du._status),
                                                  C.c_void_p(st.cuda_stream)), "rd_gz_inflate_members")
            a1.record()
            torch.cuda.synchronize()
            ims = a0.elapsed_time(a1) / 5
            ok = bool((du._status[:nm] == 0).all())
            inf = {"members": nm, "compressed_bytes": consumed, "text_bytes": ob, "ms": ims, "GB_per_s_of_text": ob / ims / 1e6, "all_members_ok": ok}
            if prof is not None:      # the inflate kernel's own stamps (per wave = per member): g_gz_prof[16 ...]
                ist = prof.cpu().tolist()[16:23]
                prof.zero_()
                itot = float(sum(ist)) or 1.0
                inames = ("header_code_lengths", "tables", "round_lookup", "walk_literals", "match", "rest", "crc")
                inf["stages"] = {k: round(x / itot, 4) for k, x in zip(inames, ist)}
                inf["stage_cycles_per_member"] = {k: round(x / 6 / max(1, nm)) for k, x in zip(inames, ist)}
            # how the rate depends on the members in flight (one wave each): the same members 1/4 ... 4 times in one launch
            import numpy as np
            mt = du._mem_dev[: nm * 24].cpu().numpy().view(np.dtype([("i", "<i8"), ("o", "<i8"), ("il", "<i4"), ("ol", "<i4")]))
            sweep = {}
            for k in ((0.25, 0.5, 1, 2, 4) if a.inflate_sweep else ()):
                reps_k = max(1, int(k))
                take = nm if k >= 1 else int(nm * k)
                tabs = []
                for r in range(reps_k):
                    t = mt[:take].copy()
                    t["o"] += r * ob
                    tabs.append(t)
                tab = torch.from_numpy(np.concatenate(tabs).view(np.uint8)).to(dev)
                txt = torch.empty(reps_k * ob + 64, dtype=torch.uint8, device=dev)
                stt = torch.empty(take * reps_k, dtype=torch.int32, device=dev)
                nbytes = sum(int(x) for x in np.concatenate(tabs)["ol"])
                for it in range(4):
                    if it == 1:
                        a0.record()
                    N.check(lib.rd_gz_inflate_members(N.ptr(du._comp_dev), consumed, N.ptr(tab), take * reps_k, N.ptr(txt), txt.numel(), N.ptr(stt),
                                                      C.c_void_p(st.cuda_stream)), "rd_gz_inflate_members")
                a1.record()
                torch.cuda.synchronize()
                kms = a0.elapsed_time(a1) / 3
                sweep["%d_members" % (take * reps_k)] = {"ms": round(kms, 3), "GB_per_s_of_text": round(nbytes / kms / 1e6, 1), "ok": bool((stt == 0).all())}
            if sweep:
                inf["members_in_flight_sweep"] = sweep
            if a.zlib_members:        # the same text as bgzip would write it: zlib level 6 members of 65,280 bytes (host-made, ~10 s)
                import struct
                raw = text[: min(int(text.numel()), 3500 * 65280)].cpu().numpy().tobytes()
                parts = []
                for o in range(0, len(raw), 65280):
                    blk = raw[o:o + 65280]
                    co = zlib.compressobj(6, zlib.DEFLATED, -15)
                    body = co.compress(blk) + co.flush()
                    parts.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(blk) & 0xffffffff, len(blk)))
                zb = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
                nmz, consz, obz, _ = du.index(zb, len(zb))
                du.inflate(zb, consz, nmz, obz)
                if prof is not None:
                    prof.zero_()
                a0.record()
                for _ in range(5):
                    N.check(lib.rd_gz_inflate_members(N.ptr(du._comp_dev), consz, N.ptr(du._mem_dev), nmz, N.ptr(du._text_dev), obz, N.ptr(du._status),
                                                      C.c_void_p(st.cuda_stream)), "rd_gz_inflate_members")
                a1.record()
                torch.cuda.synchronize()
                zms = a0.elapsed_time(a1) / 5
                same = bool((du._text_dev[:obz].cpu().numpy().tobytes() == raw))
                inf["zlib6_members"] = {"members": nmz, "compressed_bytes": consz, "text_bytes": obz, "ms": zms, "GB_per_s_of_text": obz / zms / 1e6,
                                        "all_members_ok": bool((du._status[:nmz] == 0).all()), "text_identical": same}
                if prof is not None:
                    ist = prof.cpu().tolist()[16:23]
                    prof.zero_()
                    itot = float(sum(ist)) or 1.0
                    inf["zlib6_members"]["stages"] = {k: round(x / itot, 4) for k, x in zip(inames, ist)}
                    inf["zlib6_members"]["stage_cycles_per_member"] = {k: round(x / 5 / max(1, nmz)) for k, x in zip(inames, ist)}
        except Exception as e:      # noqa: BLE001
            inf = {"error": repr(e)}
        rec[name] = {"records": nr, "text_bytes": int(text.numel()), "device_gzip_bytes": comp, "ratio": int(text.numel()) / comp, "device_inflate": inf,
                     "zlib5_bytes_est": z5, "size_vs_zlib5": comp / z5, "ms_per_chunk_both_labels": ms,
                     "GB_per_s": int(text.numel()) / ms / 1e6, "reads_per_s": nr / ms * 1e3,
                     "members": sum(int(outs[lab][1][2]) for lab in (0, 1))}
    print(json.dumps(rec))
    if a.out:
        json.dump(rec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
