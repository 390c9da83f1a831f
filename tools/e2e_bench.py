#!/usr/bin/env python
"""Timed region (iii) of SURVEY §8d - the whole `ribodetector` CLI (detect.main(): model load, prefix-table build, ingest, H2D, kernels,
D2H, output write) on FASTQ files in tmpfs (the reference flow: detect.py:464-499) - as a module (bench.py imports E2E, gzip_record,
encoder_record from here) and as a script:
    python tools/e2e_bench.py [--reads 4000000]      every flow (SE / PE, plain / gz / BGZF in, plain / gz out) on sequencer-like reads
"""
import argparse
import gzip
import json
import os
import shutil
import sys
import tempfile
import threading
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
HBM_PEAK_GBPS = 8000.0


def usable_cores():
    """host cores this process may actually use: min(affinity, cgroup cpu quota). The GPU boxes expose 256 logical CPUs
    but cap the container at 16 CPUs' worth of time (cpu.max = 1600000 100000)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def thread_cpu():
    """CPU seconds (user + system) of every thread of this process, by thread id: {tid: (name, seconds)}"""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open("/proc/self/task/%s/stat" % tid).read()
            name = st[st.index("(") + 1:st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            out[tid] = (name, (int(f[11]) + int(f[12])) / tick)
        except Exception:
            pass
    return out


def shm_free(path="/dev/shm"):
    try:
        return shutil.disk_usage(path).free
    except OSError:
        return 0


def gzip_file(src, dst, level=1):
    """src -> dst as ONE gzip member (what a sequencer's .fastq.gz is: one DEFLATE stream). libdeflate when the box has it
    (outside every timed region: the input of the gz -> gz measurement), zlib otherwise."""
    import ctypes as C
    import zlib
    data = open(src, "rb").read()
    try:
        ld = C.CDLL("libdeflate.so.0")
        ld.libdeflate_alloc_compressor.restype = C.c_void_p
        ld.libdeflate_alloc_compressor.argtypes = [C.c_int]
        ld.libdeflate_gzip_compress_bound.restype = C.c_size_t
        ld.libdeflate_gzip_compress_bound.argtypes = [C.c_void_p, C.c_size_t]
        ld.libdeflate_gzip_compress.restype = C.c_size_t
        ld.libdeflate_gzip_compress.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t]
        ld.libdeflate_free_compressor.argtypes = [C.c_void_p]
        c = ld.libdeflate_alloc_compressor(level)
        cap = ld.libdeflate_gzip_compress_bound(c, len(data))
        buf = C.create_string_buffer(cap)
        n = ld.libdeflate_gzip_compress(c, data, len(data), buf, cap)
        ld.libdeflate_free_compressor(c)
        if n == 0:
            raise OSError("libdeflate_gzip_compress failed")
        with open(dst, "wb") as fh:
            fh.write(memoryview(buf)[:n])
        return "libdeflate level %d" % level
    except OSError:
        co = zlib.compressobj(1, zlib.DEFLATED, 31)
        with open(dst, "wb") as fh:
            for i in range(0, len(data), 1 << 24):
                fh.write(co.compress(data[i:i + (1 << 24)]))
            fh.write(co.flush())
        return "zlib level 1"


class E2E:
    """The CLI on FASTQ file(s) built ONCE from a device-resident read stream (arenas: one uint8 tensor per mate; offsets / lens: the
    reads' layout), in a tmpfs directory: leg(in_kind, out_gz) runs detect.main() once untimed (it pays one-off costs of the process:
    first pinned allocations, page cache) and `timed_calls` times; the MEDIAN call is reported, with the spread, the host cores the
    call kept busy, and the steady-state rate after the first chunk (pipeline fill excluded).
      in_kind: "plain" | "gz" (ONE gzip member per file - what sequencers write) | "bgzf" (65,280-byte members - bgzip, this build's writer)
    The CLI logs at WARNING here (its per-chunk INFO lines would be most of stderr)."""

    def __init__(self, torch, synth, arenas, offsets, lens, L, ensure, style="const"):
        """style "const": SURVEY 8d's files (header @s<idx>/<mate>, quality 'I' * len; BGZF framed by this build's device writer, the single
        stream by libdeflate / zlib level 1 - the files of rounds 3-5). style "seqlike": what a user has - Illumina headers, binned
        qualities (synth.fastq_image_torch), BGZF made by zlib level 6 per 65,280-byte block (what bgzip writes), the single stream by
        zlib level 6 (synth.pgzip_file)."""
        self.torch, self.synth, self.style = torch, synth, style
        self.L, self.ensure, self.n = L, ensure, int(lens.numel())
        self.lens, self.dev = lens, arenas[0].device
        self.dir = tempfile.mkdtemp(prefix="rd_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        self.files = {"plain": []}
        self.how = {"plain": None if style == "const" else "Illumina headers, binned qualities"}
        for m, a in enumerate(arenas):
            p = os.path.join(self.dir, "r_%d.fq" % (m + 1))
            synth.fastq_image_torch(a, offsets, lens, mate=m + 1, style=style, seed=m).cpu().numpy().tofile(p)
            self.files["plain"].append(p)
        self.plain_bytes = sum(os.path.getsize(p) for p in self.files["plain"])
        self._ncall = 0

    def close(self):
        shutil.rmtree(self.dir, ignore_errors=True)

    def save(self):
        """what a child process needs to run legs on these files (E2E.from_dir)"""
        with open(os.path.join(self.dir, "e2e_meta.json"), "w") as fh:
            json.dump({"L": self.L, "ensure": self.ensure, "n": self.n, "files": self.files, "how": self.how, "plain_bytes": self.plain_bytes, "style": self.style}, fh)

    @classmethod
    def from_dir(cls, d):
        m = json.load(open(os.path.join(d, "e2e_meta.json")))
        e = cls.__new__(cls)
        e.dir, e.L, e.ensure, e.n, e.files, e.how, e.plain_bytes = d, m["L"], m["ensure"], m["n"], m["files"], m["how"], m["plain_bytes"]
        e.torch = e.synth = e.lens = e.dev = None
        e.style = m.get("style", "const")
        return e

    def leg_in_child(self, in_kind="plain", out_gz=False, timed_calls=3, threads=None, env=None):
        """the same leg in a process of its own - what a CLI invocation is. (In a process that has already made a dozen calls the HIP
        runtime's helper thread spins a full core through every later call: BGZF -> gz read 1.3 host cores as the 7th leg of one process,
        0.6 as the first.)"""
        import subprocess
        self.inputs(in_kind)
        self.save()
        cmd = [sys.executable, os.path.abspath(__file__), "--one-leg", self.dir, "--in-kind", in_kind, "--timed-calls", str(timed_calls)] + (
            ["--out-gz"] if out_gz else []) + (["--threads", str(threads)] if threads else []) + [a for k, v in (env or {}).items() for a in ("--env", "%s=%s" % (k, v))]
        r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            raise RuntimeError("e2e leg %s -> %s failed (rc %d): %s" % (in_kind, "gz" if out_gz else "plain", r.returncode, r.stderr.decode(errors="replace")[-400:]))
        return json.loads(lines[-1])

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def inputs(self, kind):
        if kind in self.files:
            return self.files[kind]
        torch = self.torch
        plain = self.files["plain"]
        if self.style == "seqlike" and kind in ("bgzf", "gz"):     # made by zlib level 6, as a user's bgzip / gzip would
            from ribodetector_amd import synth as sy
            outs = [p[:-3] + (".bgzf.fq.gz" if kind == "bgzf" else ".one.fq.gz") for p in plain]
            for p, q in zip(plain, outs):
                (sy.bgzip_file if kind == "bgzf" else sy.pgzip_file)(p, q, level=6)
            self.files[kind] = outs
            self.how[kind] = ("BGZF members of 65,280 bytes, zlib level 6 (what bgzip writes)" if kind == "bgzf" else
                              "one gzip member per file, zlib level 6 (deflated in 8 MiB pieces primed with the 32 KiB before them, as pigz does)")
            return outs
        if kind == "bgzf":          # framed by the device writer, outside every timed region
            import numpy as np
            from ribodetector_amd.gz import DeviceGzip, eof_block
            dg = DeviceGzip(self.dev)
            outs = []
            for p in plain:
                t = torch.from_numpy(np.fromfile(p, dtype=np.uint8)).to(self.dev)
                rs = torch.zeros(self.n + 1, dtype=torch.int64, device=self.dev)
                torch.cumsum(18 + 2 * self.lens.to(torch.int64), 0, out=rs[1:])
                o, info = dg.compress_selected(t, rs, torch.zeros(self.n, dtype=torch.int8, device=self.dev), 0)
                torch.cuda.synchronize(self.dev)
                q = p[:-3] + ".bgzf.fq.gz"
                with open(q, "wb") as fh:
                    fh.write(o[: int(info[0])].cpu().numpy().tobytes())
                    fh.write(eof_block())
                outs.append(q)
                del t, o
            del dg
            torch.cuda.empty_cache()
            self.files[kind], self.how[kind] = outs, "BGZF members of 65,280 bytes (device writer)"
        elif kind == "gz":
            res = [None] * len(plain)
            outs = [p[:-3] + ".one.fq.gz" for p in plain]

            def comp(i):
                res[i] = gzip_file(plain[i], outs[i])
            ths = [threading.Thread(target=comp, args=(i,)) for i in range(len(plain))]
            [t.start() for t in ths]
            [t.join() for t in ths]
            self.files[kind], self.how[kind] = outs, "one gzip member per file, " + str(res[0])
        else:
            raise ValueError(kind)
        return self.files[kind]

    def leg(self, in_kind="plain", out_gz=False, timed_calls=3, threads=None, env=None):
        from ribodetector_amd import detect
        ins = self.inputs(in_kind)
        ext = ".fq.gz" if out_gz else ".fq"
        outs = [os.path.join(self.dir, "non_%d%s" % (m + 1, ext)) for m in range(len(ins))]
        rrs = [os.path.join(self.dir, "rrna_%d%s" % (m + 1, ext)) for m in range(len(ins))]
        argv = (["-l", str(self.L), "-i", *ins, "-o", *outs, "-r", *rrs] + (["-e", self.ensure] if len(ins) == 2 else [])
                + (["-t", str(threads)] if threads else []))
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        calls = []
        try:
            for _ in range(1 + timed_calls):
                for q in outs + rrs:                      # every call writes NEW files (truncating GBs of tmpfs pages is not the CLI's work)
                    if os.path.exists(q):
                        os.remove(q)
                th0 = thread_cpu()
                t0, c0 = time.perf_counter(), time.process_time()
                pr = detect.main(argv, log_level="WARNING")
                dt, cpu = time.perf_counter() - t0, time.process_time() - c0
                th1 = thread_cpu()
                top = sorted(((th1[t][1] - th0.get(t, (None, 0.0))[1], th1[t][0]) for t in th1), reverse=True)
                tm = pr.timing
                calls.append({"seconds": round(dt, 4), "reads_per_s": len(ins) * self.n / dt,
                              "host_cores_busy": round(cpu / dt, 2),      # CPU seconds of ALL threads of the process / wall seconds
                              "main_thread_s": {k: round(v, 4) for k, v in pr._stage_s.items()},
                              "thread_cpu_s": dict(getattr(pr, "thread_cpu_s", {})),
                              "os_threads_cpu_s": [[nm, round(c, 3)] for c, nm in top if c >= 0.02][:8],     # (threads alive at the end of the call)
                              "load_model_s": round(tm["load_model_s"], 4), "detect_s": round(tm["detect_s"], 4), "prefix_k": tm["prefix_k"],
                              "ingest": tm.get("ingest"),
                              "reads_per_s_after_model_load": len(ins) * self.n / tm["detect_s"],
                              "reads_per_s_after_first_chunk": tm.get("reads_per_s_after_first_chunk")})
                del pr
            out_bytes = sum(os.path.getsize(p) for p in outs + rrs)
        finally:
            for q in outs + rrs:
                if os.path.exists(q):
                    os.remove(q)
            for k, v in old.items():
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = v
        timed = sorted(calls[1:], key=lambda c: c["seconds"])
        med = timed[len(timed) // 2]
        steady = sorted(c["reads_per_s_after_first_chunk"] or 0.0 for c in timed)[len(timed) // 2]
        return {"flow": "%s -> %s" % (in_kind, "gz" if out_gz else "plain"), "reads_per_s": med["reads_per_s"], "seconds": med["seconds"],
                "reads_per_s_after_first_chunk": steady or None, "reads_per_s_after_model_load": med["reads_per_s_after_model_load"],
                "host_cores_busy": med["host_cores_busy"], "spread": (timed[-1]["seconds"] - timed[0]["seconds"]) / med["seconds"],
                "timed_calls": timed_calls, "files": len(ins), "records_per_file": self.n, "ingest": med["ingest"],
                "input_bytes": sum(os.path.getsize(p) for p in ins), "plain_input_bytes": self.plain_bytes, "output_bytes": out_bytes,
                "input_compressor": self.how.get(in_kind), "threads_flag": threads or 10, "env": env or {},
                "warm_call": calls[0], "calls": calls[1:]}


def gzip_record(torch, synth, dev, arena, offsets, lens, labels):
    """device gzip (csrc/rd_deflate.hpp) on the FASTQ text of one step's first mate, partitioned by the step's own labels into the two
    files a CLI run writes: compressed size against zlib level 5 (the reference's writer, on a 32 MB sample), time per chunk by events"""
    import zlib
    from ribodetector_amd.gz import DeviceGzip
    text = synth.fastq_image_torch(arena, offsets, lens, mate=1)
    n = int(lens.numel())
    rs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cumsum(18 + 2 * lens.to(torch.int64), 0, out=rs[1:])
    lab = labels.view(torch.int8).contiguous()
    dg = DeviceGzip(dev)
    outs = {}
    for v in (0, 1):
        outs[v] = dg.compress_selected(text, rs, lab, v, slot=v)
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    a.record()
    for _ in range(reps):
        for v in (0, 1):
            outs[v] = dg.compress_selected(text, rs, lab, v, slot=v)
    b.record()
    torch.cuda.synchronize(dev)
    ms = a.elapsed_time(b) / reps
    comp = sum(int(outs[v][1][0]) for v in (0, 1))
    plain = sum(int(outs[v][1][1]) for v in (0, 1))
    sample = text[: min(int(text.numel()), 32 << 20)].cpu().numpy().tobytes()
    z5 = len(zlib.compress(sample, 5)) / len(sample)
    # the way back (csrc/rd_inflate_dev.hpp): the members of the larger of the two streams, inflated one wave per member
    gun = None
    try:
        import ctypes as C
        from ribodetector_amd import _native as NN
        from ribodetector_amd.gz import DeviceGunzip
        v = 0 if int(outs[0][1][0]) >= int(outs[1][1][0]) else 1
        nb0 = int(outs[v][1][0])
        cbuf = outs[v][0][:nb0].cpu().numpy()
        du = DeviceGunzip(dev)
        nm, consumed, ob, _ = du.index(cbuf, nb0)
        du.inflate(cbuf, consumed, nm, ob)
        st = torch.cuda.current_stream(dev)
        a.record()
        for _ in range(reps):
            NN.check(NN.lib().rd_gz_inflate_members(NN.ptr(du._comp_dev), consumed, NN.ptr(du._mem_dev), nm, NN.ptr(du._text_dev), ob, NN.ptr(du._status),
                                                    C.c_void_p(st.cuda_stream)), "rd_gz_inflate_members")
        b.record()
        torch.cuda.synchronize(dev)
        ims = a.elapsed_time(b) / reps
        gun = {"kernel": "rd_gz_inflate_kernel", "members": nm, "compressed_bytes": consumed, "text_bytes": ob, "ms": ims, "GB_per_s_of_text": ob / ims / 1e6,
               "all_members_ok": bool((du._status[:nm] == 0).all()),
               "bound": "latency of a wave's own chain (one wave per member; ~3,500 members = 3.4 waves per SIMD in flight)"}
    except Exception as e:      # noqa: BLE001
        gun = {"error": repr(e)}
    return {"kernel": "rd_gz_deflate_kernel (+ select, pack, compact)", "records": n, "device_gunzip": gun, "text_bytes": plain, "compressed_bytes": comp,
            "ratio": plain / max(comp, 1), "size_vs_zlib_level_5": (comp / max(plain, 1)) / z5, "ms_per_chunk_both_label_files": ms,
            "GB_per_s_of_text": plain / ms / 1e6, "reads_per_s": n / ms * 1e3, "members": sum(int(outs[v][1][2]) for v in (0, 1)),
            "bound": "dependent-issue latency at four waves per SIMD (155 KB of LDS: one workgroup of sixteen waves per CU); HBM: %.3f of 8 TB/s" % (plain / ms / 1e6 / HBM_PEAK_GBPS),
            "what": "the FASTQ text of one step's first mate (constant quality, 218 B per record) split by the step's labels into the two "
                    "gzip (BGZF) streams the CLI appends to its .gz outputs; zlib level 5 = the reference's gzip.open(..., compresslevel=5)"}


def encoder_record(torch, N, dev, arena, offs, lens, n, L):
    """standalone encoder kernels on the first n reads: algorithmic bytes / kernel time (events on the launch stream)"""
    lib, st = N.lib(), N.stream_ptr(dev)

    def timed(fn, reps=10):
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / reps * 1e-3

    rec = {"reads": n, "read_len": L, "bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "kernels": {}}

    def put(name, t, nbytes, what):
        rec["kernels"][name] = {"ms": t * 1e3, "achieved": nbytes / t / 1e9, "frac": nbytes / t / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes": what}
    codes = torch.empty((n, L), dtype=torch.uint8, device=dev)
    t = timed(lambda: N.check(lib.rd_encode_codes(N.ptr(arena), N.ptr(offs), N.ptr(lens), n, L, L, N.ptr(codes), st), "rd_encode_codes"))
    put("rd_encode_codes_kernel", t, n * (2 * L + 12), "L in + L out + 12 index per read")
    del codes
    oh = torch.empty((n, L, 4), dtype=torch.float32, device=dev)
    t = timed(lambda: N.check(lib.rd_encode_onehot_padded(N.ptr(arena), N.ptr(offs), N.ptr(lens), n, L, N.ptr(oh), st), "rd_encode_onehot_padded"))
    put("rd_encode_onehot_padded_kernel", t, n * (17 * L + 12), "L in + 16 L out + 12 index per read")
    del oh
    ws = torch.empty(int(lib.rd_classify_workspace_bytes(n, L)), dtype=torch.uint8, device=dev)
    si = torch.empty(n, dtype=torch.int64, device=dev)
    ui = torch.empty(n, dtype=torch.int64, device=dev)
    bs = torch.empty(L, dtype=torch.int64, device=dev)
    tot = torch.empty(1, dtype=torch.int64, device=dev)
    N.check(lib.rd_pack_plan(N.ptr(lens), n, L, N.ptr(si), N.ptr(ui), N.ptr(bs), N.ptr(tot), N.ptr(ws), ws.numel(), st), "rd_pack_plan")
    data = torch.empty((int(tot.item()), 4), dtype=torch.float32, device=dev)
    t = timed(lambda: N.check(lib.rd_pack_onehot(N.ptr(arena), N.ptr(offs), N.ptr(lens), n, L, N.ptr(si), N.ptr(bs), N.ptr(data), st), "rd_pack_onehot"))
    put("rd_pack_onehot_kernel", t, n * (17 * L + 16), "L in + 16 L out + 16 index per read")
    return rec


def bench_legs(a):
    """the legs bench.py reports (timed region iii), in a process of their own - what a CLI invocation is: the rank-0 stream of bench.py
    (same seeds, same slices) as FASTQ files in tmpfs; one JSON on the last stdout line"""
    import ribodetector_amd      # noqa: F401  (before torch: the runtime knobs of ribodetector_amd/__init__.py)
    import torch
    from ribodetector_amd import synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    P, RL, paired = a.pairs_per_step, a.read_len, a.paired
    nslices = 4
    r1 = [synth.reads_torch(P, RL, seed=2000 + i, device=dev) for i in range(nslices)]
    r2 = [synth.reads_torch(P, RL, seed=7000 + i, device=dev) for i in range(nslices)] if paired else None
    lens = r1[0][2]
    if a.var_len:                                           # bench.py's var300 workload: lengths ~ U{40..300}
        g = torch.Generator(device=dev)
        g.manual_seed(4)
        lens = torch.randint(40, 301, (P,), generator=g, device=dev, dtype=torch.int32)
    rec = {}
    cat = lambda ts, rep=1: torch.cat([t[0] for t in ts] * rep)      # noqa: E731
    ne = min(P, 1 << 20)
    with E2E(torch, synth, [r1[0][0][: ne * RL]] + ([r2[0][0][: ne * RL]] if paired else []), r1[0][1][: ne + 1], lens[:ne].contiguous(),
             a.max_len, a.ensure) as e:
        rec["one_step_batch"] = e.leg("plain", False, timed_calls=1)
    if a.records <= 0 or ne != P:
        rec["plain_to_plain"] = rec["one_step_batch"]
        print(json.dumps(rec))
        return
    nm = 2 if paired else 1
    rec_bytes = nm * (2 * RL + 20)
    rep = max(1, a.records // (nslices * P))
    while rep > 1 and shm_free() < 3.2 * rep * nslices * P * rec_bytes:     # plain + gz + BGZF inputs, plain outputs of one call, margin
        rep //= 2
    if shm_free() < 3.2 * rep * nslices * P * rec_bytes:
        rec["plain_to_plain"] = rec["one_step_batch"]
        rec["skipped"] = "not enough free /dev/shm for %d records per file" % (rep * nslices * P)
        print(json.dumps(rec))
        return
    nbig = rep * nslices * P
    offs_l = torch.arange(nbig + 1, dtype=torch.int64, device=dev) * RL
    arenas = [cat(r1, rep)] + ([cat(r2, rep)] if paired else [])
    with E2E(torch, synth, arenas, offs_l, lens.repeat(rep * nslices), a.max_len, a.ensure) as e:
        del arenas, r1, r2
        e.inputs("bgzf")
        e.inputs("gz")
        e.lens = None
        torch.cuda.empty_cache()
        xenv = dict(kv.split("=", 1) for kv in a.env)                        # (--env K=V: the same for every leg - A/B runs)
        legs = {"plain_to_plain": ("plain", False, None, {}), "plain_to_gz": ("plain", True, None, {}), "bgzf_to_gz": ("bgzf", True, None, {}),
                "bgzf_to_plain": ("bgzf", False, None, {}), "bgzf_to_gz_host_parse": ("bgzf", True, None, {"RD_DEVICE_PARSE": "0"}),
                "plain_to_plain_host_parse": ("plain", False, None, {"RD_DEVICE_PARSE": "0"}),
                "gz_to_gz": ("gz", True, None, {}),                                                    # the single stream decoded on the GPU (the default since round 5)
                "gz_to_gz_host_inflate": ("gz", True, None, {"RD_DEVICE_INFLATE": "members"}),          # ... by the host's parallel decoder, -t 10
                "gz_to_gz_host_inflate_all_cores": ("gz", True, usable_cores(), {"RD_DEVICE_INFLATE": "members"})}
        want = [k for k in (a.legs.split(",") if a.legs else legs) if k]
        for k in want:                  # every leg in a process of its own (the input files are built once, here)
            if k not in legs:
                continue
            kind, out_gz, threads, env = legs[k]
            rec[k] = e.leg_in_child(kind, out_gz, threads=threads, env=dict(env, **xenv))
    # the same records as a user's files would hold them (round 6): Illumina headers, binned qualities, zlib-6 BGZF / single stream
    slegs = {"seqlike_plain_to_gz": ("plain", True), "seqlike_bgzf_to_gz": ("bgzf", True), "seqlike_gz_to_gz": ("gz", True)}
    swant = [k for k in (a.legs.split(",") if a.legs else slegs) if k in slegs]
    if swant and not a.no_seqlike:
        r1 = [synth.reads_torch(P, RL, seed=2000 + i, device=dev) for i in range(nslices)]
        r2 = [synth.reads_torch(P, RL, seed=7000 + i, device=dev) for i in range(nslices)] if paired else None
        arenas = [cat(r1, rep)] + ([cat(r2, rep)] if paired else [])
        with E2E(torch, synth, arenas, offs_l, lens.repeat(rep * nslices), a.max_len, a.ensure, style="seqlike") as e:
            del arenas, r1, r2
            for kind in {slegs[k][0] for k in swant} - {"plain"}:
                e.inputs(kind)
            e.lens = None
            torch.cuda.empty_cache()
            for k in swant:
                rec[k] = e.leg_in_child(slegs[k][0], slegs[k][1], env=dict(xenv))
    print(json.dumps(rec))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4000000)
    ap.add_argument("--bench-legs", action="store_true", help="the legs of bench.py's timed region (iii), as one JSON (bench.py runs this in a child process)")
    ap.add_argument("--pairs-per-step", type=int, default=1 << 20)
    ap.add_argument("--records", type=int, default=1 << 24)
    ap.add_argument("--read-len", type=int, default=100)
    ap.add_argument("--max-len", type=int, default=100)
    ap.add_argument("--single-end", dest="paired", action="store_false")
    ap.add_argument("--var-len", action="store_true")
    ap.add_argument("--ensure", default="rrna")
    ap.add_argument("--legs", default=None, help="with --bench-legs: only these legs (comma-separated names)")
    ap.add_argument("--no-seqlike", action="store_true", help="with --bench-legs: skip the second file set (sequencer-like text, zlib-6 inputs)")
    ap.add_argument("--one-leg", default=None, metavar="DIR", help=argparse.SUPPRESS)      # child of E2E.leg_in_child
    ap.add_argument("--in-kind", default="plain", help=argparse.SUPPRESS)
    ap.add_argument("--out-gz", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--timed-calls", type=int, default=3, help=argparse.SUPPRESS)
    ap.add_argument("--threads", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--env", action="append", default=[], help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.one_leg:
        import ribodetector_amd      # noqa: F401  (before torch: the runtime knobs of ribodetector_amd/__init__.py)
        e = E2E.from_dir(a.one_leg)
        print(json.dumps(e.leg(a.in_kind, a.out_gz, timed_calls=a.timed_calls, threads=a.threads, env=dict(x.split("=", 1) for x in a.env))))
        return
    if a.bench_legs:
        return bench_legs(a)
    from ribodetector_amd import detect, synth
    d = tempfile.mkdtemp(prefix="rde2e", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    files = {}
    for mate, seed in ((1, 1), (2, 2)):
        arena, off, _ = synth.reads_numpy(a.reads, 100, seed=seed)
        p = os.path.join(d, "r_%d.fq" % mate)
        synth.write_fastq_realistic(p, arena, off, mate, seed=seed)
        with open(p, "rb") as fi, gzip.open(p + ".gz", "wb", compresslevel=5) as fo:
            shutil.copyfileobj(fi, fo, 1 << 24)
        files[mate] = p
    out = {"reads_per_file": a.reads}

    def run(tag, inputs, outputs, extra=(), n=None):
        t, c = time.perf_counter(), time.process_time()
        p = detect.main(["-l", "100", "-i", *inputs, "-o", *outputs, *extra], log_level="WARNING")      # default chunking: 1 Mi reads per chunk
        dt, cpu = time.perf_counter() - t, time.process_time() - c
        out[tag] = {"reads_per_s": len(inputs) * (n or a.reads) / dt, "seconds": dt, "host_cores_busy": round(cpu / dt, 2), "rrna": p.num_rrna, "non_rrna": p.num_nonrrna,
                    "main_thread_s": {k: round(v, 3) for k, v in p._stage_s.items()},
                    "timing": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in p.timing.items()}}

    o = lambda n: os.path.join(d, n)     # noqa: E731
    run("first_call_se_plain", [files[1]], [o("w.fq")])        # includes HIP start-up, model load, first pinned allocations
    run("first_call_pe_plain", [files[1], files[2]], [o("w1.fq"), o("w2.fq")], ["-e", "rrna"])
    run("se_plain_to_plain", [files[1]], [o("a.fq")])
    run("se_gz_to_plain", [files[1] + ".gz"], [o("b.fq")])
    run("se_gz_to_gz", [files[1] + ".gz"], [o("c.fq.gz")])
    run("pe_plain_to_plain", [files[1], files[2]], [o("d1.fq"), o("d2.fq")], ["-e", "rrna"])
    run("pe_gz_to_plain", [files[1] + ".gz", files[2] + ".gz"], [o("e1.fq"), o("e2.fq")], ["-e", "rrna"])
    run("pe_gz_to_gz", [files[1] + ".gz", files[2] + ".gz"], [o("f1.fq.gz"), o("f2.fq.gz")], ["-e", "rrna"])
    # .gz outputs are deflated on the GPU by default; the host's libdeflate writer for comparison, plain -> gz (GPU-bound:
    # recurrence + deflate), and gz -> gz with every usable core given to the inflate of the inputs (-t)
    run("se_plain_to_gz", [files[1]], [o("g.fq.gz")])
    run("pe_plain_to_gz", [files[1], files[2]], [o("h1.fq.gz"), o("h2.fq.gz")], ["-e", "rrna"])
    cores = usable_cores()
    run("pe_gz_to_gz_t%d" % cores, [files[1] + ".gz", files[2] + ".gz"], [o("i1.fq.gz"), o("i2.fq.gz")], ["-e", "rrna", "-t", str(cores)])
    # BGZF inputs = the .gz files the device wrote above (non-rRNA mates of pe_gz_to_gz): members inflated on the GPU (the default for
    # such files) or by the host's member decoder; RD_DEVICE_PARSE=0 = the round-4 route (text back to the host's parser)
    with gzip.open(o("f1.fq.gz"), "rb") as fh:
        nb = sum(chunk.count(b"\n") for chunk in iter(lambda: fh.read(1 << 24), b"")) // 4
    out["bgzf_reads_per_file"] = nb
    bg = [o("f1.fq.gz"), o("f2.fq.gz")]
    run("pe_bgzf_to_gz", bg, [o("m1.fq.gz"), o("m2.fq.gz")], ["-e", "rrna"], n=nb)
    run("pe_bgzf_to_plain", bg, [o("n1.fq"), o("n2.fq")], ["-e", "rrna"], n=nb)
    os.environ["RD_DEVICE_PARSE"] = "0"
    run("pe_bgzf_to_gz_host_parse", bg, [o("q1.fq.gz"), o("q2.fq.gz")], ["-e", "rrna"], n=nb)
    run("pe_plain_to_plain_host_parse", [files[1], files[2]], [o("s1.fq"), o("s2.fq")], ["-e", "rrna"])
    del os.environ["RD_DEVICE_PARSE"]
    os.environ["RD_DEVICE_INFLATE"] = "0"
    run("pe_bgzf_to_gz_host_inflate", bg, [o("p1.fq.gz"), o("p2.fq.gz")], ["-e", "rrna"], n=nb)
    del os.environ["RD_DEVICE_INFLATE"]
    import hashlib

    def sha(name):      # of the text: .gz files through Python's gzip (every member's CRC-32 and ISIZE checked on the way)
        h = hashlib.sha1()
        with (gzip.open(o(name), "rb") if name.endswith(".gz") else open(o(name), "rb")) as fh:
            for chunk in iter(lambda: fh.read(1 << 24), b""):
                h.update(chunk)
        return h.hexdigest()
    # the same records whatever the route
    out["same_text"] = {"plain_to_plain == gz_to_gz == plain_to_plain_host_parse": len({sha("d1.fq"), sha("f1.fq.gz"), sha("s1.fq")}) == 1,
                        "bgzf_to_plain == bgzf_to_gz == host_parse == host_inflate": len({sha("n1.fq"), sha("m1.fq.gz"), sha("q1.fq.gz"), sha("p1.fq.gz")}) == 1,
                        "second mates likewise": len({sha("d2.fq"), sha("f2.fq.gz"), sha("s2.fq")}) == 1 and len({sha("n2.fq"), sha("m2.fq.gz"), sha("q2.fq.gz"), sha("p2.fq.gz")}) == 1}
    for f in ("m1.fq.gz", "m2.fq.gz", "n1.fq", "n2.fq", "p1.fq.gz", "p2.fq.gz", "q1.fq.gz", "q2.fq.gz", "s1.fq", "s2.fq"):
        os.remove(o(f))
    os.environ["RD_DEVICE_GZIP"] = "0"
    run("se_gz_to_gz_host_deflate", [files[1] + ".gz"], [o("j.fq.gz")])
    run("pe_gz_to_gz_host_deflate", [files[1] + ".gz", files[2] + ".gz"], [o("k1.fq.gz"), o("k2.fq.gz")], ["-e", "rrna"])
    run("pe_plain_to_gz_host_deflate", [files[1], files[2]], [o("l1.fq.gz"), o("l2.fq.gz")], ["-e", "rrna"])
    del os.environ["RD_DEVICE_GZIP"]
    out["output_bytes"] = {n: os.path.getsize(o(n)) for n in ("f1.fq.gz", "k1.fq.gz", "d1.fq") if os.path.exists(o(n))}
    shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
