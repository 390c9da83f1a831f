#!/usr/bin/env python3
"""bench.py - reads/s classified, 100 bp paired-end (BASELINE.json metric) on N MI355X of one node.

A "step" = one pass of the hot path over one batch of synthetic read pairs (timed region (ii)):
    read bytes in PINNED HOST memory -> H2D on a copy stream (double buffered: the bytes of step k+1 travel while step k
    computes) -> rd_classify(R1) + rd_classify(R2) (length bucketing, fused encoder, LSTM recurrence, FC, argmax)
    -> float64 re-evaluation of the ~15 reads per million whose margin is inside the fp32 noise band (deferred pass: the recurrence
       kernel's epilogue records them, rd_sync_results evaluates them, one workgroup each, beside the next step's recurrences)
    -> rd_pair_fuse(--ensure rrna) + counters -> D2H of the 1-byte pair labels into pinned host memory,
    and for N>1 the RCCL gather of the labels to rank 0. The post-pass of a step (float64 pass, fusion, D2H, gather) runs on a side
    stream and overlaps the recurrences of the next step; everything is inside the timed region.
Workload = BASELINE.json configs[2] ("10M paired-end 100 bp reads with --ensure rrna, 1 MI355X"): with the default
--steps 10 x 1,048,576 pairs/step = 10.5 M pairs (21 M reads) are classified inside the timed region.
For N>1 every rank gets its own shard of the same size (weak scaling), as the reads shard embarrassingly.
`python bench.py --gpus N` with N>1 and no WORLD_SIZE in the environment re-launches itself under
torch.distributed.run with N ranks (backend nccl = RCCL); launched under torchrun it uses the environment as is.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  config.kernel_only_reads_per_s - timed region (i): the same steps with the read bytes already resident in HBM
  roofline     - the recurrence kernel against the dense MFMA peak: the algorithmic FLOPs of the steps the kernel EXECUTES
                 (steps x 131072 + 1024 per read; the steps a prefix-state table row stands for are not counted -
                 frac_counting_table_steps has them) / average launch duration measured with hipEvents on the launch stream inside
                 the timed region (C ABI rd_profile_*); traffic = HBM bytes per launch from FETCH_SIZE/WRITE_SIZE collected by
                 rocprofv3 --pmc passes over a child run of this same command (gfx950 correction: FETCH_SIZE x2), against SURVEY
                 §8d's algorithmic bytes (113 B per 100 bp read) with the table rows and the 8-byte offsets listed separately
  encoder      - the standalone HBM-bound encoder kernels (reference tensor layouts): achieved GB/s against 8 TB/s
  cpu_baseline - the CPU oracle's restatement of ribodetector_cpu (padded BiLSTM over all L steps, batch 1024,
                 one batch per thread) timed on this box's host cores on a bounded sample of the same reads.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READ_LEN = 100
PEAKS = {"mfma_f32": 157.3, "simple": 157.3, "mfma_f16x3_t32": 2500.0}   # dense TFLOP/s, MI355X_MICROARCH.md
MFMA_FLOPS_PER_ALGO_FLOP = {"mfma_f32": 1, "simple": 1, "mfma_f16x3_t32": 3}   # f16x3 issues three f16 products per fp32 product
HBM_PEAK_GBPS = 8000.0
WORKLOADS = {"pe100": (True, 100, 100), "se100": (False, 100, 100), "pe150": (True, 150, 150), "var300": (False, 300, 300)}


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


def cpu_baseline(arena_np, n_reads, target_s=12.0):
    """Time the oracle's batched ribodetector_cpu restatement on all usable host cores; bounded to ~target_s seconds.
    Two rates: model only (bytes -> logits; the base lookup is fused into the port's input gather) and encode + model (the
    [1024, L, 4] fp32 one-hot tensor ribodetector_cpu feeds onnxruntime, detect_cpu.py:699-700, materialised by the C port
    before the model runs - the reference builds it with Python list comprehensions, ~120 us per read and core)."""
    import numpy as np
    from oracle import oracle as O
    ora = O.load_default()
    cores = usable_cores()
    off = np.arange(n_reads + 1, dtype=np.int64) * READ_LEN
    lens = np.full(n_reads, READ_LEN, dtype=np.int32)
    probe = min(n_reads, 1024 * min(cores, 8))
    t0 = time.time()
    ora.forward_padded(arena_np, off[:probe], lens[:probe], READ_LEN, batched=True, batch=1024, nthreads=cores)
    rate = probe / max(time.time() - t0, 1e-6)
    n = int(min(n_reads, max(1024 * cores, (rate * target_s) // (1024 * cores) * 1024 * cores)))
    t0 = time.time()
    cpu_logits = ora.forward_padded(arena_np, off[:n], lens[:n], READ_LEN, batched=True, batch=1024, nthreads=cores)
    dt = time.time() - t0
    # encode leg: one-hot [n, L, 4] fp32 by the C port (single thread per call here; n/cores reads per thread equivalent)
    ne = min(n, 65536)
    t1 = time.time()
    for i in range(0, ne, 1024):
        ora.encode_padded_batch(arena_np, off[i:i + 1025], READ_LEN)
    t_enc_per_read = (time.time() - t1) / ne / cores      # the encode parallelises over reads like the model does
    enc_model = 1.0 / (1.0 / (n / dt) + t_enc_per_read)
    return {"logits": cpu_logits, "n": n, "value": n / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "encode_plus_model_reads_per_s": enc_model, "os_cpu_count": os.cpu_count() or 0,
            "sample": "first %d reads of the rank-0 R1 stream (100 bp); oracle rdo_forward_padded_batched = the ribodetector_cpu "
                      "algorithm (padded BiLSTM over all 100 steps x 2 directions, batch 1024, one batch per thread) as a hand-vectorised "
                      "AVX-512/AVX2 + OpenMP C port, %d threads = usable cores of this container (os.cpu_count() = %d), %.1f s. "
                      "`value` is MODEL ONLY (ASCII bytes -> logits, base lookup fused); encode_plus_model adds materialising the "
                      "[1024,100,4] fp32 one-hot input in C (the reference's Python encoder is ~100x slower than that and would "
                      "dominate). onnxruntime is not installed, so this port stands in for ribodetector_cpu"
                      % (n, cores, os.cpu_count() or 0, dt)}


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


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` (N>1) outside torchrun: start N ranks of this script under torch.distributed.run."""
    import torch
    shared = os.environ.get("RD_LOCAL_DEVICE")        # tests: several ranks share one GPU and exchange labels over gloo
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if shared is None and have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d needs %d visible devices, this box has %d (one rank per GPU over RCCL; for a functional "
                         "run on fewer devices set RD_DIST_BACKEND=gloo RD_LOCAL_DEVICE=0)\n" % (args.gpus, args.gpus, have))
        raise SystemExit(3)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // args.gpus)))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def pmc_traffic(args, kernel_substr, timeout_s=240):
    """HBM bytes per launch of the recurrence kernel: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one
    pass) over a child run of this script (`--pmc-child`: same workload, same batch, resident inputs, 1 warm-up + 2 steps).
    Counter collection only - no trace domain is combined with --pmc. Returns (dict | None, error | None)."""
    import csv
    import glob
    import shutil
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "rocprofv3 not found"
    out = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="rd_pmc_", dir="/tmp")
        cmd = [rp, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
               "--pmc-child", "--workload", args.workload, "--variant", args.variant, "--pairs-per-step", str(args.pairs_per_step),
               "--ensure", args.ensure]
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RD_FORCE_DIST", "MASTER_PORT"):         # the counter child is a plain one-process run
            env.pop(k, None)
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            shutil.rmtree(d, ignore_errors=True)
            return None, "rocprofv3 --pmc %s timed out" % ctr
        vals = []
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    if kernel_substr in row["Kernel_Name"] and "kernel<true>" not in row["Kernel_Name"] and row["Counter_Name"] == ctr:   # <true> = the table build
                        vals.append(float(row["Counter_Value"]))
        shutil.rmtree(d, ignore_errors=True)
        if not vals:
            return None, "rocprofv3 --pmc %s: no rows for %s (rc %d): %s" % (ctr, kernel_substr, r.returncode, r.stdout.decode(errors="replace")[-300:])
        out[ctr + "_KB"] = sum(vals) / len(vals)
        out[ctr + "_launches"] = len(vals)
    out["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE_KB"] + out["WRITE_SIZE_KB"]) * 1024.0
    out["correction"] = "gfx950 FETCH_SIZE tallies 64 B per 128-B request: x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; KB = 1024 B"
    return out, None


def shutil_free(path):
    import shutil
    try:
        return shutil.disk_usage(path).free
    except OSError:
        return 0


def gzip_file(src, dst, level=4):
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


def e2e_record(torch, synth, arenas, offsets, lens, L, ensure, timed_calls=1, gz=False, threads=None, gz_in=None):
    """Timed region (iii) of SURVEY §8d: the whole `ribodetector` CLI - detect.main(): model load (incl. building the prefix-state
    table), native FASTQ parse, H2D, kernels, label D2H, output write - on a FASTQ file (pair) built from the rank-0 stream of this
    run in tmpfs (the reference flow: detect.py:464-499). One warm call (it pays one-off costs of the process: first pinned
    allocations, page cache), then `timed_calls` calls; the MEDIAN is reported. gz: the inputs are single-member .gz files and the
    outputs are written as .gz (the reference compresses by extension at level 5, detect.py:729-741) - the form sequencer data
    arrives in, bound by the host's inflate / deflate threads."""
    import shutil
    import tempfile
    import threading
    from ribodetector_amd import detect
    d = tempfile.mkdtemp(prefix="rd_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        n = int(lens.numel())
        ins, how = [], None
        for m, a in enumerate(arenas):
            p = os.path.join(d, "r_%d.fq" % (m + 1))
            synth.fastq_image_torch(a, offsets, lens, mate=m + 1).cpu().numpy().tofile(p)
            ins.append(p)
        plain_bytes = sum(os.path.getsize(p) for p in ins)
        if gz and gz_in == "bgzf":
            # BGZF inputs (what bgzip and this build's own writer produce): framed by the device writer, outside every timed region
            from ribodetector_amd.gz import DeviceGzip, eof_block
            dg = DeviceGzip(arenas[0].device)
            for p in ins:
                import numpy as np
                t = torch.from_numpy(np.fromfile(p, dtype=np.uint8)).to(arenas[0].device)
                rs = torch.zeros(n + 1, dtype=torch.int64, device=t.device)
                torch.cumsum(18 + 2 * lens.to(torch.int64), 0, out=rs[1:])
                o, info = dg.compress_selected(t, rs, torch.zeros(n, dtype=torch.int8, device=t.device), 0)
                torch.cuda.synchronize(t.device)
                with open(p + ".gz", "wb") as fh:
                    fh.write(o[: int(info[0])].cpu().numpy().tobytes())
                    fh.write(eof_block())
                os.remove(p)
                del t, o
            del dg
            ins, how = [p + ".gz" for p in ins], "BGZF members of 65,280 bytes (device writer)"
        elif gz and gz_in is not False:
            res = [None] * len(ins)

            def comp(i):
                res[i] = gzip_file(ins[i], ins[i] + ".gz")
            ths = [threading.Thread(target=comp, args=(i,)) for i in range(len(ins))]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            for p in ins:
                os.remove(p)
            ins, how = [p + ".gz" for p in ins], res[0]
        ext = ".fq.gz" if gz else ".fq"
        outs = [os.path.join(d, "non_%d%s" % (m + 1, ext)) for m in range(len(ins))]
        rrs = [os.path.join(d, "rrna_%d%s" % (m + 1, ext)) for m in range(len(ins))]
        argv = ["-l", str(L), "-i", *ins, "-o", *outs, "-r", *rrs] + (["-e", ensure] if len(ins) == 2 else []) + (["-t", str(threads)] if threads else [])
        calls = []
        for call in range(1 + timed_calls):
            for q in outs + rrs:                      # every call writes NEW files (truncating GBs of tmpfs pages is not the CLI's work)
                if os.path.exists(q):
                    os.remove(q)
            t0, c0 = time.perf_counter(), time.process_time()
            pr = detect.main(argv)
            dt, cpu = time.perf_counter() - t0, time.process_time() - c0
            calls.append({"seconds": dt, "reads_per_s": len(ins) * n / dt, "main_thread_s": {k: round(v, 4) for k, v in pr._stage_s.items()},
                          "host_cores_busy": round(cpu / dt, 2),      # CPU seconds of ALL threads of the process / wall seconds
                          "thread_cpu_s": dict(getattr(pr, "thread_cpu_s", {})),   # the pipeline's Python threads (native helpers are not in it)
                          "load_model_s": round(pr.timing["load_model_s"], 4), "detect_s": round(pr.timing["detect_s"], 4),
                          "prefix_k": pr.timing["prefix_k"], "reads_per_s_after_model_load": len(ins) * n / pr.timing["detect_s"]})
            del pr
        timed = sorted(calls[1:], key=lambda c: c["seconds"])
        med = timed[len(timed) // 2]
        out_bytes = sum(os.path.getsize(p) for p in outs + rrs)
        return {"reads_per_s": med["reads_per_s"], "seconds": med["seconds"], "files": len(ins),
                "reads_per_s_after_model_load": med["reads_per_s_after_model_load"], "timed_calls": timed_calls,
                "host_cores_busy": med["host_cores_busy"],
                "spread": (timed[-1]["seconds"] - timed[0]["seconds"]) / med["seconds"],
                "records_per_file": n, "input_bytes": sum(os.path.getsize(p) for p in ins), "plain_input_bytes": plain_bytes,
                "output_bytes": out_bytes, "input_compressor": how,
                "threads_flag": threads or 10,
                "what": "whole detect.main() call on FASTQ in tmpfs, %s, -t %d%s: model load + prefix table build + parse + H2D + "
                        "kernels + D2H + write; median of %d call(s) after one warm call"
                        % (("BGZF -> gz (RD_DEVICE_INFLATE=%s)" % os.environ.get("RD_DEVICE_INFLATE", "auto") if gz_in == "bgzf" else
                            "gz -> gz" if gz_in is not False else "plain -> gz") if gz else "plain -> plain", threads or 10,
                           "" if threads else " (the CLI's default)", timed_calls),
                "warm_call": calls[0], "calls": calls[1:]}
    finally:
        shutil.rmtree(d, ignore_errors=True)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs-per-step", type=int, default=1 << 20)
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--ensure", default="rrna")
    ap.add_argument("--workload", default="pe100", choices=sorted(WORKLOADS),
                    help="pe100 = BASELINE configs[2] (the metric's configuration, default); se100 = configs[1]; pe150 = the per-GPU "
                         "shard of configs[3]; var300 = the per-GPU shard of configs[4] (40-300 bp, -l 300)")
    ap.add_argument("--resident-only", action="store_true", help="time region (i) only: read bytes resident in HBM (diagnostics)")
    ap.add_argument("--inline-refine", action="store_true",
                    help="leave the float64 refine pass inside rd_classify (on the main stream) instead of overlapping it with the next "
                         "step's recurrences on a side stream")
    ap.add_argument("--refine-scan", action="store_true",
                    help="A/B only: the round-3 placement of the float64 pass - rd_refine (a scan of all logits) issued by this script on the "
                         "side stream - instead of the deferred pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the short extra measurement of the exact-fp32 MFMA kernel")
    ap.add_argument("--no-encoder", action="store_true", help="skip the standalone encoder kernels")
    ap.add_argument("--no-e2e", action="store_true", help="skip timed region (iii): the whole CLI on a FASTQ file built from the rank-0 stream")
    ap.add_argument("--e2e-records", type=int, default=1 << 24,
                    help="records per file of the long region-(iii) measurement (default 16 Mi; 0 = only the one-step batch)")
    ap.add_argument("--traffic", default="live", choices=["live", "off"],
                    help="live: collect FETCH_SIZE/WRITE_SIZE with rocprofv3 --pmc over a child run of this command (adds ~40 s)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist
    from ribodetector_amd import _native as N
    from ribodetector_amd import dist as rdist
    from ribodetector_amd import synth
    from ribodetector_amd.model import model as module_arch
    from ribodetector_amd.parse_config import ConfigParser

    rank, world, local = rdist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev = torch.device("cuda", int(os.environ.get("RD_LOCAL_DEVICE", local)))   # override: several ranks on one GPU (tests only)
    torch.cuda.set_device(dev)
    multi = rdist.active()                                 # several ranks - or ONE rank under RD_FORCE_DIST=1 (a one-rank RCCL
    backend = dist.get_backend() if multi else None        # communicator: the collectives of the N>1 run, executed on a 1-GPU box)

    cfg = ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json"))
    model = cfg.init_obj("arch", module_arch)
    model.load_state_dict(cfg.load_state_dict("mcc"))
    model.to(dev).eval()
    model.set_variant(args.variant)
    model.set_prefix_table(os.environ.get("RD_PREFIX_K", "auto"))   # opt-in since round 4 (SeqModel.to() allocates nothing by itself)
    variant = "mfma_f16x3_t32" if args.variant == "auto" else args.variant

    P = args.pairs_per_step
    paired, RL, MAXLEN = WORKLOADS[args.workload]
    steps = 2 if args.pmc_child else args.steps
    nslices = max(1, min(steps, 4))                       # distinct batches, reused cyclically
    r1 = [synth.reads_torch(P, RL, seed=2000 + 100 * rank + i, device=dev) for i in range(nslices)]
    r2 = [synth.reads_torch(P, RL, seed=7000 + 100 * rank + i, device=dev) for i in range(nslices)] if paired else None
    offs = r1[0][1][:-1].contiguous()
    lens = r1[0][2]
    if args.workload == "var300":                          # lengths ~ U{40..300}; rows keep their 300-byte stride in the arena
        g = torch.Generator(device=dev)
        g.manual_seed(4 + rank)
        lens = torch.randint(40, 301, (P,), generator=g, device=dev, dtype=torch.int32)
    # FLOPs/read = T x 131072 + 1024 (forward recurrence + FC), bytes/read = T + 4 (len) + 8 (logits) + 1 (label).
    # With a prefix-state table a read whose first k bases are A/C/G/T(U) starts from its table row, k steps in:
    # those k steps are looked up, not computed. roofline.achieved counts the steps the kernel EXECUTES (what the MFMA pipe does);
    # counting every step of every read (the table's steps as if computed) is reported beside it as *_counting_table_steps.
    PK = model.prefix_k
    steps_full = torch.clamp(lens, max=MAXLEN).to(torch.int64)
    def executed(pk):
        """(steps the kernel executes per launch, table rows read per launch), averaged over the launches of the run's batches"""
        if not pk:
            return float(steps_full.sum().item()), 0.0
        isb = torch.zeros(256, dtype=torch.bool, device=dev)
        isb[[65, 67, 71, 84, 85]] = True
        se, rr, nl = 0.0, 0.0, 0
        for t in r1 + (r2 or []):
            first = isb[t[0].view(P, RL)[:, :pk].long()].all(1) & (steps_full > pk)
            se += float((steps_full - pk * first.to(torch.int64)).sum().item())
            rr += float(first.sum().item())
            nl += 1
        return se / nl, rr / nl
    steps_exec_sum, rows_read = executed(PK)
    flops_all_steps = float(steps_full.sum().item()) * 131072 + 1024.0 * P
    flops_per_launch = steps_exec_sum * 131072 + 1024.0 * P
    bytes_survey = float(steps_full.sum().item()) + P * (4 + 8 + 1)       # 113 B per 100 bp read
    bytes_offsets = P * 8.0                                               # this ABI's int64 start offset per read (not in §8d's count)
    bytes_table = rows_read * 1024.0                                      # ONE 1 KiB table row per read that starts from the table
    exec_steps_frac = steps_exec_sum / float(steps_full.sum().item())
    nm = 2 if paired else 1
    # two sets of result buffers: the post-pass of step i (side stream) runs while the recurrences of step i+1 (main stream) write
    # the other set
    lgs = [[torch.empty((P, 2), dtype=torch.float32, device=dev) for _ in range(nm)] for _ in range(2)]
    lab8s = [torch.empty((P,), dtype=torch.uint8, device=dev) for _ in range(2)]
    counts = torch.zeros(3, dtype=torch.int64, device=dev)
    gathered = torch.empty(P * world, dtype=torch.int8, device=dev) if (multi and rank == 0) else None
    cur = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(dev)
    pipelined = not (args.inline_refine or args.pmc_child)   # counter passes: one stream, so that no other kernel runs beside the one counted
    if pipelined and args.refine_scan:
        model.set_refine(0.0)
    elif pipelined:
        # deferred float64 pass (C ABI rd_set_refine_async): the recurrence kernel's epilogue records the reads inside the noise band;
        # rd_sync_results - called on the SIDE stream, in the step's post-pass - evaluates them, one workgroup each, on the model's
        # stream beside the next step's recurrences. No scan launch over the logits (rounds 2-3 issued rd_refine here: 1.4 ms each).
        model.set_refine_async(16)
    ev_post = [None, None]                                 # post-pass that last read result set k

    def compute(i, a1, a2, after=None):
        """one step: the recurrences of both mates on the main stream; on the side stream (overlapping the next step's recurrences)
        the float64 re-evaluation of the reads inside the noise band, pair fusion / counters and `after(labels)` (label D2H,
        RCCL gather). Returns the event that marks the end of the step's post-pass."""
        k = i & 1
        lg, lab8 = lgs[k], lab8s[k]
        if ev_post[k] is not None:
            cur.wait_event(ev_post[k])
        model.classify_bytes(a1, offs, lens, MAXLEN, want_labels=not paired, logits=lg[0], labels=None if paired else lab8)
        if paired:
            model.classify_bytes(a2, offs, lens, MAXLEN, want_labels=False, logits=lg[1])
        ev_main = torch.cuda.Event()
        ev_main.record(cur)
        post = side if pipelined else cur
        with torch.cuda.stream(post):
            post.wait_event(ev_main)
            if pipelined and not args.refine_scan:
                model.sync_results()                       # (current stream = the side stream: the main stream never waits)
            if args.refine_scan and pipelined and not (paired and args.ensure == "none"):
                model.refine(a1, offs, lens, MAXLEN, lg[0], None if paired else lab8)
                if paired:
                    model.refine(a2, offs, lens, MAXLEN, lg[1], None)
            if paired:
                if args.ensure == "none":                  # pair margin decides (reference detect.py:657): the scan form, with the mate
                    model.refine(a1, offs, lens, MAXLEN, lg[0], None, lg[1])
                    model.refine(a2, offs, lens, MAXLEN, lg[1], None, lg[0])
                lab = module_arch.pair_fuse(lg[0], lg[1], args.ensure, counts)
            else:
                module_arch.count_labels(lab8, counts)
                lab = lab8.view(torch.int8)
            fin = after(lab) if after else None
            ev_post[k] = torch.cuda.Event(blocking=True)   # the host SLEEPS in .synchronize() (8 ranks share 16 host cores)
            ev_post[k].record(post)
        return ev_post[k], fin

    def exchange(lab):
        if multi:
            _, fin = rdist.gather_labels(lab, P * world, dst=0, async_op=True, out=gathered)
            return fin
        return None

    def sync():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def run_resident(nsteps):
        """region (i): inputs resident in HBM"""
        pend = []
        for i in range(nsteps):
            _, f = compute(i, r1[i % nslices][0], r2[i % nslices][0] if paired else None, exchange)
            if f:
                pend.append(f)
        for f in pend:
            f()

    if args.pmc_child:                                      # profiled by the parent's rocprofv3 --pmc passes; prints nothing
        run_resident(3)
        torch.cuda.synchronize(dev)
        return

    # ---- region (ii): pinned host bytes -> H2D (copy stream, double buffered) -> kernels -> labels D2H (pinned) ----------------
    host_in = [[t[0].cpu().pin_memory() for t in r1]] + ([[t[0].cpu().pin_memory() for t in r2]] if paired else [])
    cs = torch.cuda.Stream(dev)
    dbuf = [[torch.empty_like(r1[0][0]) for _ in range(nm)] for _ in range(2)]
    host_lab = [torch.empty((P,), dtype=torch.int8).pin_memory() for _ in range(2)]
    ev_ready = [torch.cuda.Event() for _ in range(2)]      # H2D of the slot finished
    ev_free = [None, None]                                 # every kernel that reads the slot's bytes finished (= the post-pass event)

    def h2d(i):
        s = i & 1
        with torch.cuda.stream(cs):
            if ev_free[s] is not None:
                cs.wait_event(ev_free[s])
            for m in range(nm):
                dbuf[s][m].copy_(host_in[m][i % nslices], non_blocking=True)
            ev_ready[s].record(cs)

    def run_device_path(nsteps):
        pend = []
        h2d(0)
        for i in range(nsteps):
            s = i & 1
            if i + 1 < nsteps:
                h2d(i + 1)
            cur.wait_event(ev_ready[s])
            if ev_free[s] is not None:                     # the host consumed the labels of step i-2 (bounds the run-ahead)
                while not ev_free[s].query():              # sleep-poll: hipEventSynchronize spins a host core per rank, and the 8
                    time.sleep(2e-4)                       # ranks of a node share 16 of them (0.2 ms against a 60 ms step)

            def after(lab, s=s):
                host_lab[s].copy_(lab, non_blocking=True)  # 1 B per pair into pinned memory, behind the post-pass
                return exchange(lab)
            ev_free[s], f = compute(i, dbuf[s][0], dbuf[s][1] if paired else None, after)
            if f:
                pend.append(f)
        for f in pend:
            f()

    rank_info = None
    if multi:                                               # create the communicators (incl. the point-to-point ones the gather
        t0 = time.perf_counter()                            # uses) outside the timed region, whatever --warmup is
        f = exchange(torch.zeros((P,), dtype=torch.int8, device=dev))
        if f:
            f()
        sync()
        first_gather_s = time.perf_counter() - t0
        # Self-check of the exchange before anything is timed (the first N > 1 run on real hardware is a one-shot): step 0 of this
        # rank's shard once WITHOUT the gather -> this rank's own label counts, once THROUGH it -> rank 0 counts the slice it
        # received from every rank. Any difference ends the run with exit code 4 on every rank.
        keep = {}
        ev, _ = compute(0, r1[0][0], r2[0][0] if paired else None, lambda lab: keep.__setitem__("lab", lab.clone()))
        ev.synchronize()
        mine = [int((keep["lab"] == v).sum()) for v in (0, 1, -1)]
        ev, f = compute(1, r1[0][0], r2[0][0] if paired else None, exchange)
        got_all = f() if f else None                        # rank 0: the [P * world] labels (host memory under gloo)
        sync()
        try:
            uuid = str(torch.cuda.get_device_properties(dev).uuid)
        except Exception:
            uuid = "?"
        infos = [None] * world
        dist.all_gather_object(infos, {"rank": rank, "local_rank": local, "device": str(dev), "device_uuid": uuid, "pid": os.getpid(),
                                       "first_gather_s": round(first_gather_s, 4), "step0_label_counts": mine})
        verdict = [None]
        if rank == 0:
            bad = []
            if os.environ.get("RD_BENCH_CORRUPT_GATHER") == "1":      # (tests: the check must catch a wrong gather)
                got_all[P * world - 1] = 1 - got_all[P * world - 1]
            for r in range(world):
                sl = got_all[r * P:(r + 1) * P]
                got = [int((sl == v).sum()) for v in (0, 1, -1)]
                if got != infos[r]["step0_label_counts"]:
                    bad.append({"rank": r, "gathered": got, "local": infos[r]["step0_label_counts"]})
            verdict = [bad]
            if len({i["device_uuid"] for i in infos}) != world and "RD_LOCAL_DEVICE" not in os.environ:
                sys.stderr.write("bench.py: WARNING: %d ranks on %d distinct devices\n" % (world, len({i["device_uuid"] for i in infos})))
        dist.broadcast_object_list(verdict, src=0)
        if verdict[0]:
            if rank == 0:
                sys.stderr.write("bench.py: label gather self-check FAILED: %s\n" % json.dumps(verdict[0]))
            dist.barrier()
            raise SystemExit(4)
        rank_info = infos
        counts.zero_()
    timed_path = run_resident if args.resident_only else run_device_path
    timed_path(args.warmup)
    sync()
    counts.zero_()
    model.profile_enable(True)
    sync()
    th0 = thread_cpu()
    t0, c0 = time.perf_counter(), time.process_time()
    timed_path(args.steps)
    sync()
    dt, cpu_s = time.perf_counter() - t0, time.process_time() - c0      # cpu_s: user + system time of ALL threads of this rank
    th1 = thread_cpu()
    by_thread = sorted(((th1[t][1] - th0.get(t, (None, 0.0))[1], th1[t][0]) for t in th1), reverse=True)
    by_thread = [{"thread": nm, "cpu_s": round(c, 3)} for c, nm in by_thread if c >= 0.01][:6]
    launches, kms = model.profile_read()
    model.profile_enable(False)
    rdist.reduce_counts(counts)
    on_dev = not multi or backend == "nccl"
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if on_dev else "cpu")
    if multi:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    cpu_ranks = [cpu_s]
    if multi:
        cpu_ranks = [None] * world
        dist.all_gather_object(cpu_ranks, cpu_s)
    total_pairs = P * args.steps * world
    c = counts.cpu().tolist()
    assert c[0] + c[1] + c[2] == total_pairs, (c, total_pairs)
    host_check = int((host_lab[(args.steps - 1) & 1] != 0).sum()) if not args.resident_only and args.steps > 0 else None

    # ---- region (i): the same steps with the bytes resident in HBM -------------------------------------------------------------
    dt_res = None
    if not args.resident_only:
        sync()
        t1 = time.perf_counter()
        run_resident(args.steps)
        sync()
        dt_res = time.perf_counter() - t1
        tr = torch.tensor([dt_res], dtype=torch.float64, device=dev if on_dev else "cpu")
        if multi:
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        dt_res = float(tr.item())

    if rank == 0:
        mult = 2.0 if paired else 1.0
        avg_ms = kms / max(launches, 1)
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if launches else None
        base = ("mfma_f16x3_t32" if variant.startswith(("mfma_f16x3_t32", "t32_")) else "mfma_f32" if variant.startswith("mfma_f32") else variant)
        peak = PEAKS[base]
        kname = "rd_lstm_%s_kernel" % base
        traffic, traffic_src = None, "not collected (--traffic off or N>1)"
        if args.traffic == "live" and world == 1:
            tj, err = pmc_traffic(args, "rd_lstm_")
            if tj:
                traffic = tj["hbm_bytes_per_launch"]
                traffic_src = dict(tj, how="rocprofv3 --pmc passes over a child run of this command in this invocation")
            else:
                traffic_src = "failed: %s" % err
        wl_text = {"pe100": "BASELINE configs[2]: paired-end 100 bp, --ensure %s" % args.ensure,
                   "se100": "BASELINE configs[1]: single-end 100 bp",
                   "pe150": "BASELINE configs[3] per-GPU shard: paired-end 150 bp, -l 150, --ensure %s" % args.ensure,
                   "var300": "BASELINE configs[4] per-GPU shard: single-end 40-300 bp, -l 300, length-bucketed"}[args.workload]
        region = ("read bytes resident in HBM (region i)" if args.resident_only else
                  "read bytes start in pinned host memory, H2D double-buffered on a copy stream, labels D2H to pinned memory (SURVEY 8d region ii)")
        out = {
            "metric": "reads/sec classified, 100 bp paired-end" if args.workload == "pe100" else "reads/sec classified, " + args.workload,
            "value": mult * total_pairs / dt,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if base != "mfma_f16x3_t32" else "f16x3-split (f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": wl_text + ", %d %s/step/GPU x %d steps (%.1f M total); %s"
                                   % (P, "pairs" if paired else "reads", args.steps, total_pairs / 1e6, region),
                       "timed_region": "i" if args.resident_only else "ii",
                       "timed_regions": "value = (ii) pinned host bytes -> host labels; kernel_only_reads_per_s = (i) bytes resident in HBM; "
                                        "e2e_cli_reads_per_s = (iii) the whole CLI on a FASTQ file in tmpfs (SURVEY 8d)",
                       "kernel_only_reads_per_s": (mult * total_pairs / dt_res) if dt_res else None,
                       "device_path_over_kernel_only": (dt_res / dt) if dt_res else None,
                       "pairs_per_s": (total_pairs / dt) if paired else None, "per_step_per_gpu": P,
                       "read_len": RL if args.workload != "var300" else "40-300",
                       "ensure": args.ensure if paired else None,
                       "kernel_variant": variant,
                       "precision": ("every fp32 product h*w is formed as three f16 MFMA products of hi/lo parts with fp32 accumulation; "
                                     "parity_sample below is this run's check against the fp32 CPU port"
                                     if base == "mfma_f16x3_t32" else "fp32"),
                       "parallelism": "reads sharded x%d, label gather to rank 0" % world,
                       "host_cpu_seconds_per_rank_in_timed_region": [round(c, 4) for c in cpu_ranks],
                       "host_cores_busy": sum(cpu_ranks) / dt, "host_cores_usable": usable_cores(),
                       "host_cpu_seconds_by_thread_rank0": by_thread,
                       "rccl_ranks": world, "dist_backend": backend, "forced_dist": bool(multi and world == 1),
                       "ranks": rank_info, "gather_self_check": "passed" if rank_info else None,
                       "prefix_table": {"k": PK, "bytes": (4 ** PK + 1) * 1024 if PK else 0,
                                        "what": "recurrence state after every sequence of k bases, built by the kernel itself at model load; "
                                                "a read whose first k bases are A/C/G/T starts from its row (bit-identical logits, "
                                                "tests/test_gpu_prefix.py); alt_no_prefix_table = the same steps with k = 0"},
                       "refine": {"band": module_arch.SeqModel.REFINE_DEFAULT, "what": "reads whose margin is inside the band are "
                                  "re-evaluated in float64 (labels of the exact function); inside the timed region",
                                  "placement": ("rd_refine scan on the side stream (round-3 form, --refine-scan)" if (pipelined and args.refine_scan) else
                                                "deferred (rd_set_refine_async): candidates recorded by the recurrence kernel's epilogue, "
                                                "evaluated by rd_sync_results on the post-pass stream beside the next step's recurrences")
                                  if pipelined else "inline in rd_classify"},
                       "label_counts": {"non_rrna": c[0], "rrna": c[1], "unclassified": c[2]},
                       "host_labels_nonzero_last_step": host_check},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": (achieved / peak) if achieved else None, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kname, "launches": launches, "avg_launch_ms": avg_ms,
                         "what": "achieved = algorithmic FLOPs of the steps the kernel EXECUTES (steps x 131072 + 1024 per read, SURVEY 8d) / "
                                 "avg_launch_ms; the steps a prefix-state table row stands for are looked up, not computed, and are not "
                                 "counted here (frac_counting_table_steps counts them: reads/s x SURVEY 8d FLOPs per read / peak)",
                         "algorithmic_flops_per_launch": flops_per_launch,
                         "algorithmic_flops_per_launch_counting_table_steps": flops_all_steps,
                         "achieved_counting_table_steps": (flops_all_steps / (avg_ms * 1e-3) / 1e12) if launches else None,
                         "frac_counting_table_steps": (flops_all_steps / (avg_ms * 1e-3) / 1e12 / peak) if launches else None,
                         "steps_executed_over_steps": exec_steps_frac,
                         "mfma_flops_executed_per_algorithmic_flop": MFMA_FLOPS_PER_ALGO_FLOP[base],
                         "mfma_pipe_frac": (achieved * MFMA_FLOPS_PER_ALGO_FLOP[base] / peak) if achieved else None,
                         "algorithmic_bytes_per_launch": bytes_survey,
                         "table_bytes_per_launch": bytes_table,
                         "offset_bytes_per_launch": bytes_offsets,
                         "traffic_over_algorithmic": (traffic / bytes_survey) if traffic else None,
                         "traffic_over_algorithmic_plus_table_and_offsets": (traffic / (bytes_survey + bytes_table + bytes_offsets)) if traffic else None,
                         "traffic_note": "algorithmic_bytes_per_launch = SURVEY 8d (T + 4 + 8 + 1 B per read). The kernel also reads an 8-byte "
                                         "start offset per read (this ABI's index) and, with the prefix-state table, ONE 1 KiB row per read "
                                         "that starts from it (table_bytes_per_launch: ~9x the 8d bytes, read once, 40 GB/s of the 8 TB/s; "
                                         "they buy k of the T steps - alt_no_prefix_table runs without them)" if PK else
                                         "algorithmic_bytes_per_launch = SURVEY 8d (T + 4 + 8 + 1 B per read); + an 8-byte start offset per read"},
        }
        def alt_run(label):
            """the same timed region and step count as `value` with the model as it is configured now"""
            timed_path(1)
            sync()
            model.profile_enable(True)
            t1 = time.perf_counter()
            timed_path(args.steps)
            sync()
            d1 = time.perf_counter() - t1
            l2, k2 = model.profile_read()
            model.profile_enable(False)
            return d1, k2 / max(l2, 1), l2

        if world == 1 and not args.no_alt and base != "mfma_f32":
            # the same steps on the exact-fp32 MFMA kernel (v_mfma_f32_16x16x4_f32) - the strict-precision line, against the fp32-MFMA
            # roofline of SURVEY §8d. Same timed region and step count as `value`; the kernel has its own prefix-state table (round 4:
            # set_variant rebuilds the table with the kernel that will use it), so the same steps are executed / looked up.
            model.set_variant("mfma_f32")
            pk32 = model.prefix_k
            se32, _ = executed(pk32)
            d1, ms32, l2 = alt_run("fp32")
            model.set_variant(args.variant)
            f_exec, pk_f = se32 * 131072 + 1024.0 * P, PEAKS["mfma_f32"]
            out["alt_fp32_kernel"] = {"kernel": "rd_lstm_mfma_f32_kernel", "value": mult * P * args.steps / d1, "unit": "reads/s",
                                      "steps": args.steps, "ms_per_step": 1e3 * d1 / args.steps,
                                      "timed_region": "i" if args.resident_only else "ii", "dtype": "f32", "prefix_k": pk32,
                                      "what": "the same timed region and steps as `value` on the exact-fp32 MFMA kernel with its own prefix-state "
                                              "table: the line for a reader who does not credit the f16x3 split",
                                      "roofline": {"bound": "mfma", "achieved": f_exec / (ms32 * 1e-3) / 1e12, "peak": pk_f, "unit": "TFLOP/s",
                                                   "frac": f_exec / (ms32 * 1e-3) / 1e12 / pk_f, "avg_launch_ms": ms32, "launches": l2,
                                                   "frac_counting_table_steps": flops_all_steps / (ms32 * 1e-3) / 1e12 / pk_f,
                                                   "steps_executed_over_steps": se32 / float(steps_full.sum().item())}}
        if world == 1 and not args.no_alt and PK:
            # the same steps without the prefix-state table (every read steps over all its bases)
            model.set_prefix_table(0)
            d1, ms0, l2 = alt_run("k0")
            model.set_prefix_table(PK)
            a2 = flops_all_steps / (ms0 * 1e-3) / 1e12
            out["alt_no_prefix_table"] = {"kernel": kname, "value": mult * P * args.steps / d1, "unit": "reads/s", "steps": args.steps,
                                          "ms_per_step": 1e3 * d1 / args.steps, "timed_region": "i" if args.resident_only else "ii",
                                          "what": "the same timed region and steps as `value` with every read stepping over all its bases "
                                                  "(k = 0): the rate to quote if the prefix-state table is not credited",
                                          "roofline": {"bound": "mfma", "achieved": a2, "peak": peak, "unit": "TFLOP/s", "frac": a2 / peak,
                                                       "avg_launch_ms": ms0, "launches": l2}}
        if world == 1 and not args.no_encoder:
            try:
                out["encoder"] = encoder_record(torch, N, dev, r1[0][0], offs, lens, P, MAXLEN)
            except Exception as e:
                out["encoder"] = {"error": repr(e)}
        if world == 1 and not args.no_encoder and args.workload != "var300":
            try:
                lab_now = (host_lab[(args.steps - 1) & 1].to(dev) if not args.resident_only and args.steps > 0 else
                           torch.zeros(P, dtype=torch.int8, device=dev))
                out["device_gzip"] = gzip_record(torch, synth, dev, r1[(args.steps - 1) % nslices][0], r1[0][1], lens, lab_now)
            except Exception as e:
                out["device_gzip"] = {"error": repr(e)}
        if world == 1 and not args.no_e2e and not multi:
            try:
                # the batch of one step: the pipeline-fill-bound point (a 2 M-read input is 0.15 s of CLI)
                ne = min(P, 1 << 20)
                e2e = e2e_record(torch, synth, [r1[0][0][: ne * RL]] + ([r2[0][0][: ne * RL]] if paired else []),
                                 r1[0][1][: ne + 1], lens[:ne].contiguous(), MAXLEN, args.ensure)
                out["e2e_cli"] = {"one_step_batch": e2e}
                # long enough to be a number: every slice of the stream, repeated up to 16 Mi records per file (>= 1.5 s of CLI);
                # median of three calls after a warm one. Then the same flow gz -> gz on the un-repeated slices.
                if nslices > 1 and ne == P and args.e2e_records > 0:
                    free_shm = shutil_free("/dev/shm")
                    rec_bytes = nm * (2 * RL + 20)
                    rep = max(1, args.e2e_records // (nslices * P))
                    while rep > 1 and free_shm < 2.6 * rep * nslices * P * rec_bytes:
                        rep //= 2
                    if free_shm > 2.6 * rep * nslices * P * rec_bytes:
                        nbig = rep * nslices * P
                        offs_l = torch.arange(nbig + 1, dtype=torch.int64, device=dev) * RL
                        big = e2e_record(torch, synth, [torch.cat([t[0] for t in r1] * rep)] + ([torch.cat([t[0] for t in r2] * rep)] if paired else []),
                                         offs_l, lens.repeat(rep * nslices), MAXLEN, args.ensure, timed_calls=3)
                        out["e2e_cli"]["large"] = big
                        out["e2e_cli"].update({k: big[k] for k in ("reads_per_s", "seconds", "records_per_file", "files", "spread", "what")})
                        out["config"]["e2e_cli_reads_per_s"] = big["reads_per_s"]
                        ng = nslices * P
                        gzr = e2e_record(torch, synth, [torch.cat([t[0] for t in r1])] + ([torch.cat([t[0] for t in r2])] if paired else []),
                                         offs_l[: ng + 1], lens.repeat(nslices), MAXLEN, args.ensure, timed_calls=2, gz=True)
                        out["e2e_cli"]["gz_to_gz"] = gzr
                        out["e2e_cli"]["gz_to_gz_reads_per_s"] = gzr["reads_per_s"]
                        out["config"]["e2e_cli_gz_to_gz_reads_per_s"] = gzr["reads_per_s"]
                        # the same with -t = the usable host cores: with the deflate on the GPU every core can inflate the inputs
                        gza = e2e_record(torch, synth, [torch.cat([t[0] for t in r1])] + ([torch.cat([t[0] for t in r2])] if paired else []),
                                         offs_l[: ng + 1], lens.repeat(nslices), MAXLEN, args.ensure, timed_calls=2, gz=True, threads=usable_cores())
                        out["e2e_cli"]["gz_to_gz_all_cores"] = gza
                        out["config"]["e2e_cli_gz_to_gz_all_cores_reads_per_s"] = gza["reads_per_s"]
                        # plain -> gz: no inflate, so the GPU is the bound - recurrences plus the deflate of every chunk
                        p2g = e2e_record(torch, synth, [torch.cat([t[0] for t in r1])] + ([torch.cat([t[0] for t in r2])] if paired else []),
                                         offs_l[: ng + 1], lens.repeat(nslices), MAXLEN, args.ensure, timed_calls=2, gz=True, gz_in=False)
                        out["e2e_cli"]["plain_to_gz"] = p2g
                        out["config"]["e2e_cli_plain_to_gz_reads_per_s"] = p2g["reads_per_s"]
                        # BGZF -> gz: the members of the inputs inflated on the GPU (the default for such files), then by the host's
                        # member decoder on the same files
                        old = os.environ.get("RD_DEVICE_INFLATE")
                        try:
                            for key, v in (("bgzf_to_gz", None), ("bgzf_to_gz_host_inflate", "0")):
                                os.environ.pop("RD_DEVICE_INFLATE", None)
                                if v is not None:
                                    os.environ["RD_DEVICE_INFLATE"] = v
                                out["e2e_cli"][key] = e2e_record(torch, synth, [torch.cat([t[0] for t in r1])] + ([torch.cat([t[0] for t in r2])] if paired else []),
                                                                 offs_l[: ng + 1], lens.repeat(nslices), MAXLEN, args.ensure, timed_calls=2, gz=True, gz_in="bgzf")
                                out["config"]["e2e_cli_%s_reads_per_s" % key] = out["e2e_cli"][key]["reads_per_s"]
                        finally:
                            os.environ.pop("RD_DEVICE_INFLATE", None)
                            if old is not None:
                                os.environ["RD_DEVICE_INFLATE"] = old
                if "reads_per_s" not in out["e2e_cli"]:
                    out["e2e_cli"].update({k: e2e[k] for k in ("reads_per_s", "seconds", "records_per_file", "files", "spread", "what")})
                    out["config"]["e2e_cli_reads_per_s"] = e2e["reads_per_s"]
            except Exception as e:
                out["e2e_cli"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1 and args.workload in ("pe100", "se100"):
            try:
                nb = min(P, 400000)
                cb = cpu_baseline(r1[0][0][: nb * READ_LEN].cpu().numpy(), nb)
                cpu_logits, ns = cb.pop("logits"), cb.pop("n")
                out["cpu_baseline"] = cb
                out["config"]["gpu_over_cpu"] = out["value"] / cb["value"]
                # the baseline's logits double as a parity check of the timed kernel on the same reads (same semantics:
                # ribodetector_cpu's padded input = the 'padded' switch of the HIP path)
                import numpy as np
                model.set_semantics("padded")
                model.set_refine(module_arch.SeqModel.REFINE_DEFAULT)      # the library default: refine pass inside rd_classify
                g_logits, g_labels = model.classify_bytes(r1[0][0], offs[:ns].contiguous(), lens[:ns].contiguous(), MAXLEN)
                model.set_semantics("packed")
                g_logits = g_logits.cpu().numpy()
                err = np.abs(g_logits - cpu_logits).max(axis=1)
                margin = np.abs(cpu_logits[:, 1] - cpu_logits[:, 0])
                diff = np.flatnonzero(g_labels.cpu().numpy() != (cpu_logits[:, 1] > cpu_logits[:, 0]))
                out["parity_sample"] = {"reads": int(ns), "vs": "cpu_baseline logits (fp32 C port of ribodetector_cpu), same reads",
                                        "max_abs_logit_err": float(err.max()), "p9999_abs_logit_err": float(np.quantile(err, 0.9999)),
                                        "label_mismatches": int(len(diff)),
                                        "largest_margin_among_mismatches": float(margin[diff].max()) if len(diff) else None}
            except Exception as e:  # the checker is not the product: report, don't hide
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
        sys.stdout.flush()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
