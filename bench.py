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
import ribodetector_amd      # noqa: E402,F401  (sets ROC_SIGNAL_POOL_SIZE before the HIP runtime starts: ribodetector_amd/__init__.py)
sys.path.insert(1, os.path.join(ROOT, "tools"))
from e2e_bench import encoder_record, gzip_record, thread_cpu, usable_cores      # noqa: E402  (tools/e2e_bench.py)

READ_LEN = 100
PEAKS = {"mfma_f32": 157.3, "simple": 157.3, "mfma_f16x3_t32": 2500.0}   # dense TFLOP/s, MI355X_MICROARCH.md
MFMA_FLOPS_PER_ALGO_FLOP = {"mfma_f32": 1, "simple": 1, "mfma_f16x3_t32": 3}   # f16x3 issues three f16 products per fp32 product
HBM_PEAK_GBPS = 8000.0
WORKLOADS = {"pe100": (True, 100, 100), "se100": (False, 100, 100), "pe150": (True, 150, 150), "var300": (False, 300, 300)}


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
            "sample_short": "first %d reads of rank 0's R1 stream; C port (AVX + OpenMP) of the padded 100-step BiLSTM, batch 1024/thread, %d threads, %.1f s" % (n, cores, dt),
            "sample": "first %d reads of the rank-0 R1 stream (100 bp); oracle rdo_forward_padded_batched = the ribodetector_cpu "
                      "algorithm (padded BiLSTM over all 100 steps x 2 directions, batch 1024, one batch per thread) as a hand-vectorised "
                      "AVX-512/AVX2 + OpenMP C port, %d threads = usable cores of this container (os.cpu_count() = %d), %.1f s. "
                      "`value` is MODEL ONLY (ASCII bytes -> logits, base lookup fused); encode_plus_model adds materialising the "
                      "[1024,100,4] fp32 one-hot input in C (the reference's Python encoder is ~100x slower than that and would "
                      "dominate). onnxruntime is not installed, so this port stands in for ribodetector_cpu"
                      % (n, cores, os.cpu_count() or 0, dt)}


class GpuSampler:
    """Shader clock and package power of one GPU while the timed region runs, from the amdgpu hwmon files of its PCI function (freq1_input
    in Hz, power1_average / power1_input in microwatts; a read costs microseconds, no tool is started): what tells a 34 M box from a 36 M
    box - the recurrence kernel is bound by the package power cap, so its rate follows the clock the firmware grants."""

    def __init__(self, torch, dev, period=0.05):
        import glob
        import threading
        self.files, self.samples, self.period = {}, {"sclk_mhz": [], "power_w": []}, period
        try:
            p = torch.cuda.get_device_properties(dev)
            base = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
            for hw in sorted(glob.glob(base + "/hwmon/hwmon*")):
                for key, names in (("sclk_mhz", ("freq1_input",)), ("power_w", ("power1_average", "power1_input"))):
                    for nm in names:
                        if key not in self.files and os.path.exists(os.path.join(hw, nm)):
                            self.files[key] = os.path.join(hw, nm)
        except Exception:      # noqa: BLE001 - (no such files: the record says null)
            pass
        self._stop = threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True) if self.files else None

    def _run(self):
        while not self._stop.is_set():
            for key, path in self.files.items():
                try:
                    with open(path) as fh:
                        self.samples[key].append(float(fh.read().strip()) / 1e6)      # Hz -> MHz, microwatts -> W
                except (OSError, ValueError):
                    pass
            self._stop.wait(self.period)

    def start(self):
        if self._th:
            self._th.start()
        return self

    def stop(self):
        self._stop.set()
        if self._th:
            self._th.join()
        out = {}
        for key, v in self.samples.items():
            out[key] = {"avg": round(sum(v) / len(v), 1), "min": round(min(v), 1), "max": round(max(v), 1), "samples": len(v)} if v else None
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


def e2e_legs(args, P, RL, MAXLEN, paired):
    """timed region (iii): the whole CLI on FASTQ files built from the rank-0 stream of this run (same seeds, same slices), in a CHILD
    process (tools/e2e_bench.py --bench-legs): a CLI invocation is a process of its own, and inside this one the runtime's helper
    thread - after the timed region, the alt kernels and the probes have run - spins a full core through every call (BGZF -> gz read
    1.3 host cores here, 0.6 in a process of its own).
    one_step_batch: the batch of one step (the pipeline-fill-bound point), one call after a warm one. Then every flow on files of
    --e2e-records records (default 16 Mi: >= 1 s of CLI per call), median of three calls after a warm one: plain -> plain, plain -> gz,
    BGZF -> gz / plain (text stays on the device), the same two through the host parser, gz -> gz (one member per file: the stream decoded on the GPU; and by
    the host's decoders with -t 10 / -t = cores: RD_DEVICE_INFLATE=members)."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "e2e_bench.py"), "--bench-legs", "--pairs-per-step", str(P), "--records", str(args.e2e_records),
           "--read-len", str(RL), "--max-len", str(MAXLEN), "--ensure", args.ensure] + ([] if paired else ["--single-end"]) + (
               ["--var-len"] if args.workload == "var300" else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RD_FORCE_DIST", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("tools/e2e_bench.py --bench-legs failed (rc %d): %s" % (r.returncode, r.stderr.decode(errors="replace")[-400:]))
    return json.loads(lines[-1])


def compact(full):
    """the ONE stdout line: the driver keeps an 8,000-byte tail of stdout, so the line stays under 4,000 bytes (asserted in
    tests/test_gpu_bench.py): contract keys, a dozen config scalars, the roofline's numbers, cpu_baseline, parity_sample and one
    scalar per alt / e2e leg. Everything else is in bench_full.json (--full-out)."""
    r3 = lambda x: (float("%.5g" % x) if isinstance(x, float) else x)      # noqa: E731
    c, rf = full["config"], full["roofline"]
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    out["config"] = {k: r3(c[k]) for k in ("workload", "timed_region", "per_step_per_gpu", "read_len", "ensure", "kernel_variant", "parallelism",
                                           "kernel_only_reads_per_s", "host_cores_busy", "host_cores_usable", "dist_backend", "rccl_ranks",
                                           "gather_self_check", "forced_dist", "gpu_over_cpu", "host_labels_nonzero_last_step") if k in c}
    out["config"]["prefix_k"] = c["prefix_table"]["k"]
    out["config"]["refine_band"] = c["refine"]["band"]
    out["config"]["label_counts"] = [c["label_counts"][k] for k in ("non_rrna", "rrna", "unclassified")]
    out["roofline"] = {k: r3(rf[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "launches", "avg_launch_ms", "traffic",
                                              "traffic_over_algorithmic", "steps_executed_over_steps", "frac_counting_table_steps",
                                              "mfma_pipe_frac", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch", "effective_clock_ghz",
                                              "package_power_w") if k in rf}
    if isinstance(rf.get("traffic_source"), str):
        out["roofline"]["traffic_source"] = rf["traffic_source"][:120]
    for name, key in (("alt_fp32", "alt_fp32_kernel"), ("alt_no_prefix_table", "alt_no_prefix_table")):
        if key in full:
            out[name + "_reads_per_s"] = r3(full[key]["value"])
            out[name + "_frac"] = r3(full[key]["roofline"]["frac"])
    e = full.get("e2e_cli") or {}
    if "error" in e:
        out["e2e_error"] = e["error"][:200]
    for k, v in e.items():          # (the *_host_parse legs - the round-4 route - stay in the full record: the line must stay under 4,000 bytes)
        if isinstance(v, dict) and "reads_per_s" in v and k != "one_step_batch" and not k.endswith("_host_parse"):
            out["e2e_" + k] = {"rps": r3(v["reads_per_s"]), "steady_rps": r3(v.get("reads_per_s_after_first_chunk")),
                               "host_cores_busy": v["host_cores_busy"], "spread": r3(v["spread"])}
    if "plain_to_plain" in e:
        out["e2e_what"] = "whole CLI call, median of %d; steady = after first chunk; %s records/file%s" % (
            e["plain_to_plain"].get("timed_calls", 0), e["plain_to_plain"].get("records_per_file"),
            "; seqlike = Illumina-like text, zlib-6 inputs" if any(k.startswith("seqlike_") for k in e) else "")
    if "encoder" in full and "kernels" in full["encoder"]:
        out["encoder_GBps"] = {k.replace("rd_", "").replace("_kernel", ""): r3(v["achieved"]) for k, v in full["encoder"]["kernels"].items()}
    g = full.get("device_gzip") or {}
    if "GB_per_s_of_text" in g:
        out["device_gzip"] = {"deflate_ms_per_chunk": r3(g["ms_per_chunk_both_label_files"]), "deflate_GBps_text": r3(g["GB_per_s_of_text"]),
                              "size_vs_zlib5": r3(g["size_vs_zlib_level_5"]),
                              "inflate_GBps_text": r3((g.get("device_gunzip") or {}).get("GB_per_s_of_text"))}
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = ({k: r3(cb[k]) for k in ("value", "unit", "cores", "kind", "encode_plus_model_reads_per_s") if k in cb}
                               if "error" not in cb else {"error": cb["error"][:200]})
        if "sample" in cb:
            out["cpu_baseline"]["sample"] = cb["sample_short"]
    if "parity_sample" in full:
        out["parity_sample"] = {k: r3(v) for k, v in full["parity_sample"].items() if k != "vs"}
    out["full_record"] = os.path.basename(full.get("_full_out", "bench_full.json"))
    return out


def emit(full, args):
    """bench_full.json (everything) + the compact line as the LAST line of stdout"""
    path = args.full_out or os.path.join(ROOT, "bench_full.json")
    full["_full_out"] = path
    try:
        with open(path, "w") as fh:
            json.dump(full, fh, indent=1)
    except OSError as e:
        sys.stderr.write("bench.py: cannot write %s: %s\n" % (path, e))
    line = json.dumps(compact(full), separators=(",", ":"))
    if args.verbose:
        sys.stderr.write(json.dumps(full) + "\n")
    sys.stderr.flush()
    sys.stdout.write(line + "\n")
    sys.stdout.flush()


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
    ap.add_argument("--full-out", default=None, help="where the full record goes (default: bench_full.json next to this script)")
    ap.add_argument("--verbose", action="store_true", help="also print the full record on stderr")
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
    pinned = None
    if world > 1:                                          # a rank's threads on its share of the CPUs next to its GPU (RD_PIN=0: off)
        pinned, how = rdist.pin_rank_cpus(dev.index, local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        pinned = {"cpus": pinned, "how": how}
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
    sampler = GpuSampler(torch, dev).start()
    t0, c0 = time.perf_counter(), time.process_time()
    timed_path(args.steps)
    sync()
    dt, cpu_s = time.perf_counter() - t0, time.process_time() - c0      # cpu_s: user + system time of ALL threads of this rank
    gpu_state = sampler.stop()
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
    cpu_ranks, gpu_states = [cpu_s], [gpu_state]
    if multi:
        both = [None] * world
        dist.all_gather_object(both, (cpu_s, gpu_state, pinned))
        cpu_ranks, gpu_states = [b[0] for b in both], [b[1] for b in both]
        if rank_info:
            for i, b in zip(rank_info, both):
                i["gpu_state_in_timed_region"], i["cpus"] = b[1], b[2]
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
                  "pinned host bytes -> H2D -> kernels -> labels D2H (SURVEY 8d region ii)")
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
                         "effective_clock_ghz": round(gpu_state["sclk_mhz"]["avg"] / 1e3, 3) if gpu_state.get("sclk_mhz") else None,
                         "package_power_w": gpu_state["power_w"]["avg"] if gpu_state.get("power_w") else None,
                         "gpu_state_in_timed_region": gpu_states if world > 1 else gpu_state,
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
                torch.cuda.empty_cache()
                out["e2e_cli"] = e2e_legs(args, P, RL, MAXLEN, paired)
                for k, v in out["e2e_cli"].items():
                    if isinstance(v, dict) and "reads_per_s" in v:
                        out["config"]["e2e_cli_%s_reads_per_s" % k] = v["reads_per_s"]
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
        emit(out, args)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
