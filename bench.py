#!/usr/bin/env python3
"""bench.py - reads/s classified, 100 bp paired-end (BASELINE.json metric) on N MI355X of one node.

A "step" = one pass of the hot path over one batch of synthetic read pairs that is already resident in HBM:
    rd_classify(R1) + rd_classify(R2)  (length bucketing, fused encoder, LSTM recurrence, FC, argmax)
    rd_pair_fuse(--ensure rrna) + counters, and for N>1 the RCCL gather of the 1-byte pair labels to rank 0.
Workload = BASELINE.json configs[2] ("10M paired-end 100 bp reads with --ensure rrna, 1 MI355X"): with the default
--steps 10 x 1,048,576 pairs/step = 10.5 M pairs (21 M reads) are classified inside the timed region.
For N>1 every rank gets its own shard of the same size (weak scaling), as the reads shard embarrassingly.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline     - the recurrence kernel against the fp32-MFMA peak: algorithmic FLOPs (T*131072+1024 per read)
                 / average launch duration measured with hipEvents on the launch stream (C ABI rd_profile_*)
  cpu_baseline - the CPU oracle's restatement of ribodetector_cpu (padded BiLSTM over all L steps, batch 1024,
                 one batch per thread) timed on this box's host cores on a bounded sample of the same reads.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READ_LEN = 100
FLOPS_PER_READ = READ_LEN * 131072 + 1024        # forward recurrence h.W_hh^T + FC
BYTES_PER_READ = READ_LEN + 4 + 8 + 8 + 1        # ASCII + len + offset + logits + label
PEAKS = {"mfma_f32": 157.3, "simple": 157.3, "mfma_f16x3": 2500.0, "mfma_f16x3_t32": 2500.0}   # dense TFLOP/s, MI355X_MICROARCH.md
MFMA_FLOPS_PER_ALGO_FLOP = {"mfma_f32": 1, "simple": 1, "mfma_f16x3": 3, "mfma_f16x3_t32": 3}   # f16x3 issues three f16 products per fp32 product


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


def cpu_baseline(arena_np, n_reads, target_s=15.0):
    """Time the oracle's batched ribodetector_cpu restatement on all usable host cores; bounded to ~target_s seconds."""
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
    return {"logits": cpu_logits, "n": n, "value": n / dt, "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "first %d reads of the rank-0 R1 stream (100 bp), oracle rdo_forward_padded_batched = ribodetector_cpu "
                      "algorithm (padded BiLSTM over all 100 steps x 2 directions, batch 1024, one batch per thread, "
                      "%d OpenMP threads = usable cores of this container; os.cpu_count() = %d), %.1f s; onnxruntime is not installed, so this C port "
                      "stands in for it" % (n, cores, os.cpu_count() or 0, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs-per-step", type=int, default=1 << 20)
    ap.add_argument("--variant", default="auto")
    ap.add_argument("--ensure", default="rrna")
    ap.add_argument("--workload", default="pe100", choices=["pe100", "se100", "pe150", "var300"],
                    help="pe100 = BASELINE configs[2] (the metric's configuration, default); se100 = configs[1]; pe150 = the per-GPU "
                         "shard of configs[3]; var300 = the per-GPU shard of configs[4] (40-300 bp, -l 300)")
    ap.add_argument("--pcie", action="store_true", help="also time the step with the read bytes starting in pinned host memory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the short extra measurement of the exact-fp32 MFMA kernel")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ribodetector_amd import dist as rdist
    from ribodetector_amd import synth
    from ribodetector_amd.model import model as module_arch
    from ribodetector_amd.parse_config import ConfigParser

    rank, world, local = rdist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dev = torch.device("cuda", int(os.environ.get("RD_LOCAL_DEVICE", local)))   # override: several ranks on one GPU (tests only)
    torch.cuda.set_device(dev)

    cfg = ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json"))
    model = cfg.init_obj("arch", module_arch)
    model.load_state_dict(cfg.load_state_dict("mcc"))
    model.to(dev).eval()
    model.set_variant(args.variant)
    variant = "mfma_f16x3_t32" if args.variant == "auto" else args.variant

    P = args.pairs_per_step
    WL = {"pe100": (True, 100, 100), "se100": (False, 100, 100), "pe150": (True, 150, 150), "var300": (False, 300, 300)}[args.workload]
    paired, RL, MAXLEN = WL
    nslices = max(1, min(args.steps, 10))                 # distinct batches resident in HBM, reused cyclically
    r1 = [synth.reads_torch(P, RL, seed=2000 + 100 * rank + i, device=dev) for i in range(nslices)]
    r2 = [synth.reads_torch(P, RL, seed=7000 + 100 * rank + i, device=dev) for i in range(nslices)] if paired else None
    offs = r1[0][1][:-1].contiguous()
    lens = r1[0][2]
    if args.workload == "var300":                          # lengths ~ U{40..300}; rows keep their 300-byte stride in the arena
        g = torch.Generator(device=dev)
        g.manual_seed(4 + rank)
        lens = torch.randint(40, 301, (P,), generator=g, device=dev, dtype=torch.int32)
    flops_per_launch = float((torch.clamp(lens, max=MAXLEN).to(torch.float64) * 131072 + 1024).sum().item())
    bytes_per_launch = float(lens.to(torch.float64).sum().item()) + P * (4 + 8 + 8 + 1)
    lg1 = torch.empty((P, 2), dtype=torch.float32, device=dev)
    lg2 = torch.empty((P, 2), dtype=torch.float32, device=dev)
    counts = torch.zeros(3, dtype=torch.int64, device=dev)
    gathered = torch.empty(P * world, dtype=torch.int8, device=dev) if (world > 1 and rank == 0) else None

    lab8 = torch.empty((P,), dtype=torch.uint8, device=dev)

    def step(i, a1=None, a2=None):
        a1 = r1[i % nslices][0] if a1 is None else a1
        if paired:
            a2 = r2[i % nslices][0] if a2 is None else a2
            model.classify_bytes(a1, offs, lens, MAXLEN, want_labels=False, logits=lg1)
            model.classify_bytes(a2, offs, lens, MAXLEN, want_labels=False, logits=lg2)
            lab = module_arch.pair_fuse(lg1, lg2, args.ensure, counts)
        else:
            model.classify_bytes(a1, offs, lens, MAXLEN, want_labels=True, logits=lg1, labels=lab8)
            module_arch.count_labels(lab8, counts)
            lab = lab8.view(torch.int8)
        if world > 1:
            _, fin = rdist.gather_labels(lab, P * world, dst=0, async_op=True, out=gathered)
            return fin
        return None

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        f = step(i)
        if f:
            f()
    counts.zero_()
    model.profile_enable(True)
    sync()
    t0 = time.perf_counter()
    pend = []
    for i in range(args.steps):
        f = step(i)
        if f:
            pend.append(f)
    for f in pend:
        f()
    sync()
    dt = time.perf_counter() - t0
    launches, kms = model.profile_read()
    model.profile_enable(False)
    rdist.reduce_counts(counts)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if world == 1 or dist.get_backend() == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    total_pairs = P * args.steps * world
    c = counts.cpu().tolist()
    assert c[0] + c[1] + c[2] == total_pairs, (c, total_pairs)

    if rank == 0:
        reads_per_launch = P
        avg_ms = kms / max(launches, 1)
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if launches else None
        base = ("mfma_f16x3_t32" if variant.startswith("mfma_f16x3_t32") else "mfma_f16x3" if variant.startswith("mfma_f16x3")
                else "mfma_f32" if variant.startswith("mfma_f32") else variant)
        peak = PEAKS[base]
        traffic = None
        try:   # HBM bytes per launch from the PMC passes of tools/profile_round.sh (bench.py cannot collect PMC itself)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["rd_lstm_%s_kernel" % base]
            traffic = tj["hbm_bytes_per_launch"] * bytes_per_launch / (tj["reads_per_launch"] * (tj["read_len"] + 21))
        except Exception:
            pass
        out = {
            "metric": "reads/sec classified, 100 bp paired-end" if args.workload == "pe100" else "reads/sec classified, " + args.workload,
            "value": (2.0 if paired else 1.0) * total_pairs / dt,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if not variant.startswith("mfma_f16x3") else "f16x3-split (f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": {"pe100": "BASELINE configs[2]: paired-end 100 bp, --ensure %s" % args.ensure,
                                    "se100": "BASELINE configs[1]: single-end 100 bp",
                                    "pe150": "BASELINE configs[3] per-GPU shard: paired-end 150 bp, -l 150, --ensure %s" % args.ensure,
                                    "var300": "BASELINE configs[4] per-GPU shard: single-end 40-300 bp, -l 300, length-bucketed"}[args.workload]
                                   + ", %d %s/step/GPU x %d steps (%.1f M total), inputs resident in HBM"
                                   % (P, "pairs" if paired else "reads", args.steps, total_pairs / 1e6),
                       "pairs_per_s": (total_pairs / dt) if paired else None, "per_step_per_gpu": P, "read_len": RL if args.workload != "var300" else "40-300",
                       "ensure": args.ensure if paired else None,
                       "kernel_variant": variant,
                       "precision": ("every fp32 product h*w is formed as three f16 MFMA products of hi/lo parts with fp32 accumulation; "
                                     "logit error against a float64 evaluation equals the fp32 reference's own (DESIGN.md 4, "
                                     "tests/test_gpu_parity.py::test_error_against_float64_truth); parity_sample below is this run's check"
                                     if variant.startswith("mfma_f16x3") else "fp32"), "parallelism": "reads sharded x%d, RCCL label gather" % world,
                       "label_counts": {"non_rrna": c[0], "rrna": c[1], "unclassified": c[2]}},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": (achieved / peak) if achieved else None, "traffic": traffic,
                         "kernel": "rd_lstm_%s_kernel" % base, "launches": launches, "avg_launch_ms": avg_ms,
                         "algorithmic_flops_per_launch": flops_per_launch,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "mfma_flops_executed_per_algorithmic_flop": MFMA_FLOPS_PER_ALGO_FLOP[base],
                         "mfma_pipe_frac": (achieved * MFMA_FLOPS_PER_ALGO_FLOP[base] / peak) if achieved else None},
        }
        if world == 1 and not args.no_alt and base != "mfma_f32":
            # the same step on the exact-fp32 MFMA kernel (v_mfma_f32_16x16x4_f32), for the fp32-MFMA roofline of SURVEY §8d
            model.set_variant("mfma_f32")
            model.profile_enable(True)
            sync()
            t1 = time.perf_counter()
            for i in range(2):
                step(i)
            sync()
            d1 = time.perf_counter() - t1
            l2, k2 = model.profile_read()
            model.profile_enable(False)
            model.set_variant(args.variant)
            a2 = flops_per_launch / (k2 / max(l2, 1) * 1e-3) / 1e12
            out["alt_fp32_kernel"] = {"kernel": "rd_lstm_mfma_f32_kernel", "value": (2.0 if paired else 1.0) * P * 2 / d1, "unit": "reads/s", "steps": 2,
                                      "roofline": {"bound": "mfma", "achieved": a2, "peak": PEAKS["mfma_f32"], "unit": "TFLOP/s",
                                                   "frac": a2 / PEAKS["mfma_f32"], "avg_launch_ms": k2 / max(l2, 1)}}
        if args.pcie and world == 1:
            # same step, but every batch's bytes start in pinned host memory: H2D on a copy stream, double buffered
            hosts = [[t[0].cpu().pin_memory() for t in (r1[:2] if not paired else r1[:2] + r2[:2])]]
            h = hosts[0]
            cs = torch.cuda.Stream(dev)
            bufs = [[torch.empty_like(r1[0][0]) for _ in range(2 if paired else 1)] for _ in range(2)]
            evs = [torch.cuda.Event() for _ in range(2)]
            def h2d(i):
                with torch.cuda.stream(cs):
                    bufs[i & 1][0].copy_(h[i & 1], non_blocking=True)
                    if paired:
                        bufs[i & 1][1].copy_(h[2 + (i & 1)], non_blocking=True)
                    evs[i & 1].record(cs)
            counts.zero_()
            sync()
            t2 = time.perf_counter()
            h2d(0)
            for i in range(args.steps):
                if i + 1 < args.steps:
                    h2d(i + 1)
                torch.cuda.current_stream(dev).wait_event(evs[i & 1])
                step(i, bufs[i & 1][0], bufs[i & 1][1] if paired else None)
                cs.wait_stream(torch.cuda.current_stream(dev))    # buffer (i&1) is reused by h2d(i+2)
            lab_host = torch.empty((P,), dtype=torch.int8).pin_memory()
            sync()
            d2 = time.perf_counter() - t2
            out["config"]["pcie_inclusive_reads_per_s"] = (2.0 if paired else 1.0) * P * args.steps / d2
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
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
