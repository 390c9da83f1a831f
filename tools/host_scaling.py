#!/usr/bin/env python
"""Do W ranks fit the host of ONE box? (VERDICT r2 next #1c)

The 8-GPU node of the scaling run gives every rank its own GPU but the same cgroup of host cores as the 1-GPU boxes have
(bench.py usable_cores(): 16). This tool runs what can be run on a 1-GPU box - W ranks SHARING the one GPU, labels exchanged
over gloo - and records the host side of it: CPU seconds per rank inside the timed region, cores busy, and the ratio device
path / kernel-only, for bench.py at W = 1, 2, 4, 8 and for the CLI (plain and gz input) at W = 1 and 8. The GPU is shared, so
reads/s do NOT scale here; what the numbers show is that W x (launch thread + reader + writers) stay below the core budget.
Writes gpurun_out/r06_host_scaling.json.      python tools/host_scaling.py [--reads 4000000]"""
import argparse
import json
import os
import resource
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def child_cpu():
    r = resource.getrusage(resource.RUSAGE_CHILDREN)
    return r.ru_utime + r.ru_stime


def run_bench(world, pairs):
    env = dict(os.environ, RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0", RD_PREFIX_K="12")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    full = os.path.join(tempfile.gettempdir(), "rd_hs_full_%d_%d.json" % (os.getpid(), world))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2", "--pairs-per-step", str(pairs),
           "--no-alt", "--no-cpu-baseline", "--no-encoder", "--no-e2e", "--traffic", "off", "--full-out", full]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    if r.returncode != 0 or not os.path.exists(full):
        return {"error": (r.stdout + r.stderr)[-1500:]}
    j = json.load(open(full))       # (the full record: the stdout line is the compact one)
    os.remove(full)
    c = j["config"]
    return {"ranks": world, "pairs_per_step_per_rank": pairs, "reads_per_s_all_ranks_one_gpu": j["value"], "ms_per_step": j["ms_per_step"],
            "device_path_over_kernel_only": c["device_path_over_kernel_only"], "cpu_seconds_per_rank": c["host_cpu_seconds_per_rank_in_timed_region"],
            "host_cores_busy": c["host_cores_busy"], "host_cores_usable": c["host_cores_usable"], "dist_backend": c["dist_backend"]}


def run_cli(world, inputs, outdir, tag, threads, shared_decode="1", gz_out=False, extra_env=None, share_gpu=True):
    """share_gpu: the ranks share GPU 0 and exchange over gloo (a 1-GPU box); False: one GPU per rank over RCCL (tools/scale_cli.py)"""
    outs = [os.path.join(outdir, "%s_w%d_%d.fq%s" % (tag, world, i, ".gz" if gz_out else "")) for i in range(len(inputs))]
    base = ["-l", "100", "-i", *inputs, "-o", *outs, "-t", str(threads)] + (["-e", "rrna"] if len(inputs) == 2 else [])
    tfile = os.path.join(outdir, "timing_%s_w%d" % (tag, world))
    env = dict(os.environ, RD_PREFIX_K=os.environ.get("RD_PREFIX_K", "12"), RD_SHARED_DECODE=shared_decode, RD_TIMING_OUT=tfile, **(extra_env or {}))
    if world == 1:
        cmd = [sys.executable, "-m", "ribodetector_amd.detect"] + base
    else:
        if share_gpu:
            env.update(RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), "-m", "ribodetector_amd.detect"] + base
    c0, t0 = child_cpu(), time.perf_counter()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    dt, cpu = time.perf_counter() - t0, child_cpu() - c0
    if r.returncode != 0:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "hs_err_%s_w%d.log" % (tag, world)), "a") as fh:
            fh.write((r.stdout + r.stderr)[-60000:])
        import re
        tb = re.findall(r"(?:Error|error|Exception)[^\n]*", r.stdout + r.stderr)
        return {"error": (r.stdout + r.stderr)[-600:], "error_lines": tb[:12], "env": extra_env or {}}
    import gzip
    import hashlib
    sha = [hashlib.sha1((gzip.open if gz_out else open)(o, "rb").read()).hexdigest() for o in outs]      # (of the text)
    for o in outs:
        os.remove(o)
    # what the ranks measured themselves (RD_TIMING_OUT): the run without interpreter start-up and model load
    import glob
    tj = [json.load(open(f)) for f in sorted(glob.glob(tfile + ".rank*"))]
    for f in glob.glob(tfile + ".rank*"):
        os.remove(f)
    det = max((t["detect_s"] for t in tj), default=None)
    nread = max((t["num_read"] for t in tj), default=0) * len(inputs)
    return {"ranks": world, "threads_flag": threads, "wall_s_whole_process": dt, "cpu_s_all_ranks": cpu, "cores_busy": cpu / dt, "output_sha1": sha,
            "detect_s_max_over_ranks": det, "reads_per_s_detect": nread / det if det else None,
            "gz_ranges_s": [t.get("gz_ranges_s") for t in tj], "detect_s_by_rank": [round(t["detect_s"], 3) for t in tj],
            "slowest_rank": max(tj, key=lambda t: t["detect_s"], default=None), "ingest_modes": sorted({str((v.get("feeder") or {}).get("mode") or v.get("path")) for t in tj for v in (t.get("ingest") or {}).values()}),
            "env": extra_env or {}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4000000)
    ap.add_argument("--legs", default="", help="comma-separated CLI legs (default: all)")
    ap.add_argument("--no-bench", action="store_true")
    ap.add_argument("--gz-repeat", type=int, default=1, help="the single-stream .gz inputs hold the plain files' text this many times")
    a = ap.parse_args()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from e2e_bench import usable_cores
    out = {"host_cores_usable": usable_cores(), "os_cpu_count": os.cpu_count(), "note": "all ranks share ONE GPU (RD_LOCAL_DEVICE=0, gloo): host-side evidence only"}
    out["bench"] = [] if a.no_bench else [run_bench(w, 1 << 18) for w in (1, 2, 4, 8)]
    import gzip
    import shutil
    from ribodetector_amd import synth
    d = tempfile.mkdtemp(prefix="rd_hs_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        files = []
        for mate, seed in ((1, 1), (2, 2)):
            arena, off, _ = synth.reads_numpy(a.reads, 100, seed=seed)
            p = os.path.join(d, "r_%d.fq" % mate)
            synth.write_fastq_realistic(p, arena, off, mate, seed=seed)
            synth.pgzip_file(p, p + ".gz", level=6, repeat=a.gz_repeat)      # ONE gzip member (deflated in parallel, pigz-style)
            files.append(p)
        # the same files as BGZF (framed by the device writer): under several ranks every rank inflates the members of its own share
        import numpy as np
        import torch
        from ribodetector_amd.gz import DeviceGzip, eof_block
        dgz = DeviceGzip("cuda:0")
        for p in files:
            t = torch.from_numpy(np.fromfile(p, dtype=np.uint8)).cuda()
            nl = torch.nonzero(t == 10).flatten()
            rs = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), nl[3::4] + 1])
            o, info = dgz.compress_selected(t, rs, torch.zeros(rs.numel() - 1, dtype=torch.int8, device="cuda"), 0)
            torch.cuda.synchronize()
            os.makedirs(os.path.join(d, "bgzf"), exist_ok=True)
            with open(os.path.join(d, "bgzf", os.path.basename(p) + ".gz"), "wb") as fh:
                fh.write(o[: int(info[0])].cpu().numpy().tobytes())
                fh.write(eof_block())
            del t, nl, rs, o
        del dgz
        torch.cuda.empty_cache()
        out["cli_reads_per_file"] = a.reads
        out["gz_reads_per_file"] = a.reads * a.gz_repeat
        out["cli"] = {}
        bg = [os.path.join(d, "bgzf", os.path.basename(f) + ".gz") for f in files]
        legs = (("pe_plain", files), ("pe_gz", [f + ".gz" for f in files]), ("pe_gz_to_gz", [f + ".gz" for f in files]), ("pe_bgzf", bg), ("pe_bgzf_to_gz", bg),
                ("pe_plain_to_gz", files))
        for tag, ins in legs:
            if a.legs and tag not in a.legs.split(","):
                continue
            rows = []
            for w in (1, 8):
                rows.append(run_cli(w, ins, d, tag, threads=10 if w == 1 else 2, gz_out=tag.endswith("_to_gz")))
            if tag.startswith("pe_gz"):
                # round 6: the default at W = 8 above = every rank decodes its own range of the stream on the GPU. For comparison: one decode
                # per node into /dev/shm (rounds 3-5: RD_GZ_SHARD=0), and - pe_gz only - every rank decodes the whole stream itself (round 2)
                rows.append(dict(run_cli(8, ins, d, tag, threads=2, gz_out=tag.endswith("_to_gz"), extra_env={"RD_GZ_SHARD": "0"}), one_decode_per_node=True))
                if tag == "pe_gz":
                    rows.append(dict(run_cli(8, ins, d, tag, threads=2, shared_decode="0", extra_env={"RD_GZ_SHARD": "0"}), every_rank_decodes=True))
            same = all("output_sha1" in r for r in rows) and all(r["output_sha1"] == rows[0]["output_sha1"] for r in rows)
            out["cli"][tag] = {"runs": rows, "outputs_identical_w1_w8": same}
        g, p = out["cli"].get("pe_gz", {}).get("runs", []), out["cli"].get("pe_plain", {}).get("runs", [])
        if len(g) >= 4 and len(p) >= 2 and all("cpu_s_all_ranks" in r for r in g + p):
            # what reading .gz costs on top of reading plain text, in CPU seconds: once at W = 1; at W = 8 the same amount if the
            # stream is decoded once per node, eight times that if every rank decodes it (each process also pays ~2.5 s of
            # interpreter + torch start-up, which is why the totals are compared as differences)
            out["gz_over_plain_cpu_s"] = {"w1": g[0]["cpu_s_all_ranks"] - p[0]["cpu_s_all_ranks"],
                                          "w8_ranges_on_the_gpus": g[1]["cpu_s_all_ranks"] - p[1]["cpu_s_all_ranks"],
                                          "w8_one_decode_per_node": g[2]["cpu_s_all_ranks"] - p[1]["cpu_s_all_ranks"],
                                          "w8_every_rank_decodes": g[3]["cpu_s_all_ranks"] - p[1]["cpu_s_all_ranks"]}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_host_scaling.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
