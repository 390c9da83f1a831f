#!/usr/bin/env python
"""The CLI's flows at N ranks, one GPU per rank (tools/scale_sweep.sh calls this after the bench.py points of the same N): sequencer-like
mate files built once in tmpfs - plain, BGZF by zlib level 6 (what bgzip writes), ONE gzip member by zlib level 6 - then, per flow, the
whole `ribodetector` run under torch.distributed.run: reads/s of detect() (max over ranks, model load excluded), wall time of the whole
process group, host cores busy, and the outputs' SHA-1 against the N = 1 run's.
    python tools/scale_cli.py --make DIR [--records 8388608]
    python tools/scale_cli.py --run DIR --gpus N [--flows bgzf_to_gz,plain_to_plain,gz_to_gz] [--share-gpu]
--share-gpu: all ranks on GPU 0 over gloo (a functional run of the sweep on a 1-GPU box)."""
import argparse
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

FLOWS = {"plain_to_plain": ("plain", False), "plain_to_gz": ("plain", True), "bgzf_to_gz": ("bgzf", True), "gz_to_gz": ("gz", True), "gz_to_plain": ("gz", False)}


def make(d, records):
    import torch
    from ribodetector_amd import synth
    os.makedirs(d, exist_ok=True)
    dev = torch.device("cuda", 0)
    meta = {"records": records, "files": {"plain": [], "bgzf": [], "gz": []}}
    for mate, seed in ((1, 2000), (2, 7000)):
        arena, off, lens = synth.reads_torch(records, 100, seed=seed, device=dev)
        p = os.path.join(d, "r_%d.fq" % mate)
        synth.fastq_image_torch(arena, off, lens, mate=mate, style="seqlike", seed=mate).cpu().numpy().tofile(p)
        del arena, off, lens
        torch.cuda.empty_cache()
        synth.bgzip_file(p, p[:-3] + ".bgzf.fq.gz", level=6)
        synth.pgzip_file(p, p[:-3] + ".one.fq.gz", level=6)
        meta["files"]["plain"].append(p)
        meta["files"]["bgzf"].append(p[:-3] + ".bgzf.fq.gz")
        meta["files"]["gz"].append(p[:-3] + ".one.fq.gz")
    meta["bytes"] = {k: [os.path.getsize(f) for f in v] for k, v in meta["files"].items()}
    json.dump(meta, open(os.path.join(d, "meta.json"), "w"))
    print(json.dumps(meta))


def run(d, gpus, flows, share_gpu):
    from host_scaling import run_cli
    meta = json.load(open(os.path.join(d, "meta.json")))
    out = {"n_gpus": gpus, "records_per_file": meta["records"], "share_gpu": share_gpu, "flows": {}}
    import torch
    if not share_gpu and torch.cuda.device_count() < gpus:
        out["skipped"] = "%d GPUs asked for, %d visible" % (gpus, torch.cuda.device_count())
        print(json.dumps(out))
        return
    for flow in flows:
        kind, gz_out = FLOWS[flow]
        r = run_cli(gpus, meta["files"][kind], d, flow, threads=10 if gpus == 1 else 2, gz_out=gz_out, share_gpu=share_gpu)
        if "error" in r:
            out["flows"][flow] = {"error": r["error"][-400:], "error_lines": r.get("error_lines")}
            continue
        out["flows"][flow] = {"cli_reads_per_s": r["reads_per_s_detect"], "detect_s": r["detect_s_max_over_ranks"], "wall_s_whole_job": r["wall_s_whole_process"],
                              "cores_busy": r["cores_busy"], "ingest_modes": r["ingest_modes"], "gz_ranges_s": r["gz_ranges_s"], "output_sha1": r["output_sha1"]}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--make", default=None)
    ap.add_argument("--run", default=None)
    ap.add_argument("--records", type=int, default=1 << 23)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--flows", default="bgzf_to_gz,plain_to_plain,gz_to_gz")
    ap.add_argument("--share-gpu", action="store_true")
    a = ap.parse_args()
    if a.make:
        return make(a.make, a.records)
    run(a.run, a.gpus, [f for f in a.flows.split(",") if f], a.share_gpu)


if __name__ == "__main__":
    main()
