#!/usr/bin/env python
"""Where does a chunk's GPU time go in the whole CLI? One detect.main() call under `rocprofv3 --kernel-trace` on FASTQ files in tmpfs
(paired-end 100 bp, 2^20 pairs per chunk), then the trace of the steady-state chunks is accounted for: per chunk the time the
recurrence kernels run, the time other kernels run while NO recurrence kernel does (the recurrence wave holds all 512 registers of its
SIMD and 126 KB of LDS: nothing shares a CU with it), the time no kernel runs at all, and every other kernel's own
duration by name.       python tools/cli_gpu_account.py --in plain|bgzf|gz --out plain|gz [--pairs N] [--json out.json]"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def union(iv):
    iv = sorted(iv)
    out, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                out += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return out + (ce - cs if cs is not None else 0)


def subtract(iv, holes):
    """total length of the intervals `iv` outside the (sorted, disjoint) intervals `holes`"""
    tot = 0
    for s, e in iv:
        cur = s
        for hs, he in holes:
            if he <= cur:
                continue
            if hs >= e:
                break
            if hs > cur:
                tot += hs - cur
            cur = max(cur, he)
            if cur >= e:
                break
        if cur < e:
            tot += e - cur
    return tot


def merge(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def account(trace_dir, pairs_per_chunk=1 << 20):
    rows = []
    for p in glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]))
    rows.sort()
    lstm = [r for r in rows if "rd_lstm_mfma" in r[2] and (r[1] - r[0]) > 5e6]        # (the table-building launches are short)
    if len(lstm) < 8:
        return {"error": "too few recurrence launches: %d" % len(lstm)}
    a, b = len(lstm) // 4 // 2 * 2, len(lstm) * 7 // 8 // 2 * 2                       # whole chunks (two launches each), ends trimmed
    t0, t1 = lstm[a][0], lstm[b][0]
    chunks = (b - a) / 2
    W = t1 - t0
    clip = lambda r: (max(r[0], t0), min(r[1], t1))
    inw = [r for r in rows if r[1] > t0 and r[0] < t1]
    liv = merge([clip(r) for r in inw if "rd_lstm_mfma" in r[2]])
    oiv = [clip(r) for r in inw if "rd_lstm_mfma" not in r[2]]
    busy_l = sum(e - s for s, e in liv)
    other_alone = subtract(merge(oiv), liv)
    any_busy = union([tuple(x) for x in liv] + oiv)
    by = {}
    for r in inw:
        if "rd_lstm_mfma" in r[2]:
            continue
        s, e = clip(r)
        d = by.setdefault(r[2], [0, 0, 0])
        d[0] += 1
        d[1] += e - s
        d[2] += subtract([(s, e)], liv)
    ms = lambda x: round(x / chunks / 1e6, 3)
    return {"chunks": chunks, "pairs_per_chunk": pairs_per_chunk, "ms_per_chunk": ms(W), "reads_per_s_steady": round(2 * pairs_per_chunk * chunks / (W / 1e9)),
            "recurrence_ms": ms(busy_l), "other_kernels_with_no_recurrence_running_ms": ms(other_alone), "no_kernel_running_ms": ms(W - any_busy),
            "other_kernels": {k: {"launches_per_chunk": round(v[0] / chunks, 2), "ms": ms(v[1]), "ms_outside_recurrence": ms(v[2])}
                              for k, v in sorted(by.items(), key=lambda kv: -kv[1][1]) if v[1] / chunks > 2e4}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--in", dest="in_kind", default="plain", choices=["plain", "bgzf", "gz"])
    ap.add_argument("--out", dest="out_kind", default="plain", choices=["plain", "gz"])
    ap.add_argument("--pairs", type=int, default=10 << 20)
    ap.add_argument("--json", default=None)
    ap.add_argument("--env", action="append", default=[])
    a = ap.parse_args()
    import torch
    from ribodetector_amd import synth
    from e2e_bench import E2E
    dev = torch.device("cuda", 0)
    arenas = []
    for seed in (2000, 7000):
        ar, off, lens = synth.reads_torch(a.pairs, 100, seed=seed, device=dev)
        arenas.append(ar)
    with E2E(torch, synth, arenas, off, lens, 100, "rrna") as e:
        ins = e.inputs(a.in_kind)
        ext = ".fq.gz" if a.out_kind == "gz" else ".fq"
        outs = [os.path.join(e.dir, "o%d%s" % (k, ext)) for k in (1, 2)]
        del arenas, ar
        torch.cuda.empty_cache()
        tr = os.path.join(e.dir, "tr")
        env = dict(os.environ, PYTHONPATH=ROOT, TMPDIR="/tmp")
        for kv in a.env:
            k, v = kv.split("=", 1)
            env[k] = v
        cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", tr, "-o", "x", "--", sys.executable, "-m", "ribodetector_amd.detect",
               "-l", "100", "-i", *ins, "-o", *outs, "-e", "rrna"]
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        if r.returncode:
            print(r.stderr.decode(errors="replace")[-1500:], file=sys.stderr)
            return 1
        rec = {"flow": "%s -> %s" % (a.in_kind, a.out_kind), "pairs": a.pairs, "env": a.env, "under": "rocprofv3 --kernel-trace"}
        rec.update(account(tr))
    print(json.dumps(rec, indent=1))
    if a.json:
        json.dump(rec, open(a.json, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
