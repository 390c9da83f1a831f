"""bench.py contract on the GPU: one JSON line with every field the driver reads, at N=1 and under torchrun x2.
The x2 case shares this box's single GPU (RD_LOCAL_DEVICE=0) and exchanges labels over gloo - it checks the multi-rank
code path (shard seeds, async label gather, counter all-reduce, max-over-ranks timing), not the speed."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline")         # cpu_baseline is skipped here (--no-cpu-baseline): it takes ~20 s


def _line(out, full_path=None):
    """the compact line: the ONLY JSON line, the LAST line of stdout, under 4,000 bytes (the driver keeps an 8,000-byte tail of
    stdout: round 4's 22 KB line was cut and the round went unmeasured). With full_path: (line, the full record written there)."""
    all_lines = out.splitlines()
    lines = [l for l in all_lines if l.startswith("{")]
    assert len(lines) == 1 and all_lines[-1] == lines[0], out[-2000:]
    assert len(lines[0]) < 4000, len(lines[0])
    j = json.loads(lines[0])
    if full_path is None:
        return j
    return j, json.load(open(full_path))


def test_bench_single_gpu_contract(tmp_path):
    fp = str(tmp_path / "full.json")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--pairs-per-step", "65536", "--no-alt",
                        "--no-cpu-baseline", "--traffic", "off", "--e2e-records", "0", "--full-out", fp], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stderr) < 6000, r.stderr[-3000:]                     # the e2e legs log at WARNING: no per-chunk lines
    line, j = _line(r.stdout, fp)
    for k in KEYS:
        assert k in j and k in line, k
    # the compact line carries the same numbers as the full record
    assert line["value"] == j["value"] and line["ms_per_step"] == j["ms_per_step"] and line["config"]["timed_region"] == "ii"
    assert abs(line["roofline"]["frac"] - j["roofline"]["frac"]) < 1e-4 and line["roofline"]["kernel"] == j["roofline"]["kernel"]
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_launch_ms", "steps_executed_over_steps"}
    assert line["e2e_plain_to_plain"]["rps"] > 1e5 and "workload" in line["config"] and "model" not in line["config"]
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["unit"] == "reads/s" and j["scaling"] == "weak"
    assert j["metric"] == "reads/sec classified, 100 bp paired-end" and j["higher_is_better"] is True
    assert j["value"] > 1e6 and abs(j["value"] - 2 * 65536 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]
    rf = j["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert "workload" in j["config"] and "model" not in j["config"]
    # the timed region is SURVEY 8d (ii): pinned host bytes -> H2D -> kernels -> labels D2H; (i) rides along
    assert j["config"]["timed_region"] == "ii" and j["config"]["kernel_only_reads_per_s"] > 1e6
    assert j["config"]["host_labels_nonzero_last_step"] > 0           # the labels really arrived in host memory
    assert j["ms_per_step"] >= 2 * rf["avg_launch_ms"] * 0.999        # a step = two launches of the recurrence kernel
    # round 3: the prefix-state table is reported with what the MFMAs execute; timed region (iii) = the whole CLI on a FASTQ built
    # from the same stream rides along (--no-alt here, so no alt_* records)
    pt = j["config"]["prefix_table"]
    assert pt["k"] in range(0, 14) and (pt["k"] == 0 or pt["bytes"] == (4 ** pt["k"] + 1) * 1024)
    assert 0.8 < rf["steps_executed_over_steps"] <= 1.0 and (pt["k"] == 0) == (rf["steps_executed_over_steps"] == 1.0)
    # round 4: frac counts the steps the kernel EXECUTES (a table row's steps are looked up, not computed); three byte fields
    assert abs(rf["frac"] - rf["mfma_pipe_frac"] / 3) < 0.005 and rf["mfma_flops_executed_per_algorithmic_flop"] == 3
    assert abs(rf["frac"] - rf["frac_counting_table_steps"] * (rf["algorithmic_flops_per_launch"] / rf["algorithmic_flops_per_launch_counting_table_steps"])) < 1e-9
    assert rf["frac"] <= rf["frac_counting_table_steps"]
    assert rf["algorithmic_bytes_per_launch"] == 65536 * 113 and rf["offset_bytes_per_launch"] == 65536 * 8
    assert (pt["k"] == 0) == (rf["table_bytes_per_launch"] == 0) and rf["table_bytes_per_launch"] <= 65536 * 1024
    assert j["config"]["refine"]["placement"].startswith("deferred")
    e = j["e2e_cli"]["one_step_batch"]
    assert j["config"]["e2e_cli_plain_to_plain_reads_per_s"] > 1e5 and e["records_per_file"] == 65536 and e["files"] == 2 and e["timed_calls"] == 1
    assert e["calls"][0]["prefix_k"] == 8                                    # the CLI sizes its table by its input
    assert 0 < j["config"]["host_cores_busy"] < 4
    enc = j["encoder"]["kernels"]
    assert set(enc) == {"rd_encode_codes_kernel", "rd_encode_onehot_padded_kernel", "rd_pack_onehot_kernel"}
    assert all(v["achieved"] > 50 for v in enc.values())
    gz = j["device_gzip"]                                               # round 4: the .gz outputs' deflate runs on the device
    assert gz["records"] == 65536 and gz["text_bytes"] == 65536 * 218 and gz["members"] >= 218
    assert gz["size_vs_zlib_level_5"] < 1.10 and gz["ratio"] > 4 and gz["GB_per_s_of_text"] > 1


def test_bench_reports_the_rate_without_the_prefix_table(tmp_path):
    """alt_no_prefix_table: same timed region, same steps, k = 0 - lower rate, same roofline definition; alt_fp32_kernel rides along"""
    fp = str(tmp_path / "full.json")
    r = subprocess.run([sys.executable, "bench.py", "--steps", "4", "--warmup", "2", "--pairs-per-step", "524288", "--no-cpu-baseline",
                        "--no-encoder", "--no-e2e", "--traffic", "off", "--full-out", fp], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line, j = _line(r.stdout, fp)
    assert abs(line["alt_no_prefix_table_reads_per_s"] - j["alt_no_prefix_table"]["value"]) < 1e-4 * line["alt_no_prefix_table_reads_per_s"]
    assert abs(line["alt_fp32_frac"] - j["alt_fp32_kernel"]["roofline"]["frac"]) < 1e-4
    a = j["alt_no_prefix_table"]
    assert j["config"]["prefix_table"]["k"] >= 4 and a["timed_region"] == "ii" and a["steps"] == 4
    assert 0.70 * j["value"] < a["value"] < 1.05 * j["value"], (a["value"], j["value"])     # (wall clock over 4 steps: noisy)
    assert a["roofline"]["avg_launch_ms"] > 1.05 * j["roofline"]["avg_launch_ms"]             # (hipEvents around the launches: not noisy)
    # the strict-fp32 line: same region, same steps, its own prefix-state table, executed-work roofline
    f = j["alt_fp32_kernel"]
    assert f["steps"] == j["steps"] == 4 and f["timed_region"] == "ii" and f["ms_per_step"] > 0 and f["prefix_k"] == j["config"]["prefix_table"]["k"]
    assert f["roofline"]["frac"] > 0.5 and f["roofline"]["launches"] == 2 * 4
    assert abs(f["roofline"]["steps_executed_over_steps"] - j["roofline"]["steps_executed_over_steps"]) < 1e-9
    assert abs(f["value"] - 2 * 524288 * 4 / (f["ms_per_step"] * 4e-3)) < 1e-6 * f["value"]


def test_bench_self_launches_two_ranks_one_gpu(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE: bench.py starts its own two ranks (torch.distributed.run). Here they
    share the one GPU and exchange labels over gloo; on an N-GPU node the same path runs one rank per GPU over RCCL."""
    env = dict(os.environ, RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    fp = str(tmp_path / "full.json")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs-per-step", "65536", "--no-alt",
           "--no-cpu-baseline", "--full-out", fp]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line, j = _line(r.stdout, fp)
    assert line["n_gpus"] == 2 and line["value"] == j["value"] and line["config"]["gather_self_check"] == "passed" and len(j["config"]["ranks"]) == 2
    assert j["n_gpus"] == 2 and j["value"] > 1e6 and j["config"]["rccl_ranks"] == 2 and j["config"]["dist_backend"] == "gloo"
    assert abs(j["value"] - 2 * 2 * 65536 * 3 / (j["ms_per_step"] * 3e-3)) < 1e-6 * j["value"]     # whole-job aggregate


def test_bench_self_launches_eight_ranks_one_gpu(tmp_path):
    """the driver's largest point (--gpus 8) as far as a 1-GPU box can take it: eight ranks share the GPU, labels over gloo"""
    env = dict(os.environ, RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    fp = str(tmp_path / "full.json")
    cmd = [sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--pairs-per-step", "16384", "--no-alt",
           "--no-cpu-baseline", "--full-out", fp]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line, j = _line(r.stdout, fp)             # (the N = 8 line is under 4,000 bytes too: the per-rank table is in the full record)
    assert line["n_gpus"] == 8 and len(j["config"]["ranks"]) == 8 and line["config"]["label_counts"] == [j["config"]["label_counts"][k] for k in ("non_rrna", "rrna", "unclassified")]
    assert j["n_gpus"] == 8 and j["config"]["rccl_ranks"] == 8
    assert abs(j["value"] - 2 * 8 * 16384 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]
    c = j["config"]["label_counts"]
    assert c["non_rrna"] + c["rrna"] + c["unclassified"] == 8 * 16384 * 2


def test_bench_under_torchrun_two_ranks_one_gpu():
    """the driver's own launch form: python -m torch.distributed.run ... bench.py --gpus 2"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs-per-step", "65536",
           "--no-alt", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and j["value"] > 1e6


def test_scale_sweep_script_on_one_gpu(tmp_path):
    """tools/scale_sweep.sh - the N = 1, 2, 4, 8 sweep the first multi-GPU node will run - end to end on this box: the ranks of
    every point share the one GPU and exchange labels over gloo. Every point passes bench.py's gather self-check (labels gathered on
    rank 0 against each rank's own recomputation) and lists its ranks' devices; the JSON carries efficiency against N = 1 and the
    agreement of the N = 1 rank-group run with the plain line."""
    out = tmp_path / "sweep.json"
    env = dict(os.environ, RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0", RD_PREFIX_K="8", RD_SWEEP_SHARE_GPU="1", RD_SWEEP_CLI_RECORDS="262144",
               RD_SWEEP_CLI_NS="1 8")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_sweep.sh"), str(out), "--pairs-per-step", "16384", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    j = json.load(open(out))
    pts = {p["n_gpus"]: p for p in j["points"]}
    assert sorted(pts) == [1, 2, 4, 8] and not any(p.get("skipped") for p in j["points"]), j
    for n, p in pts.items():
        assert p["gather_self_check"] == "passed" and p["rccl_ranks"] == n and len(p["ranks"]) == n and p["reads_per_s"] > 1e5
        assert all(rk["device"] == "cuda:0" and rk["first_gather_s"] >= 0 for rk in p["ranks"])
        assert p["efficiency_vs_n1"] is not None
    assert j["plain_line_reads_per_s"] > 1e5 and "n1_group_over_plain" in j
    # round 6: the CLI's flows ride along at the same N (here N = 1 and 8): BGZF -> gz, plain -> plain, single-stream gz -> gz, every one of them
    # sharded (at N = 8 the .gz is decoded range by range on the ranks' GPU), outputs equal to the N = 1 run's
    for n in (1, 8):
        fl = pts[n]["cli"]
        assert set(fl) == {"bgzf_to_gz", "plain_to_plain", "gz_to_gz"}, fl
        assert all(v["cli_reads_per_s"] > 1e4 and v["outputs_equal_n1"] and v["cores_busy"] > 0 for v in fl.values()), fl
    assert pts[8]["cli"]["gz_to_gz"]["ingest_modes"] == ["gz-range"] and pts[1]["cli"]["gz_to_gz"]["ingest_modes"] == ["device"]
    assert all(c is None or c.get("how") for c in pts[8]["cpus"])


def test_gather_self_check_catches_a_wrong_gather():
    """bench.py --gpus N refuses to time a run whose gathered labels differ from the ranks' own: with RD_BENCH_CORRUPT_GATHER=1 rank 0
    flips one gathered label of the check step and every rank must leave with exit code 4"""
    env = dict(os.environ, RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0", RD_PREFIX_K="8", RD_BENCH_CORRUPT_GATHER="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--pairs-per-step", "16384", "--no-alt",
                        "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "label gather self-check FAILED" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_needs_n_devices():
    """--gpus 2 on a box with one device (and no shared-GPU override) stops with a clear message, before any rank starts"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has >= 2 devices")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RD_LOCAL_DEVICE", "RD_DIST_BACKEND")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "needs 2 visible devices" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_two_gpu_rccl_when_available():
    """one rank per GPU over RCCL (backend nccl): only on a box with >= 2 devices"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RD_LOCAL_DEVICE", "RD_DIST_BACKEND")}
    import tempfile
    fp = os.path.join(tempfile.mkdtemp(), "full.json")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-alt", "--no-cpu-baseline", "--full-out", fp],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    _, j = _line(r.stdout, fp)
    assert j["n_gpus"] == 2 and j["config"]["dist_backend"] == "nccl" and j["config"]["gather_self_check"] == "passed"
    assert len({rk["device_uuid"] for rk in j["config"]["ranks"]}) == 2
    _cli_two_rank_cases(env)


def _cli_two_rank_cases(env):
    # the CLI across the two devices against the one-process run: single-stream gz input through the range decoder (round 6: every rank
    # decodes its own range on ITS GPU, maps all-gathered over RCCL), the same with RD_GZ_SHARD=0 (one decode per node through shared memory,
    # label gather over RCCL), plain input (sharded parse, device gzip of every rank's part), and paired gz mates whose compressed
    # positions drift (the records in front of the common cut travel to the rank before: dist.shift_to_prev over RCCL point-to-point)
    import gzip
    import tempfile
    sys.path.insert(0, ROOT)
    from ribodetector_amd import synth

    def run(pre, args, extra):
        r = subprocess.run(pre + args, cwd=ROOT, env=dict(env, **extra), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return r.stdout + r.stderr
    one = [sys.executable, "-m", "ribodetector_amd.detect"]
    two = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29731",
           "-m", "ribodetector_amd.detect"]
    with tempfile.TemporaryDirectory() as d:
        a, o, _ = synth.reads_numpy(200000, (40, 140), seed=5, rrna_frac=0.3)
        a2, o2, _ = synth.reads_numpy(200000, (60, 100), seed=6, rrna_frac=0.3)
        for ext, extra, ranges in ((".fq.gz", {}, True), (".fq.gz", {"RD_GZ_SHARD": "0"}, False), (".fq", {}, False)):
            inp = os.path.join(d, "in" + ext)
            synth.write_fastq(inp, a, o, 1)
            got = {}
            for tag, pre in (("one", one), ("two", two)):
                out, rr = os.path.join(d, tag + ".non.fq.gz"), os.path.join(d, tag + ".rrna.fq")
                log = run(pre, ["-l", "100", "-i", inp, "-o", out, "-r", rr, "--chunk_size", "4", "-m", "3"], extra)
                got[tag] = (gzip.open(out, "rb").read(), open(rr, "rb").read())
                if tag == "two":
                    assert ("ranges of one DEFLATE stream" in log) == ranges, log[-1500:]
            assert got["one"] == got["two"] and len(got["one"][0]) > 0 and len(got["one"][1]) > 0
        i1, i2 = os.path.join(d, "m_1.fq.gz"), os.path.join(d, "m_2.fq.gz")
        synth.write_fastq(i1, a, o, 1)
        synth.write_fastq(i2, a2, o2, 2, prefix="the_second_mate_has_longer_headers")
        got = {}
        for tag, pre in (("one", one), ("two", two)):
            outs = [os.path.join(d, "%s.%s" % (tag, x)) for x in ("n1.fq", "n2.fq.gz", "r1.fq.gz", "r2.fq")]
            log = run(pre, ["-l", "100", "-i", i1, i2, "-o", *outs[:2], "-r", *outs[2:], "-e", "rrna", "--chunk_size", "4", "-m", "3"], {})
            got[tag] = [(gzip.open(x, "rb") if x.endswith("gz") else open(x, "rb")).read() for x in outs]
            if tag == "two":
                assert "ranges of one DEFLATE stream" in log and len(set(__import__("re").findall(r"Rank (\d) decoded", log))) == 2
        assert got["one"] == got["two"] and all(len(x) > 0 for x in got["one"])



def test_two_rank_cli_cases_on_one_gpu():
    """the CLI cases of test_two_gpu_rccl_when_available with both ranks on this box's one GPU over gloo (RD_TEST_FULL=1 only: the flows
    themselves are covered by tests/test_gpu_cli.py; this keeps the two-GPU test's own code from rotting on 1-GPU boxes)"""
    from conftest import FULL
    if not FULL:
        pytest.skip("RD_TEST_FULL=1")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    _cli_two_rank_cases(dict(env, RD_DIST_BACKEND="gloo", RD_LOCAL_DEVICE="0"))
