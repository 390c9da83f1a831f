"""bench.py's stdout line without a GPU: compact() applied to a full record of the size a real run produces (profiles/r04_bench.json:
22 KB, the record whose one-line form was cut by the driver's 8,000-byte stdout tail in round 4) stays under 4,000 bytes and keeps
every contract key, the roofline's numbers and cpu_baseline. The GPU contract tests assert the same on live runs."""
import importlib.util
import json
import os

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _bench():
    spec = importlib.util.spec_from_file_location("rd_bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _full(n_ranks=1):
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    legs = {}
    for k, v in full["e2e_cli"].items():            # the round-4 record's legs under this round's names, + the three new ones
        if isinstance(v, dict) and "reads_per_s" in v:
            legs["plain_to_plain" if k == "large" else k] = dict(v, reads_per_s_after_first_chunk=1.07 * v["reads_per_s"])
    for extra in ("bgzf_to_plain", "bgzf_to_gz_host_parse", "plain_to_plain_host_parse", "seqlike_plain_to_gz", "seqlike_bgzf_to_gz", "seqlike_gz_to_gz"):
        legs[extra] = dict(legs["plain_to_gz"])          # (round 6: the three legs on sequencer-like files; the *_host_parse legs left the line)
    full["roofline"].update(effective_clock_ghz=1.873, package_power_w=1288.4)
    full["e2e_cli"] = legs
    full["cpu_baseline"]["sample_short"] = "first 229376 reads of rank 0's R1 stream; C port (AVX + OpenMP) of the padded 100-step BiLSTM, batch 1024/thread, 16 threads, 12.3 s"
    full["n_gpus"] = n_ranks
    full["config"]["ranks"] = [{"rank": r, "local_rank": r, "device": "cuda:%d" % r, "device_uuid": "GPU-%032x" % r, "pid": 1000 + r,
                                "first_gather_s": 0.1234, "step0_label_counts": [1000000, 48576, 0]} for r in range(n_ranks)]
    return full


def test_compact_line_fits_the_drivers_tail():
    b = _bench()
    for n in (1, 8):
        full = _full(n)
        line = json.dumps(b.compact(full), separators=(",", ":"))
        assert len(line) < 3500, len(line)
        j = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                  "data", "config", "roofline", "cpu_baseline"):
            assert k in j, k
        assert j["value"] == full["value"] and j["ms_per_step"] == full["ms_per_step"] and "workload" in j["config"]
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(j["roofline"])
        assert {"value", "unit", "cores", "kind", "sample"} <= set(j["cpu_baseline"])
        assert abs(j["roofline"]["frac"] - j["roofline"]["achieved"] / j["roofline"]["peak"]) < 1e-4
        assert j["e2e_bgzf_to_gz"]["rps"] > 0 and j["e2e_plain_to_plain"]["host_cores_busy"] > 0 and "ranks" not in j["config"]
        assert j["e2e_seqlike_gz_to_gz"]["rps"] > 0 and "e2e_bgzf_to_gz_host_parse" not in j and j["roofline"]["effective_clock_ghz"] == 1.873
        assert "parity_sample" in j and j["alt_fp32_frac"] > 0.5


def test_emit_writes_the_full_record_and_prints_one_line(tmp_path, capsys):
    b = _bench()

    class A:
        full_out = str(tmp_path / "full.json")
        verbose = False
    full = _full()
    b.emit(full, A)
    out = capsys.readouterr().out
    assert out.endswith("\n") and out.count("\n") == 1 and len(out) < 4000
    assert json.load(open(A.full_out))["e2e_cli"]["gz_to_gz"]["calls"]          # the nested per-call records live in the file
