#!/bin/bash
# scale_sweep.sh - the metric's N > 1 points in one go: `python bench.py --gpus N` for N = 1, 2, 4, 8 back to back (each run
# launches its own ranks, one per GPU over RCCL, and checks its label gather against a single-rank recomputation before timing),
# plus the plain one-process line; writes ONE JSON with reads/s, efficiency against N = 1 and the agreement of the N = 1 rank-group
# run with the plain line (must be within 2 %).
#   tools/scale_sweep.sh [out.json] [extra bench.py flags ...]
# On a box with fewer devices than N the point is skipped (bench.py exits 3) - or, for a functional run of the whole sweep on ONE
# GPU (tests): RD_DIST_BACKEND=gloo RD_LOCAL_DEVICE=0 tools/scale_sweep.sh out.json --pairs-per-step 16384 --steps 2
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/scale_sweep.json}
shift || true
NS=${RD_SWEEP_NS:-"1 2 4 8"}
FLAGS="--no-cpu-baseline --no-alt --no-encoder --no-e2e --traffic off $*"
TMP=$(mktemp -d)
cd "$R"
# a throw-away run first: the first process on a box pays cold caches and clocks, and the plain line is compared with the N = 1 group
env -u RD_FORCE_DIST -u WORLD_SIZE -u RANK -u LOCAL_RANK python bench.py --gpus 1 $FLAGS --steps 2 > /dev/null 2>&1
# the plain line (no process group at all)
# (the full record of every run - per-rank table included - goes to --full-out; stdout carries the compact line only)
env -u RD_FORCE_DIST -u WORLD_SIZE -u RANK -u LOCAL_RANK python bench.py --gpus 1 $FLAGS --full-out $TMP/plain.json > /dev/null 2> $TMP/plain.err
for N in $NS; do
  if [ "$N" = 1 ]; then   # one rank THROUGH the collectives (a one-rank group): what the N > 1 runs add to a rank, at N = 1
    RD_FORCE_DIST=1 python bench.py --gpus 1 $FLAGS --full-out $TMP/n1.json > /dev/null 2> $TMP/n1.err; echo "N=1 rc=$?" >> $TMP/rc
    # ... and the plain line once more, BEHIND the group run: the box keeps warming up over the first runs of a fresh lease, so the
    # N = 1 group is compared with the mean of the plain runs on either side of it (same flags, same pairs per step)
    env -u RD_FORCE_DIST -u WORLD_SIZE -u RANK -u LOCAL_RANK python bench.py --gpus 1 $FLAGS --full-out $TMP/plain2.json > /dev/null 2>> $TMP/plain.err
  else
    env -u RD_FORCE_DIST python bench.py --gpus $N $FLAGS --full-out $TMP/n$N.json > /dev/null 2> $TMP/n$N.err; echo "N=$N rc=$?" >> $TMP/rc
  fi
done
# the CLI's flows at the same N (round 6): sequencer-like files in tmpfs, built once; BGZF -> gz, plain -> plain, single-stream gz -> gz
# under torch.distributed.run, one GPU per rank (RD_SWEEP_CLI=0: skip; RD_SWEEP_SHARE_GPU=1: all ranks on GPU 0 over gloo, for a 1-GPU box)
if [ "${RD_SWEEP_CLI:-1}" != 0 ]; then
  CLID=$(mktemp -d -p /dev/shm rd_sweep_XXXX 2>/dev/null || mktemp -d)
  python tools/scale_cli.py --make $CLID --records ${RD_SWEEP_CLI_RECORDS:-8388608} > $TMP/cli_make.json 2> $TMP/cli_make.err
  for N in ${RD_SWEEP_CLI_NS:-$NS}; do
    python tools/scale_cli.py --run $CLID --gpus $N ${RD_SWEEP_SHARE_GPU:+--share-gpu} > $TMP/cli$N.json 2> $TMP/cli$N.err
  done
  rm -rf $CLID
fi
python - "$TMP" "$OUT" $NS <<'PY'
import json, os, sys
tmp, out, ns = sys.argv[1], sys.argv[2], [int(x) for x in sys.argv[3:]]
def load(name):
    try:
        return json.load(open(os.path.join(tmp, name)))
    except Exception:
        return None
plain, plain2 = load("plain.json"), load("plain2.json")
if plain and plain2:
    plain = dict(plain, value=0.5 * (plain["value"] + plain2["value"]), values=[plain["value"], plain2["value"]])
rec = {"plain_line_reads_per_s": plain and plain["value"], "plain_runs": plain and plain.get("values"), "points": [],
       "rc": open(os.path.join(tmp, "rc")).read().split("\n")[:-1]}
base = None
cli_ref = {}
for n in ns:
    j = load("n%d.json" % n)
    if j is None:
        err = open(os.path.join(tmp, "n%d.err" % n)).read()[-400:]
        rec["points"].append({"n_gpus": n, "skipped": True, "stderr_tail": err})
        continue
    if n == 1:
        base = j["value"]
    pt = {"n_gpus": n, "reads_per_s": j["value"], "ms_per_step": j["ms_per_step"], "dist_backend": j["config"]["dist_backend"],
          "rccl_ranks": j["config"]["rccl_ranks"], "host_cores_busy": j["config"]["host_cores_busy"],
          "gather_self_check": j["config"].get("gather_self_check"),
          "ranks": [{k: r[k] for k in ("rank", "device", "device_uuid", "first_gather_s")} for r in (j["config"].get("ranks") or [])],
          "efficiency_vs_n1": (j["value"] / (n * base)) if base else None,
          "gpu_state": j["roofline"].get("gpu_state_in_timed_region"), "cpus": [r.get("cpus") for r in (j["config"].get("ranks") or [])]}
    cli = load("cli%d.json" % n)
    if cli:          # the CLI's flows at this N: reads/s of detect() over the ranks, host cores busy, outputs against the N = 1 run's
        pt["cli"] = cli["flows"]
        for name, fl in cli["flows"].items():
            if "output_sha1" in fl:
                cli_ref.setdefault(name, fl["output_sha1"])
                fl["outputs_equal_n1"] = fl.pop("output_sha1") == cli_ref[name]
    rec["points"].append(pt)
if plain and base:
    rec["n1_group_over_plain"] = base / plain["value"]
    rec["n1_agrees_with_plain_within_2pct"] = abs(base / plain["value"] - 1) <= 0.02
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
PY
rm -rf $TMP
