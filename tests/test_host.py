"""Host-side mirror of the reference interface (no GPU): config surface, SeqModel argument handling, synthetic data."""
import os

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _cfg():
    from ribodetector_amd.parse_config import ConfigParser
    return ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json"))


def test_config_surface():
    cfg = _cfg()
    assert cfg["n_gpu"] == 1 and cfg["arch"]["type"] == "SeqModel"
    assert dict(cfg["arch"]["args"]) == dict(input_size=4, hidden_size=128, num_layers=1, num_classes=2, batch_first=True,
                                             bidirectional=True, pack_seq=True)
    assert set(cfg["state_file"]) == {"mcc", "recall"}
    sd = cfg.load_state_dict("recall")
    assert sorted(sd) == sorted(["rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0",
                                 "rnn.weight_ih_l0_reverse", "rnn.weight_hh_l0_reverse", "rnn.bias_ih_l0_reverse",
                                 "rnn.bias_hh_l0_reverse", "out.weight", "out.bias"])
    assert sum(v.size for v in sd.values()) == 137730


def test_weights_digest():
    import hashlib
    import json
    dg = json.load(open(os.path.join(ROOT, "ribodetector_amd", "data", "weights_digest.json")))
    assert dg["source_sha256"] == "54d58c03967f2425b0bebde7a6b1a81d12ef55ca539e388480c9aea5aa401b78"
    sd = _cfg().load_state_dict("mcc")
    for k, v in dg["tensors"].items():
        assert list(sd[k].shape) == v["shape"]
        assert hashlib.sha256(np.ascontiguousarray(sd[k]).tobytes()).hexdigest() == v["sha256"], k


def test_init_obj_and_arg_validation():
    from ribodetector_amd.model import model as module_arch
    cfg = _cfg()
    m = cfg.init_obj("arch", module_arch)
    assert m.hidden_size == 128 and m.pack_seq
    with pytest.raises(AssertionError):
        cfg.init_obj("arch", module_arch, hidden_size=64)          # overriding config kwargs is not allowed
    assert not module_arch.SeqModel(**dict(cfg["arch"]["args"], pack_seq=False)).pack_seq      # forward2: padded Tensor input
    for bad in (dict(num_layers=2), dict(bidirectional=False), dict(pack_seq=False, batch_first=False), dict(hidden_size=64)):
        args = dict(cfg["arch"]["args"])
        args.update(bad)
        with pytest.raises(NotImplementedError):
            module_arch.SeqModel(**args)
    sd = cfg.load_state_dict("mcc")
    m.load_state_dict(sd)
    assert set(m.state_dict()) == set(sd)
    broken = dict(sd)
    broken.pop("out.bias")
    with pytest.raises(RuntimeError, match="Missing key"):
        m.load_state_dict(broken)
    pref = {"module." + k: v for k, v in sd.items()}            # the reference's DataParallel failure mode
    with pytest.raises(RuntimeError, match="Missing key"):
        m.load_state_dict(pref)
    with pytest.raises(RuntimeError, match="GPU only"):
        m.to("cpu")


def test_synth_deterministic():
    from ribodetector_amd import synth
    a1, o1, l1 = synth.reads_numpy(500, (40, 300), seed=3)
    a2, o2, l2 = synth.reads_numpy(500, (40, 300), seed=3)
    assert (a1 == a2).all() and (o1 == o2).all() and (l1 == l2).all()
    assert l1.min() >= 40 and l1.max() <= 300 and o1[-1] == len(a1)
    assert set(np.unique(a1)) <= set(b"ACGTN")
    a3, _, _ = synth.reads_numpy(500, 100, seed=4)
    assert abs((a3 == ord("N")).mean() - 0.001) < 0.002


def test_gate_math_schedule_is_reproducible():
    """the hand-interleave tables of the default kernel (EW_CELL / EW_STAGE) are what tools/gen_ew_schedule.py generates"""
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_ew_schedule.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_part_files_of_the_sharded_cli(tmp_path):
    """every rank writes '<output>.part<r>' (gzip outputs keep a name that ends in 'gz': the writer compresses by name) and rank 0
    joins the parts in rank order; a concatenation of gzip members is a valid gzip file"""
    import gzip
    from ribodetector_amd.detect import part_path
    from ribodetector_amd.data_loader import fastx_parser as fx
    assert part_path("out/n.fq", 3) == "out/n.fq.part3"
    assert part_path("out/n.fq.gz", 0) == "out/n.fq.part0.gz" and part_path("o.fq.unclassified.gz", 7).endswith(".part7.gz")
    final = str(tmp_path / "joined.fq.gz")
    parts = [part_path(final, r) for r in range(3)]
    blobs = [b"@a\nAC\n+\nII\n" * 1000, b"", b"@b\nGT\n+\nII\n" * 7]
    for p, b in zip(parts, blobs):
        with gzip.open(p, "wb") as fh:
            fh.write(b)
    fx.concatenate_parts(final, parts)
    assert gzip.open(final, "rb").read() == b"".join(blobs)
    assert not any(os.path.exists(p) for p in parts)
    # the concurrent form: the final file is created at its full size, every part is placed at its offset (any order)
    final2 = str(tmp_path / "placed.fq.gz")
    parts2 = [part_path(final2, r) for r in range(3)]
    comp = []
    for p, b in zip(parts2, blobs):
        with gzip.open(p, "wb") as fh:
            fh.write(b)
        comp.append(os.path.getsize(p))
    with open(final2, "wb") as fh:
        fh.truncate(sum(comp))
    for r in (2, 0, 1):
        fx.place_part(final2, parts2[r], sum(comp[:r]))
    assert gzip.open(final2, "rb").read() == b"".join(blobs) and not any(os.path.exists(p) for p in parts2)
    plain = str(tmp_path / "joined.fq")
    pp = [part_path(plain, r) for r in range(2)]
    open(pp[0], "wb").write(b"x" * 5)
    open(pp[1], "wb").write(b"y" * 3)
    fx.concatenate_parts(plain, pp)
    assert open(plain, "rb").read() == b"xxxxxyyy"


# ---- round 4: advisor findings of round 3 (shared-memory chunk slots, cleanup on SIGTERM, LOCAL_WORLD_SIZE) ----------------------------

def _predictor(tmp_path=None):
    import argparse
    from ribodetector_amd import detect
    from ribodetector_amd.parse_config import ConfigParser
    args = argparse.Namespace(log=None, chunk_size=None, len=100, ensure="none", deviceid=None, semantics=None)
    return detect.Predictor(ConfigParser.from_json(os.path.join(ROOT, "ribodetector_amd", "config.json")), args)


def test_shm_slot_is_reserved_not_sparse(tmp_path, monkeypatch):
    """a chunk slot in /dev/shm is posix_fallocate'd: where the space is missing the reader gets an OSError with a hint (which travels
    the parser-error path to the other ranks) instead of a SIGBUS at the first store; nothing is left behind"""
    import errno
    from ribodetector_amd.data_loader import fastx_parser as fx
    a = fx.ShmArena("rd_0_%d_f0" % os.getpid())
    views, sl = a.alloc(1 << 16, 100)
    assert os.path.getsize(sl["path"]) >= (1 << 16) and os.stat(sl["path"]).st_blocks * 512 >= (1 << 16)   # pages are reserved
    a.close()
    assert not os.path.exists(sl["path"])

    def no_space(fd, off, n):
        raise OSError(errno.ENOSPC, "No space left on device")
    monkeypatch.setattr(os, "posix_fallocate", no_space)
    b = fx.ShmArena("rd_0_%d_f1" % os.getpid())
    with pytest.raises(OSError, match="RD_SHARED_DECODE=0"):
        b.alloc(1 << 16, 100)
    assert not [f for f in os.listdir(b.dir) if f.startswith("rd_0_%d_f1" % os.getpid())]
    ok, need = fx.ShmArena.fits(2, 1 << 20, 280)
    assert need > 2 * 6 * (1 << 20) * 280 and isinstance(ok, bool)
    assert fx.ShmArena.fits(2, 1 << 40, 280)[0] is False                       # more than any /dev/shm holds


def test_stale_shm_slots_of_dead_processes_are_swept(tmp_path):
    import subprocess
    import sys
    from ribodetector_amd.data_loader import fastx_parser as fx
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    p = subprocess.Popen([sys.executable, "-c", "pass"])
    p.wait()
    dead = os.path.join(d, "rd_12345_%d_f0.0" % p.pid)                         # a slot of a process that is gone
    alive = os.path.join(d, "rd_12345_%d_f0.0" % os.getpid())                  # ... and of one that is not
    for f in (dead, alive):
        open(f, "wb").write(b"x")
    try:
        assert fx.ShmArena.sweep_stale() >= 1
        assert not os.path.exists(dead) and os.path.exists(alive)
    finally:
        for f in (dead, alive):
            if os.path.exists(f):
                os.remove(f)


def test_sigterm_removes_slots_and_part_files(tmp_path):
    """when ANOTHER rank fails, torch.distributed.run sends this one SIGTERM: the handler Predictor installs removes the shared-memory
    slots and this rank's part files before the process goes"""
    import signal
    import subprocess
    import sys
    code = r'''
import os, sys, signal, argparse
sys.path.insert(0, %r)
from ribodetector_amd import detect
from ribodetector_amd.data_loader import fastx_parser as fx
from ribodetector_amd.parse_config import ConfigParser
args = argparse.Namespace(log=None, chunk_size=None, len=100, ensure="none", deviceid=None, semantics=None)
p = detect.Predictor(ConfigParser.from_json(os.path.join(%r, "ribodetector_amd", "config.json")), args)
a = fx.ShmArena("rd_0_%%d_f0" %% os.getpid())
views, sl = a.alloc(1 << 16, 100)
p._arenas.append(a)
part = os.path.join(%r, "out.fq.part1")
open(part, "w").write("x")
p._part_files.append(part)
print(sl["path"], flush=True)
os.kill(os.getpid(), signal.SIGTERM)
import time; time.sleep(30)
''' % (ROOT, ROOT, str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == -signal.SIGTERM, (r.returncode, r.stderr[-1500:])
    slot = r.stdout.strip().splitlines()[-1]
    assert slot and not os.path.exists(slot) and not os.path.exists(str(tmp_path / "out.fq.part1"))


def test_shared_decode_needs_local_world_size(monkeypatch):
    """one decode per node only when the launcher says the ranks ARE on one node: LOCAL_WORLD_SIZE set and equal to the world size
    (a launcher that sets only RANK / WORLD_SIZE may have spread the ranks over hosts, whose /dev/shm are different memories)"""
    p = _predictor()
    p.multi, p.world, p.rank, p.sharded_parse = True, 2, 1, False
    p.input = ["x.fq.gz"]
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert p._shared_decode() is False                                        # (no collective is entered for the answer)
    p._shared = None
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    assert p._shared_decode() is False
    p._shared = None
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    monkeypatch.setenv("RD_SHARED_DECODE", "0")
    assert p._shared_decode() is False


def test_which_output_files_are_gzip(tmp_path):
    """by name, like the reference's writer (detect.py:738); --ensure both adds the two '.unclassified.gz' files (detect.py:390-400)"""
    from ribodetector_amd.detect import Predictor
    f = Predictor.gz_output_files
    assert f(["o.fq"], None, False, "none") == []
    assert f(["o.fq.gz"], None, False, "none") == [(0, 0)]
    assert f(["o.fq"], ["r.fq.gz"], False, "none") == [(0, 1)]
    assert f(["o1.fq.gz", "o2.fq"], ["r1.fq", "r2.fastq.gz"], True, "rrna") == [(1, 1), (0, 0)]
    assert f(["o1.fq", "o2.fq"], None, True, "both") == [(0, -1), (1, -1)]
    assert f(["o1.fqgz", "o2.fq"], None, True, "none") == [(0, 0)]            # ends with 'gz', as the reference tests it
