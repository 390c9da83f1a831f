"""Multi-GPU sharding logic on CPU: world_size-2 gloo processes (the RCCL path differs only in the backend string).
Invariant: the gathered label vector with W ranks == the W=1 label vector, bit for bit, and the
all-reduced counters equal the global counts."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, ensure, tmp, weighted=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from oracle import oracle as O
    from ribodetector_amd import dist as rdist
    from ribodetector_amd import synth
    r, w, _ = rdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    ora = O.load_default()
    a1, o1, l1 = synth.reads_numpy(n, (50, 110), seed=5)
    a2, o2, l2 = synth.reads_numpy(n, (50, 110), seed=6)
    bounds = None
    if weighted:                                  # equal bases per rank (what detect.py does under torchrun)
        bounds = rdist.shard_bounds(n, world, np.minimum(l1, 100).astype(np.int64) + np.minimum(l2, 100))
        lo, hi = bounds[rank], bounds[rank + 1]
    else:
        lo, hi = rdist.shard_range(n, rank, world)
    # per-rank classifier = the CPU oracle on this rank's contiguous shard of the pairs
    g1 = ora.forward_packed(a1, o1[lo:hi + 1], l1[lo:hi], 100)
    g2 = ora.forward_packed(a2, o2[lo:hi + 1], l2[lo:hi], 100)
    lab = torch.from_numpy(ora.pair_fuse(g1, g2, ensure))
    counts = torch.tensor(ora.count_labels(lab.numpy()), dtype=torch.int64)
    full = rdist.gather_labels(lab, n, dst=0, bounds=bounds)
    _, fin = rdist.gather_labels(lab, n, dst=0, async_op=True, bounds=bounds)
    full2 = fin()
    rdist.reduce_counts(counts)
    if rank == 0:
        assert torch.equal(full, full2)
        np.save(os.path.join(tmp, "labels.npy"), full.numpy())
    np.save(os.path.join(tmp, "counts%d.npy" % rank), counts.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,ensure,weighted", [(101, "both", False), (64, "rrna", False), (97, "none", True)])
def test_two_rank_gather_matches_single(tmp_path, oracle, n, ensure, weighted):
    from ribodetector_amd import synth
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, ensure, str(tmp_path), weighted), nprocs=2, join=True)
    a1, o1, l1 = synth.reads_numpy(n, (50, 110), seed=5)
    a2, o2, l2 = synth.reads_numpy(n, (50, 110), seed=6)
    want = oracle.pair_fuse(oracle.forward_packed(a1, o1, l1, 100), oracle.forward_packed(a2, o2, l2, 100), ensure)
    got = np.load(str(tmp_path / "labels.npy"))
    assert got.dtype == np.int8 and (got == want).all()
    c = oracle.count_labels(want)
    for r in range(2):
        assert np.load(str(tmp_path / ("counts%d.npy" % r))).tolist() == c


def _var_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ribodetector_amd import dist as rdist
    rdist.init_from_env(backend="gloo")
    rng = np.random.default_rng(100 + rank)
    # three "files" per chunk with sizes that differ per rank, one of them empty on some ranks, one empty everywhere
    for chunk in range(3):
        mine = [int(rng.integers(0, 5000)) if (rank + chunk) % 2 else 0, int(rng.integers(1, 70000)), 0]
        bufs = [torch.from_numpy(rng.integers(0, 256, max(m, 1) + 17, dtype=np.uint8)) for m in mine]   # (longer than the payload, like the device buffers)
        sizes = rdist.all_gather_sizes(mine)
        assert sizes.shape == (world, 3) and sizes[rank].tolist() == mine
        for f in range(3):
            got = rdist.gather_var_bytes(bufs[f], mine[f], sizes[:, f].tolist(), dst=0)
            np.save(os.path.join(tmp, "sent_%d_%d_%d.npy" % (chunk, f, rank)), bufs[f][: mine[f]].numpy())
            if rank == 0:
                assert len(got) == world
                for r in range(world):
                    np.save(os.path.join(tmp, "got_%d_%d_%d.npy" % (chunk, f, r)), got[r].numpy())
            else:
                assert got is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_variable_size_byte_gather(tmp_path, world):
    """gather_var_bytes / all_gather_sizes: the compressed members every rank makes of its shard of a chunk reach rank 0 byte for byte,
    whatever their sizes (empty ones included), in rank order"""
    mp.spawn(_var_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for chunk in range(3):
        for f in range(3):
            for r in range(world):
                sent = np.load(str(tmp_path / ("sent_%d_%d_%d.npy" % (chunk, f, r))))
                got = np.load(str(tmp_path / ("got_%d_%d_%d.npy" % (chunk, f, r))))
                assert sent.dtype == np.uint8 and np.array_equal(sent, got)


def test_shard_ranges_cover_and_order():
    from ribodetector_amd import dist as rdist
    for n in (0, 1, 7, 64, 1000003):
        for w in (1, 2, 3, 8):
            rs = [rdist.shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sz = rdist.shard_sizes(n, w)
            assert sum(sz) == n and max(sz) - min(sz) <= 1


def test_weighted_bounds_balance_bases():
    from ribodetector_amd import dist as rdist
    rng = np.random.default_rng(3)
    for n, w in ((1000, 8), (17, 3), (5, 8), (0, 4), (1, 2)):
        lens = rng.integers(0, 301, n)
        lens[: n // 2] = np.sort(lens[: n // 2])              # a sorted stretch: equal read counts would be unbalanced
        b = rdist.shard_bounds(n, w, lens)
        assert b[0] == 0 and b[-1] == n and all(b[i] <= b[i + 1] for i in range(w))
        if n >= 100:
            per = [int(lens[b[r]:b[r + 1]].sum()) for r in range(w)]
            assert max(per) - min(per) <= 2 * 300, per         # within one or two reads' worth of bases
    assert rdist.shard_bounds(10, 2) == [0, 5, 10]
    assert rdist.shard_bounds(4, 2, [0, 0, 0, 0]) in ([0, 4, 4], [0, 0, 4])   # all-empty reads: any monotone split


def test_single_process_passthrough():
    from ribodetector_amd import dist as rdist
    x = torch.arange(5, dtype=torch.int8)
    assert rdist.gather_labels(x, 5) is x
    c = torch.tensor([1, 2, 3])
    assert rdist.reduce_counts(c) is c


def _shm_worker(rank, world, port, path, tmp):
    """the per-chunk message of the one-decode-per-node mode (detect.Predictor._chunk_stream): rank 0 parses a .gz once into
    shared-memory slots and broadcasts where each chunk lies; the other ranks map the slot"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import hashlib
    from ribodetector_amd import dist as rdist
    from ribodetector_amd.data_loader import fastx_parser as fx
    rdist.init_from_env(backend="gloo")
    h = hashlib.sha256()
    n_rec = n_chunks = 0
    if rank == 0:
        arena = fx.ShmArena("rd_test_%d_%d" % (port, os.getpid()))
        try:
            held = []
            for c in fx.get_seq_chunks(path, chunk_size=1000, first_chunk=250, arena=arena):
                dist.broadcast_object_list([[c.shm]], src=0)
                h.update(c.buf.tobytes()); h.update(c.seq_off.tobytes()); h.update(c.seq_len.tobytes())
                n_rec += len(c.seq_len)
                n_chunks += 1
                held.append(c)
                if len(held) > 2:                       # the writer of the CLI releases a slot once its chunk is written
                    dist.barrier()                      # (here: once every rank has hashed the chunk)
                    held.pop(0).release()
                else:
                    dist.barrier()
            dist.broadcast_object_list([None], src=0)
            slots = len(arena.slots)
        finally:
            arena.close()
        assert slots <= 7 and slots < n_chunks // 2, (slots, n_chunks)   # slots are reused (the ramp of chunk sizes adds a few small ones)
    else:
        while True:
            msg = [None]
            dist.broadcast_object_list(msg, src=0)
            if msg[0] is None:
                break
            c = fx.ShmArena.attach(msg[0][0])
            h.update(c.buf.tobytes()); h.update(c.seq_off.tobytes()); h.update(c.seq_len.tobytes())
            n_rec += len(c.seq_len)
            n_chunks += 1
            dist.barrier()
    open(os.path.join(tmp, "shm%d.txt" % rank), "w").write("%s %d %d" % (h.hexdigest(), n_rec, n_chunks))
    dist.barrier()
    dist.destroy_process_group()


def test_shared_memory_chunks_reach_the_other_ranks(tmp_path):
    import hashlib
    from ribodetector_amd import synth
    from ribodetector_amd.data_loader import fastx_parser as fx
    a, o, _ = synth.reads_numpy(20000, (30, 150), seed=8)
    path = str(tmp_path / "r.fq.gz")
    synth.write_fastq(path, a, o, 1)
    port = _free_port()
    mp.spawn(_shm_worker, args=(3, port, path, str(tmp_path)), nprocs=3, join=True)
    h = hashlib.sha256()
    n = 0
    for c in fx.get_seq_chunks(path, chunk_size=1000, first_chunk=250):
        h.update(c.buf.tobytes()); h.update(c.seq_off.tobytes()); h.update(c.seq_len.tobytes())
        n += len(c.seq_len)
    got = [open(str(tmp_path / ("shm%d.txt" % r))).read().split() for r in range(3)]
    assert all(g[0] == h.hexdigest() and int(g[1]) == n == 20000 for g in got), got
    assert int(got[0][2]) >= 20                                                  # many chunks, ramped sizes
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("rd_test_%d_" % port)]


def _shift_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from ribodetector_amd import dist as rdist
    rdist.init_from_env(backend="gloo")
    # rank r sends r * 1000 + 7 bytes of value r to rank r - 1 (rank 2 sends nothing: None); rank 0's own buffer goes nowhere
    buf = None if rank == 2 else torch.full((rank * 1000 + 7,), rank, dtype=torch.uint8)
    got = rdist.shift_to_prev(buf)
    np.save(os.path.join(tmp, "got%d.npy" % rank), np.zeros(0, np.uint8) if got is None else got.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shift_to_prev_three_ranks(tmp_path):
    """dist.shift_to_prev (round 6: the mate records in front of a rank's common cut travel to the rank before, data_loader/gz_shard.py):
    rank r receives exactly what rank r + 1 sent, the last rank and the neighbour of a rank that sent nothing receive nothing"""
    port = _free_port()
    mp.spawn(_shift_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    got = [np.load(str(tmp_path / ("got%d.npy" % r))) for r in range(4)]
    assert got[0].size == 1007 and (got[0] == 1).all()
    assert got[1].size == 0                       # rank 2 sent nothing
    assert got[2].size == 3007 and (got[2] == 3).all()
    assert got[3].size == 0


def test_pin_rank_cpus_deals_equal_shares():
    """dist.pin_rank_cpus without a GPU (no NUMA node to go by): the usable CPUs in equal contiguous shares, the process bound to its own;
    RD_PIN=0 and a single rank leave the affinity alone"""
    from ribodetector_amd import dist as rdist
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity")
    before = sorted(os.sched_getaffinity(0))
    try:
        assert rdist.pin_rank_cpus(0, 0, 1) == (None, "off")
        assert rdist.pin_rank_cpus(0, 1, 4, mode="0") == (None, "off")
        assert sorted(os.sched_getaffinity(0)) == before
        if len(before) >= 4:
            cpus, how = rdist.pin_rank_cpus(0, 1, 4)
            k = len(before)
            assert cpus == before[k // 4:2 * k // 4] and "share 2 of 4" in how and sorted(os.sched_getaffinity(0)) == cpus
            os.sched_setaffinity(0, before)
            cpus, how = rdist.pin_rank_cpus(0, 3, 4, mode="node")
            assert cpus == before and "share" not in how
    finally:
        os.sched_setaffinity(0, before)
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), before)
            except OSError:
                pass
    assert rdist._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
