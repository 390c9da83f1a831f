"""Multi-GPU sharding of the read stream.

Reads are independent, so a chunk of n reads (or pairs - R1[i] and R2[i] stay on the same rank) is split into W
contiguous sub-ranges; rank r classifies [r*n/W, (r+1)*n/W). The only exchange is (1) a gather of the 1-byte labels to
rank 0, which concatenated in rank order ARE the labels in input order, and (2) an all-reduce(SUM) of the three
counters (non-rRNA, rRNA, unclassified - reference detect.py:331-333,388-389,400). Backend "nccl" (= RCCL over xGMI on
ROCm) for device tensors, "gloo" for the CPU tests. The reference has no working multi-GPU path to mirror (its
DataParallel wrap fails to load the checkpoint).
"""
import os

import torch
import torch.distributed as dist


def forced():
    """RD_FORCE_DIST=1: take the collective branches even with ONE rank (a one-rank RCCL communicator), so that the code a
    multi-GPU node runs - init_process_group("nccl"), the label gather and the counter all-reduce on device tensors - also
    executes on a 1-GPU box (tests/test_gpu_dist.py, `RD_FORCE_DIST=1 python bench.py`)."""
    return os.environ.get("RD_FORCE_DIST", "") == "1"


def active(group=None):
    """True when the collectives below really run: several ranks, or one rank under RD_FORCE_DIST=1"""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or forced())


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:   # a forced one-rank group outside torchrun: any free port
            import socket
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s.getsockname()[1])
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:                      # RD_DIST_BACKEND=gloo: label exchange over host memory (e.g. several ranks
            backend = os.environ.get("RD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")   # sharing one GPU)
        if backend == "nccl":
            dev = int(os.environ.get("RD_LOCAL_DEVICE", local))
            torch.cuda.set_device(dev)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n, rank, world):
    """contiguous sub-range of rank `rank`: [r*n//W, (r+1)*n//W)"""
    return (rank * n) // world, ((rank + 1) * n) // world


def shard_sizes(n, world):
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def shard_bounds(n, world, weights=None):
    """W+1 boundaries of the contiguous shards of n reads. Without weights: r*n//W (equal read counts). With weights
    (per-read work, e.g. min(len, max_len) summed over the mates): equal total work per rank, which is what balances the
    ranks on variable-length input. Deterministic in (n, world, weights), so every rank computes
    the same split from its own copy of the chunk."""
    if weights is None or n == 0:
        return [(r * n) // world for r in range(world + 1)]
    import numpy as np
    cs = np.cumsum(np.asarray(weights, dtype=np.int64))
    total = int(cs[-1])
    b = [0] + [int(np.searchsorted(cs, (r * total) // world, side="right")) for r in range(1, world)] + [n]
    for r in range(1, world + 1):
        b[r] = max(b[r], b[r - 1])
    return b


def gather_labels(local_labels, n_total, dst=0, group=None, async_op=False, out=None, bounds=None):
    """Gather the per-rank label vectors (int8/uint8, 1 B per read or pair) to rank `dst`, in input order.

    Every rank passes its shard's labels (length shard_range(n_total, rank, world), or bounds[rank+1]-bounds[rank] when the
    shard_bounds() of a weighted split are given). Returns the [n_total] tensor on `dst` (None elsewhere); with
    async_op=True returns (tensor_or_None, finish) where finish() waits and trims."""
    if not active(group):
        return (local_labels, (lambda: local_labels)) if async_op else local_labels
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if local_labels.is_cuda and dist.get_backend(group) == "gloo":
        local_labels, out = local_labels.cpu(), None
    sizes = shard_sizes(n_total, world) if bounds is None else [bounds[r + 1] - bounds[r] for r in range(world)]
    assert sum(sizes) == n_total
    assert local_labels.numel() == sizes[rank], (local_labels.numel(), sizes[rank])
    mx = max(sizes)
    send = local_labels
    if send.numel() != mx:                       # pad to the common size (1 B per read: the padding is noise)
        send = torch.zeros(mx, dtype=local_labels.dtype, device=local_labels.device)
        send[: local_labels.numel()] = local_labels
    recv = None
    if rank == dst:
        buf = out if out is not None and out.numel() == mx * world else torch.empty(mx * world, dtype=send.dtype, device=send.device)
        recv = list(buf.view(world, mx).unbind(0))
    work = dist.gather(send, recv, dst=dst, group=group, async_op=True)

    def finish():
        work.wait()
        if rank != dst:
            return None
        if all(s == mx for s in sizes):
            return buf
        return torch.cat([recv[r][: sizes[r]] for r in range(world)])
    if async_op:
        return (None if rank != dst else buf), finish
    return finish()


def gather_var_bytes(buf, nbytes, sizes, dst=0, group=None):
    """Gather byte strings of different lengths (the gzip members every rank made of its shard of a chunk, ribodetector_amd/gz.py) to
    rank `dst`: `buf` uint8 tensor (device or host) whose first `nbytes` bytes are this rank's, `sizes` the byte counts of all ranks
    (every rank knows them: all_gather_sizes). Returns on `dst` a list of W uint8 tensors (views of one receive buffer: device
    memory under nccl, host memory under gloo), None elsewhere. One collective, padded to the largest string (compressed data:
    tens of MB per chunk at most)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [int(x) for x in sizes]
    assert len(sizes) == world and sizes[rank] == int(nbytes)
    mx = max(sizes)
    if mx == 0:
        return [torch.empty(0, dtype=torch.uint8) for _ in range(world)] if rank == dst else None
    on_host = dist.get_backend(group) == "gloo"
    dev = "cpu" if on_host else buf.device
    send = torch.zeros(mx, dtype=torch.uint8, device=dev)
    if nbytes:
        send[:nbytes] = buf[:nbytes].cpu() if on_host else buf[:nbytes]
    recv = None
    if rank == dst:
        big = torch.empty(mx * world, dtype=torch.uint8, device=dev)
        recv = list(big.view(world, mx).unbind(0))
    dist.gather(send, recv, dst=dst, group=group)
    return [recv[r][: sizes[r]] for r in range(world)] if rank == dst else None


def all_gather_sizes(values, group=None):
    """every rank's list of integers -> int64 tensor [W, len(values)] on every rank (host memory)"""
    world = dist.get_world_size(group)
    on_host = dist.get_backend(group) == "gloo"
    t = torch.tensor(list(values), dtype=torch.int64)
    if not on_host:
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return torch.stack(out).cpu()


def reduce_counts(counts, group=None):
    """all-reduce(SUM) of the int64[3] counters; every rank gets the totals."""
    if active(group):
        if counts.is_cuda and dist.get_backend(group) == "gloo":
            host = counts.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            counts.copy_(host)
        else:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


def shift_to_prev(buf, group=None):
    """every rank r > 0 sends `buf` (uint8 tensor, device or host; None = nothing) to rank r - 1; returns what rank r + 1 sent (uint8
    tensor - device memory under nccl, host memory under gloo - or None). The mate records in front of a rank's common cut travel this
    way to the rank before (data_loader/gz_shard.py): one all-gather of the sizes, then one send / receive pair per boundary."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = int(buf.numel()) if buf is not None else 0
    sizes = all_gather_sizes([n], group=group)[:, 0].tolist()
    on_host = dist.get_backend(group) == "gloo"
    ops, recv, keep = [], None, None
    if rank > 0 and n > 0:
        keep = buf.cpu().contiguous() if on_host else buf.contiguous()
        ops.append(dist.P2POp(dist.isend, keep, rank - 1, group))
    if rank + 1 < world and sizes[rank + 1] > 0:
        recv = torch.empty(int(sizes[rank + 1]), dtype=torch.uint8, device="cpu" if on_host else torch.device("cuda", torch.cuda.current_device()))
        ops.append(dist.P2POp(dist.irecv, recv, rank + 1, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return recv


def _cpulist(text):
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_node(device_index):
    """NUMA node of a visible GPU (its PCI function's sysfs entry), or None when the platform does not say (-1: one node, a VM)"""
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as fh:
            n = int(fh.read().strip())
        return n if n >= 0 else None
    except Exception:      # noqa: BLE001 - (no such attribute / file: nothing to go by)
        return None


def pin_rank_cpus(device_index, local_rank, local_world, mode=None):
    """Keep a rank's host threads near its GPU and out of the other ranks' way: the usable CPUs (sched_getaffinity) of the GPU's NUMA
    node - all usable CPUs when the platform names no node - are dealt out in equal contiguous shares to the ranks whose GPUs sit on
    that node, and every thread of this process (the runtime's helper threads exist already; the readers, writers and feeders come
    later and inherit) is bound to this rank's share. RD_PIN=node binds to the whole node instead, RD_PIN=0 leaves the scheduler alone.
    Nothing in the reference to match (its DataParallel wrap is dead code, detect.py:95-96). Returns (sorted CPU list, what was done)."""
    mode = (mode or os.environ.get("RD_PIN", "share")).lower()
    if mode in ("0", "off", "no") or local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None, "off"
    usable = set(os.sched_getaffinity(0))
    node = gpu_numa_node(device_index)
    cpus, peers = usable, list(range(local_world))
    if node is not None:
        try:
            with open("/sys/devices/system/node/node%d/cpulist" % node) as fh:
                on_node = _cpulist(fh.read()) & usable
            if on_node:
                cpus = on_node
                nodes = [gpu_numa_node(i) if i < torch.cuda.device_count() else node for i in range(local_world)]
                same = [i for i in range(local_world) if nodes[i] == node]
                if local_rank in same:
                    peers = same
        except OSError:
            pass
    cpus = sorted(cpus)
    what = "node %s" % node if node is not None else "all usable CPUs"
    if mode != "node":
        k, m = peers.index(local_rank) if local_rank in peers else 0, max(1, len(peers))
        share = cpus[k * len(cpus) // m:(k + 1) * len(cpus) // m]
        if share:
            cpus, what = share, "share %d of %d of %s" % (k + 1, m, what)
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), cpus)
            except OSError:
                pass
        os.sched_setaffinity(0, cpus)
    except OSError as e:
        return None, "failed: %s" % e
    return cpus, what
