#!/usr/bin/env python3
"""`ribodetector` command line on MI355X - same flags, config.json lookup, log lines, output files and label rules as
the reference's GPU product (reference detect.py:34-809), with the per-batch host work (one-hot + pack_sequence in
DataLoader workers, detect.py:666-726) replaced by raw bytes -> HBM -> fused HIP kernels.

Flow per chunk (reference run_with_chunks, detect.py:326-523):
    FASTQ/FASTA chunk as one byte arena + offsets (data_loader/fastx_parser.py)
    -> pinned host buffer -> H2D on a copy stream (double buffered: chunk k+1 is parsed/copied while chunk k computes)
    -> rd_classify (R1 [, R2]) -> rd_pair_fuse / argmax -> labels D2H (1 B/read)
    -> records written by label in input order.
Under torchrun (WORLD_SIZE > 1), one process per GPU:
  * plain input files: every rank parses only its own byte range (record-aligned, mates cut at the same record index:
    data_loader/fastx_parser.plan_ranges), classifies it, writes its own part of every output file; the counters are
    all-reduced (RCCL) and every rank copies its part to its offset in the final file (rank order = input order);
  * gzip input (one DEFLATE stream, not splittable): every rank parses the stream, classifies a contiguous shard of each
    chunk, and rank 0 gathers the 1-byte labels over RCCL and writes (ribodetector_amd/dist.py).
"""
import argparse
import math
import os
import threading
import time
import queue
from argparse import RawTextHelpFormatter

import numpy as np
import torch

from . import __version__
from . import _native as _native_mod
from . import dist as rdist
from .data_loader import device_reader as dr
from .data_loader import fastx_parser as fx
from .model import model as module_arch
from .parse_config import ConfigParser

cd = os.path.dirname(os.path.abspath(__file__))
DEFAULT_CHUNK_READS = 1 << 20      # records per chunk when --chunk_size is not given (reference: whole file in RAM)


class colors:
    HEADER = '\033[95m'
    OKBLUE = '\033[94m'
    OKCYAN = '\033[96m'
    OKGREEN = '\033[92m'
    OKYELLOW = '\033[33m'
    WARNING = '\033[93m'
    FAIL = '\033[91m'
    ENDC = '\033[0m'
    BOLD = '\033[1m'


def part_path(path, rank):
    """name of rank `rank`'s part of an output file; keeps a trailing 'gz' because the writer compresses by name
    (reference detect.py:738: read_file.endswith('gz'))"""
    if path.endswith('gz'):
        return '%s.part%d.gz' % (path[:-2].rstrip('.'), rank)
    return '%s.part%d' % (path, rank)


class Predictor:
    """Main class of predictor for rRNA, non-rRNA sequences (interface of reference detect.py:34-43)."""

    GZ_RING = 6          # device gzip: sets of output buffers in flight (submitted x2, queued x2, being written, + 1)

    def __init__(self, config, args, log_level=None):
        self.config = config
        self.args = args
        # under torchrun only rank 0 owns the log file (the other ranks log to the console only)
        self.logger = config.get_logger('predict', 1, self.args.log if int(os.environ.get('RANK', '0')) == 0 else None)
        log_level = log_level or os.environ.get('RD_LOG_LEVEL')     # (benchmarks call main() with "WARNING": no per-chunk lines)
        import logging
        self.logger.setLevel(getattr(logging, str(log_level).upper()) if log_level else logging.NOTSET)
        self.chunk_size = self.args.chunk_size
        self.thread_cpu_s = {}
        self.rank, self.world, self.local_rank = 0, 1, 0
        self.multi = False                       # collectives in use: several ranks (or one rank under RD_FORCE_DIST=1, dist.py)
        self._arenas = []                        # shared-memory chunk arenas of this rank (rank 0, gzip input, several ranks)
        self._part_files = []                    # this rank's part files of a sharded-parse run (removed if the run fails)
        self.sharded_parse = False               # several ranks, plain input: every rank parses its own byte range
        self._shared = None                      # gzip input, several ranks of one node: one decode per node (decided once per run)
        self._chunk_reads = None
        self._use_device = None                  # does the text of the inputs stay on the device? decided once per run
        self._install_cleanup()

    # ---- cleanup ------------------------------------------------------------------------------------
    def cleanup(self):
        """what a run must not leave behind, whichever way it ends: the shared-memory chunk slots (they are RAM) and this rank's
        '<out>.partN' / '<out>.joining' files"""
        self._close_arenas()
        for f in self._part_files:
            try:
                os.remove(f)
            except OSError:
                pass
        self._part_files = []

    def _install_cleanup(self):
        """atexit + SIGTERM: when ANOTHER rank fails, torch.distributed.run sends this one SIGTERM - the default action would end the
        process without running any Python, and rank 0's slot files would stay in /dev/shm until the node reboots"""
        import atexit
        import signal
        import weakref
        ref = weakref.ref(self)

        def run():
            me = ref()
            if me is not None:
                me.cleanup()
        atexit.register(run)
        if threading.current_thread() is threading.main_thread():
            prev = signal.getsignal(signal.SIGTERM)

            def on_term(signum, frame):
                run()
                signal.signal(signal.SIGTERM, prev if callable(prev) or prev in (signal.SIG_DFL, signal.SIG_IGN) else signal.SIG_DFL)
                os.kill(os.getpid(), signal.SIGTERM)
            try:
                signal.signal(signal.SIGTERM, on_term)
            except (ValueError, OSError):
                pass

    # ---- model -------------------------------------------------------------------------------------
    def get_state_dict(self):
        """'recall' weights iff --ensure norrna, else 'mcc' (reference detect.py:45-82)."""
        self.len = self.args.len
        if self.len < 40:
            self.logger.info('The accuracy will drop with reads shorter than 40.')
        model_file_ext = 'recall' if self.args.ensure == 'norrna' else 'mcc'
        self.state_key = model_file_ext
        self.state_file = self.config.state_file(model_file_ext)
        self.logger.info('Using high {} model'.format(model_file_ext.upper()))
        self.logger.info('Log file: {}'.format(self.args.log))

    def kernel_config(self):
        """the `kernel` block of config.json (specific to this build), validated before anything touches the GPU"""
        kcfg = dict(self.config.config.get('kernel', {}))
        variant = kcfg.get('variant', 'auto')
        if variant not in ('auto', 'mfma_f32', 'simple', 'mfma_f16x3_t32'):
            raise RuntimeError("config.json kernel.variant must be one of auto, mfma_f16x3_t32, mfma_f32, simple; got %r" % (variant,))
        sem = getattr(self.args, 'semantics', None) or kcfg.get('semantics', 'gpu')
        if sem not in ('gpu', 'cpu', 'packed', 'padded'):
            raise RuntimeError("config.json kernel.semantics must be gpu or cpu; got %r" % (sem,))
        refine = float(kcfg.get('refine', module_arch.SeqModel.REFINE_DEFAULT))
        if not 0.0 <= refine <= 1.0:
            raise RuntimeError("config.json kernel.refine must be in [0, 1]; got %r" % (refine,))
        pk = kcfg.get('prefix_k', None)           # prefix-state table: absent = the model's default (RD_PREFIX_K or "auto")
        if pk is not None and pk != 'auto' and not (isinstance(pk, int) and (pk == 0 or 4 <= pk <= 13)):
            raise RuntimeError("config.json kernel.prefix_k must be \"auto\", 0 or an integer in [4, 13]; got %r" % (pk,))
        gz = os.environ.get("RD_DEVICE_GZIP") or kcfg.get("gzip", "device")     # who deflates .gz outputs: the GPU (BGZF members) or the host
        gz = {"1": "device", "0": "host"}.get(gz, gz)
        if gz not in ("device", "host"):
            raise RuntimeError("config.json kernel.gzip must be \"device\" or \"host\"; got %r" % (gz,))
        return {"variant": variant, "semantics": sem, "refine": refine, "prefix_k": pk, "gzip": gz}

    def prefix_k_for_input(self):
        """k of the prefix-state table that pays off for THIS run: a row saves k steps per read, level k costs 4^k one-step
        workgroup slots to build (measured on MI355X, tools/time_setup.py: k = 8 / 10 / 11 / 12 take 0.4 / 1.7 / 6.3 / 22 ms with
        their allocation; one read-step is worth 0.3 ns), so k+1 beats k from about 3 * 4^k reads on. The read count is estimated
        from the input sizes (gzip: x4.5) and -l; under torchrun every rank sees its share."""
        from .data_loader import fastx_parser as fx
        n = 0.0
        for path in self.args.input or []:
            try:
                size, gz = fx.file_info(path)
                fa = fx.get_seq_format(path).startswith("fa")
            except Exception:      # the run itself reports unreadable inputs
                continue
            n += size * (4.5 if gz else 1.0) / ((1 if fa else 2) * max(self.args.len, 30) + 30)
        n /= max(1, int(os.environ.get("WORLD_SIZE", "1")))
        return 8 if n < 2e6 else 10 if n < 14e6 else 11 if n < 49e6 else 12

    def load_model(self):
        """Load the model onto the GPU (reference detect.py:84-119). Raises RuntimeError without a visible device."""
        kcfg = self.kernel_config()
        if self.args.deviceid is not None:
            os.environ["HIP_VISIBLE_DEVICES"] = self.args.deviceid
            os.environ["CUDA_VISIBLE_DEVICES"] = self.args.deviceid
        self.rank, self.world, self.local_rank = rdist.init_from_env()
        self.multi = rdist.active()
        self.get_state_dict()
        model = self.config.init_obj('arch', module_arch)
        if not torch.cuda.is_available():
            self.logger.error('{}No visible GPU devices!{} This build runs the HIP kernels only; the CPU product of the '
                              'reference is ribodetector_cpu'.format(colors.FAIL, colors.ENDC))
            raise RuntimeError("Set HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES or use CPU inference.")
        if self.multi:                           # one process per GPU; RD_LOCAL_DEVICE pins a rank elsewhere (ranks sharing a GPU)
            self.device = torch.device('cuda', int(os.environ.get('RD_LOCAL_DEVICE', self.local_rank)))
            torch.cuda.set_device(self.device)
        else:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.has_cuda = True
        if self.multi and self.world > 1:        # (round 6) a rank's threads stay on its share of the CPUs next to its GPU; RD_PIN=0: off
            lw = int(os.environ.get("LOCAL_WORLD_SIZE", self.world))
            self.pinned_cpus, how = rdist.pin_rank_cpus(self.device.index, self.local_rank, lw)
            if self.pinned_cpus is not None:
                self.logger.info('Rank {} runs on CPUs {} ({})'.format(self.rank, ",".join(str(c) for c in self.pinned_cpus), how))
        model.load_state_dict(self.config.load_state_dict(self.state_key))
        self.logger.info('Model using {} for read length {}{}{}{} loaded'.format(
            self.device, colors.BOLD, colors.OKCYAN, self.len, colors.ENDC))
        if kcfg["prefix_k"] is not None:
            model.set_prefix_table(kcfg["prefix_k"])     # recorded now, built by .to()
        elif "RD_PREFIX_K" not in os.environ:
            model.set_prefix_table("auto", cap=self.prefix_k_for_input())
        self.model = model.to(self.device)
        self.model.set_variant(kcfg["variant"])
        self.model.set_semantics(kcfg["semantics"])
        # margin band of the float64 re-evaluation (config.json kernel.refine; 0 = off; default 2.5e-4), as the deferred pass of the
        # C ABI (rd_set_refine_async): the recurrence kernel's epilogue records the reads inside the band, and submit_chunk's
        # post-pass evaluates them (rd_sync_results on the post stream) beside the next chunk's recurrences.
        self.refine_band = kcfg["refine"]
        self.gzip_on_device = kcfg["gzip"] == "device"
        self.model.set_refine(self.refine_band)
        self.model.set_refine_async(16 if self.refine_band > 0 else 0)
        self.model.eval()

    # ---- classification of one chunk ------------------------------------------------------------------
    def _to_device(self, chunk, lo, hi, stream):
        """async H2D of records [lo, hi) of a chunk: only the bytes spanning those reads travel. The native reader parsed
        straight into pinned buffers, so there is no staging copy."""
        b0 = int(chunk.rec_start[lo])
        b1 = int(chunk.rec_start[hi])
        tbuf, toff, tlen = chunk.tensors[:3]
        with torch.cuda.stream(stream):
            arena = tbuf[b0:b1].to(self.device, non_blocking=True) if b1 > b0 else torch.zeros(1, dtype=torch.uint8, device=self.device)
            off = toff[lo:hi].to(self.device, non_blocking=True) - b0
            ln = tlen[lo:hi].to(self.device, non_blocking=True)
        return arena, off, ln

    def submit_chunk(self, chunks):
        """Enqueue one chunk: H2D on the copy stream, kernels on the compute stream, D2H of the labels into pinned memory.
        Nothing here waits for the GPU, so the next chunk's H2D overlaps this chunk's kernels. Returns a ticket for
        collect_chunk()."""
        n = len(chunks[0].seq_len)
        bounds = None
        if self.multi and not self.sharded_parse:   # equal bases (= recurrence steps) per rank, not equal read counts
            work = sum(np.minimum(np.asarray(c.seq_len, dtype=np.int64), self.len) for c in chunks)
            bounds = rdist.shard_bounds(n, self.world, work)
        lo, hi = (0, n) if bounds is None else (bounds[self.rank], bounds[self.rank + 1])
        cs = self._copy_stream
        cur = torch.cuda.current_stream(self.device)
        on_dev = isinstance(chunks[0], dr.DeviceChunk)      # text and index already in HBM (data_loader/device_reader.py): nothing to copy
        if on_dev:
            dev_in = [c.dev[:3] for c in chunks]
            for c in chunks:
                cur.wait_event(c.ready)
        else:
            dev_in = [self._to_device(c, lo, hi, cs) for c in chunks]
            cur.wait_stream(cs)
        # (dev_in stays referenced by the ticket: its tensors were allocated on the copy stream and must not return to that
        # stream's pool while other streams read them - and the deferred float64 pass reads the bases until the post-pass has run)
        outs = [self.model.classify_bytes(a, o, l, self.len, want_labels=not self.is_paired) for a, o, l in dev_in]
        main_done = torch.cuda.Event()
        main_done.record(cur)
        # post-pass on its own stream: it overlaps the recurrences of the next chunk (the float64 refine pass has the latency of
        # one read with most of the GPU idle)
        post = self._post_stream
        with torch.cuda.stream(post):
            post.wait_event(main_done)
            self.model.sync_results()                # the reads inside the noise band, in float64 (current stream = post)
            if self.is_paired and self.args.ensure == 'none' and self.refine_band > 0:
                # pair label = argmax of the SUMMED logits (reference detect.py:657): also the reads whose PAIR margin is inside the
                # band - only both mates' logits together say which, so this mode keeps the scan form of the pass (rd_refine)
                for k, (a, o, l) in enumerate(dev_in):
                    self.model.refine(a, o, l, self.len, outs[k][0], outs[k][1], outs[1 - k][0], thresh=self.refine_band)
            if self.is_paired:
                labels = module_arch.pair_fuse(outs[0][0], outs[1][0], self.args.ensure)
            else:
                labels = outs[0][1].view(torch.int8)
            host = finish = None
            gzparts = {}
            if not self.multi or self.sharded_parse:
                host = torch.empty(labels.shape, dtype=torch.int8, pin_memory=True)
                host.copy_(labels, non_blocking=True)
            if on_dev:
                # a chunk whose text lives on the device: every output file's records are selected there - deflated (.gz outputs, as
                # below) or packed into one contiguous text (plain outputs, rd_select_pack) - and only those bytes travel to the host
                ring = self._gz_seq % self.GZ_RING
                self._gz_seq += 1
                lab8 = labels.view(torch.int8)
                for e, lab in self._out_files:
                    text, rs = chunks[e].dev[0], chunks[e].dev[3]
                    if (e, lab) in self._gz_files:
                        out, info = self._gz.compress_selected(text, rs, lab8, lab, slot=(e, lab, ring))
                    else:
                        out, info = self._sel.pack_selected(text, rs, lab8, lab, slot=(e, lab, ring))
                    ih = torch.empty(4, dtype=torch.int64, pin_memory=True)
                    ih.copy_(info, non_blocking=True)
                    gzparts[(e, lab)] = [(out, ih)]
            # .gz outputs: the records of every label file of this chunk - under the label gather: of this rank's shard of it - are
            # deflated here, where the text already is (BGZF members, csrc/rd_deflate.hpp), beside the next chunk's recurrences; the
            # writer threads fetch the compressed bytes and append them (reference: gzip.open(..., compresslevel=5) on the host,
            # detect.py:729-741). Every rank takes the same decision (the chunks of a shared decode are the same chunks).
            if not on_dev and self._gz_files and all(c.tensors is not None and len(c.tensors) > 3 and c.verbatim for c in chunks):
                ring = self._gz_seq % self.GZ_RING
                self._gz_seq += 1
                for e, lab in self._gz_files:
                    c = chunks[e]
                    b0 = int(c.rec_start[lo])
                    rs = c.tensors[3][lo:hi + 1].to(self.device, non_blocking=True) - b0
                    # (own writer: half of the worst-case output size is reserved - an incompressible chunk falls back to the host's
                    # deflate; under the label gather the full bound, every rank must take the same path)
                    out, info = self._gz.compress_selected(dev_in[e][0], rs, labels.view(torch.int8), lab, slot=(e, lab, ring),
                                                           out_frac=1.0 if (self.multi and not self.sharded_parse) else 0.5)
                    ih = torch.empty(4, dtype=torch.int64, pin_memory=True)
                    ih.copy_(info, non_blocking=True)
                    gzparts[(e, lab)] = [(out, ih)]
            if self.multi and not self.sharded_parse:   # label gather (1 B per read) queued behind the kernels, collected later
                _, finish = rdist.gather_labels(labels, n, dst=0, bounds=bounds, async_op=True)
            done = _native_mod.new_event()
            done.record(post)
        return {"n": n, "bounds": bounds, "labels": labels, "host": host, "finish": finish, "done": done, "keep": (dev_in, outs),
                "gz": gzparts, "totals": [c.total for c in chunks] if on_dev else ()}

    def collect_chunk(self, tk):
        """Labels of a submitted chunk: int8 numpy on rank 0 (whole chunk, input order), None elsewhere."""
        if self.multi and not self.sharded_parse:
            labels = tk["finish"]()
            if tk.get("gz"):
                # the members every rank made of its shard travel to rank 0, which appends them in rank order = input order (sizes
                # first: one small all-gather per chunk; then one padded gather per output file)
                _native_mod.wait_event(tk["done"])
                mine = [int(tk["gz"][key][0][1][0]) for key in self._gz_files]
                sizes = rdist.all_gather_sizes(mine)
                gathered = {}
                for f, key in enumerate(self._gz_files):
                    out, _ = tk["gz"][key][0]
                    if mine[f] > out.numel():
                        raise RuntimeError("device gzip: output buffer too small (%d > %d)" % (mine[f], out.numel()))
                    parts = rdist.gather_var_bytes(out, mine[f], sizes[:, f].tolist(), dst=0)
                    if self.rank == 0:
                        gathered[key] = [(t, None) for t in parts]
                tk["gz"] = gathered
            return None if self.rank != 0 else labels.cpu().numpy()
        _native_mod.wait_event(tk["done"])         # (a blocking event: the thread sleeps in the driver, it does not spin a host core)
        for t in tk.get("totals", ()):
            if int(t[0]) < 0:
                raise RuntimeError("device chunk assembly failed (rd_fastq_gather)")
        return tk["host"].numpy()

    def classify_chunk(self, chunks):
        """chunks: (c1,) or (c1, c2). Returns the int8 labels of the whole chunk on rank 0 (numpy), None elsewhere."""
        return self.collect_chunk(self.submit_chunk(chunks))

    # ---- drivers ----------------------------------------------------------------------------------------
    # Host pipeline: one parser thread per input file -> GPU (main thread) -> one writer thread per mate. The C++ reader,
    # the kernels and the C++ writer all release the GIL, so the three stages overlap (the reference parses the two mates
    # with Pool(2), detect.py:131-132, but encodes, classifies and writes in lock-step).
    @staticmethod
    def _spawn(target, *a):
        th = threading.Thread(target=target, args=a, daemon=True)
        th.start()
        return th

    def _reader_queue(self, path, chunk_reads, depth=2, byte_range=None, arena=None, schedule=None):
        q = queue.Queue(maxsize=depth)

        def work():
            try:
                self._timeline.append(("reader_thread_%s" % os.path.basename(str(path)), round(time.perf_counter() - self._t_run, 4)))
                # FASTQ whose text can stay on the device (plain files, BGZF): H2D of the file's bytes, members inflated and records
                # framed there - no parser thread at all (data_loader/device_reader.py; RD_DEVICE_PARSE=0 keeps the host parser)
                if arena is None and self._device_parse(path):
                    st = self.ingest.setdefault(os.path.basename(str(path)), {"path": "device"})
                    stream = dr.get_seq_chunks_device(path, chunk_size=chunk_reads, byte_range=byte_range, first_chunk=1 << 17, schedule=schedule,
                                                      device=self.device, stats=st)
                # one plain input file: its parser thread was the slowest stage of the pipeline - two readers over byte segments,
                # small first chunks (mate files keep one reader each and exact chunk sizes: their chunks must pair up)
                elif arena is None and len(self.input) == 1 and not fx.file_info(path)[1] and int(self.args.threads) >= 4:
                    stream = fx.get_seq_chunks_parallel(path, chunk_size=chunk_reads, byte_range=byte_range, workers=2)
                else:
                    stream = fx.get_seq_chunks(path, chunk_size=chunk_reads, byte_range=byte_range, first_chunk=1 << 17, arena=arena,
                                               schedule=schedule, device=self.device)
                for c in stream:
                    q.put(c)
                q.put(None)
            except BaseException as e:      # surface parser errors on the main thread
                q.put(e)
            finally:
                self.thread_cpu_s["reader:" + os.path.basename(str(path))] = round(time.thread_time(), 4)
        self._spawn(work)
        return q

    def _device_parse(self, path):
        """does this input's text stay on the device? FASTQ, plain or BGZF, when every record a rank reads is a record it classifies
        (one rank, or the sharded parse) - under the label gather the chunk's lengths are needed on the host for the shard bounds"""
        if self._use_device is None:      # ONE decision for the run: the mates' chunks must be of one kind (submit_chunk looks at the first)
            self._use_device = (not self.multi or self.sharded_parse) and all(dr.device_parse_wanted(p) for p in self.input)
        return self._use_device

    def _shared_decode(self):
        """several ranks of ONE node on gzip input: rank 0 inflates and parses the stream once into shared memory (fx.ShmArena)
        and tells the others where each chunk lies; they map it and take their share of the records. (Ranks spread over several
        nodes cannot share memory: there every rank decodes the stream itself, as in round 2.)"""
        if self._shared is None:
            import torch.distributed as dist
            # one node only - and only when the launcher SAYS so (torchrun sets LOCAL_WORLD_SIZE; a launcher that sets just
            # RANK / WORLD_SIZE may have spread the ranks over several hosts, whose /dev/shm are different memories)
            ok = (self.multi and not self.sharded_parse and self.world > 1 and os.environ.get("RD_SHARED_DECODE", "1") != "0" and
                  os.environ.get("LOCAL_WORLD_SIZE") is not None and int(os.environ["LOCAL_WORLD_SIZE"]) == self.world)
            if ok:                                   # rank 0 owns the slots: its /dev/shm must hold them (all ranks take its answer)
                msg = [None]
                if self.rank == 0:
                    fx.ShmArena.sweep_stale()
                    chunk = self._chunk_reads or DEFAULT_CHUNK_READS
                    fits, need = fx.ShmArena.fits(len(self.input), chunk, 2 * max(self.len, 50) + 80)
                    msg = [bool(fits)]
                    if not fits:
                        self.logger.info('Shared gzip decode needs about {} MB of /dev/shm, which is not free: every rank decodes '
                                         'the input itself'.format(need >> 20))
                dist.broadcast_object_list(msg, src=0)
                ok = bool(msg[0])
            self._shared = ok
        return self._shared

    def _chunk_stream(self, chunk_reads):
        import torch.distributed as dist
        from . import _native
        shared = self._shared_decode()
        if shared and self.rank != 0:                # chunks arrive as descriptions of rank 0's shared-memory slots
            while True:
                msg = [None]
                t0 = time.perf_counter()
                dist.broadcast_object_list(msg, src=0)
                self._stage_s["wait_reader"] += time.perf_counter() - t0
                if msg[0] is None:
                    return
                if isinstance(msg[0], str):
                    raise RuntimeError("rank 0: " + msg[0])
                yield tuple(fx.ShmArena.attach(d) for d in msg[0])
        ranges = self._ranges if self.sharded_parse else [None] * len(self.input)
        # -t/--threads also bounds the decoder threads of .gz inputs (parallel DEFLATE decoding, csrc/rd_pgzip.h): what is left after
        # the parser threads and this one, divided among the input files - and among the ranks when every rank decodes for itself
        sharers = len(self.input) * (1 if shared else self.world)
        _native.host_lib().rd_host_set_gz_threads(max(2, min(12, (int(self.args.threads) - 2) // max(1, sharers))))
        arenas = [None] * len(self.input)
        if shared:
            tag = "rd_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getpid())
            arenas = [fx.ShmArena("%s_f%d" % (tag, i)) for i in range(len(self.input))]
            self._arenas += arenas
        schedule = None
        if len(self.input) == 2 and not shared and not any(fx.file_info(p)[1] for p in self.input):
            # plain mate files: one schedule of chunk sizes for both readers (small first AND last chunks), from the first file
            schedule = fx.chunk_schedule(self.input[0], chunk_reads, byte_range=ranges[0]) or None
        qs = [self._reader_queue(p, chunk_reads, byte_range=r, arena=a, schedule=schedule) for p, r, a in zip(self.input, ranges, arenas)]
        while True:
            cs = []
            err = None
            for q in qs:
                t0 = time.perf_counter()
                c = q.get()
                self._stage_s["wait_reader"] += time.perf_counter() - t0
                if isinstance(c, BaseException):
                    err = c
                    break
                cs.append(c)
            if err is None and not all(c is None for c in cs) and (any(c is None for c in cs) or len({len(c.seq_len) for c in cs}) != 1):
                err = ValueError("paired-end files have different numbers of records")
            if shared:                                # (an error is passed on, so that no rank waits for a chunk that never comes)
                dist.broadcast_object_list([str(err) if err is not None else None if cs[0] is None else [c.shm for c in cs]], src=0)
            if err is not None:
                raise err
            if all(c is None for c in cs):
                return
            yield tuple(cs)

    def run_with_chunks(self, chunk_reads=None):
        """Classify the input in chunks and write the outputs (reference detect.py:326-523)."""
        if chunk_reads is None:
            chunk_reads = self.batch_size * self.chunk_size
        self._chunk_reads, self._shared, self._use_device = chunk_reads, None, None
        self._timeline, self._t_run = [], time.perf_counter()
        # plain inputs under several ranks: every rank parses, classifies and writes its own byte range (no label exchange)
        plain = not any(fx.file_info(p)[1] for p in self.input)
        # ... and BGZF FASTQ inputs likewise: their members are independent, so every rank inflates (on its own GPU), parses, classifies
        # and writes the members of its share (positions in the decompressed stream, fx.BgzfView). RD_BGZF_SHARD=0: one decoding rank
        bgzf = (self.multi and not plain and os.environ.get("RD_BGZF_SHARD", "1") != "0"
                and all(fx.get_seq_format(p) in ("fqgz", "fagz") and fx.bgzf_all_the_way(p) and fx.device_inflate_wanted(p) for p in self.input))
        views = None
        if bgzf:                    # the member index: rank 0 walks the headers, the others receive the three arrays per file
            import torch.distributed as dist
            idx = [None]
            if self.rank == 0:
                try:
                    idx = [[fx.BgzfView.build_index(p) for p in self.input]]
                except ValueError as e:     # not BGZF all the way (a plain member in the middle): the one-decoder path reads such files
                    idx = [str(e)]
            dist.broadcast_object_list(idx, src=0)
            if isinstance(idx[0], str):
                if self.rank == 0:
                    self.logger.info('{}: one rank decodes'.format(idx[0]))
                bgzf = False
            else:
                views = [fx.BgzfView(p, index=i) for p, i in zip(self.input, idx[0])]
        def all_gather(obj):
            import torch.distributed as dist
            out = [None] * self.world
            dist.all_gather_object(out, obj)
            return out
        # ... and single-stream .gz inputs (what sequencers write; round 6): every rank decodes its own compressed range on its own GPU -
        # symbols first, bytes once the ranks have exchanged the 64 KiB maps of their ranges (data_loader/gz_shard.py). What the device
        # decoder does not take (or RD_GZ_SHARD=0) stays with the one-decode path below.
        gzr = None
        if (self.multi and not plain and not bgzf and os.environ.get("RD_GZ_SHARD", "1") != "0"
                and all(dr.device_ingest_kind(p) == "stream" for p in self.input)):
            from .data_loader import gz_shard
            t0 = time.perf_counter()
            gzr, why = gz_shard.prepare(self.input, self.rank, self.world, self.device, [fx.get_seq_format(p).startswith("fa") for p in self.input],
                                        all_gather, rdist.shift_to_prev)
            if gzr is None and self.rank == 0:
                self.logger.info('{}: one rank decodes'.format(why))
            self.gz_shard_s = time.perf_counter() - t0
            if gzr is not None:                      # (every rank says so itself: the line is what shows that no rank read another's bytes)
                self.logger.info('Rank {} decoded {} compressed bytes into {} bytes of text on {} in {:.2f} s'.format(
                    self.rank, ", ".join(str(x.comp_bytes) for x in gzr), ", ".join(str(x.stats["text_bytes"]) for x in gzr), self.device, self.gz_shard_s))
        self.sharded_parse = self.multi and (plain or bgzf or gzr is not None)
        self.bytes_parsed = None
        if self.sharded_parse:
            if gzr is not None:
                self._ranges = gzr
            else:
                self._ranges = fx.plan_ranges(self.input, self.rank, self.world, all_gather, views=views)
            self.bytes_parsed = [e - b for b, e in self._ranges]
            totals = [v.size for v in views] if views else [fx.file_info(p)[0] for p in self.input]
            for r, bp in enumerate(all_gather(self.bytes_parsed)):
                if self.rank == 0:
                    self.logger.info('Rank {} parses {} bytes of {}{}'.format(
                        r, ", ".join(str(b) for b in bp), ", ".join(str(t) for t in totals),
                        " (decompressed; BGZF members)" if bgzf else " (compressed; ranges of one DEFLATE stream)" if gzr is not None else ""))
        def part(path):
            if not self.sharded_parse:
                return path
            self._part_files.append(part_path(path, self.rank))
            return self._part_files[-1]
        writer = self.rank == 0 or self.sharded_parse
        ends = (0, 1) if self.is_paired else (0,)
        fhs = {}
        from . import _native
        _native.host_lib().rd_host_set_threads(int(self.args.threads))   # -t/--threads: gzip workers; read when a writer is opened
        log = self.logger.info if self.rank == 0 else (lambda *a, **k: None)
        finals = []                                # final output paths, in the order the handles are opened
        if writer:
            if self.rrna is not None:
                log('Writing output rRNA sequences into file: {}{}{}'.format(colors.OKBLUE, ", ".join(self.rrna), colors.ENDC))
                fhs[1] = [fx.open_for_write(part(self.rrna[e])) for e in ends]
                finals += [self.rrna[e] for e in ends]
            log('Writing output non-rRNA sequences into file: {}{}{}'.format(colors.OKBLUE, ", ".join(self.output), colors.ENDC))
            fhs[0] = [fx.open_for_write(part(self.output[e])) for e in ends]
            finals += [self.output[e] for e in ends]
            if self.is_paired and self.args.ensure == 'both':
                unclf = [self.output[e] + '.unclassified.gz' for e in ends]
                fhs[-1] = [fx.open_for_write(part(u)) for u in unclf]
                finals += unclf
                log('Writing unclassified sequences into file: {}{}{}'.format(colors.OKYELLOW, ", ".join(unclf), colors.ENDC))
        if writer and self.sharded_parse:          # parts are joined below: the joined file gets ONE BGZF end-of-file block, at its end
            for handles in fhs.values():
                for fh in handles:
                    fh.set_eof_marker(False)
        num_read = num_nonrrna = num_rrna = num_unknown = 0
        self._timeline.append(("outputs_open", round(time.perf_counter() - self._t_run, 4)))
        self._stage_s = {"wait_reader": 0.0, "classify": 0.0, "wait_writer": 0.0}   # main-thread seconds per pipeline stage
        self.thread_cpu_s = {}                                                      # CPU seconds of the pipeline's Python threads, by role
        main_cpu0 = time.thread_time()
        self._first_chunk = None
        self.ingest = {}                           # per input file: which reader took it, and the device feeder's stage times
        from . import gz as _gzmod
        self._copy_stream = _gzmod.acquire_stream(self.device)      # (pooled: the allocator's cache is per stream, gz.acquire_stream)
        self._post_stream = _gzmod.acquire_stream(self.device)
        wr_streams = []
        # which (mate, label) files are gzip outputs deflated on the device: every rank deflates the records it classified - and writes
        # them itself (one rank, or the sharded parse of plain inputs) or, under the label gather, sends the members to rank 0
        self._gz_files, self._gz_seq = [], 0
        if self.gzip_on_device:                    # (the same list on every rank: it is derived from the arguments)
            from .gz import DeviceGzip
            self._gz_files = self.gz_output_files(self.output, self.rrna, self.is_paired, self.args.ensure)
            if self._gz_files:
                self._gz = DeviceGzip(self.device)
        # every (mate, label) file this run writes; for chunks on the device the plain ones are packed there (rd_select_pack)
        self._out_files = [(e, lab) for lab in fhs for e in ends]
        if any(self._device_parse(p) for p in self.input):
            from .gz import DeviceSelect
            self._sel = DeviceSelect(self.device)

        # writer threads (rank 0): one per mate, records of every label file in input order
        wq, werr, wth = [], [], []
        if writer:
            def write_end(e, q):
                # Two sets of pinned staging buffers: the bytes an item's files get from the GPU (gzip members / packed records) are
                # fetched for item k+1 while item k is being written. The fetch is a kernel on this thread's stream (C ABI
                # rd_copy_bytes), not a DMA copy - an SDMA queue is shared in order with copies that wait for kernels.
                from . import _native
                stages = [[], []]
                torch.cuda.set_device(self.device)          # (the current device is per thread)
                gz_copy = _gzmod.acquire_stream(self.device)
                wr_streams.append(gz_copy)

                def issue(item, k):
                    """queue the D2H of everything the item's files take from the GPU; returns the jobs to complete() in file order"""
                    chunk, labels, gzparts = item
                    jobs, slot = [], 0
                    for lab, handles in fhs.items():
                        part = gzparts.get((e, lab)) if gzparts else None
                        if part is None:
                            jobs.append(("selected", handles[e], lab, None, None))
                            continue
                        as_text = (e, lab) not in self._gz_files      # packed records (rd_select_pack) instead of gzip members
                        put = handles[e].write_text if as_text else handles[e].write_members
                        for out, info in part:      # made on the GPU (one piece per rank under the label gather): fetch, append
                            nb = int(info[1 if as_text else 0]) if info is not None else int(out.numel())
                            if info is not None and int(info[3]):
                                raise RuntimeError("device %s: the chunk's record table does not describe its text" % ("select" if as_text else "gzip"))
                            if nb > out.numel():        # text that does not compress into the reserved half: the host deflates this piece
                                jobs.append(("selected", handles[e], lab, None, None))
                            elif nb and not out.is_cuda:
                                jobs.append(("host", put, out, nb, None))
                            elif nb:
                                st = stages[k]
                                if slot == len(st):
                                    st.append(None)
                                if st[slot] is None or st[slot].numel() < nb:
                                    st[slot] = None
                                    st[slot] = torch.empty(max(nb, 1 << 22) * 5 // 4, dtype=torch.uint8, pin_memory=True)
                                _native.copy_bytes(st[slot], out, nb, gz_copy)
                                done = _native.new_event()
                                done.record(gz_copy)
                                jobs.append(("dev", put, st[slot], nb, done))
                                slot += 1
                    return item, jobs

                def complete(pending):
                    (chunk, labels, _), jobs = pending
                    for kind, put, src, nb, done in jobs:
                        if kind == "selected":
                            put.write_selected(chunk, labels, src)
                            continue
                        if done is not None:
                            _native.wait_event(done)    # (sleeping in the driver: stream.synchronize() would spin a core)
                        put(src.data_ptr(), nb)
                    if chunk.release is not None:   # a shared-memory slot: free for the next chunk once its text is written
                        chunk.release()
                try:
                    pending, k = None, 0
                    while True:
                        try:
                            item = q.get() if pending is None else q.get_nowait()
                        except queue.Empty:         # nothing to prefetch: write what is in hand, then wait
                            complete(pending)
                            pending = None
                            continue
                        if item is None:
                            if pending is not None:
                                complete(pending)
                            return
                        nxt = issue(item, k)
                        k ^= 1
                        if pending is not None:
                            complete(pending)
                        pending = nxt
                except BaseException as ex:
                    werr.append(ex)
                    while q.get() is not None:   # keep draining so that the producer never blocks
                        pass
                finally:
                    self.thread_cpu_s["writer:%d" % e] = round(time.thread_time(), 4)
            for e in ends:
                q = queue.Queue(maxsize=2)
                wq.append(q)
                wth.append(self._spawn(write_end, e, q))
        def in_flight(stream):
            """chunk k+1 is submitted (its H2D starts) before the labels of chunk k are waited for"""
            prev = None
            for chunks in stream:
                t0 = time.perf_counter()
                if len(self._timeline) < 30:          # (the first chunks' way through the pipeline, seconds since the run started: tools/first_chunk_probe.py)
                    self._timeline.append(("chunk_of_%d_read" % len(chunks[0].seq_len), round(t0 - self._t_run, 4)))
                tk = self.submit_chunk(chunks)
                if len(self._timeline) < 30:
                    self._timeline.append(("submitted", round(time.perf_counter() - self._t_run, 4)))
                self._stage_s["classify"] += time.perf_counter() - t0
                if prev is not None:
                    yield prev
                prev = (chunks, tk)
            if prev is not None:
                yield prev
        self._timeline.append(("writers_started", round(time.perf_counter() - self._t_run, 4)))
        try:
            for chunks, tk in in_flight(self._chunk_stream(chunk_reads)):
                t0 = time.perf_counter()
                labels = self.collect_chunk(tk)
                if len(self._timeline) < 30:
                    self._timeline.append(("labels", round(time.perf_counter() - self._t_run, 4)))
                self._stage_s["classify"] += time.perf_counter() - t0
                num_read += len(chunks[0].seq_len)
                if self._first_chunk is None:
                    self._first_chunk = (time.perf_counter(), num_read)
                if writer:
                    if werr:
                        raise werr[0]
                    num_nonrrna += int((labels == 0).sum())
                    num_rrna += int((labels == 1).sum())
                    num_unknown += int((labels == -1).sum())
                    t0 = time.perf_counter()
                    for e in ends:
                        wq[e].put((chunks[e], labels, tk.get("gz")))
                    self._stage_s["wait_writer"] += time.perf_counter() - t0
                    log('{}{}{} sequences finished!'.format(colors.OKGREEN, num_read, colors.ENDC))
        finally:
            for q in wq:
                q.put(None)
            for th in wth:
                th.join()
            for st in [self._copy_stream, self._post_stream] + wr_streams:
                try:
                    st.synchronize()
                except Exception:      # noqa: BLE001 - (a failed run: the stream is given back all the same)
                    pass
                _gzmod.release_stream(st)
        if werr:
            raise werr[0]
        self.thread_cpu_s["main"] = round(time.thread_time() - main_cpu0, 4)
        self._close_arenas()
        if writer:
            self.writer_threads = sorted({fh.threads for handles in fhs.values() for fh in handles})
            for handles in fhs.values():
                for fh in handles:
                    fh.close()
        if self.sharded_parse:                     # totals over the ranks; the parts are joined in rank order = input order
            import torch.distributed as dist
            tot = torch.tensor([num_read, num_nonrrna, num_rrna, num_unknown], dtype=torch.int64,
                               device=self.device if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)          # also the barrier: every part file is closed before the merge
            num_read, num_nonrrna, num_rrna, num_unknown = (int(x) for x in tot.cpu().tolist())
            # join the parts (rank order = input order) concurrently: every rank copies its own part to the offset that the sizes
            # of the lower ranks' parts give; gzip parts are complete members, whose concatenation is a valid gzip file
            mine = [os.path.getsize(part_path(path, self.rank)) for path in finals]
            sizes = [None] * self.world
            dist.all_gather_object(sizes, mine)
            # the join goes into '<final>.joining' and is renamed after the last barrier: a run that dies while the parts are
            # being placed never leaves a full-size, partly zero-filled file under the final name
            tmps = [path + '.joining' for path in finals]
            self._part_files += tmps if self.rank == 0 else []
            # .gz files whose members were made on the device are BGZF: one end-of-file block (an empty member) behind the last part
            from .gz import eof_block
            tails = [eof_block() if (self.gzip_on_device and path.endswith('gz')) else b'' for path in finals]
            if self.rank == 0:
                for f, tmp in enumerate(tmps):
                    with open(tmp, 'wb') as fh:
                        body = sum(sz[f] for sz in sizes)
                        fh.truncate(body + len(tails[f]))
                        if tails[f]:
                            fh.seek(body)
                            fh.write(tails[f])
            dist.barrier()
            for f, path in enumerate(finals):
                fx.place_part(tmps[f], part_path(path, self.rank), sum(sizes[r][f] for r in range(self.rank)))
            dist.barrier()
            if self.rank == 0:
                for tmp, path in zip(tmps, finals):
                    os.replace(tmp, path)
            self._part_files = []
            dist.barrier()
        if self.rank == 0:
            self.logger.info('Processed {}{}{}{} sequences in total'.format(colors.BOLD, colors.OKCYAN, num_read, colors.ENDC))
            self.logger.info('Detected {}{}{}{} non-rRNA sequences'.format(colors.BOLD, colors.OKCYAN, num_nonrrna, colors.ENDC))
            self.logger.info('Detected {}{}{}{} rRNA sequences'.format(colors.BOLD, colors.OKCYAN, num_rrna, colors.ENDC))
            if self.is_paired and self.args.ensure == 'both':
                self.logger.info('Discarded {}{}{}{} unclassified sequences'.format(
                    colors.BOLD, colors.OKCYAN, num_unknown, colors.ENDC))
        self.num_read, self.num_nonrrna, self.num_rrna, self.num_unknown = num_read, num_nonrrna, num_rrna, num_unknown

    @staticmethod
    def gz_output_files(output, rrna, is_paired, ensure):
        """(mate, label) of every output file that is written gzip-compressed - by name, like the reference's writer
        (detect.py:738: read_file.endswith('gz')); the '<out>.unclassified.gz' files of --ensure both (detect.py:390-400) always are.
        The same list on every rank (it depends on the arguments only): the files whose records are deflated on the device."""
        ends = (0, 1) if is_paired else (0,)
        files = [(e, lab) for lab, names in ((1, rrna), (0, output)) if names is not None for e in ends if names[e].endswith('gz')]
        if is_paired and ensure == 'both':
            files += [(e, -1) for e in ends]
        return files

    def _close_arenas(self):
        for a in self._arenas:
            a.close()
        self._arenas = []

    def run(self):
        """Whole-file mode of the reference (detect.py:121-324): same outputs; streamed here in 1 Mi-record chunks
        instead of holding the parsed file in host RAM."""
        self.run_with_chunks(chunk_reads=DEFAULT_CHUNK_READS)

    def detect(self):
        """Argument checks, batch-size rule, dispatch (reference detect.py:525-584)."""
        self.input = self.args.input
        self.output = self.args.output
        self.rrna = self.args.rrna
        self.pack_seq = self.config['arch']['args']['pack_seq']
        num_inputs = len(self.input)
        num_rrna_outputs = None if self.rrna is None else len(self.rrna)
        if num_inputs != len(self.output) or num_inputs > 2 or num_inputs < 1:
            self.logger.error('{}The number of input and output sequence files is invalid!{}'.format(colors.FAIL, colors.ENDC))
            raise RuntimeError(
                "Input or output should have no more than two files and they should have the same number of files.")
        if num_rrna_outputs is not None and num_rrna_outputs != num_inputs:
            self.logger.error('{}The number of output rRNA sequence files is invalid!{}'.format(colors.FAIL, colors.ENDC))
            raise RuntimeError(
                "Ouput rRNA should have no more than two files and they should the same number with input files.")
        self.is_paired = (num_inputs == 2)
        # reference batch-size heuristic (detect.py:558-568); kept because --chunk_size is expressed in these batches
        denom = (2 * self.len * 6.4) if self.is_paired else (self.len * 6.4)
        self.batch_size = 2 ** math.floor(math.log2(((self.args.memory - 2) * 1024 * 1024) / denom))
        self.logger.info('Choose batch size: {}{}{}{} based on the given GPU RAM size {}GB and max read length {}'.format(
            colors.BOLD, colors.OKCYAN, self.batch_size, colors.ENDC, self.args.memory, self.len))
        if self.chunk_size is None:
            self.run()
        else:
            self.run_with_chunks()

    # ---- label helpers with the reference's signatures (host lists; used by tests / API users) -----------------
    @staticmethod
    def separate_reads(reads, labels):
        """{label: [reads]} (reference detect.py:600-614)"""
        out = {}
        for read, label in zip(reads, labels):
            out.setdefault(int(label), []).append(read)
        return out

    def separate_paired_reads(self, r1_reads, r1_outs, r2_reads, r2_outs):
        """Pair fusion on the device (rd_pair_fuse) then the reference's dict-of-lists result (detect.py:616-663)."""
        lab = module_arch.pair_fuse(r1_outs.contiguous(), r2_outs.contiguous(), self.args.ensure).cpu().tolist()
        return Predictor.separate_reads(r1_reads, lab), Predictor.separate_reads(r2_reads, lab)


def build_parser():
    args = argparse.ArgumentParser(description='rRNA sequence detector', formatter_class=RawTextHelpFormatter)
    args.add_argument('-c', '--config', default=None, type=str, help='Path of config file')
    args.add_argument('-d', '--deviceid', default=None, type=str,
                      help='Indices of GPUs to enable. Quotated comma-separated device ID numbers. (default: all)')
    args.add_argument('-l', '--len', type=int, required=True,
                      help='Sequencing read length. Note: the accuracy reduces for reads shorter than 40.')
    args.add_argument('-i', '--input', default=None, type=str, nargs='*', required=True,
                      help='Path of input sequence files (fasta and fastq), the second file will be considered as second end if two files given.')
    args.add_argument('-o', '--output', default=None, type=str, nargs='*', required=True,
                      help='Path of the output sequence files after rRNAs removal (same number of files as input). \n(Note: 2 times slower to write gz files)')
    args.add_argument('-r', '--rrna', default=None, type=str, nargs='*',
                      help='Path of the output sequence file of detected rRNAs (same number of files as input)')
    args.add_argument('-e', '--ensure', default="none", type=str, choices=['rrna', 'norrna', 'both', 'none'],
                      help='''Ensure which classificaion has high confidence for paired end reads.
norrna: output only high confident non-rRNAs, the rest are clasified as rRNAs;
rrna: vice versa, only high confident rRNAs are classified as rRNA and the rest output as non-rRNAs;
both: both non-rRNA and rRNA prediction with high confidence;
none: give label based on the mean probability of read pair.
      (Only applicable for paired end reads, discard the read pair when their predicitons are discordant)''')
    args.add_argument('-t', '--threads', default=10, type=int, help='Number of threads to use. (default: 10)')
    args.add_argument('-m', '--memory', default=32, type=int, help='Amount (GB) of GPU RAM. (default: 12)')
    args.add_argument('--chunk_size', default=None, type=int,
                      help='Use this parameter when having low memory. Parsing the file in chunks.\n{}.\n{}.'.format(
                          'Not needed when free RAM >=5 * your_file_size (uncompressed, sum of paired ends)',
                          'When chunk_size=256, memory=16 it will load 256 * 16 * 1024 reads each chunk (use ~20 GB for 100bp paired end)'))
    args.add_argument('--log', default=None, type=str, help='Log file name')
    args.add_argument('--semantics', default=None, choices=['gpu', 'cpu'],
                      help='(extension) which reference product to reproduce for reads shorter than --len or ending in N:\n'
                           'gpu = ribodetector (packed sequences, default); cpu = ribodetector_cpu (zero-padded input).')
    args.add_argument('-v', '--version', action='version', version='%(prog)s {version}'.format(version=__version__))
    return args


def main(argv=None, log_level=None):
    args = build_parser().parse_args(argv)
    config_file = os.path.join(cd, 'config.json') if args.config is None else args.config
    config = ConfigParser.from_json(config_file)
    seq_pred = Predictor(config, args, log_level=log_level)
    try:
        t0 = time.perf_counter()
        seq_pred.load_model()
        t1 = time.perf_counter()
        seq_pred.detect()
        t2 = time.perf_counter()
        fc = getattr(seq_pred, "_first_chunk", None)      # (time its labels arrived, its records): the rate after the pipeline filled
        steady = None
        if fc is not None and seq_pred.num_read > fc[1] and t2 > fc[0]:
            steady = len(seq_pred.input) * (seq_pred.num_read - fc[1]) / (t2 - fc[0])
        seq_pred.timing = {"load_model_s": t1 - t0, "detect_s": t2 - t1, "prefix_k": seq_pred.model.prefix_k,
                           "reads_per_s_after_first_chunk": steady, "ingest": getattr(seq_pred, "ingest", None),
                           "gz_ranges_s": getattr(seq_pred, "gz_shard_s", None), "pinned_cpus": getattr(seq_pred, "pinned_cpus", None),
                           "first_chunks_timeline": getattr(seq_pred, "_timeline", None), "run_started_at": getattr(seq_pred, "_t_run", None)}
        if os.environ.get("RD_TIMING_OUT"):          # (tools/host_scaling.py, tools/scale_sweep.sh: what a torchrun child measured, per rank)
            import json
            with open("%s.rank%d" % (os.environ["RD_TIMING_OUT"], seq_pred.rank), "w") as fh:
                json.dump(dict(seq_pred.timing, rank=seq_pred.rank, world=seq_pred.world, num_read=seq_pred.num_read,
                               thread_cpu_s=seq_pred.thread_cpu_s, process_cpu_s=time.process_time(), main_thread_s=getattr(seq_pred, "_stage_s", None),
                               phases_s=getattr(seq_pred, "phases_s", None)), fh, default=str)
    except BaseException:
        seq_pred.cleanup()                       # a failed run leaves no slot, '<out>.partN' or '<out>.joining' files behind
        if seq_pred.multi:
            # a rank that fails must not leave the others waiting in a collective (and must not wait in one itself while the
            # interpreter shuts down): report and leave at once - torch.distributed.run then tears the other ranks down
            import sys
            import traceback
            traceback.print_exc()
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(1)
        raise
    return seq_pred


if __name__ == '__main__':
    main()
