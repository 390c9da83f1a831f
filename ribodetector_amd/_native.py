"""ctypes binding of the C ABI in include/ribodetector_amd.h (librd_hip.so, built by __graft_entry__.build()).

There is NO CPU fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RD_HIP_LIB") or os.path.join(_HERE, "csrc", "librd_hip.so")   # env override: A/B of two builds

ENSURE_MODES = {"none": 0, "rrna": 1, "norrna": 2, "both": 3}
SEMANTICS = {"packed": 0, "gpu": 0, "padded": 1, "cpu": 1}
PREFIX_K_MIN, PREFIX_K_MAX, PREFIX_K_AUTO = 4, 13, 12     # include/ribodetector_amd.h RD_PREFIX_K_*; AUTO: 16 GiB, -12 % steps at 100 bp
VARIANTS = {"auto": 0, "mfma_f32": 1, "simple": 2, "mfma_f16x3_t32": 4}      # what librd_hip.so (the product build) accepts
# A/B and diagnostic instantiations: only in librd_hip_diag.so (built with -DRD_DIAG by __graft_entry__.build_diag(), selected
# with RD_HIP_LIB=.../librd_hip_diag.so by the scripts under tools/). Several compute wrong results by design.
DIAG_VARIANTS = {"mfma_f32_a0s0": 10, "mfma_f32_a1s0": 11, "mfma_f32_a0s1": 12, "mfma_f32_a1s1": 13,
                 "mfma_f32_diag_noew": 20, "mfma_f32_diag_nomfma": 21, "mfma_f32_diag_mfmaonly": 22, "mfma_f32_diag_mfmabar": 23}
if os.path.basename(LIB_PATH).startswith("librd_hip_diag"):
    VARIANTS = dict(VARIANTS, **DIAG_VARIANTS)

# every symbol include/ribodetector_amd.h declares (tests/test_abi.py checks the .so exports all of them)
SYMBOLS = [
    "rd_model_create", "rd_model_destroy", "rd_set_variant", "rd_variant_available", "rd_set_semantics", "rd_set_refine", "rd_set_refine_async", "rd_sync_results", "rd_refine", "rd_prefix_table_bytes", "rd_prefix_scratch_bytes", "rd_set_prefix_table", "rd_prefix_k", "rd_classify_workspace_bytes", "rd_classify",
    "rd_pair_fuse", "rd_count_labels", "rd_encode_codes", "rd_encode_onehot_padded", "rd_pack_plan",
    "rd_pack_onehot", "rd_profile_enable", "rd_profile_read", "rd_last_error", "rd_version",
    "rd_gz_workspace_bytes", "rd_gz_out_bound", "rd_gz_compress_selected", "rd_gz_eof_block", "rd_gz_inflate_members",
    "rd_fastq_index_workspace_bytes", "rd_fastq_index", "rd_fastq_gather", "rd_fastq_sample", "rd_fastq_strip_mark", "rd_fasta_index_workspace_bytes", "rd_fasta_index", "rd_fasta_gather", "rd_fasta_sample", "rd_select_workspace_bytes", "rd_select_pack", "rd_stream_create", "rd_stream_destroy", "rd_copy_bytes", "rd_gz_stream_workspace_bytes", "rd_gz_stream_inflate",
    "rd_gz_range_workspace_bytes", "rd_gz_range_decode", "rd_gz_range_resolve_workspace_bytes", "rd_gz_range_resolve",
]


class RdWeights(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in
                ("w_ih", "w_hh", "b_ih", "b_hh", "w_ih_r", "w_hh_r", "b_ih_r", "b_hh_r", "w_out", "b_out")] + \
               [("input_size", C.c_int32), ("hidden_size", C.c_int32), ("num_classes", C.c_int32)]


_lib = None


def lib():
    """Load librd_hip.so (once). Raises RuntimeError if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "ribodetector_amd: HIP extension %s is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % LIB_PATH)
    # torch first: librd_hip.so must bind to the HIP runtime torch ships and initialises (its bundled libamdhip64), not to a
    # second copy from /opt/rocm - with two runtimes in one process the second one sees no device.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i64, i32, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_size_t
    L.rd_model_create.argtypes = [C.POINTER(RdWeights), C.c_int, C.POINTER(vp)]
    L.rd_model_create.restype = C.c_int
    L.rd_model_destroy.argtypes = [vp]
    L.rd_model_destroy.restype = None
    L.rd_set_variant.argtypes = [vp, C.c_int]
    L.rd_variant_available.argtypes = [C.c_int]
    L.rd_set_semantics.argtypes = [vp, C.c_int]
    L.rd_set_refine.argtypes = [vp, C.c_float]
    L.rd_set_refine_async.argtypes = [vp, C.c_int]
    L.rd_sync_results.argtypes = [vp, vp]
    L.rd_refine.argtypes = [vp, vp, vp, vp, i64, i32, vp, vp, vp, C.c_float, vp]
    L.rd_prefix_table_bytes.argtypes = [i32]
    L.rd_prefix_table_bytes.restype = sz
    L.rd_prefix_scratch_bytes.argtypes = [i32]
    L.rd_prefix_scratch_bytes.restype = sz
    L.rd_set_prefix_table.argtypes = [vp, i32, vp, sz, vp, sz, vp]
    L.rd_prefix_k.argtypes = [vp]
    L.rd_classify_workspace_bytes.argtypes = [i64, i32]
    L.rd_classify_workspace_bytes.restype = sz
    L.rd_classify.argtypes = [vp, vp, vp, vp, i64, i32, vp, vp, vp, sz, vp]
    L.rd_pair_fuse.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    L.rd_count_labels.argtypes = [vp, i64, vp, vp]
    L.rd_encode_codes.argtypes = [vp, vp, vp, i64, i32, i32, vp, vp]
    L.rd_encode_onehot_padded.argtypes = [vp, vp, vp, i64, i32, vp, vp]
    L.rd_pack_plan.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, sz, vp]
    L.rd_pack_onehot.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp, vp]
    L.rd_gz_workspace_bytes.argtypes = [i64, i64]
    L.rd_gz_workspace_bytes.restype = sz
    L.rd_gz_out_bound.argtypes = [i64]
    L.rd_gz_out_bound.restype = sz
    L.rd_gz_compress_selected.argtypes = [vp, i64, vp, vp, i64, i32, vp, sz, vp, vp, sz, vp]
    L.rd_gz_eof_block.argtypes = [vp, sz]
    L.rd_gz_inflate_members.argtypes = [vp, i64, vp, i64, vp, i64, vp, vp]
    L.rd_fastq_index_workspace_bytes.argtypes = [i64]
    L.rd_fastq_index_workspace_bytes.restype = sz
    L.rd_fastq_index.argtypes = [vp, i64, i64, vp, vp, i32, vp, i64, vp, vp, sz, vp]
    L.rd_fastq_gather.argtypes = [vp, vp, vp, i64, i64, i64, vp, i64, vp, vp, vp, vp, vp, vp]
    L.rd_fastq_strip_mark.argtypes = [vp, vp, vp, i64, vp, vp]
    L.rd_fastq_sample.argtypes = [vp, vp, i64, vp, i64, vp]
    L.rd_fasta_index_workspace_bytes.argtypes = [i64, i64]
    L.rd_fasta_index_workspace_bytes.restype = sz
    L.rd_fasta_index.argtypes = [vp, i64, i64, vp, vp, i32, vp, i64, vp, i64, vp, vp, i64, vp, vp, sz, vp]
    L.rd_fasta_gather.argtypes = [vp, vp, vp, vp, i64, i64, i64, vp, i64, vp, vp, vp, vp, vp, vp]
    L.rd_fasta_sample.argtypes = [vp, vp, i64, vp, i64, vp]
    L.rd_select_workspace_bytes.argtypes = [i64]
    L.rd_select_workspace_bytes.restype = sz
    L.rd_select_pack.argtypes = [vp, i64, vp, vp, i64, i32, vp, sz, vp, vp, sz, vp]
    L.rd_stream_create.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.rd_stream_destroy.argtypes = [vp]
    L.rd_copy_bytes.argtypes = [vp, vp, i64, i32, vp]
    L.rd_gz_stream_workspace_bytes.argtypes = [i64, i32, i32, i64]
    L.rd_gz_stream_workspace_bytes.restype = sz
    L.rd_gz_stream_inflate.argtypes = [vp, i64, i64, i64, i32, i32, C.c_uint32, vp, i64, i32, vp, vp, vp, i64, vp, vp, sz, vp]
    L.rd_gz_range_workspace_bytes.argtypes = [i64, i32, i32, i64]
    L.rd_gz_range_workspace_bytes.restype = sz
    L.rd_gz_range_decode.argtypes = [vp, i64, i64, i64, i32, i32, C.c_uint32, vp, i64, i32, vp, vp, vp, i64, vp, vp, sz, vp]
    L.rd_gz_range_resolve_workspace_bytes.argtypes = [i64]
    L.rd_gz_range_resolve_workspace_bytes.restype = sz
    L.rd_gz_range_resolve.argtypes = [vp, i64, vp, C.c_uint32, vp, vp, vp, sz, vp]
    L.rd_profile_enable.argtypes = [vp, C.c_int]
    L.rd_profile_read.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_double)]
    L.rd_last_error.restype = C.c_char_p
    L.rd_version.restype = C.c_char_p
    for name in SYMBOLS:
        getattr(L, name)          # AttributeError here = the .so does not export a symbol the header declares
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib().rd_last_error().decode()))


def stream_ptr(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def new_event():
    import torch
    return torch.cuda.Event(blocking=True)


def wait_event(ev):
    """sleep until `ev` has happened: the event is queried every 0.5 ms. hipEventSynchronize SPINS a host core on this stack even for
    an event created with the blocking-sync flag (measured: the waiting thread's CPU time = its wall time), and the ranks of a node
    share 16 cores with their readers and writers. RD_EVENT_WAIT=sync uses it all the same."""
    if os.environ.get("RD_EVENT_WAIT") == "sync":
        ev.synchronize()
        return
    import time
    while not ev.query():
        time.sleep(5e-4)


def copy_bytes(dst, src, nbytes, stream, workgroups=0):
    """dst[:nbytes] = src[:nbytes] (uint8 tensors: device memory or pinned host memory) by a kernel on `stream` (C ABI rd_copy_bytes);
    RD_COPY_ENGINE=1 keeps hipMemcpyAsync (the DMA engines)"""
    import torch
    if os.environ.get("RD_COPY_ENGINE") == "1":
        with torch.cuda.stream(stream):
            dst[:nbytes].copy_(src[:nbytes], non_blocking=True)
        return
    dev = dst.device if dst.is_cuda else src.device
    workgroups = workgroups or int(os.environ.get("RD_COPY_WGS", "0"))
    with torch.cuda.device(dev):
        check(lib().rd_copy_bytes(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), int(nbytes), int(workgroups), C.c_void_p(stream.cuda_stream)),
              "rd_copy_bytes")


def ptr(t):
    """device/host pointer of a torch tensor (None -> NULL)"""
    return C.c_void_p(0 if t is None else t.data_ptr())


# ---- host ingest library (librd_host.so: C++ + zlib, no GPU) -------------------------------------------------------
HOST_LIB_PATH = os.path.join(_HERE, "csrc", "librd_host.so")
HOST_SYMBOLS = ["rd_reader_open", "rd_reader_close", "rd_reader_next", "rd_host_file_info", "rd_host_find_record_start",
                "rd_host_count_records", "rd_host_skip_records", "rd_reader_open_range", "rd_writer_open", "rd_writer_write_selected",
                "rd_writer_write_members", "rd_writer_close", "rd_reader_open_feed", "rd_reader_feed", "rd_reader_feed_end", "rd_reader_feed_abort", "rd_reader_set_flush_empty_tail", "rd_writer_write_text", "rd_writer_set_eof_marker", "rd_host_gz_index", "rd_writer_threads", "rd_host_last_error", "rd_host_set_threads", "rd_host_set_gz_threads", "rd_host_gunzip", "rd_host_gunzip_parallel"]
_host = None


def host_lib():
    """Load librd_host.so (once); raises RuntimeError if it has not been built."""
    global _host
    if _host is not None:
        return _host
    if not os.path.exists(HOST_LIB_PATH):
        raise RuntimeError("ribodetector_amd: host library %s is missing - run __graft_entry__.build()" % HOST_LIB_PATH)
    L = C.CDLL(HOST_LIB_PATH)
    vp, i64 = C.c_void_p, C.c_int64
    L.rd_reader_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.rd_reader_close.argtypes = [vp]
    L.rd_host_file_info.argtypes = [C.c_char_p, C.POINTER(i64), C.POINTER(C.c_int32)]
    L.rd_host_find_record_start.argtypes = [C.c_char_p, C.c_int, i64, C.POINTER(i64)]
    L.rd_host_count_records.argtypes = [C.c_char_p, C.c_int, i64, i64, C.POINTER(i64)]
    L.rd_host_skip_records.argtypes = [C.c_char_p, C.c_int, i64, i64, C.POINTER(i64)]
    L.rd_reader_open_range.argtypes = [C.c_char_p, C.c_int, i64, i64, C.POINTER(vp)]
    L.rd_reader_close.restype = None
    L.rd_reader_next.argtypes = [vp, i64, vp, i64, vp, vp, vp, C.POINTER(i64), C.POINTER(i64)]
    L.rd_reader_open_feed.argtypes = [C.c_int, C.POINTER(vp)]
    L.rd_reader_feed.argtypes = [vp, vp, i64]
    L.rd_reader_feed_end.argtypes = [vp, C.c_char_p]
    L.rd_reader_feed_abort.argtypes = [vp]
    L.rd_reader_set_flush_empty_tail.argtypes = [vp, C.c_int]
    L.rd_writer_write_text.argtypes = [vp, vp, i64]
    L.rd_writer_set_eof_marker.argtypes = [vp, C.c_int]
    L.rd_host_gz_index.argtypes = [vp, i64, i64, i64, vp, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.rd_writer_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.rd_writer_write_selected.argtypes = [vp, vp, vp, i64, vp, C.c_int32]
    L.rd_writer_write_members.argtypes = [vp, vp, i64]
    L.rd_writer_close.argtypes = [vp]
    L.rd_writer_threads.argtypes = [vp]
    L.rd_host_set_threads.argtypes = [C.c_int]
    L.rd_host_set_gz_threads.argtypes = [C.c_int]
    L.rd_host_gunzip.argtypes = [C.c_char_p, vp, i64, C.POINTER(i64)]
    L.rd_host_gunzip_parallel.argtypes = [C.c_char_p, vp, i64, C.POINTER(i64), C.c_int, i64, vp]
    L.rd_host_last_error.restype = C.c_char_p
    _host = L
    return L


def host_check(rc, what):
    if rc != 0:
        raise ValueError("%s: %s" % (what, host_lib().rd_host_last_error().decode()))
