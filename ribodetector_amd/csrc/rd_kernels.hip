// rd_kernels.hip - CDNA4 (gfx950) kernels + C ABI of the RiboDetector BiLSTM inference path.
//
// Reference path being replaced (paths relative to /root/reference/ribodetector):
//   data_loader/seq_encoder.py:11-18,126-127   one-hot nucleotide encoder
//   detect.py:666-726                          collate: truncate to max_len, one-hot, pack_sequence
//   model/model.py:32-37,114-119               forward1: BiLSTM, last-timestep gather, Linear(256->2)
//   detect.py:288,481                          argmax
//   detect.py:616-663                          paired-end label fusion
//
// Structural facts used (verified by tests against the reference's own outputs):
//   * forward1 gathers timestep len-1. There the reverse LSTM has made exactly ONE step from the zero state,
//     so its contribution to the logits is a 5-entry table keyed by the last base (rev_lut).
//   * the input is one-hot/zero, so W_ih x + b_ih + b_hh is a 5-row table (in_lut) - a gather, not a GEMM.
//   => the only dense work is the forward recurrence h[B,128] . W_hh^T[128,512] per timestep.
//
// Kernel inventory (DESIGN.md §3 has the roofline of each)
//   rd_prep_kernel, rd_revtab_kernel   weight pre-packing (once per model); reverse-half table of the padded semantics
//   rd_steps_kernel, rd_bucket_scan/scatter_kernel   steps per read + length bucketing for rd_classify (longest first)
//   rd_len_hist/scan/scatter_kernel    stable counting sort = pack_sequence's sort, for rd_pack_plan
//   rd_lstm_mfma_f16x3_t32_kernel      DEFAULT recurrence: split-precision f16 MFMA 32x32x16, weights resident in AGPRs,
//                                      hand-interleaved gate math; fused encoder + FC + argmax epilogue
//   rd_lstm_mfma_f32_kernel            exact-fp32 MFMA recurrence (A/B reference for the split-precision kernels)
//   rd_lstm_simple_kernel              plain-FMA cross-check of the same function
//   rd_refine_kernel                   float64 re-evaluation of the reads whose margin is inside the fp32 noise band
//   rd_encode_* / rd_pack_onehot       standalone encoder kernels (reference tensor layouts), HBM-bound
//   rd_pair_fuse_kernel, rd_count_kernel
//   rd_gz_*                            records of one label -> gzip (BGZF) members on the device (rd_deflate.hpp)
//   rd_fq_* / rd_fa_*                  FASTQ records framed, FASTA batches re-written and indexed in HBM (rd_fastq_index.hpp, rd_fasta_index.hpp)
#include <stdlib.h>
#include "rd_common.hpp"
#include "rd_prep.hpp"
#include "rd_recurrence.hpp"
#include "rd_sort.hpp"
#include "rd_lstm_f32.hpp"
#include "rd_lstm_t32.hpp"
#include "rd_refine.hpp"
#include "rd_encode.hpp"
#include "rd_deflate.hpp"
#include "rd_inflate_dev.hpp"
#include "rd_inflate_stream.hpp"
#include "rd_fastq_index.hpp"
#include "rd_fasta_index.hpp"

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char *rd_last_error(void) { return g_err; }
const char *rd_version(void) { return "ribodetector_amd 0.1.0 (gfx950)"; }

int rd_model_create(const rd_weights *w, int device, rd_model **out) {
    if (!w || !out) RD_FAIL(RD_E_INVALID, "rd_model_create: null argument");
    if (w->input_size != 4 || w->hidden_size != HID || w->num_classes != 2)
        RD_FAIL(RD_E_UNSUPPORTED, "rd_model_create: kernels cover input_size=4, hidden_size=128, num_classes=2 (got %d,%d,%d)",
                w->input_size, w->hidden_size, w->num_classes);
    const float *src[10] = {w->w_ih, w->w_hh, w->b_ih, w->b_hh, w->w_ih_r, w->w_hh_r, w->b_ih_r, w->b_hh_r, w->w_out, w->b_out};
    const int offs[11] = {OFF_WIH, OFF_WHH, OFF_BIH, OFF_BHH, OFF_WIHR, OFF_WHHR, OFF_BIHR, OFF_BHHR, OFF_WOUT, OFF_BOUT, RAW_FLOATS};
    for (int i = 0; i < 10; ++i)
        if (!src[i]) RD_FAIL(RD_E_INVALID, "rd_model_create: weight pointer %d is null", i);
    RD_HIP(hipSetDevice(device));
    rd_model *m = new rd_model();
    memset(m, 0, sizeof(*m));
    m->device = device;
    m->variant = RD_VARIANT_MFMA_F16X3_T32;
    m->refine_thresh = RD_REFINE_DEFAULT;
    m->prefix_k = 0;
    float *host = new float[RAW_FLOATS];
    for (int i = 0; i < 10; ++i) memcpy(host + offs[i], src[i], sizeof(float) * (size_t)(offs[i + 1] - offs[i]));
    hipError_t e = hipSuccess;
    auto A = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    A((void **)&m->d.raw, sizeof(float) * RAW_FLOATS);
    A((void **)&m->d.wpack32, sizeof(float) * 4 * 8 * 32 * 64);
    A((void **)&m->d.wt_hh, sizeof(float) * HID * G4);
    A((void **)&m->d.wpack16b, sizeof(uint16_t) * 4 * 2 * 4 * 8 * 64 * 8);
    A((void **)&m->d.in_lut, sizeof(float) * 5 * G4);
    A((void **)&m->d.lut_t32, sizeof(float) * 4 * 2 * 4 * 4 * 6 * 4);
    A((void **)&m->d.rev_lut, sizeof(float) * 10);
    A((void **)&m->d.rev_tab, sizeof(float) * (size_t)MAX_LEN_LIMIT * 10);
    A((void **)&m->d.w_out, sizeof(float) * 512);
    A((void **)&m->d.b_out, sizeof(float) * 2);
    A((void **)&m->d.zero_row, PFX_ROW);
    if (e == hipSuccess) e = hipMemset(m->d.zero_row, 0, PFX_ROW);
    if (e == hipSuccess) e = hipMemcpy(m->d.raw, host, sizeof(float) * RAW_FLOATS, hipMemcpyHostToDevice);
    delete[] host;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rd_prep_kernel, dim3(64), dim3(256), 0, 0, m->d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "rd_model_create: %s", hipGetErrorString(e));
        rd_model_destroy(m);
        return RD_E_HIP;
    }
    m->ptab = m->d.zero_row;
    *out = m;
    return RD_OK;
}

void rd_model_destroy(rd_model *m) {
    if (!m) return;
    hipSetDevice(m->device);
    if (m->side) { hipStreamSynchronize(m->side); hipStreamDestroy(m->side); }
    if (m->ev_fork) hipEventDestroy(m->ev_fork);
    for (int x = 0; x < 2; ++x) {
        if (m->ev_join[x]) hipEventDestroy(m->ev_join[x]);
        if (m->q_e[x]) hipFree(m->q_e[x]);
        if (m->q_count[x]) hipFree(m->q_count[x]);
    }
    hipFree(m->d.raw); hipFree(m->d.wpack32); hipFree(m->d.wt_hh); hipFree(m->d.in_lut);
    hipFree(m->d.rev_lut); hipFree(m->d.rev_tab); hipFree(m->d.w_out); hipFree(m->d.b_out);
    if (m->d.wpack16b) hipFree(m->d.wpack16b);
    if (m->d.lut_t32) hipFree(m->d.lut_t32);
    if (m->d.zero_row) hipFree(m->d.zero_row);
    for (int i = 0; i < 2 * 512; ++i)
        if (m->prof_ev[i]) hipEventDestroy(m->prof_ev[i]);
    delete m;
}

// Product build: the three kernels that compute the function (AUTO = the split-precision MFMA kernel). The A/B and diagnostic
// instantiations of the exact-fp32 kernel (ids 10-23; several compute WRONG results by design: gate math or MFMAs removed) exist
// only in the -DRD_DIAG build (librd_hip_diag.so, tools/), never in librd_hip.so. (The experiment copy of the default kernel of
// rounds 1-3, rd_lstm_t32_diag.hpp, was removed in round 4: ONE body of that kernel; its results are in profiles/ and DESIGN.md §8.)
static bool rd_variant_known(int v) {
    if (v == RD_VARIANT_MFMA_F32 || v == RD_VARIANT_SIMPLE || v == RD_VARIANT_MFMA_F16X3_T32) return true;
#ifdef RD_DIAG
    switch (v) {
    case 10: case 11: case 12: case 13: case 20: case 21: case 22: case 23: return true;
    default: break;
    }
#endif
    return false;
}

int rd_variant_available(int variant) { return (variant == RD_VARIANT_AUTO || rd_variant_known(variant)) ? 1 : 0; }

int rd_set_variant(rd_model *m, int variant) {
    if (!m) RD_FAIL(RD_E_INVALID, "rd_set_variant: null model");
    if (variant == RD_VARIANT_AUTO) variant = RD_VARIANT_MFMA_F16X3_T32;
    if (!rd_variant_known(variant))
        RD_FAIL(RD_E_UNSUPPORTED, "rd_set_variant: variant %d not available in this build (product build: 0 auto, 1 mfma_f32, 2 simple, 4 mfma_f16x3_t32)", variant);
    m->variant = variant;
    return RD_OK;
}

int rd_set_semantics(rd_model *m, int semantics) {
    if (!m) RD_FAIL(RD_E_INVALID, "rd_set_semantics: null model");
    if (semantics != RD_SEM_PACKED && semantics != RD_SEM_PADDED) RD_FAIL(RD_E_INVALID, "rd_set_semantics: unknown semantics %d", semantics);
    m->semantics = semantics;
    return RD_OK;
}

int rd_set_refine(rd_model *m, float thresh) {
    if (!m) RD_FAIL(RD_E_INVALID, "rd_set_refine: null model");
    if (!(thresh >= 0.0f) || thresh > 1.0f) RD_FAIL(RD_E_INVALID, "rd_set_refine: threshold %g out of range [0, 1]", (double)thresh);
    m->refine_thresh = thresh;
    return RD_OK;
}

constexpr uint32_t RD_REFINE_QCAP = 8192;   // candidates a queue holds (an overflow is caught by the second tier of rd_async_flush)

static RefineQueue rd_queue(const rd_model *m, int x) { return RefineQueue{(RefineEntry *)m->q_e[x], m->q_count[x], RD_REFINE_QCAP}; }

// evaluate the candidates waiting in queue x on the model's stream, beside whatever `st` does next
static int rd_async_flush(rd_model *m, hipStream_t st, int x) {
    RD_HIP(hipEventRecord(m->ev_fork, st));
    RD_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0));
    hipLaunchKernelGGL(rd_refine_flush_kernel, dim3(REFINE_FLUSH_WGS), dim3(1024), 0, m->side, m->d, rd_queue(m, x));
    for (int k = 0; k < m->q_npend[x]; ++k) {   // second tier: launches that return at once unless the queue overflowed
        const auto &pd = m->q_pend[x][k];
        ReadBatch rb{(const uint8_t *)pd.p[0], (const int64_t *)pd.p[1], (const int32_t *)pd.p[2], nullptr, nullptr, pd.n, pd.max_len, pd.sem,
                     m->d.rev_tab, nullptr, nullptr, 0, RefineQueue{nullptr, nullptr, 0}, 0.0f};
        int64_t nb = (pd.n + REFINE_SLICE - 1) / REFINE_SLICE;
        if (nb > 16) nb = 16;   // the workgroups walk the slices (gate closed: sixteen workgroups look at it and leave)
        hipLaunchKernelGGL(rd_refine_kernel, dim3((unsigned)nb), dim3(1024), 0, m->side, m->d, rb, (const float2 *)nullptr, pd.thresh,
                           (float *)pd.p[3], (uint8_t *)pd.p[4], rd_queue(m, x), (const uint32_t *)m->q_count[x]);
    }
    RD_HIP(hipGetLastError());
    RD_HIP(hipMemsetAsync(m->q_count[x], 0, sizeof(uint32_t), m->side));
    RD_HIP(hipEventRecord(m->ev_join[x], m->side));
    m->q_flushing[x] = 1;
    m->q_wait[x] = 1;
    return RD_OK;
}

// queue x is about to take the candidates of a call issued on `st`: its last flush (and the reset of its counter) comes first. The
// stream that joined that flush has waited already; a rd_sync_results on ANOTHER stream (a caller's post-processing stream) has not
// ordered `st` behind it - the flush is long finished by then (it takes 0.3 ms, a recurrence launch tens), this makes it formal.
static int rd_async_acquire(rd_model *m, hipStream_t st, int x) {
    if (m->q_wait[x]) {
        RD_HIP(hipStreamWaitEvent(st, m->ev_join[x], 0));
        m->q_wait[x] = 0;
    }
    return RD_OK;
}

// everything issued on `st` from here on sees the results of the calls whose candidates were in queue x
static int rd_async_join(rd_model *m, hipStream_t st, int x) {
    if (m->q_flushing[x]) {
        RD_HIP(hipStreamWaitEvent(st, m->ev_join[x], 0));
        m->q_flushing[x] = 0;
        m->q_npend[x] = 0;
    }
    return RD_OK;
}

int rd_sync_results(rd_model *m, void *stream) {
    if (!m) RD_FAIL(RD_E_INVALID, "rd_sync_results: null model");
    if (!m->refine_async) return RD_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc = RD_OK;
    if (m->q_calls > 0) {
        rc = rd_async_flush(m, st, m->q_cur);
        if (rc) return rc;
        m->q_cur ^= 1;   // calls issued from here on (on whatever stream) record into the other queue while this one is evaluated
        m->q_calls = 0;
    }
    for (int x = 0; x < 2 && !rc; ++x) rc = rd_async_join(m, st, x);
    return rc;
}

int rd_set_refine_async(rd_model *m, int calls) {
    if (!m) RD_FAIL(RD_E_INVALID, "rd_set_refine_async: null model");
    if (calls < 0 || calls > 16) RD_FAIL(RD_E_INVALID, "rd_set_refine_async: calls per group %d out of range [0, 16]", calls);
    if (m->q_calls || m->q_flushing[0] || m->q_flushing[1])
        RD_FAIL(RD_E_INVALID, "rd_set_refine_async: calls are pending - rd_sync_results first");
    if (calls && !m->side) {
        RD_HIP(hipSetDevice(m->device));
        // highest priority: when a CU becomes free its workgroups go first - a workgroup of the float64 pass needs a whole CU (1,024
        // threads), and so does every workgroup of the recurrence kernel that the pass is meant to run beside
        int lo = 0, hi = 0;
        RD_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        const char *pr = getenv("RD_REFINE_STREAM_PRIORITY");   // (A/B knob: "low" / "normal"; default high)
        const int prio = pr && !strcmp(pr, "low") ? lo : pr && !strcmp(pr, "normal") ? (lo + hi) / 2 : hi;
        RD_HIP(hipStreamCreateWithPriority(&m->side, hipStreamNonBlocking, prio));
        RD_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
        for (int x = 0; x < 2; ++x) {
            RD_HIP(hipEventCreateWithFlags(&m->ev_join[x], hipEventDisableTiming));
            RD_HIP(hipMalloc(&m->q_e[x], sizeof(RefineEntry) * RD_REFINE_QCAP));
            RD_HIP(hipMalloc((void **)&m->q_count[x], sizeof(uint32_t)));
            RD_HIP(hipMemset(m->q_count[x], 0, sizeof(uint32_t)));
        }
    }
    m->refine_async = calls;
    return RD_OK;
}

static int rd_refine_launch(rd_model *m, const ReadBatch &rb, float *logits, uint8_t *labels, const float *mate_logits, float thresh,
                            hipStream_t st) {
    const int64_t nb = (rb.n + REFINE_SLICE - 1) / REFINE_SLICE;
    hipLaunchKernelGGL(rd_refine_kernel, dim3((unsigned)nb), dim3(1024), 0, st, m->d, rb, (const float2 *)mate_logits, thresh, logits, labels,
                       RefineQueue{nullptr, nullptr, 0}, (const uint32_t *)nullptr);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_refine(const rd_model *cm, const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n, int32_t max_len,
              float *logits, uint8_t *labels, const float *mate_logits, float thresh, void *stream) {
    rd_model *m = const_cast<rd_model *>(cm);
    if (!m || !logits) RD_FAIL(RD_E_INVALID, "rd_refine: null model or logits");
    if (n < 0 || n > 0x7fffffffLL) RD_FAIL(RD_E_INVALID, "rd_refine: n=%lld out of range", (long long)n);
    if (max_len < 1 || max_len > MAX_LEN_LIMIT) RD_FAIL(RD_E_INVALID, "rd_refine: max_len=%d out of range [1,%d]", max_len, MAX_LEN_LIMIT);
    if (!(thresh <= 1.0f)) RD_FAIL(RD_E_INVALID, "rd_refine: threshold %g out of range", (double)thresh);
    if (thresh <= 0.0f) thresh = m->refine_thresh;   // the model's band
    if (n == 0 || thresh <= 0.0f) return RD_OK;
    if (!arena || !seq_off || !seq_len) RD_FAIL(RD_E_INVALID, "rd_refine: null input pointer");
    hipStream_t st = (hipStream_t)stream;
    if (m->refine_async) { int rc0 = rd_sync_results(m, stream); if (rc0) return rc0; }   // the logits it reads may have candidates waiting
    // padded semantics: the reverse-half table is built by rd_classify for ITS max_len, on ITS stream. Rebuilding it here - typically
    // on a side stream, while a recurrence kernel of the main stream reads it - would be a data race (advisor finding, round 2), so
    // the pass refuses a max_len the table was not built for: it re-evaluates reads of an earlier rd_classify call by definition.
    if (m->semantics == RD_SEM_PADDED && m->rev_tab_len != max_len)
        RD_FAIL(RD_E_INVALID, "rd_refine: padded semantics needs a preceding rd_classify with the same max_len (table built for %d, got %d)",
                m->rev_tab_len, max_len);
    ReadBatch rb{arena, seq_off, seq_len, nullptr, nullptr, n, max_len, m->semantics, m->d.rev_tab, nullptr, nullptr, 0,
                 RefineQueue{nullptr, nullptr, 0}, 0.0f};
    return rd_refine_launch(m, rb, logits, labels, mate_logits, thresh, st);
}

size_t rd_prefix_table_bytes(int32_t k) {
    if (k < RD_PREFIX_K_MIN || k > RD_PREFIX_K_MAX) return 0;
    return (((size_t)1 << (2 * k)) + 1) * PFX_ROW;
}

size_t rd_prefix_scratch_bytes(int32_t k) {
    if (k < RD_PREFIX_K_MIN || k > RD_PREFIX_K_MAX) return 0;
    return ((size_t)1 << (2 * (k - 1))) * PFX_ROW;
}

int rd_prefix_k(const rd_model *m) { return m ? m->prefix_k : 0; }

int rd_set_prefix_table(rd_model *m, int32_t k, void *table, size_t table_bytes, void *scratch, size_t scratch_bytes, void *stream) {
    if (!m) RD_FAIL(RD_E_INVALID, "rd_set_prefix_table: null model");
    if (k == 0) {
        m->prefix_k = 0;
        m->ptab = m->d.zero_row;
        m->ptab_variant = 0;
        return RD_OK;
    }
    if (m->variant != RD_VARIANT_MFMA_F16X3_T32 && m->variant != RD_VARIANT_MFMA_F32)
        RD_FAIL(RD_E_UNSUPPORTED, "rd_set_prefix_table: kernel variant %d has no prefix-state table (mfma_f16x3_t32 and mfma_f32 do)", m->variant);
    const size_t need = rd_prefix_table_bytes(k), need_s = rd_prefix_scratch_bytes(k);
    if (!need) RD_FAIL(RD_E_INVALID, "rd_set_prefix_table: k=%d out of range [%d,%d] (or 0 = none)", k, RD_PREFIX_K_MIN, RD_PREFIX_K_MAX);
    if (!table || ((uintptr_t)table & 255) || !scratch || ((uintptr_t)scratch & 255))
        RD_FAIL(RD_E_INVALID, "rd_set_prefix_table: table and scratch must be 256-byte aligned device pointers");
    if (table_bytes < need) RD_FAIL(RD_E_WORKSPACE, "rd_set_prefix_table: table too small for k=%d: %zu < %zu", k, table_bytes, need);
    if (scratch_bytes < need_s) RD_FAIL(RD_E_WORKSPACE, "rd_set_prefix_table: scratch too small for k=%d: %zu < %zu", k, scratch_bytes, need_s);
    hipStream_t st = (hipStream_t)stream;
    uint8_t *tab = (uint8_t *)table, *scr = (uint8_t *)scratch;
    // level j (the states after every j-base prefix) from level j-1, one step per launch; the levels alternate between the table
    // and the scratch so that level k lands in the table (level k-1, a part of its size, is the largest one in the scratch)
    const uint8_t *prev = m->d.zero_row;
    for (int j = 1; j <= k; ++j) {
        uint8_t *dst = ((k - j) & 1) ? scr : tab;
        const int64_t rows = (int64_t)1 << (2 * j);
        ReadBatch rb{nullptr, nullptr, nullptr, nullptr, nullptr, rows, 1, RD_SEM_PACKED, nullptr, prev, nullptr, j,
                     RefineQueue{nullptr, nullptr, 0}, 0.0f};
        const dim3 grid((unsigned)((rows + 63) / 64)), blk(256);
        if (m->variant == RD_VARIANT_MFMA_F32)   // the rows are the state of the kernel that builds them: each kernel its own table
            hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<2, 0, 0, true>), grid, blk, 0, st, m->d, rb, (float *)dst, (uint8_t *)nullptr);
        else
            hipLaunchKernelGGL(rd_lstm_mfma_f16x3_t32_kernel<true>, grid, blk, 0, st, m->d, rb, (float *)dst, (uint8_t *)nullptr);
        prev = dst;
    }
    RD_HIP(hipGetLastError());
    RD_HIP(hipMemsetAsync(tab + ((size_t)1 << (2 * k)) * PFX_ROW, 0, PFX_ROW, st));   // the zero row: reads that cannot use a prefix row
    RD_HIP(hipStreamSynchronize(st));
    m->prefix_k = k;
    m->ptab = tab;
    m->ptab_variant = m->variant;
    return RD_OK;
}

size_t rd_classify_workspace_bytes(int64_t n, int32_t max_len) {
    if (n < 0 || max_len < 1) return 0;
    return sort_plan(n, max_len).total;
}

int rd_profile_enable(rd_model *m, int enable) {
    if (!m) RD_FAIL(RD_E_INVALID, "rd_profile_enable: null model");
    m->prof_enabled = enable;
    m->prof_count = 0;
    m->prof_ms_accum = 0;
    m->prof_launches_accum = 0;
    return RD_OK;
}

static int rd_profile_drain(rd_model *m) {
    for (int i = 0; i < m->prof_count; ++i) {
        float ms = 0;
        RD_HIP(hipEventSynchronize(m->prof_ev[2 * i + 1]));
        RD_HIP(hipEventElapsedTime(&ms, m->prof_ev[2 * i], m->prof_ev[2 * i + 1]));
        m->prof_ms_accum += ms;
        m->prof_launches_accum += 1;
    }
    m->prof_count = 0;
    return RD_OK;
}

int rd_profile_read(rd_model *m, int64_t *launches, double *total_ms) {
    if (!m) RD_FAIL(RD_E_INVALID, "rd_profile_read: null model");
    int rc = rd_profile_drain(m);
    if (rc) return rc;
    if (launches) *launches = m->prof_launches_accum;
    if (total_ms) *total_ms = m->prof_ms_accum;
    return RD_OK;
}

int rd_classify(const rd_model *cm, const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n,
                int32_t max_len, float *logits, uint8_t *labels, void *workspace, size_t workspace_bytes, void *stream) {
    rd_model *m = const_cast<rd_model *>(cm);
    if (!m || !logits) RD_FAIL(RD_E_INVALID, "rd_classify: null model or logits");
    if (n < 0 || n > 0x7fffffffLL) RD_FAIL(RD_E_INVALID, "rd_classify: n=%lld out of range", (long long)n);
    if (max_len < 1 || max_len > MAX_LEN_LIMIT) RD_FAIL(RD_E_INVALID, "rd_classify: max_len=%d out of range [1,%d]", max_len, MAX_LEN_LIMIT);
    if (n == 0) return RD_OK;
    if (!arena || !seq_off || !seq_len || !workspace) RD_FAIL(RD_E_INVALID, "rd_classify: null input pointer");
    hipStream_t st = (hipStream_t)stream;
    const SortPlan sp = sort_plan(n, max_len);
    if (workspace_bytes < sp.total) RD_FAIL(RD_E_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, sp.total);
    // deferred float64 pass (rd_set_refine_async): not inside a stream capture (a captured call stays self-contained)
    bool deferred = false;
    int join_after_launch = -1;
    if (m->refine_async && m->refine_thresh > 0.0f) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(st, &cap);
        deferred = cap == hipStreamCaptureStatusNone;
    }
    if (deferred) {
        // a call that touches a buffer of a call whose candidates still wait (or are being evaluated) gets the finished results first
        bool clash = false;
        for (int x = 0; x < 2 && !clash; ++x)
            for (int k = 0; k < m->q_npend[x] && !clash; ++k) {
                const auto &pd = m->q_pend[x][k];
                const char *lo = (const char *)pd.p[3], *la = (const char *)pd.p[4];
                clash = (m->semantics == RD_SEM_PADDED && pd.max_len != max_len) ||   // (the padded table would be rebuilt under the flush)
                        arena == pd.p[0] || seq_off == pd.p[1] || seq_len == pd.p[2] ||
                        ((const char *)logits < lo + 8 * pd.n && lo < (const char *)logits + 8 * n) ||
                        (labels && la && (const char *)labels < la + pd.n && la < (const char *)labels + n);
            }
        if (clash) { int rc0 = rd_sync_results(m, stream); if (rc0) return rc0; }
        if (m->q_calls >= m->refine_async) {   // the group is complete: its candidates are evaluated beside THIS call's recurrence
            const int old = m->q_cur;
            int rc0 = rd_async_flush(m, st, old);
            if (rc0) return rc0;
            m->q_cur ^= 1;
            m->q_calls = 0;
            rc0 = rd_async_join(m, st, m->q_cur);   // the other queue's flush was issued a whole group ago
            if (rc0) return rc0;
            join_after_launch = old;
        }
        int rc1 = rd_async_acquire(m, st, m->q_cur);
        if (rc1) return rc1;
    }
    if (m->semantics == RD_SEM_PADDED && m->rev_tab_len != max_len) {   // (after the block above: candidates of another max_len that
        hipLaunchKernelGGL(rd_revtab_kernel, dim3(1), dim3(512), 0, st, m->d, max_len);   // still waited have been evaluated by now)
        m->rev_tab_len = max_len;
    }
    int32_t *steps = nullptr, *order = nullptr, *pfx = nullptr;
    const int pk = m->variant == m->ptab_variant ? m->prefix_k : 0;      // the rows hold the state of the kernel that built them
    int rc = run_steps_and_buckets(arena, seq_off, seq_len, n, max_len, m->semantics, workspace, workspace_bytes, steps, order, pk, pfx, st);
    if (rc) return rc;
    ReadBatch rb{arena, seq_off, seq_len, steps, order, n, max_len, m->semantics, m->d.rev_tab, pk > 0 ? m->ptab : m->d.zero_row, pfx, pk,
                 deferred ? rd_queue(m, m->q_cur) : RefineQueue{nullptr, nullptr, 0}, m->refine_thresh};
    hipEvent_t *ev = nullptr;
    if (m->prof_enabled) {
        if (m->prof_count == 512) { rc = rd_profile_drain(m); if (rc) return rc; }
        ev = &m->prof_ev[2 * m->prof_count];
        for (int i = 0; i < 2; ++i)
            if (!ev[i]) RD_HIP(hipEventCreate(&ev[i]));
        RD_HIP(hipEventRecord(ev[0], st));
    }
    if (m->variant == RD_VARIANT_SIMPLE) {
        const int64_t nwg = (n + SB - 1) / SB;
        hipLaunchKernelGGL(rd_lstm_simple_kernel, dim3((unsigned)nwg), dim3(512), 0, st, m->d, rb, logits, labels);
    } else {
        const int64_t nwg = (n + BT - 1) / BT;
        const dim3 grid((unsigned)nwg), blk(256);
        switch (m->variant) {
        case RD_VARIANT_MFMA_F16X3_T32: hipLaunchKernelGGL(rd_lstm_mfma_f16x3_t32_kernel<false>, grid, blk, 0, st, m->d, rb, logits, labels); break;
        case RD_VARIANT_MFMA_F32: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<2, 0>), grid, blk, 0, st, m->d, rb, logits, labels); break;
#ifdef RD_DIAG
        case 10: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<0, 0>), grid, blk, 0, st, m->d, rb, logits, labels); break;
        case 11: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<1, 0>), grid, blk, 0, st, m->d, rb, logits, labels); break;
        case 12: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<0, 1>), grid, blk, 0, st, m->d, rb, logits, labels); break;
        case 13: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<1, 1>), grid, blk, 0, st, m->d, rb, logits, labels); break;
        case 20: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<1, 0, 1>), grid, blk, 0, st, m->d, rb, logits, labels); break;
        case 21: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<1, 0, 2>), grid, blk, 0, st, m->d, rb, logits, labels); break;
        case 22: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<1, 0, 3>), grid, blk, 0, st, m->d, rb, logits, labels); break;
        case 23: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<1, 0, 4>), grid, blk, 0, st, m->d, rb, logits, labels); break;
#endif
        default: RD_FAIL(RD_E_UNSUPPORTED, "rd_classify: variant %d not available in this build", m->variant);
        }
    }
    RD_HIP(hipGetLastError());
    if (ev) { RD_HIP(hipEventRecord(ev[1], st)); m->prof_count++; }
    if (join_after_launch >= 0) {   // everything issued on `st` from here on sees the final results of the group just evaluated
        rc = rd_async_join(m, st, join_after_launch);
        if (rc) return rc;
    }
    if (m->refine_thresh > 0.0f) {
        if (!deferred) return rd_refine_launch(m, rb, logits, labels, nullptr, m->refine_thresh, st);
        auto &pd = m->q_pend[m->q_cur][m->q_npend[m->q_cur]++];   // (the recurrence kernel's epilogue recorded the candidates)
        pd.p[0] = arena; pd.p[1] = seq_off; pd.p[2] = seq_len; pd.p[3] = logits; pd.p[4] = labels;
        pd.n = n; pd.max_len = max_len; pd.sem = m->semantics; pd.thresh = m->refine_thresh;
        m->q_calls++;
    }
    return RD_OK;
}

int rd_pair_fuse(const float *logits1, const float *logits2, int64_t n, int32_t ensure_mode, int8_t *pair_labels,
                 uint64_t *counts, void *stream) {
    if (n < 0 || ensure_mode < 0 || ensure_mode > 3) RD_FAIL(RD_E_INVALID, "rd_pair_fuse: bad n or ensure_mode");
    if (n == 0) return RD_OK;
    if (!logits1 || !logits2 || !pair_labels) RD_FAIL(RD_E_INVALID, "rd_pair_fuse: null pointer");
    // few, fat workgroups: every workgroup ends with three global atomics on the same counters
    int64_t nb = (n + 2047) / 2048;
    if (nb > 1024) nb = 1024;
    const bool vec = (((uintptr_t)logits1 | (uintptr_t)logits2) & 15) == 0 && ((uintptr_t)pair_labels & 1) == 0;
    if (vec)
        hipLaunchKernelGGL(rd_pair_fuse_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float2 *)logits1,
                           (const float2 *)logits2, n, ensure_mode, pair_labels, counts);
    else
        hipLaunchKernelGGL(rd_pair_fuse_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float2 *)logits1,
                           (const float2 *)logits2, n, ensure_mode, pair_labels, counts);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_count_labels(const uint8_t *labels, int64_t n, uint64_t *counts, void *stream) {
    if (n < 0) RD_FAIL(RD_E_INVALID, "rd_count_labels: bad n");
    if (n == 0) return RD_OK;
    if (!labels || !counts) RD_FAIL(RD_E_INVALID, "rd_count_labels: null pointer");
    int64_t nb = (n + 16383) / 16384;
    if (nb > 1024) nb = 1024;
    if (((uintptr_t)labels & 15) == 0)
        hipLaunchKernelGGL(rd_count_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, labels, n, counts);
    else
        hipLaunchKernelGGL(rd_count_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, labels, n, counts);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_encode_codes(const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n, int32_t max_len,
                    int32_t stride, uint8_t *codes, void *stream) {
    if (n < 0 || max_len < 1 || stride < max_len) RD_FAIL(RD_E_INVALID, "rd_encode_codes: bad n/max_len/stride");
    if (n == 0) return RD_OK;
    if (!arena || !seq_off || !seq_len || !codes) RD_FAIL(RD_E_INVALID, "rd_encode_codes: null pointer");
    if ((int64_t)ENC_R * stride > 0x7fffffffLL) RD_FAIL(RD_E_INVALID, "rd_encode_codes: stride too large");
    int64_t nb = (n + ENC_R - 1) / ENC_R;
    if (nb > 256 * 64) nb = 256 * 64;
    if (((uintptr_t)codes & 3) == 0 && (stride & 3) == 0)   // every piece of every row starts on a dword boundary
        hipLaunchKernelGGL(rd_encode_codes_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, arena, seq_off, seq_len,
                           n, max_len, stride, codes);
    else
        hipLaunchKernelGGL(rd_encode_codes_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, arena, seq_off, seq_len,
                           n, max_len, stride, codes);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_encode_onehot_padded(const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n,
                            int32_t max_len, float *onehot, void *stream) {
    if (n < 0 || max_len < 1) RD_FAIL(RD_E_INVALID, "rd_encode_onehot_padded: bad n/max_len");
    if (n == 0) return RD_OK;
    if (!arena || !seq_off || !seq_len || !onehot) RD_FAIL(RD_E_INVALID, "rd_encode_onehot_padded: null pointer");
    if (max_len > MAX_LEN_LIMIT) RD_FAIL(RD_E_INVALID, "rd_encode_onehot_padded: max_len=%d out of range [1,%d]", max_len, MAX_LEN_LIMIT);
    int64_t nb = (n + ENC_R - 1) / ENC_R;
    if (nb > 256 * 64) nb = 256 * 64;
    hipLaunchKernelGGL(rd_encode_onehot_padded_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, arena, seq_off,
                       seq_len, n, max_len, (f32x4 *)onehot);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_pack_plan(const int32_t *seq_len, int64_t n, int32_t max_len, int64_t *sorted_idx, int64_t *unsorted_idx,
                 int64_t *batch_sizes, int64_t *total_steps, void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 1 || max_len < 1 || max_len > MAX_LEN_LIMIT) RD_FAIL(RD_E_INVALID, "rd_pack_plan: bad n/max_len");
    if (!seq_len || !sorted_idx || !unsorted_idx || !batch_sizes || !total_steps || !workspace)
        RD_FAIL(RD_E_INVALID, "rd_pack_plan: null pointer");
    int32_t *order = nullptr;
    int64_t *len_start = nullptr;
    return run_sort(seq_len, n, max_len, workspace, workspace_bytes, order, sorted_idx, unsorted_idx, batch_sizes, total_steps,
                    len_start, (hipStream_t)stream);
}

int rd_pack_onehot(const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n, int32_t max_len,
                   const int64_t *sorted_idx, const int64_t *batch_sizes, float *data, void *stream) {
    if (n < 1 || max_len < 1) RD_FAIL(RD_E_INVALID, "rd_pack_onehot: bad n/max_len");
    if (!arena || !seq_off || !seq_len || !sorted_idx || !batch_sizes || !data) RD_FAIL(RD_E_INVALID, "rd_pack_onehot: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = (n + ENC_R - 1) / ENC_R;
    if (nb > 0x7fffffffLL) RD_FAIL(RD_E_INVALID, "rd_pack_onehot: n too large");
    hipLaunchKernelGGL(rd_pack_onehot_kernel, dim3((unsigned)nb), dim3(256), 0, st, arena, seq_off, seq_len, n, max_len, sorted_idx,
                       batch_sizes, (f32x4 *)data);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

// ---- device-side gzip of the label-partitioned records (rd_deflate.hpp) ------------------------------------------------------------
namespace {
struct GzPlan {
    int nb;                 // scan blocks
    int64_t cap_members;    // members the chunk can have at most (every record selected)
    int grid;               // workgroups of the deflate kernel
    size_t off_bytes, bsum_bytes, plain_bytes, toks_bytes, slots_bytes, msize_bytes, moff_bytes, total;
};
GzPlan gz_plan(int64_t n, int64_t text_bytes) {
    GzPlan p;
    p.nb = (int)((n + 1 + GZ_SCAN_ITEMS - 1) / GZ_SCAN_ITEMS);
    p.cap_members = (text_bytes + GZ_MEMBER - 1) / GZ_MEMBER;
    if (p.cap_members < 1) p.cap_members = 1;
    p.grid = (int)(p.cap_members < GZ_MAX_GRID ? p.cap_members : GZ_MAX_GRID);
    p.off_bytes = align_up((size_t)(n + 1) * 8, 256);
    p.bsum_bytes = align_up((size_t)p.nb * 8, 256);
    p.plain_bytes = align_up((size_t)p.cap_members * GZ_MEMBER + 256, 256);
    p.toks_bytes = align_up((size_t)p.grid * GZ_MEMBER * 4, 256);
    p.slots_bytes = (size_t)p.cap_members * GZ_SLOT;
    p.msize_bytes = align_up((size_t)p.cap_members * 4, 256);
    p.moff_bytes = align_up((size_t)p.cap_members * 8, 256);
    p.total = p.off_bytes + p.bsum_bytes + p.plain_bytes + p.toks_bytes + p.slots_bytes + p.msize_bytes + p.moff_bytes;
    return p;
}
}  // namespace

size_t rd_gz_workspace_bytes(int64_t n, int64_t text_bytes) {
    if (n < 0 || text_bytes < 0) return 0;
    return gz_plan(n, text_bytes).total;
}

size_t rd_gz_out_bound(int64_t text_bytes) {
    if (text_bytes < 0) return 0;
    const int64_t members = (text_bytes + GZ_MEMBER - 1) / GZ_MEMBER;
    return (size_t)members * (GZ_HDR + 5 + GZ_TRL) + (size_t)text_bytes;   // every member as a stored block
}

int rd_gz_eof_block(uint8_t *dst, size_t cap) {   // BGZF's end-of-file marker: an empty member (28 bytes)
    static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!dst || cap < sizeof(eof)) RD_FAIL(RD_E_INVALID, "rd_gz_eof_block: need 28 bytes");
    memcpy(dst, eof, sizeof(eof));
    return (int)sizeof(eof);
}

int rd_gz_compress_selected(const uint8_t *text, int64_t text_bytes, const int64_t *rec_start, const int8_t *labels, int64_t n,
                            int32_t label, uint8_t *out, size_t out_cap, int64_t *info, void *workspace, size_t workspace_bytes,
                            void *stream) {
    if (n < 0 || n > 0x7fffffffLL || text_bytes < 0) RD_FAIL(RD_E_INVALID, "rd_gz_compress_selected: bad n or text_bytes");
    if (!info) RD_FAIL(RD_E_INVALID, "rd_gz_compress_selected: null info");
    if (label < -128 || label > 127) RD_FAIL(RD_E_INVALID, "rd_gz_compress_selected: label %d is not an int8 value", label);
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        RD_HIP(hipMemsetAsync(info, 0, 4 * sizeof(int64_t), st));
        return RD_OK;
    }
    if ((!text && text_bytes > 0) || !rec_start || !labels || !out || !workspace) RD_FAIL(RD_E_INVALID, "rd_gz_compress_selected: null pointer");
    if (((uintptr_t)workspace & 255) || ((uintptr_t)out & 255)) RD_FAIL(RD_E_INVALID, "rd_gz_compress_selected: workspace and out must be 256-byte aligned");
    const GzPlan p = gz_plan(n, text_bytes);
    if (workspace_bytes < p.total) RD_FAIL(RD_E_WORKSPACE, "rd_gz_compress_selected: workspace too small: %zu < %zu", workspace_bytes, p.total);
    char *w = (char *)workspace;
    int64_t *out_off = (int64_t *)w; w += p.off_bytes;
    int64_t *bsum = (int64_t *)w; w += p.bsum_bytes;
    uint8_t *plain = (uint8_t *)w; w += p.plain_bytes;
    uint32_t *toks = (uint32_t *)w; w += p.toks_bytes;
    uint8_t *slots = (uint8_t *)w; w += p.slots_bytes;
    uint32_t *msize = (uint32_t *)w; w += p.msize_bytes;
    int64_t *moff = (int64_t *)w;
    hipLaunchKernelGGL(rd_gz_sel_sum_kernel, dim3(p.nb), dim3(256), 0, st, rec_start, labels, n, label, bsum);
    hipLaunchKernelGGL(rd_gz_sel_base_kernel, dim3(1), dim3(256), 0, st, bsum, p.nb, info, text_bytes);
    hipLaunchKernelGGL(rd_gz_sel_off_kernel, dim3(p.nb), dim3(256), 0, st, rec_start, labels, n, label, bsum, out_off);
    hipLaunchKernelGGL(rd_gz_pack_kernel, dim3((unsigned)((n + GZ_PACK_RECS - 1) / GZ_PACK_RECS)), dim3(256), 0, st, text, rec_start, out_off, n, plain, info);
    hipLaunchKernelGGL(rd_gz_deflate_kernel, dim3(p.grid), dim3(GZ_THREADS), 0, st, plain, info, toks, slots, msize);
    hipLaunchKernelGGL(rd_gz_moff_kernel, dim3(1), dim3(256), 0, st, msize, moff, info);
    hipLaunchKernelGGL(rd_gz_compact_kernel, dim3(p.grid), dim3(256), 0, st, slots, msize, moff, info, out, (int64_t)out_cap);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_gz_inflate_members(const uint8_t *comp, int64_t comp_bytes, const rd_gz_member *members, int64_t n, uint8_t *text, int64_t text_bytes,
                          uint32_t *status, void *stream) {
    static_assert(sizeof(rd_gz_member) == sizeof(GzMemberIn), "rd_gz_member layout");
    if (n < 0 || comp_bytes < 0 || text_bytes < 0) RD_FAIL(RD_E_INVALID, "rd_gz_inflate_members: bad size");
    if (n == 0) return RD_OK;
    if (!comp || !members || !status || (!text && text_bytes > 0)) RD_FAIL(RD_E_INVALID, "rd_gz_inflate_members: null pointer");
    int64_t grid = (n + GZI_WAVES - 1) / GZI_WAVES;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(rd_gz_inflate_kernel, dim3((unsigned)grid), dim3(64 * GZI_WAVES), 0, (hipStream_t)stream, comp, comp_bytes,
                       (const GzMemberIn *)members, n, text, text_bytes, status);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

// ---- one DEFLATE stream inflated on the device (rd_inflate_stream.hpp) ----------------------------------------------------------------------
size_t rd_gz_stream_workspace_bytes(int64_t data_bytes, int32_t section_bytes, int32_t cap_syms, int64_t text_cap) {
    if (data_bytes < 0 || section_bytes < 1024 || cap_syms < 1024 || text_cap < 0) return 0;
    return gzs_plan(data_bytes, section_bytes, cap_syms, text_cap).total;
}

int rd_gz_stream_inflate(const uint8_t *comp, int64_t comp_bytes, int64_t data_bytes, int64_t valid_bytes, int32_t section_bytes, int32_t cap_syms,
                         uint32_t first_start_bit, const rd_gzs_state *carry, int64_t carry_delta_bits, int32_t at_eof, const uint8_t *win_in,
                         uint8_t *win_out, uint8_t *text, int64_t text_cap, rd_gzs_state *state, void *workspace, size_t workspace_bytes, void *stream) {
    if (!comp || !win_out || !text || !state || !workspace) RD_FAIL(RD_E_INVALID, "rd_gz_stream_inflate: null pointer");
    if (((uintptr_t)comp & 3) || ((uintptr_t)workspace & 255) || ((uintptr_t)text & 7))
        RD_FAIL(RD_E_INVALID, "rd_gz_stream_inflate: comp must be 4-byte aligned, text 8-byte aligned, workspace 256-byte aligned");
    if (data_bytes <= 0 || valid_bytes < data_bytes || comp_bytes < valid_bytes || valid_bytes >= (1LL << 28) || section_bytes < 1024 || (section_bytes & 3) ||
        cap_syms < 1024 || text_cap < 0)
        RD_FAIL(RD_E_INVALID, "rd_gz_stream_inflate: bad sizes (a batch holds < 256 MiB of compressed bytes)");
    const GzsPlan p = gzs_plan(data_bytes, section_bytes, cap_syms, text_cap);
    if (workspace_bytes < p.total) RD_FAIL(RD_E_WORKSPACE, "rd_gz_stream_inflate: workspace too small: %zu < %zu", workspace_bytes, p.total);
    char *w = (char *)workspace;
    uint32_t *found = (uint32_t *)w; w += p.found_bytes;
    GzsSec *sec = (GzsSec *)w; w += p.sec_bytes;
    int64_t *off = (int64_t *)w; w += p.off_bytes;
    int32_t *wslot = (int32_t *)w; w += p.wslot_bytes;
    int32_t *plist = (int32_t *)w; w += p.plist_bytes;
    uint32_t *tcrc = (uint32_t *)w; w += p.crc_bytes;
    uint16_t *windows16 = (uint16_t *)w; w += p.windows_bytes;
    uint16_t *gmaps = (uint16_t *)w; w += p.gmaps_bytes;
    uint16_t *gwin = (uint16_t *)w; w += p.gwin_bytes;
    uint16_t *syms = (uint16_t *)w;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t end_bits = (uint32_t)(valid_bytes * 8), sec_bits = (uint32_t)section_bytes * 8u;
    GzsState *S = (GzsState *)state;
    const GzsState *C = (const GzsState *)carry;
    hipLaunchKernelGGL(rd_gzs_search_kernel, dim3((unsigned)((p.nsec + 1 + GZS_WAVES - 1) / GZS_WAVES)), dim3(64 * GZS_WAVES), 0, st, comp, comp_bytes, end_bits, sec_bits,
                       p.nsec, first_start_bit, C, carry_delta_bits, found);
    hipLaunchKernelGGL(rd_gzs_decode_kernel, dim3((unsigned)((p.nsec + GZS_WAVES - 1) / GZS_WAVES)), dim3(64 * GZS_WAVES), 0, st, comp, comp_bytes, end_bits, p.nsec, found,
                       syms, (int)cap_syms, sec);
    hipLaunchKernelGGL(rd_gzs_scan_kernel, dim3(1), dim3(64), 0, st, sec, found, p.nsec, (int)at_eof, text_cap, off, wslot, plist, S, C, 0);
    // the window chain on symbols: the groups of sections side by side, then the groups in order (rd_inflate_stream.hpp)
    hipLaunchKernelGGL(rd_gzs_symwin_kernel, dim3((unsigned)p.ngroups), dim3(1024), 0, st, syms, (int)cap_syms, sec, plist, wslot, p.nsec, S, windows16, gmaps);
    hipLaunchKernelGGL(rd_gzs_chain_kernel, dim3(1), dim3(1024), 0, st, gmaps, wslot, p.nsec, S, win_in, (const uint16_t *)nullptr, 1, gwin, (uint16_t *)nullptr, win_out);
    hipLaunchKernelGGL(rd_gzs_resolve_kernel<false>, dim3((unsigned)(p.nsec * p.tiles_per_sec)), dim3(256), 0, st, syms, (int)cap_syms, sec, found, off, wslot,
                       p.tiles_per_sec, windows16, gwin, S, text, (uint16_t *)nullptr);
    hipLaunchKernelGGL(rd_gzs_crc_kernel, dim3((unsigned)((p.ctiles + 3) / 4)), dim3(256), 0, st, text, S, tcrc);
    hipLaunchKernelGGL(rd_gzs_fold_kernel, dim3(1), dim3(64), 0, st, tcrc, S);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

// ---- a range of one DEFLATE stream: symbols first, bytes once the window in front of the range is known (rd_inflate_stream.hpp) -----------
size_t rd_gz_range_workspace_bytes(int64_t data_bytes, int32_t section_bytes, int32_t cap_syms, int64_t text_cap) {
    if (data_bytes < 0 || section_bytes < 1024 || cap_syms < 1024 || text_cap < 0) return 0;
    return gzs_plan(data_bytes, section_bytes, cap_syms, text_cap).total;
}

int rd_gz_range_decode(const uint8_t *comp, int64_t comp_bytes, int64_t data_bytes, int64_t valid_bytes, int32_t section_bytes, int32_t cap_syms,
                       uint32_t first_start_bit, const rd_gzs_state *carry, int64_t carry_delta_bits, int32_t at_eof, const uint16_t *map_in, uint16_t *map_out,
                       uint16_t *sym_text, int64_t text_cap, rd_gzs_state *state, void *workspace, size_t workspace_bytes, void *stream) {
    if (!comp || !map_out || !sym_text || !state || !workspace) RD_FAIL(RD_E_INVALID, "rd_gz_range_decode: null pointer");
    if (((uintptr_t)comp & 3) || ((uintptr_t)workspace & 255) || ((uintptr_t)sym_text & 15))
        RD_FAIL(RD_E_INVALID, "rd_gz_range_decode: comp must be 4-byte aligned, sym_text 16-byte aligned, workspace 256-byte aligned");
    if (data_bytes <= 0 || valid_bytes < data_bytes || comp_bytes < valid_bytes || valid_bytes >= (1LL << 28) || section_bytes < 1024 || (section_bytes & 3) ||
        cap_syms < 1024 || text_cap < 0)
        RD_FAIL(RD_E_INVALID, "rd_gz_range_decode: bad sizes (a batch holds < 256 MiB of compressed bytes)");
    if ((carry == nullptr) != (map_in == nullptr)) RD_FAIL(RD_E_INVALID, "rd_gz_range_decode: carry and map_in go together (both null: the range's first batch)");
    const GzsPlan p = gzs_plan(data_bytes, section_bytes, cap_syms, text_cap);
    if (workspace_bytes < p.total) RD_FAIL(RD_E_WORKSPACE, "rd_gz_range_decode: workspace too small: %zu < %zu", workspace_bytes, p.total);
    char *w = (char *)workspace;
    uint32_t *found = (uint32_t *)w; w += p.found_bytes;
    GzsSec *sec = (GzsSec *)w; w += p.sec_bytes;
    int64_t *off = (int64_t *)w; w += p.off_bytes;
    int32_t *wslot = (int32_t *)w; w += p.wslot_bytes;
    int32_t *plist = (int32_t *)w; w += p.plist_bytes;
    w += p.crc_bytes;
    uint16_t *windows16 = (uint16_t *)w; w += p.windows_bytes;
    uint16_t *gmaps = (uint16_t *)w; w += p.gmaps_bytes;
    uint16_t *gwin = (uint16_t *)w; w += p.gwin_bytes;
    uint16_t *syms = (uint16_t *)w;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t end_bits = (uint32_t)(valid_bytes * 8), sec_bits = (uint32_t)section_bytes * 8u;
    GzsState *S = (GzsState *)state;
    const GzsState *C = (const GzsState *)carry;
    const int search0 = (C == nullptr && first_start_bit == GZS_SEARCH) ? 1 : 0;
    hipLaunchKernelGGL(rd_gzs_search_kernel, dim3((unsigned)((p.nsec + 1 + GZS_WAVES - 1) / GZS_WAVES)), dim3(64 * GZS_WAVES), 0, st, comp, comp_bytes, end_bits, sec_bits,
                       p.nsec, first_start_bit, C, carry_delta_bits, found);
    hipLaunchKernelGGL(rd_gzs_decode_kernel, dim3((unsigned)((p.nsec + GZS_WAVES - 1) / GZS_WAVES)), dim3(64 * GZS_WAVES), 0, st, comp, comp_bytes, end_bits, p.nsec, found,
                       syms, (int)cap_syms, sec);
    hipLaunchKernelGGL(rd_gzs_scan_kernel, dim3(1), dim3(64), 0, st, sec, found, p.nsec, (int)at_eof, text_cap, off, wslot, plist, S, C, search0);
    hipLaunchKernelGGL(rd_gzs_symwin_kernel, dim3((unsigned)p.ngroups), dim3(1024), 0, st, syms, (int)cap_syms, sec, plist, wslot, p.nsec, S, windows16, gmaps);
    hipLaunchKernelGGL(rd_gzs_chain_kernel, dim3(1), dim3(1024), 0, st, gmaps, wslot, p.nsec, S, (const uint8_t *)nullptr, map_in, 0, gwin, map_out, (uint8_t *)nullptr);
    hipLaunchKernelGGL(rd_gzs_resolve_kernel<true>, dim3((unsigned)(p.nsec * p.tiles_per_sec)), dim3(256), 0, st, syms, (int)cap_syms, sec, found, off, wslot,
                       p.tiles_per_sec, windows16, gwin, S, (uint8_t *)nullptr, sym_text);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

size_t rd_gz_range_resolve_workspace_bytes(int64_t n) { return n < 0 ? 0 : ((size_t)(n / GZS_CTILE + 2) * 4 + 255) / 256 * 256; }

int rd_gz_range_resolve(const uint16_t *sym_text, int64_t n, const uint8_t *window, uint32_t win_valid, uint8_t *text, rd_gzs_state *state, void *workspace,
                        size_t workspace_bytes, void *stream) {
    if (n < 0 || !state || (n > 0 && (!sym_text || !text || !workspace))) RD_FAIL(RD_E_INVALID, "rd_gz_range_resolve: bad argument");
    if (win_valid > 0 && !window) RD_FAIL(RD_E_INVALID, "rd_gz_range_resolve: win_valid > 0 needs a window");
    if (((uintptr_t)sym_text & 15) || ((uintptr_t)text & 15)) RD_FAIL(RD_E_INVALID, "rd_gz_range_resolve: sym_text and text must be 16-byte aligned");
    if (workspace_bytes < rd_gz_range_resolve_workspace_bytes(n)) RD_FAIL(RD_E_WORKSPACE, "rd_gz_range_resolve: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    GzsState *S = (GzsState *)state;
    int64_t grid = n / (8 * 256 * 4) + 1;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(rd_gzs_symtext_kernel, dim3((unsigned)grid), dim3(256), 0, st, sym_text, n, window, win_valid, text, S);
    const int ctiles = (int)((n + GZS_CTILE - 1) / GZS_CTILE) + 1;
    hipLaunchKernelGGL(rd_gzs_crc_kernel, dim3((unsigned)((ctiles + 3) / 4)), dim3(256), 0, st, text, S, (uint32_t *)workspace);
    hipLaunchKernelGGL(rd_gzs_fold_kernel, dim3(1), dim3(64), 0, st, (const uint32_t *)workspace, S);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

// ---- streams restricted to a set of compute units --------------------------------------------------------------------------------------------
int rd_stream_create(int device, const uint32_t *cu_mask, int words, int priority, void **stream) {
    if (!stream || words < 0 || (words > 0 && !cu_mask)) RD_FAIL(RD_E_INVALID, "rd_stream_create: bad argument");
    RD_HIP(hipSetDevice(device));
    hipStream_t st = nullptr;
    if (words > 0) {
        RD_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, cu_mask));      // (no priority argument in this entry point: default priority)
    } else {
        int lo = 0, hi = 0;
        RD_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        RD_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, priority < 0 ? hi : lo));
    }
    *stream = (void *)st;
    return RD_OK;
}

int rd_stream_destroy(void *stream) {
    if (stream) RD_HIP(hipStreamDestroy((hipStream_t)stream));
    return RD_OK;
}

int rd_copy_bytes(void *dst, const void *src, int64_t n, int32_t workgroups, void *stream) {
    if (n < 0 || (n > 0 && (!dst || !src))) RD_FAIL(RD_E_INVALID, "rd_copy_bytes: bad argument");
    if (n == 0) return RD_OK;
    int64_t grid = workgroups > 0 ? workgroups : 8;
    const int64_t need = n / (16 * COPY_THREADS) + 1;
    if (grid > need) grid = need;
    hipLaunchKernelGGL(rd_copy_kernel, dim3((unsigned)grid), dim3(COPY_THREADS), 0, (hipStream_t)stream, (uint8_t *)dst, (const uint8_t *)src, n);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

// ---- FASTQ record index on the device (rd_fastq_index.hpp) ----------------------------------------------------------------------------------
size_t rd_fastq_index_workspace_bytes(int64_t text_end) {
    if (text_end < 0 || text_end >= 0x7fffffffLL) return 0;
    return fq_plan(text_end).total;
}

int rd_fastq_index(uint8_t *text, int64_t pad, int64_t end, const uint8_t *prev_text, const rd_fq_summary *prev, int32_t final, int32_t *line_end,
                   int64_t cap_lines, rd_fq_summary *summary, void *workspace, size_t workspace_bytes, void *stream) {
    if (!text || !line_end || !summary || !workspace) RD_FAIL(RD_E_INVALID, "rd_fastq_index: null pointer");
    if (pad < 0 || end < pad || end >= 0x7fffffffLL - 64 || cap_lines < 0) RD_FAIL(RD_E_INVALID, "rd_fastq_index: bad pad / end / cap_lines (a batch buffer is < 2 GiB)");
    if ((prev == nullptr) != (prev_text == nullptr)) RD_FAIL(RD_E_INVALID, "rd_fastq_index: prev and prev_text go together");
    if (((uintptr_t)text & 63) || ((uintptr_t)workspace & 255)) RD_FAIL(RD_E_INVALID, "rd_fastq_index: text must be 64-byte aligned, workspace 256-byte aligned");
    const FqPlan p = fq_plan(end);
    if (workspace_bytes < p.total) RD_FAIL(RD_E_WORKSPACE, "rd_fastq_index: workspace too small: %zu < %zu", workspace_bytes, p.total);
    hipStream_t st = (hipStream_t)stream;
    uint32_t *tiles = (uint32_t *)workspace;
    FqSummary *sum = (FqSummary *)summary;
    hipLaunchKernelGGL(rd_fq_begin_kernel, dim3(1), dim3(FQ_THREADS), 0, st, text, pad, end, prev_text, (const FqSummary *)prev, (int)final, sum);
    hipLaunchKernelGGL(rd_fq_count_kernel, dim3(p.ntiles), dim3(FQ_THREADS), 0, st, text, sum, tiles);
    hipLaunchKernelGGL(rd_fq_scan_kernel, dim3(1), dim3(FQ_THREADS), 0, st, tiles, p.ntiles, sum, cap_lines);
    hipLaunchKernelGGL(rd_fq_fill_kernel, dim3(p.ntiles), dim3(FQ_THREADS), 0, st, text, sum, tiles, line_end);
    int grid = (int)((end - pad) / (64 * FQ_THREADS)) + 1;      // one thread per ~64 bytes of new text is more than one per record
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(rd_fq_check_kernel, dim3(grid), dim3(FQ_THREADS), 0, st, text, line_end, sum, (int)final);
    hipLaunchKernelGGL(rd_fq_verdict_kernel, dim3(1), dim3(1), 0, st, sum);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_fastq_gather(const uint8_t *text, const int32_t *line_end, const rd_fq_summary *summary, int64_t rec_lo, int64_t rec_hi, int64_t max_bytes,
                    uint8_t *out_text, int64_t out_cap, const int64_t *cursor_in, int64_t *cursor_out, int64_t *rec_start, int64_t *seq_off,
                    int32_t *seq_len, void *stream) {
    if (!text || !line_end || !summary || !out_text || !cursor_in || !cursor_out || !rec_start || !seq_off || !seq_len)
        RD_FAIL(RD_E_INVALID, "rd_fastq_gather: null pointer");
    if (rec_lo < 0 || rec_hi < rec_lo || max_bytes < 0 || out_cap < 0 || cursor_in == cursor_out) RD_FAIL(RD_E_INVALID, "rd_fastq_gather: bad range");
    if ((uintptr_t)out_text & 15) RD_FAIL(RD_E_INVALID, "rd_fastq_gather: out_text must be 16-byte aligned");
    int64_t grid = max_bytes / (16 * FQ_THREADS * 4) + 1;       // four 16-byte pieces per thread
    const int64_t grid_r = (rec_hi - rec_lo) / FQ_THREADS + 1;
    if (grid < grid_r) grid = grid_r;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(rd_fq_gather_kernel, dim3((unsigned)grid), dim3(FQ_THREADS), 0, (hipStream_t)stream, text, line_end, (const FqSummary *)summary, rec_lo,
                       rec_hi, out_text, out_cap, cursor_in, cursor_out, rec_start, seq_off, seq_len);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_fastq_sample(const int32_t *line_end, const rd_fq_summary *summary, int64_t every, int32_t *samples, int64_t cap, void *stream) {
    if (!line_end || !summary || !samples || every < 1 || cap < 0) RD_FAIL(RD_E_INVALID, "rd_fastq_sample: bad argument");
    if (cap == 0) return RD_OK;
    int64_t grid = cap / FQ_THREADS + 1;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(rd_fq_sample_kernel, dim3((unsigned)grid), dim3(FQ_THREADS), 0, (hipStream_t)stream, line_end, (const FqSummary *)summary, every, samples, cap);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_fastq_strip_mark(const uint8_t *text, const int32_t *line_end, const rd_fq_summary *summary, int64_t max_lines, uint8_t *del, void *stream) {
    if (!text || !line_end || !summary || !del || max_lines < 0) RD_FAIL(RD_E_INVALID, "rd_fastq_strip_mark: bad argument");
    int64_t grid = max_lines / FQ_THREADS + 1;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(rd_fq_strip_mark_kernel, dim3((unsigned)grid), dim3(FQ_THREADS), 0, (hipStream_t)stream, text, line_end, (const FqSummary *)summary, del);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

size_t rd_fasta_index_workspace_bytes(int64_t text_end, int64_t cap_lines) {
    if (text_end < 0 || text_end >= 0x7fffffffLL || cap_lines < 0) return 0;
    return fa_plan(text_end, cap_lines).total;
}

int rd_fasta_index(uint8_t *text, int64_t pad, int64_t end, const uint8_t *prev_text, const rd_fq_summary *prev, int32_t final, int32_t *line_end,
                   int64_t cap_lines, uint8_t *norm, int64_t norm_cap, int64_t *rec_tab, int32_t *hdr_tab, int64_t cap_records, rd_fq_summary *summary,
                   void *workspace, size_t workspace_bytes, void *stream) {
    if (!text || !line_end || !norm || !rec_tab || !hdr_tab || !summary || !workspace) RD_FAIL(RD_E_INVALID, "rd_fasta_index: null pointer");
    if (pad < 0 || end < pad || end >= 0x7fffffffLL - 64 || cap_lines < 0 || norm_cap < 0 || cap_records < 1)
        RD_FAIL(RD_E_INVALID, "rd_fasta_index: bad pad / end / cap_lines / norm_cap / cap_records (a batch buffer is < 2 GiB)");
    if ((prev == nullptr) != (prev_text == nullptr)) RD_FAIL(RD_E_INVALID, "rd_fasta_index: prev and prev_text go together");
    if (((uintptr_t)text & 63) || ((uintptr_t)workspace & 255) || ((uintptr_t)norm & 15))
        RD_FAIL(RD_E_INVALID, "rd_fasta_index: text must be 64-byte aligned, workspace 256-byte aligned, norm 16-byte aligned");
    const FaPlan p = fa_plan(end, cap_lines);
    if (workspace_bytes < p.total) RD_FAIL(RD_E_WORKSPACE, "rd_fasta_index: workspace too small: %zu < %zu", workspace_bytes, p.total);
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    uint32_t *tiles = (uint32_t *)ws;
    int32_t *info_a = (int32_t *)(ws + p.a_off);
    uint32_t *info_k = (uint32_t *)(ws + p.k_off);
    unsigned long long *blk = (unsigned long long *)(ws + p.blk_off);
    FaScratch *sc = (FaScratch *)(ws + p.sc_off);
    FqSummary *sum = (FqSummary *)summary;
    hipLaunchKernelGGL(rd_fq_begin_kernel, dim3(1), dim3(FQ_THREADS), 0, st, text, pad, end, prev_text, (const FqSummary *)prev, (int)final, sum);
    hipLaunchKernelGGL(rd_fq_count_kernel, dim3(p.fq.ntiles), dim3(FQ_THREADS), 0, st, text, sum, tiles);
    hipLaunchKernelGGL(rd_fq_scan_kernel, dim3(1), dim3(FQ_THREADS), 0, st, tiles, p.fq.ntiles, sum, cap_lines);
    hipLaunchKernelGGL(rd_fq_fill_kernel, dim3(p.fq.ntiles), dim3(FQ_THREADS), 0, st, text, sum, tiles, line_end);
    hipLaunchKernelGGL(rd_fa_init_kernel, dim3(1), dim3(FQ_THREADS), 0, st, sc);
    hipLaunchKernelGGL(rd_fa_lines_kernel, dim3(p.nblk), dim3(FQ_THREADS), 0, st, text, line_end, sum, info_a, info_k, blk, sc);
    hipLaunchKernelGGL(rd_fa_base_kernel, dim3(1), dim3(FQ_THREADS), 0, st, blk, line_end, sum, sc, (int)final, norm_cap, cap_records, norm, rec_tab, hdr_tab);
    hipLaunchKernelGGL(rd_fa_emit_kernel, dim3(p.nblk), dim3(FQ_THREADS), 0, st, text, info_a, info_k, blk, sum, sc, (int)final, norm, rec_tab, hdr_tab);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_fasta_gather(const uint8_t *norm, const int64_t *rec_tab, const int32_t *hdr_tab, const rd_fq_summary *summary, int64_t rec_lo, int64_t rec_hi,
                    int64_t max_bytes, uint8_t *out_text, int64_t out_cap, const int64_t *cursor_in, int64_t *cursor_out, int64_t *rec_start,
                    int64_t *seq_off, int32_t *seq_len, void *stream) {
    if (!norm || !rec_tab || !hdr_tab || !summary || !out_text || !cursor_in || !cursor_out || !rec_start || !seq_off || !seq_len)
        RD_FAIL(RD_E_INVALID, "rd_fasta_gather: null pointer");
    if (rec_lo < 0 || rec_hi < rec_lo || max_bytes < 0 || out_cap < 0 || cursor_in == cursor_out) RD_FAIL(RD_E_INVALID, "rd_fasta_gather: bad range");
    if ((uintptr_t)out_text & 15) RD_FAIL(RD_E_INVALID, "rd_fasta_gather: out_text must be 16-byte aligned");
    int64_t grid = max_bytes / (16 * FQ_THREADS * 4) + 1;
    const int64_t grid_r = (rec_hi - rec_lo) / FQ_THREADS + 1;
    if (grid < grid_r) grid = grid_r;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(rd_fa_gather_kernel, dim3((unsigned)grid), dim3(FQ_THREADS), 0, (hipStream_t)stream, norm, rec_tab, hdr_tab, (const FqSummary *)summary,
                       rec_lo, rec_hi, out_text, out_cap, cursor_in, cursor_out, rec_start, seq_off, seq_len);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_fasta_sample(const int64_t *rec_tab, const rd_fq_summary *summary, int64_t every, int32_t *samples, int64_t cap, void *stream) {
    if (!rec_tab || !summary || !samples || every < 1 || cap < 0) RD_FAIL(RD_E_INVALID, "rd_fasta_sample: bad argument");
    if (cap == 0) return RD_OK;
    int64_t grid = cap / FQ_THREADS + 1;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(rd_fa_sample_kernel, dim3((unsigned)grid), dim3(FQ_THREADS), 0, (hipStream_t)stream, rec_tab, (const FqSummary *)summary, every, samples, cap);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

// the selected records of a chunk as one contiguous text (the selection scan and the pack kernel of the device gzip, without the deflate)
size_t rd_select_workspace_bytes(int64_t n) {
    if (n < 0) return 0;
    const GzPlan p = gz_plan(n, 0);
    return p.off_bytes + p.bsum_bytes;
}

int rd_select_pack(const uint8_t *text, int64_t text_bytes, const int64_t *rec_start, const int8_t *labels, int64_t n, int32_t label, uint8_t *out,
                   size_t out_cap, int64_t *info, void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 0 || n > 0x7fffffffLL || text_bytes < 0) RD_FAIL(RD_E_INVALID, "rd_select_pack: bad n or text_bytes");
    if (!info) RD_FAIL(RD_E_INVALID, "rd_select_pack: null info");
    if (label < -128 || label > 127) RD_FAIL(RD_E_INVALID, "rd_select_pack: label %d is not an int8 value", label);
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        RD_HIP(hipMemsetAsync(info, 0, 4 * sizeof(int64_t), st));
        return RD_OK;
    }
    if ((!text && text_bytes > 0) || !rec_start || !labels || !out || !workspace) RD_FAIL(RD_E_INVALID, "rd_select_pack: null pointer");
    if (((uintptr_t)workspace & 255) || ((uintptr_t)out & 15)) RD_FAIL(RD_E_INVALID, "rd_select_pack: workspace must be 256-byte aligned, out 16-byte aligned");
    const GzPlan p = gz_plan(n, 0);
    if (workspace_bytes < p.off_bytes + p.bsum_bytes) RD_FAIL(RD_E_WORKSPACE, "rd_select_pack: workspace too small: %zu < %zu", workspace_bytes, p.off_bytes + p.bsum_bytes);
    int64_t *out_off = (int64_t *)workspace;
    int64_t *bsum = (int64_t *)((char *)workspace + p.off_bytes);
    const int64_t limit = text_bytes < (int64_t)out_cap ? text_bytes : (int64_t)out_cap;   // more selected bytes than text, or than `out` holds: info[3] = 1, nothing packed
    hipLaunchKernelGGL(rd_gz_sel_sum_kernel, dim3(p.nb), dim3(256), 0, st, rec_start, labels, n, label, bsum);
    hipLaunchKernelGGL(rd_gz_sel_base_kernel, dim3(1), dim3(256), 0, st, bsum, p.nb, info, limit);
    hipLaunchKernelGGL(rd_gz_sel_off_kernel, dim3(p.nb), dim3(256), 0, st, rec_start, labels, n, label, bsum, out_off);
    hipLaunchKernelGGL(rd_gz_pack_kernel, dim3((unsigned)((n + GZ_PACK_RECS - 1) / GZ_PACK_RECS)), dim3(256), 0, st, text, rec_start, out_off, n, out, info);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

#ifdef RD_DIAG
// diagnostic build only (not in include/ribodetector_amd.h): [dev] uint64[32] that receives the cycles per stage of rd_gz_deflate_kernel ([0..15]) and rd_gz_inflate_kernel ([16..31])
RD_API int rd_gz_diag_set_profile(unsigned long long *dev_buf) {
    RD_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_gz_prof), &dev_buf, sizeof(dev_buf)));
    return RD_OK;
}
#endif

}  // extern "C"
