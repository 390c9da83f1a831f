/*
 * ribodetector_amd.h - C ABI of the MI355X-native RiboDetector inference path (librd_hip.so).
 *
 * The reference (hzi-bifo/RiboDetector v0.3.1) is pure Python and has no FFI of its own; the seam it offers
 * is a Python duck type resolved from config.json. Each entry point below names the reference
 * interface it replaces (paths relative to /root/reference/ribodetector). INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer marked [dev] is device (HBM) memory owned by the caller,
 *     [host] is host memory. Nothing is allocated behind the caller's back except inside rd_model_create.
 *   - every launch takes an explicit hipStream_t (passed as void*; 0 = the null stream) and is asynchronous;
 *     the caller synchronises.
 *   - return value: 0 = ok, <0 = error (RD_E_*); rd_last_error() returns a thread-local message.
 *     No C++ exception crosses this boundary.
 *   - reads are handed over as raw ASCII: one byte arena + per-read start offset + per-read length
 *     (exactly what a FASTQ/FASTA chunk in memory looks like). Only the first `max_len` bases of a read are
 *     used (reference detect.py:682,714,717 `read[1][:max_len]`).
 */
#ifndef RIBODETECTOR_AMD_H
#define RIBODETECTOR_AMD_H

#include <stddef.h>
#include <stdint.h>

/* every entry point is exported explicitly: the libraries are built with -fvisibility=hidden, so `nm -D` lists exactly these */
#ifndef RD_API
#define RD_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define RD_OK 0
#define RD_E_INVALID (-1)   /* bad argument                                   */
#define RD_E_HIP (-2)       /* a HIP runtime call failed                      */
#define RD_E_UNSUPPORTED (-3) /* architecture / arg combination not covered   */
#define RD_E_WORKSPACE (-4) /* workspace too small                            */

/* --ensure modes of reference detect.py:616-663 */
#define RD_ENSURE_NONE 0
#define RD_ENSURE_RRNA 1
#define RD_ENSURE_NORRNA 2
#define RD_ENSURE_BOTH 3

/* LSTM kernel variants (rd_set_variant): all compute the same function within the 1e-4 logit bound */
#define RD_VARIANT_AUTO 0       /* = MFMA_F16X3_T32                                                            */
#define RD_VARIANT_MFMA_F32 1   /* persistent-weight fp32 MFMA recurrence (v_mfma_f32_16x16x4_f32)              */
#define RD_VARIANT_SIMPLE 2     /* plain fp32 FMA kernel, correctness cross-check                              */
/* id 3 (the first 16x16x32 tiling of the split-precision kernel) was removed in 0.2: RD_E_UNSUPPORTED          */
#define RD_VARIANT_MFMA_F16X3_T32 4 /* fp32 product as 3 fp16 MFMA products (hi*hi + hi*lo + lo*hi) on v_mfma_f32_32x32x16_f16,
                                       fp32 accumulate, 32-read tiles, hand-interleaved gate math                */
/* Any other id is rejected with RD_E_UNSUPPORTED by librd_hip.so. The A/B and diagnostic instantiations used by tools/
 * (ids >= 10, some of them wrong by design) exist only in the separately built librd_hip_diag.so (-DRD_DIAG). */

/* output-row semantics (rd_set_semantics) */
#define RD_SEM_PACKED 0 /* reference GPU product: PackedSequence, gather at timestep min(len,max_len)-1 (model/model.py:32-37)  */
#define RD_SEM_PADDED 1 /* reference CPU product `ribodetector_cpu`: input zero-padded to max_len rows, BiLSTM over all rows,
                           gather at the last non-zero row (model/model_cpu.py:29-37,57-62, detect_cpu.py:699-700)            */

typedef struct rd_model rd_model; /* opaque: device-resident, pre-packed weights */

/* Host pointers to the 10 tensors of the reference state_dict (reference model/model.py:16-24; names as in
 * torch: rnn.weight_ih_l0 [4H,I], rnn.weight_hh_l0 [4H,H], rnn.bias_ih_l0 [4H], rnn.bias_hh_l0 [4H], the same
 * four with suffix _reverse, out.weight [C,2H], out.bias [C]); row-major fp32, torch gate order i,f,g,o. */
typedef struct rd_weights {
    const float *w_ih, *w_hh, *b_ih, *b_hh;
    const float *w_ih_r, *w_hh_r, *b_ih_r, *b_hh_r;
    const float *w_out, *b_out;
    int32_t input_size;   /* must be 4   */
    int32_t hidden_size;  /* must be 128 */
    int32_t num_classes;  /* must be 2   */
} rd_weights;

/* Replaces: SeqModel(**arch.args); load_state_dict(...); .to('cuda'); .eval()
 * (reference detect.py:93,115-119, model/model.py:10-29). Uploads the weights to `device` and pre-packs them
 * (per-lane MFMA operand order for W_hh, fused input table  W_ih[:,base]+b_ih+b_hh, reverse-direction table).
 * Synchronous. */
RD_API int rd_model_create(const rd_weights *w, int device, rd_model **out);
RD_API void rd_model_destroy(rd_model *m);

/* Select the recurrence kernel (RD_VARIANT_*); default AUTO = MFMA_F16X3_T32. Ids this build does not contain are
 * refused with RD_E_UNSUPPORTED (the model keeps its current kernel). rd_variant_available: 1 if `variant` can be selected. */
RD_API int rd_set_variant(rd_model *m, int variant);
RD_API int rd_variant_available(int variant);

/* Select which of the reference's two products rd_classify reproduces (RD_SEM_*); default RD_SEM_PACKED. The two differ
 * only for reads shorter than max_len or ending in non-ACGT bases. */
RD_API int rd_set_semantics(rd_model *m, int semantics);

/* Label stability. The label is argmax(logits) (reference detect.py:288,481); every fp32 evaluation of the recurrence - the
 * reference's own included - carries up to ~1e-4 of rounding noise on the logits, so for the few reads per million whose margin
 * |logit1 - logit0| is smaller than that the label is decided by noise. rd_classify therefore re-evaluates every read whose
 * margin is below `thresh` (default RD_REFINE_DEFAULT, ~15 reads per million) in float64 - the same function, reference
 * model/model.py:32-37, with all products, sums and activations in double - and replaces its logits and label: labels are
 * those of the exact function, independent of kernel variant and batch split. thresh = 0 switches the pass off.
 * rd_refine is the same pass as a separate call (thresh <= 0: the model's band):
 *   - for callers that hold logits of BOTH mates: with `mate_logits` ([dev] float[n*2], row i = the mate of read i) a read is
 *     also re-evaluated when the PAIR margin |(l1+m1) - (l0+m0)| is below 2*thresh, which is what decides the pair label under
 *     --ensure none (detect.py:657); call it once per mate before rd_pair_fuse;
 *   - for throughput: the pass has the latency of one read (~0.3 ms: 100 dependent float64 steps on one CU) with the rest of
 *     the GPU idle. A pipelined caller switches the inline pass off (rd_set_refine(m, 0)) and issues rd_refine(..., thresh)
 *     on a second stream, where it overlaps the next batch's recurrence (bench.py and the CLI do).
 * One launch: each workgroup scans 512 logit rows and re-evaluates the candidates among them itself.
 * rd_set_refine_async(m, K), K in [1, 16], moves the pass off the caller's critical path without a second stream on the caller's
 * side (opt-in; 0 = back to the inline pass): rd_classify then only RECORDS its candidates (where their bases lie, where their
 * results go) in a queue the model owns, and the candidates of K consecutive calls are evaluated together - one workgroup each, so
 * all of them in the 0.3 ms one takes - on a stream the model owns, beside the recurrence of the call that follows the group.
 * A call's logits / labels are final, and its input buffers may be overwritten, for work issued on the caller's stream
 *   (a) after the K-th call that follows it has been issued (K = 1: after the next call), or
 *   (b) after rd_sync_results(m, stream) - which evaluates whatever waits, on the model's stream, and makes `stream` wait for it.
 *       `stream` need not be the stream the calls were issued on, as long as it is ordered behind them (an event): a pipelined
 *       caller gives its post-processing stream, so that the evaluation and what consumes it (pair fusion, label D2H) run beside the
 *       next batch's recurrences while the calls' own stream never waits (bench.py and the CLI do, since round 4). Calls issued
 *       after it record into the model's other queue. All rd_classify calls of a model in this mode go to ONE stream.
 * Until then the buffers of those calls must stay as they are: a streaming caller cycles K + 1 sets of buffers and consumes the
 * results of a call K calls later (what the reference's loop cannot do: it synchronises on .tolist() every batch, reference
 * detect.py:288,481). A call that passes a pointer of a waiting call, or overlapping output ranges, is detected and synchronises
 * first (correct results, no overlap). rd_refine synchronises first too. A call captured in a hipGraph keeps the pass inline.
 * Candidates beyond the queue's 8,192 entries are evaluated inside rd_classify. rd_model_destroy / changing K require
 * rd_sync_results first (candidates still waiting at destroy keep their fp32 results). */
#define RD_REFINE_DEFAULT 2.5e-4f
RD_API int rd_set_refine(rd_model *m, float thresh);
RD_API int rd_set_refine_async(rd_model *m, int calls_per_group);
RD_API int rd_sync_results(rd_model *m, void *stream);
RD_API int rd_refine(const rd_model *m, const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n,
              int32_t max_len, float *logits, uint8_t *labels, const float *mate_logits, float thresh, void *stream);

/* Prefix-state table (an extension; nothing in the reference to replace - its cuDNN call steps over every
 * base, reference model/model.py:33). The forward recurrence is a pure function of the bases read so far, and there are only 4^k
 * sequences of k bases: row p of the table is the recurrence state (h as the kernel's two fp16 arrays, the cell state in fp32;
 * 1 KiB) after the k bases whose base-4 number is p, computed by the classifying kernel itself, so that a read whose first k bases
 * are A/C/G/T(U) starts from its row and runs min(len,max_len) - k steps with bit-identical results (reads with another letter in
 * the prefix, or no step left after it, run all their steps from the zero state as before). This is the reverse-direction table
 * (one step, 5 rows) carried k steps forward: k = 12 takes 16 GiB of the 288 GB of HBM and removes 12 % of the steps of a 100 bp read.
 *   rd_prefix_table_bytes(k): bytes of [dev] memory for k in [RD_PREFIX_K_MIN, RD_PREFIX_K_MAX] ((4^k + 1) KiB); 0 otherwise.
 *   rd_prefix_scratch_bytes(k): bytes of [dev] scratch the build needs (4^(k-1) KiB: the level below), free again afterwards.
 *   rd_set_prefix_table: builds the table for the model's weights into caller-owned `table` (256-byte aligned, >= that many
 *     bytes; it must stay allocated until the model is destroyed or another / no table is set) and attaches it: k launches, level
 *     j = the states after every j-base prefix from level j-1 by ONE step (4/3 4^k steps in all: 15 ms at k = 12). k = 0 (pointers
 *     may be NULL) detaches. Synchronous on `stream`. The rows are the state of the kernel that builds them - the model's CURRENT
 *     variant: RD_VARIANT_MFMA_F16X3_T32 (h as two fp16 arrays + fp32 cell state) or RD_VARIANT_MFMA_F32 (h and c in fp32; same 1 KiB,
 *     same addressing); RD_VARIANT_SIMPLE has none (RD_E_UNSUPPORTED). rd_classify uses the table only while the model's variant is
 *     the one that built it (after rd_set_variant to another kernel the reads run all their steps until the table is rebuilt).
 *   rd_prefix_k: k of the attached table, 0 if none. */
#define RD_PREFIX_K_MIN 4
#define RD_PREFIX_K_MAX 13
RD_API size_t rd_prefix_table_bytes(int32_t k);
RD_API size_t rd_prefix_scratch_bytes(int32_t k);
RD_API int rd_set_prefix_table(rd_model *m, int32_t k, void *table, size_t table_bytes, void *scratch, size_t scratch_bytes, void *stream);
RD_API int rd_prefix_k(const rd_model *m);

/* Bytes of [dev] scratch rd_classify needs for n reads with truncation length max_len. */
RD_API size_t rd_classify_workspace_bytes(int64_t n, int32_t max_len);

/* Replaces: collate (one-hot + pack_sequence, reference detect.py:666-689) + model(x) (model/model.py:32-37
 * forward1: BiLSTM over min(len,max_len) steps, last-timestep gather, Linear) + torch.argmax (detect.py:288,481).
 *   arena   [dev] ASCII bytes;  seq_off [dev] int64[n] start of read i in arena;  seq_len [dev] int32[n]
 *   logits  [dev] float[n*2], row i <-> read i (input order);  labels [dev] uint8[n] or NULL (1 = rRNA)
 *   workspace [dev] >= rd_classify_workspace_bytes(n, max_len), 256-byte aligned */
RD_API int rd_classify(const rd_model *m, const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n,
                int32_t max_len, float *logits, uint8_t *labels, void *workspace, size_t workspace_bytes, void *stream);

/* Replaces: Predictor.separate_paired_reads label logic (reference detect.py:616-663).
 *   logits1/logits2 [dev] float[n*2];  pair_labels [dev] int8[n] in {0,1,-1}
 *   counts [dev] uint64[3] or NULL: (non-rRNA, rRNA, unclassified) ADDED to the existing values
 *   (the running counters of detect.py:331-333,388-389,400). */
RD_API int rd_pair_fuse(const float *logits1, const float *logits2, int64_t n, int32_t ensure_mode, int8_t *pair_labels,
                 uint64_t *counts, void *stream);

/* Replaces: Predictor.separate_reads counters for single-end (reference detect.py:600-614,485-486).
 *   labels [dev] uint8[n]; counts [dev] uint64[3], ADDED to. */
RD_API int rd_count_labels(const uint8_t *labels, int64_t n, uint64_t *counts, void *stream);

/* Replaces: SeqEncoder.encode_read / BASE_DICT (reference data_loader/seq_encoder.py:11-18,126-127) for a batch.
 * codes [dev] uint8[n*stride]: 0 A, 1 C, 2 G, 3 T/U, 4 anything else; positions >= min(len,max_len) hold 4.
 * stride >= max_len. */
RD_API int rd_encode_codes(const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n, int32_t max_len,
                    int32_t stride, uint8_t *codes, void *stream);

/* Replaces: encode_variable_len_read + np.array(..., float32) (reference seq_encoder.py:130-145,
 * detect_cpu.py:699-700): onehot [dev] float[n*max_len*4], zero rows after the read. */
RD_API int rd_encode_onehot_padded(const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n,
                            int32_t max_len, float *onehot, void *stream);

/* Replaces: torch.FloatTensor(encode_read(...)) + pack_sequence(enforce_sorted=False) (reference
 * detect.py:681-685). Two calls:
 *   rd_pack_plan: fills sorted_idx [dev] int64[n] (lengths descending, ties by input index), unsorted_idx [dev]
 *     int64[n], batch_sizes [dev] int64[max_len] (entries past the longest read are 0) and total_steps [dev]
 *     int64[1] = sum_i min(len_i,max_len).  workspace as for rd_classify. API parity only: rd_classify does NOT call it (it
 *     buckets by step count with atomics, order inside a bucket free).
 *   rd_pack_onehot: writes data [dev] float[total_steps*4], time-major over the sorted reads. */
RD_API int rd_pack_plan(const int32_t *seq_len, int64_t n, int32_t max_len, int64_t *sorted_idx, int64_t *unsorted_idx,
                 int64_t *batch_sizes, int64_t *total_steps, void *workspace, size_t workspace_bytes, void *stream);
RD_API int rd_pack_onehot(const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n, int32_t max_len,
                   const int64_t *sorted_idx, const int64_t *batch_sizes, float *data, void *stream);

/* Device-side gzip of the output files (round 4). Replaces, for the GPU path: gzip.open(out, 'wt', compresslevel=5) fed with the
 * records of one label in input order (reference detect.py:485-492,729-741: the writer compresses by file extension). The records of
 * a chunk (its text is already in HBM: the read bytes travel as the chunk's text) whose label equals `label` become a sequence of
 * complete gzip members in `out`, ready to be appended to the file - in BGZF framing (one member per 65,280 input bytes, 'B','C'
 * extra subfield), so that the file is also what bgzip writes: indexable, and decodable member by member in parallel. DEFLATE is
 * done by hand-written kernels: LZ77 with per-wave hash tables in LDS, one dynamic-Huffman block per member, CRC-32 on the device.
 *   text [dev] the chunk's bytes; rec_start [dev] int64[n+1]: record i = bytes [rec_start[i], rec_start[i+1]) (verbatim, with its
 *   newline); labels [dev] int8[n] (rd_pair_fuse's pair labels, or rd_classify's uint8 labels); label: the value to select.
 *   out [dev] >= rd_gz_out_bound(text_bytes) bytes is always enough (256-byte aligned); info [dev] int64[4]:
 *     info[0] = bytes written to out (if > out_cap: out was too small and is incomplete), info[1] = uncompressed bytes,
 *     info[2] = members, info[3] != 0: rec_start does not describe `text` (the selected records hold more bytes than the text:
 *     nothing was compressed). Asynchronous on `stream`; the caller copies info and out[0, info[0]) to the host afterwards.
 *   rd_gz_eof_block: [host] BGZF's 28-byte end-of-file marker (an empty member), to be appended once when the file is closed. */
RD_API size_t rd_gz_workspace_bytes(int64_t n, int64_t text_bytes);
RD_API size_t rd_gz_out_bound(int64_t text_bytes);
RD_API int rd_gz_compress_selected(const uint8_t *text, int64_t text_bytes, const int64_t *rec_start, const int8_t *labels, int64_t n,
                            int32_t label, uint8_t *out, size_t out_cap, int64_t *info, void *workspace, size_t workspace_bytes,
                            void *stream);
RD_API int rd_gz_eof_block(uint8_t *dst, size_t cap);

/* The input side of the same idea (round 4): a .gz whose members say how long they are - BGZF (bgzip / htslib, and every .gz this
 * build's CLI writes) - is a list of independent DEFLATE streams, and they are inflated on the device, one wave per member
 * (replaces, for such files: gzip.open(path, 'rt') in reference data_loader/seq_encoder.py:21-39 / fastx_parser.py:15-55). The host
 * walks the member headers (no decoding needed: the size is in the 'B','C' subfield, ISIZE in the trailer) and fills one rd_gz_member
 * per member; every DEFLATE block type is handled; each member's CRC-32 and ISIZE are checked on the device.
 *   comp [dev] the compressed bytes (at least 8 readable bytes behind the last member's data: its trailer);
 *   members [dev] rd_gz_member[n] (in_len at most 256 MiB; an entry that points outside comp / text gets RD_GZI_MEMBER, nothing is read
 *   or written for it); text [dev] receives member i's out_len bytes at out_off;
 *   status [dev] uint32[n]: 0 = ok, else RD_GZI_* (the member's output is then undefined). Asynchronous on `stream`. */
typedef struct rd_gz_member {
    int64_t in_off;   /* first byte of the member's raw DEFLATE data in comp (behind the gzip header) */
    int64_t out_off;  /* where its bytes go in text */
    int32_t in_len;   /* bytes of raw DEFLATE data; the 8-byte trailer (CRC-32, ISIZE) follows them */
    int32_t out_len;  /* ISIZE */
} rd_gz_member;
#define RD_GZI_OK 0
#define RD_GZI_BAD_BLOCK 1
#define RD_GZI_BAD_CODE 2
#define RD_GZI_BAD_LENGTHS 3
#define RD_GZI_OVERRUN 4
#define RD_GZI_BAD_DISTANCE 5
#define RD_GZI_TRUNCATED 6
#define RD_GZI_SIZE 7
#define RD_GZI_CRC 8
#define RD_GZI_STORED 9
#define RD_GZI_MEMBER 10   /* the descriptor itself: offsets / lengths outside comp or text */
RD_API int rd_gz_inflate_members(const uint8_t *comp, int64_t comp_bytes, const rd_gz_member *members, int64_t n, uint8_t *text, int64_t text_bytes,
                          uint32_t *status, void *stream);

/* The FASTQ record index, on the device (round 5). Replaces, for text that already lies in HBM - BGZF members inflated there by
 * rd_gz_inflate_members, or a plain file's bytes copied there - the parser's FASTQ state machine (reference
 * data_loader/fastx_parser.py:15-37: four lines per record, each rstrip()-ed, header / '+' / quality verbatim, bases not upper-cased),
 * so that inflate -> index -> classify -> select -> deflate never returns text to the host. The stream arrives in batches; a batch
 * buffer `text` [dev] has `pad` spare bytes in front of its new bytes [pad, end) and >= 65 readable / writable bytes behind `end`:
 *   rd_fastq_index: the bytes behind the last complete record of the batch before (prev_text / prev: that batch's buffer and its
 *     summary, both [dev], read on the device - or both NULL for the first batch) are copied in front of `pad`; the window [begin,
 *     end) is framed: line_end [dev] int32[cap_lines] receives the offset in `text` of the '\n' ending line j (cap_lines >= end - pad
 *     + carry + 1 is always enough); record r = lines 4r .. 4r + 3, its text = [start of line 4r, line_end[4r + 3] + 1), its bases =
 *     line 4r + 1. final != 0: the stream ends with this batch - a last line without '\n' gets one, and fewer than four trailing lines
 *     are tolerated when blank. summary [dev] rd_fq_summary: status RD_FQ_* (HEADER: record `bad_record` does not start with '@';
 *     TRUNCATED: the stream ends inside a record; CARRY: a record longer than `pad`; CHAIN: the batch before had failed);
 *     dirty != 0: some line has trailing whitespace (CR LF files) - the records are NOT verbatim ranges then: the caller strips the
 *     window (rd_fastq_strip_mark + a compaction of the marked bytes) and indexes the result with final = 1; `consumed` / `end`
 *     of the ORIGINAL summary stay the carry of the next batch. Asynchronous on `stream`; no host round trip between batches.
 *   rd_fastq_gather: records [rec_lo, rec_hi) of an indexed batch appended to a chunk: their text is copied to out_text [dev,
 *     16-byte aligned] at *cursor_in [dev] and *cursor_out [dev, another address] = the position behind it (-1 if it did not fit
 *     out_cap, or the batch / the cursor was bad: every later piece then fails too); rec_start [dev] int64[rec_hi - rec_lo + 1],
 *     seq_off int64[..], seq_len int32[..] = the chunk's entries for these records (offsets into out_text) - the arrays rd_classify and
 *     rd_gz_compress_selected take. max_bytes: an upper bound of the bytes (sizes the launch), e.g. the batch's window size.
 *   rd_fastq_strip_mark: del [dev, zeroed by the caller] gets 1 at every byte rstrip() removes from lines [0, n_lines) of the batch.
 *   rd_fastq_sample: samples [dev] int32[cap]: samples[k] = offset in `text` where record k * every starts (-1 beyond the batch's
 *     records; entry n / every ... the summary's `consumed` closes the last interval): copied to the host with the summary, they bound
 *     the bytes of any record range, so that a caller sizes a chunk's buffer without a round trip per range. */
typedef struct rd_fq_summary {
    int64_t begin, end;       /* the window framed: [begin, end) in the batch buffer (begin = pad - carry; end includes an added '\n') */
    int64_t n_lines, n_records;
    int64_t consumed;         /* offset behind the last complete record (final: = end): the next batch's carry is [consumed, end) */
    uint64_t bad_record;      /* ~0, or the first record whose header line does not start with '@' */
    int32_t status, dirty;
    int64_t reserved;
} rd_fq_summary;
#define RD_FQ_OK 0
#define RD_FQ_HEADER 1
#define RD_FQ_TRUNCATED 2
#define RD_FQ_CARRY 3
#define RD_FQ_CHAIN 4
#define RD_FQ_LINES 5      /* more lines than cap_lines */
RD_API size_t rd_fastq_index_workspace_bytes(int64_t text_end);
RD_API int rd_fastq_index(uint8_t *text, int64_t pad, int64_t end, const uint8_t *prev_text, const rd_fq_summary *prev, int32_t final, int32_t *line_end,
                   int64_t cap_lines, rd_fq_summary *summary, void *workspace, size_t workspace_bytes, void *stream);
RD_API int rd_fastq_gather(const uint8_t *text, const int32_t *line_end, const rd_fq_summary *summary, int64_t rec_lo, int64_t rec_hi, int64_t max_bytes,
                    uint8_t *out_text, int64_t out_cap, const int64_t *cursor_in, int64_t *cursor_out, int64_t *rec_start, int64_t *seq_off,
                    int32_t *seq_len, void *stream);
RD_API int rd_fastq_sample(const int32_t *line_end, const rd_fq_summary *summary, int64_t every, int32_t *samples, int64_t cap, void *stream);
RD_API int rd_fastq_strip_mark(const uint8_t *text, const int32_t *line_end, const rd_fq_summary *summary, int64_t max_lines, uint8_t *del, void *stream);

/* FASTA records on the device (round 5). Replaces the FASTA half of the parser for text that lies in HBM (reference
 * data_loader/fastx_parser.py:39-55: every line strip()-ed, blank lines skipped, '>' starts a record, the other lines of a record joined
 * and upper-cased; a record is yielded at the next header, at the end of the file only if its sequence is not empty) together with the
 * writer's '\n'.join(record) + '\n' (detect.py:489-492). A FASTA record is not a verbatim range of its file, so the batch is
 * RE-WRITTEN: norm [dev, norm_cap bytes, 16-byte aligned] receives header '\n' SEQUENCE '\n' per record (what the host reader puts into
 * its chunk buffer), rec_tab [dev] int64[cap_records] the offset in `norm` where record r starts (entry n_records = the end),
 * hdr_tab [dev] int32[cap_records] the length of its header line. text / pad / end / prev_text / prev / final / line_end / cap_lines /
 * summary as rd_fastq_index (the batches of a stream chain the same way: the carry is the raw text from the last header line on - that
 * record is complete only in a final batch; final = 2: the stream is a SHARE of a file that goes on behind it - its last record counts
 * even without a sequence, which final = 1, the end of the file, drops like the reference); summary->reserved = the bytes of `norm` that belong to records; status RD_FQ_LINES also
 * when norm_cap or cap_records is too small (norm_cap >= window + lines + 2, cap_records >= lines + 2 is always enough). Sequence lines
 * in front of the first header (a malformed file) become part of the first record's sequence, a stream without any header ONE record with
 * an empty header line - what the reference's parser yields. rd_fasta_gather / rd_fasta_sample: as rd_fastq_gather / rd_fastq_sample,
 * over `norm` and the two tables. */
RD_API size_t rd_fasta_index_workspace_bytes(int64_t text_end, int64_t cap_lines);
RD_API int rd_fasta_index(uint8_t *text, int64_t pad, int64_t end, const uint8_t *prev_text, const rd_fq_summary *prev, int32_t final, int32_t *line_end,
                   int64_t cap_lines, uint8_t *norm, int64_t norm_cap, int64_t *rec_tab, int32_t *hdr_tab, int64_t cap_records, rd_fq_summary *summary,
                   void *workspace, size_t workspace_bytes, void *stream);
RD_API int rd_fasta_gather(const uint8_t *norm, const int64_t *rec_tab, const int32_t *hdr_tab, const rd_fq_summary *summary, int64_t rec_lo, int64_t rec_hi,
                    int64_t max_bytes, uint8_t *out_text, int64_t out_cap, const int64_t *cursor_in, int64_t *cursor_out, int64_t *rec_start,
                    int64_t *seq_off, int32_t *seq_len, void *stream);
RD_API int rd_fasta_sample(const int64_t *rec_tab, const rd_fq_summary *summary, int64_t every, int32_t *samples, int64_t cap, void *stream);

/* The records of a chunk that carry one label as ONE contiguous text in `out` [dev, 16-byte aligned], in input order - what the
 * reference's writer joins on the host (detect.py:485-492: fh.write('\n'.join(selected) + '\n')) - for plain (uncompressed) outputs of
 * chunks whose text lives on the device; arguments as rd_gz_compress_selected. info [dev] int64[4]: info[1] = bytes written,
 * info[3] != 0: the selection holds more bytes than the text or than out_cap (nothing written). */
RD_API size_t rd_select_workspace_bytes(int64_t n);
RD_API int rd_select_pack(const uint8_t *text, int64_t text_bytes, const int64_t *rec_start, const int8_t *labels, int64_t n, int32_t label, uint8_t *out,
                   size_t out_cap, int64_t *info, void *workspace, size_t workspace_bytes, void *stream);

/* ONE DEFLATE stream - a plain .gz, the format sequencers write - inflated on the device (round 5; the CLI's default for such FASTQ, RD_DEVICE_INFLATE=members keeps the host's decoders).
 * Replaces, for such files: gzip.open(path, 'rt') of reference data_loader/seq_encoder.py:21-39. The two-pass scheme of pugz (this
 * build's host reader: csrc/rd_pgzip.h) with one wave per SECTION of `section_bytes` compressed bytes: block starts are searched on the
 * device, every section is decoded from its start to the next section's start with an unknown window into 16-bit symbols, the windows
 * are resolved in order and all sections in parallel. The stream arrives in batches:
 *   comp [dev, 4-byte aligned] holds `valid_bytes` bytes of the file from the batch's first byte (comp_bytes readable); the sections
 *   cover [0, data_bytes) and one more section behind them is searched for the start the NEXT batch begins with (so valid_bytes >=
 *   data_bytes + section_bytes unless the file ends; the next batch's buffer starts `data_bytes` further: carry_delta_bits =
 *   8 * data_bytes). first_start_bit: where the member's first block starts (first batch: behind the gzip header; else ignored and taken
 *   from `carry` = the state the batch before left [dev]). win_in / win_out [dev] 32 KiB: the text behind the batch before / this one.
 *   text [dev, 8-byte aligned] receives state->n_text bytes (<= text_cap). state [dev] rd_gzs_state: status RD_GZS_* (MISMATCH: a section did not
 *   end where the next one starts - nothing speculative is ever accepted; OVERFLOW: a section made more than cap_syms symbols; ...),
 *   final: the member ended in this batch at bit `end_bit` of comp (its trailer - CRC-32, ISIZE - follows at the next byte boundary: the
 *   caller compares them with state->crc and total_len). Asynchronous on `stream`. */
typedef struct rd_gzs_state {
    uint64_t total_len;
    int64_t n_text;
    uint32_t crc, status, final, end_bit, next_start, win_valid, bad_section, n_sections;
    uint64_t reserved[2];
} rd_gzs_state;
#define RD_GZS_OK 0
#define RD_GZS_DECODE 1
#define RD_GZS_MISMATCH 2
#define RD_GZS_OVERFLOW 3
#define RD_GZS_NOSTOP 4
#define RD_GZS_WINDOW 5
#define RD_GZS_TEXTCAP 6
#define RD_GZS_NOSTART 7
RD_API size_t rd_gz_stream_workspace_bytes(int64_t data_bytes, int32_t section_bytes, int32_t cap_syms, int64_t text_cap);
RD_API int rd_gz_stream_inflate(const uint8_t *comp, int64_t comp_bytes, int64_t data_bytes, int64_t valid_bytes, int32_t section_bytes, int32_t cap_syms,
                         uint32_t first_start_bit, const rd_gzs_state *carry, int64_t carry_delta_bits, int32_t at_eof, const uint8_t *win_in,
                         uint8_t *win_out, uint8_t *text, int64_t text_cap, rd_gzs_state *state, void *workspace, size_t workspace_bytes, void *stream);

/* Round 6 - a RANGE of such a stream decoded before the text in front of it is known: what lets the ranks of a node share ONE .gz
 * (rank r decodes the compressed bytes [r S / W, (r + 1) S / W) on its own GPU while rank r - 1 still decodes its share). Replaces, for W > 1,
 * the same gzip.open(path, 'rt') of reference data_loader/seq_encoder.py:21-39,75-87 that every rank would otherwise repeat (or one rank
 * perform for all). rd_gz_range_decode is rd_gz_stream_inflate up to the window chain, which runs on SYMBOLS: sym_text [dev, 16-byte
 * aligned] receives state->n_text 16-bit symbols - a byte, or 0x8000 | i = "byte i of the 32 KiB in front of the RANGE" - and map_out [dev]
 * uint16[32768] the 32 KiB behind the batch in the same form: the range's MAP so far. Batches of a range chain through carry / map_in (both
 * null for its first batch). first_start_bit = RD_GZS_SEARCH: the range starts inside the stream, its first block start is searched like
 * every section's (state->reserved[0] = the bit found: it must equal the next_start the range before reports, nothing speculative is
 * accepted); a rank whose range holds the gzip header passes the bit behind it. state: as above, crc / total_len / win_valid untouched.
 * The ranks exchange their maps, apply them in rank order on the host (window[r + 1][i] = map[r][i] is a byte ? it : window[r][map[r][i] &
 * 0x7fff]) and call rd_gz_range_resolve per batch: text [dev, 16-byte aligned] = the n symbols as bytes, window [dev] uint8[32768] the
 * 32 KiB in front of the range of which the LAST win_valid bytes are text of the member (a marker in front of them: state->status =
 * RD_GZS_WINDOW); state [dev, zeroed by the caller before the range's first call] accumulates crc / total_len over the calls (CRC-32 of the
 * range's text alone: the ranks' values are combined like zlib's crc32_combine and compared with the member's trailer). */
#define RD_GZS_SEARCH 0xfffffffeu
RD_API size_t rd_gz_range_workspace_bytes(int64_t data_bytes, int32_t section_bytes, int32_t cap_syms, int64_t text_cap);
RD_API int rd_gz_range_decode(const uint8_t *comp, int64_t comp_bytes, int64_t data_bytes, int64_t valid_bytes, int32_t section_bytes, int32_t cap_syms,
                       uint32_t first_start_bit, const rd_gzs_state *carry, int64_t carry_delta_bits, int32_t at_eof, const uint16_t *map_in, uint16_t *map_out,
                       uint16_t *sym_text, int64_t text_cap, rd_gzs_state *state, void *workspace, size_t workspace_bytes, void *stream);
RD_API size_t rd_gz_range_resolve_workspace_bytes(int64_t n);
RD_API int rd_gz_range_resolve(const uint16_t *sym_text, int64_t n, const uint8_t *window, uint32_t win_valid, uint8_t *text, rd_gzs_state *state, void *workspace,
                        size_t workspace_bytes, void *stream);

/* n bytes moved by a kernel on `stream` instead of a DMA engine: dst / src [dev, or pinned host memory mapped into the device]. The
 * feeder's H2D of file bytes and the writers' D2H of output bytes use it: an SDMA queue is shared in order with other streams' copies,
 * and a copy that waits for kernels (a label D2H behind two recurrence launches) held a 96 MB H2D back for 60-100 ms. workgroups (of 1,024 threads): 0 = 8 - few and fat, because a
 * workgroup that copies keeps its CU from the recurrence kernel for as long as the copy runs. */
RD_API int rd_copy_bytes(void *dst, const void *src, int64_t n, int32_t workgroups, void *stream);

/* A HIP stream whose kernels run on a subset of the compute units (hipExtStreamCreateWithCUMask): cu_mask [host] uint32[words], bit i =
 * CU i may be used (MI355X: 256 CUs = 8 words; the driver spreads consecutive bits over the XCDs). words == 0: an ordinary stream,
 * priority < 0 = the device's highest. Why: the latency-bound side kernels (inflate, FASTQ index, deflate) otherwise queue behind a
 * recurrence launch that holds every register file for ~30 ms; on CUs of their own they run beside it. The caller
 * wraps the handle (torch.cuda.ExternalStream) and destroys it when done. */
RD_API int rd_stream_create(int device, const uint32_t *cu_mask, int words, int priority, void **stream);
RD_API int rd_stream_destroy(void *stream);

/* Timing of the dominant kernel, for bench.py's roofline: rd_classify records hipEvents around the recurrence
 * kernel on the launch stream when enabled. rd_profile_read synchronises those events and returns the number of
 * recorded launches and their total duration. */
RD_API int rd_profile_enable(rd_model *m, int enable);
RD_API int rd_profile_read(rd_model *m, int64_t *launches, double *total_ms);

RD_API const char *rd_last_error(void);
RD_API const char *rd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RIBODETECTOR_AMD_H */
