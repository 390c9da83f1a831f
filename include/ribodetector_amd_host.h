/*
 * ribodetector_amd_host.h - C ABI of the host-side ingest / output library (librd_host.so, plain C++ + zlib, no GPU).
 *
 * the reference's FASTQ/FASTA reader and label-partitioned writer are Python loops over
 * per-record tuples (0.5 M reads/s parse, 0.2-0.34 M reads/s write); the kernels classify ~27 M reads/s. This
 * library produces exactly the arrays rd_classify consumes - one byte arena + offsets - and writes the selected records
 * back in input order.
 *
 * Reference interfaces replaced (paths relative to /root/reference/ribodetector):
 *   rd_reader_*   data_loader/seq_encoder.py:21-39,75-92 (get_seq_format, get_seq_chunks) over
 *                 data_loader/fastx_parser.py:15-55 (seq_parser): FASTQ = 4-line records, every line rstrip()-ed, bases NOT
 *                 upper-cased; FASTA = multi-line sequences joined and upper-cased, blank lines skipped; gzip by extension.
 *   rd_writer_*   detect.py:729-741 (open_for_write: gzip level 5 iff the name ends with "gz") and
 *                 detect.py:485-492 (fh.write('\n'.join(selected_records) + '\n')).
 *
 * All buffers are caller-owned (e.g. pinned torch tensors); return 0 = ok (rd_reader_next: also 1 = end of file),
 * <0 = error + rd_host_last_error().
 */
#ifndef RIBODETECTOR_AMD_HOST_H
#define RIBODETECTOR_AMD_HOST_H

#include <stddef.h>
#include <stdint.h>

/* every entry point is exported explicitly: the libraries are built with -fvisibility=hidden, so `nm -D` lists exactly these */
#ifndef RD_API
#define RD_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rd_reader rd_reader;
typedef struct rd_writer rd_writer;

/* format: 0 = FASTQ, 1 = FASTA, -1 = decide from the file name like get_seq_format (.fq/.fastq/.fa/.fasta/.fna/.fas [+ .gz])
 * A plain regular file is mapped and parsed in place (RD_READER_MMAP=0 in the environment: read through buffers instead, as pipes
 * and gzip input always are). */
RD_API int rd_reader_open(const char *path, int format, rd_reader **out);
RD_API void rd_reader_close(rd_reader *r);

/* Byte-range ingest for the multi-rank CLI (one process per GPU): every rank parses only its own part of a PLAIN input file.
 *   rd_host_file_info          size in bytes; is_gzip = 1 if the file starts with the gzip magic (no byte ranges then)
 *   rd_host_find_record_start  first record boundary at or after byte pos (FASTQ: a line starting with '@' whose second-next
 *                              line starts with '+'; FASTA: a line starting with '>'); the file size if none follows
 *   rd_host_count_records      records starting in [start, end), start being a boundary
 *   rd_host_skip_records       the boundary k records after the boundary `start`
 *   rd_reader_open_range       a reader over [start, end) (both boundaries): parses exactly like a file holding those bytes
 * Record semantics stay those of the reference parser (fastx_parser.py:15-55). */
RD_API int rd_host_file_info(const char *path, int64_t *size, int32_t *is_gzip);
RD_API int rd_host_find_record_start(const char *path, int format, int64_t pos, int64_t *out);
RD_API int rd_host_count_records(const char *path, int format, int64_t start, int64_t end, int64_t *n);
RD_API int rd_host_skip_records(const char *path, int format, int64_t start, int64_t k, int64_t *out);
RD_API int rd_reader_open_range(const char *path, int format, int64_t start, int64_t end, rd_reader **out);

/* Parse up to max_records records into caller buffers.
 *   buf[0 .. *nbytes)      normalised record text: for record i, buf[rec_start[i] .. rec_start[i+1]) is exactly
 *                          '\n'.join(record_lines) + '\n' as the reference would write it
 *   rec_start int64[max_records+1], seq_off int64[max_records] (start of the bases in buf), seq_len int32[max_records]
 * Stops early when the next record would not fit into buf_cap (it is delivered by the next call).
 * *n = number of records delivered. Returns 1 when the end of the file was reached (no record follows those delivered),
 * 0 when more may follow, <0 on error. *n == 0 with return 0 means: the next record does not fit into the remaining buffer;
 * *nbytes then holds the bytes that record needs, so that the caller can grow the buffer (the reference parser has no
 * record-size limit, fastx_parser.py:15-55). EVERY caller must handle that case - grow and call again, or stop: calling again
 * with the same buffer returns the same answer forever (examples/classify_fastq.cpp stops, fastx_parser.py grows). */
RD_API int rd_reader_next(rd_reader *r, int64_t max_records, uint8_t *buf, int64_t buf_cap, int64_t *rec_start, int64_t *seq_off,
                   int32_t *seq_len, int64_t *n, int64_t *nbytes);

/* Decompress a whole gzip file (all members) into out[0..cap) with the reader's own DEFLATE decoder (csrc/rd_inflate.h);
 * *n = bytes produced. Errors follow Python's gzip module, which the reference reads .gz input with: truncated stream,
 * CRC / length mismatch, bad magic -> -1 + message. Used by the tests and for small side files. */
RD_API int rd_host_gunzip(const char *path, uint8_t *out, int64_t cap, int64_t *n);

/* The same through the parallel decoder (csrc/rd_pgzip.h: sections of `section_bytes` compressed bytes decoded by `threads`
 * threads with an unknown window, markers resolved in order; 0 = defaults). The reader uses it for large .gz inputs; anything it
 * cannot handle (binary payload, damaged data, further members) is finished by the sequential decoder, so results and error
 * messages are those of rd_host_gunzip. stats[0..3] (may be null) = sections used, sections dropped (a block start that was not
 * one), batches, 1 if the sequential decoder took over inside the first member. */
RD_API int rd_host_gunzip_parallel(const char *path, uint8_t *out, int64_t cap, int64_t *n, int threads, int64_t section_bytes, int64_t *stats);

/* Worker threads for gzip output (independent level-5 members compressed in parallel, by libdeflate.so.0 when the system
 * has it - bound at run time - else zlib; RD_HOST_ZLIB=1 forces zlib); 0 = auto (usable cores, <= 32).
 * Mirrors the reference's -t/--threads flag (detect.py:787). */
RD_API int rd_host_set_threads(int threads);

/* Decoder threads per .gz input opened from now on (csrc/rd_pgzip.h; files >= 16 MB): 0 = the sequential decoder, < 0 = auto.
 * The CLI divides its -t value among its input files. */
RD_API int rd_host_set_gz_threads(int threads);

/* A reader whose bytes are fed by the caller instead of read from a file: the decompressed text of gzip members inflated elsewhere
 * (on the GPU: librd_hip.so rd_gz_inflate_members). Records come out of rd_reader_next exactly as from a file of those bytes.
 *   rd_reader_open_feed(format: 0 FASTQ / 1 FASTA); rd_reader_feed: hands `len` bytes over and returns when the reader has copied
 *   them (call it from a thread other than the one in rd_reader_next); rd_reader_feed_end: end of the stream (error non-empty: the
 *   stream is damaged - rd_reader_next fails with that text after the records before it). */
RD_API int rd_reader_open_feed(int format, rd_reader **out);
RD_API int rd_reader_feed(rd_reader *r, const uint8_t *bytes, int64_t len);
RD_API int rd_reader_feed_end(rd_reader *r, const char *error);
/* Closing a feed reader before its stream ended (an error elsewhere, the consumer stopped): rd_reader_feed_abort wakes a feeder that
 * waits inside rd_reader_feed (which then returns -1) and fails every later feed call; nothing is freed. The owner then JOINS its
 * feeder threads and only then calls rd_reader_close (which frees the reader: no thread may still be inside a feed call). */
RD_API int rd_reader_feed_abort(rd_reader *r);
/* FASTA only: the fed text is a share of a stream that continues behind it (another rank reads on): the share's last record is
 * yielded even when its sequence is empty - the reference yields a record at the NEXT header (fastx_parser.py:39-55) */
RD_API int rd_reader_set_flush_empty_tail(rd_reader *r, int on);

/* Walk gzip members that carry their own size - BGZF ('B','C') and this library's writer ('R','D') - without decoding them: one
 * entry per non-empty member (layout = rd_gz_member of include/ribodetector_amd.h), offsets relative to in_base / out_base.
 * Stops at an incomplete member or after `cap` entries (*consumed = bytes walked; call again with more bytes); returns 1 when it
 * meets a member without a size subfield (*consumed points at it: the rest is for the streaming decoder), < 0 on damage. */
typedef struct rd_host_gz_member {
    int64_t in_off, out_off;
    int32_t in_len, out_len;
} rd_host_gz_member;
RD_API int rd_host_gz_index(const uint8_t *buf, int64_t len, int64_t in_base, int64_t out_base, rd_host_gz_member *out, int64_t cap, int64_t *n,
                     int64_t *consumed, int64_t *out_bytes);

RD_API int rd_writer_open(const char *path, rd_writer **out);
/* compressor threads this writer was opened with (= the rd_host_set_threads value in force at rd_writer_open) */
RD_API int rd_writer_threads(const rd_writer *w);
/* append, in input order, every record i of the chunk with labels[i] == want */
RD_API int rd_writer_write_selected(rd_writer *w, const uint8_t *buf, const int64_t *rec_start, int64_t n, const int8_t *labels,
                             int32_t want);
/* append complete gzip members made elsewhere (the GPU: librd_hip.so rd_gz_compress_selected) to a gzip output, as they are; data
 * the host path has buffered for the file is compressed and written first (input order). A file that received such members is
 * closed with BGZF's end-of-file marker. */
RD_API int rd_writer_write_members(rd_writer *w, const uint8_t *members, int64_t len);
/* append text that already is the selected records in input order (packed on the GPU: librd_hip.so rd_select_pack) - the same bytes
 * rd_writer_write_selected would have gathered (reference detect.py:485-492: fh.write('\n'.join(selected) + '\n')) */
RD_API int rd_writer_write_text(rd_writer *w, const uint8_t *text, int64_t len);
/* on (default): a file that received device-made members is closed with BGZF's end-of-file marker. off: for '<out>.partN.gz' files
 * of a multi-rank run, which are joined afterwards - the joined file gets ONE marker at its end */
RD_API int rd_writer_set_eof_marker(rd_writer *w, int on);
RD_API int rd_writer_close(rd_writer *w);

RD_API const char *rd_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
