// rd_fasta_index.hpp - FASTA records of a text buffer in HBM, normalised the way the reference's parser yields them (rd_fa_* kernels)
// Part of the single translation unit rd_kernels.hip (included from there, in order); DESIGN.md §3.12.
//
// Replaces, for text that is already on the device, the FASTA half of the host parser - reference data_loader/fastx_parser.py:39-55:
// every line strip()-ed, blank lines skipped, a line that starts with '>' begins a record, the other lines of a record are joined and
// upper-cased; a record is yielded at the NEXT header and at the end of the file only if its sequence is not empty - and the writer's
// '\n'.join(record) + '\n' (detect.py:489-492). A FASTA record is therefore NOT a verbatim range of the input (multi-line sequences,
// lower case, CR LF, indentation): the batch is re-written into a second buffer, `norm`, as header '\n' SEQUENCE '\n' per record - what
// csrc/rd_host.cpp's reader puts into its chunk buffer - and indexed there:
//
//   rd_fq_begin / count / scan / fill (rd_fastq_index.hpp)   carry of the batch before, line table of the raw text
//   rd_fa_lines_kernel   per line: stripped range, kind (blank / header / sequence), bytes it contributes to `norm`; per 2,048 lines the
//                        sum of (bytes, headers) as one 64-bit word
//   rd_fa_base_kernel    one workgroup: those sums -> exclusive bases; the batch's totals and verdict
//   rd_fa_emit_kernel    per line again: position in `norm` = scan of the contributions; header and sequence bytes copied (sequence
//                        upper-cased), the record table written: rec_start[r], hdr_len[r]; the thread of the LAST header line closes the
//                        batch: the record it starts is complete only in a final batch - otherwise it is the carry of the next one
//   rd_fa_gather_kernel  records [lo, hi) of a batch's `norm` -> a chunk (text + rec_start / seq_off / seq_len), like rd_fq_gather_kernel
// A line contributes: header = 1 ('\n' that closes the record before; the very first lands at position -1 and is not written) + its
// bytes + 1; sequence = its bytes; blank = 0. Sequence lines IN FRONT of the first header (a malformed file): the reference glues them
// to the first record's sequence, and yields them under an EMPTY header when the file has no header at all - so do the kernels: the
// first header's bytes go to position 0 and the lines in front of it behind them; without any header the stream's end makes one record
// with an empty header line (until then everything is carried).
#pragma once
#include "rd_fastq_index.hpp"

namespace {

constexpr int FA_LINES = 8;                              // lines per thread
constexpr int FA_BLOCK = FQ_THREADS * FA_LINES;          // 2,048 lines per workgroup
constexpr unsigned long long FA_NONE = ~0ull;
constexpr int FA_LONG = 512;                             // bytes from which a line is copied by its workgroup instead of its thread

struct FaScratch {            // device scratch of one rd_fasta_index call (64 bytes)
    unsigned long long first_hdr, first_seq;   // index of the first header line / sequence line (FA_NONE: none)
    unsigned long long total;                  // headers << 32 | bytes of `norm` (with the closing '\n' of every header, the first included)
    unsigned long long leading;                // sequence lines in front of the first header
    unsigned long long reserved[4];
};

__device__ __forceinline__ void fa_line(const uint8_t *__restrict__ text, const int32_t *__restrict__ line_end, int64_t j, int64_t begin, int &a, int &len, int &kind) {
    int64_t ls = fq_line_start(line_end, j, begin), le = line_end[j];
    while (le > ls && fq_is_ws(text[le - 1])) --le;
    while (ls < le && fq_is_ws(text[ls])) ++ls;
    a = (int)ls;
    len = (int)(le - ls);
    kind = len == 0 ? 0 : text[ls] == '>' ? 1 : 2;
}

__global__ __launch_bounds__(FQ_THREADS) void rd_fa_init_kernel(FaScratch *__restrict__ sc) {
    if (threadIdx.x == 0) { sc->first_hdr = FA_NONE; sc->first_seq = FA_NONE; sc->total = 0; }
}

// info_a[j] = first byte of the stripped line, info_k[j] = its length | kind << 30; blk[b] = headers << 32 | bytes of workgroup b's lines
__global__ __launch_bounds__(FQ_THREADS) void rd_fa_lines_kernel(const uint8_t *__restrict__ text, const int32_t *__restrict__ line_end, const FqSummary *__restrict__ sum,
                                                                int32_t *__restrict__ info_a, uint32_t *__restrict__ info_k, unsigned long long *__restrict__ blk,
                                                                FaScratch *__restrict__ sc) {
    __shared__ int64_t sh[4];
    const int64_t L = sum->status == RD_FQ_OK ? sum->n_lines : 0, begin = sum->begin;
    const int64_t j0 = (int64_t)blockIdx.x * FA_BLOCK + (int64_t)threadIdx.x * FA_LINES;
    if ((int64_t)blockIdx.x * FA_BLOCK >= L) return;
    unsigned long long s = 0, fh = FA_NONE, fs = FA_NONE;
#pragma unroll 1
    for (int k = 0; k < FA_LINES; ++k) {
        const int64_t j = j0 + k;
        if (j >= L) break;
        int a, len, kind;
        fa_line(text, line_end, j, begin, a, len, kind);
        info_a[j] = a;
        info_k[j] = (uint32_t)len | ((uint32_t)kind << 30);
        if (kind == 1) { s += (1ull << 32) + (unsigned long long)len + 2ull; if (fh == FA_NONE) fh = (unsigned long long)j; }
        else if (kind == 2) { s += (unsigned long long)len; if (fs == FA_NONE) fs = (unsigned long long)j; }
    }
    if (fh != FA_NONE && fh < sc->first_hdr) atomicMin(&sc->first_hdr, fh);
    if (fs != FA_NONE && fs < sc->first_seq) atomicMin(&sc->first_seq, fs);
    int64_t total;
    gz_block_scan((int64_t)s, sh, total);
    if (threadIdx.x == 0) blk[blockIdx.x] = (unsigned long long)total;
}

// one workgroup: blk[] -> exclusive bases in place; totals; what can be said about the batch before a line is copied
__global__ __launch_bounds__(FQ_THREADS) void rd_fa_base_kernel(unsigned long long *__restrict__ blk, const int32_t *__restrict__ line_end, FqSummary *__restrict__ sum,
                                                               FaScratch *__restrict__ sc, int final, int64_t norm_cap, int64_t cap_records,
                                                               uint8_t *__restrict__ norm, int64_t *__restrict__ rec_start, int32_t *__restrict__ hdr_len) {
    __shared__ int64_t sh[4];
    __shared__ int64_t run_s;
    if (sum->status != RD_FQ_OK) return;
    const int64_t L = sum->n_lines;
    const int nb = (int)((L + FA_BLOCK - 1) / FA_BLOCK);
    if (threadIdx.x == 0) run_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += FQ_THREADS) {
        const int b = b0 + threadIdx.x;
        const int64_t v = b < nb ? (int64_t)blk[b] : 0;
        int64_t total;
        const int64_t ex = gz_block_scan(v, sh, total);
        const int64_t run = run_s;
        if (b < nb) blk[b] = (unsigned long long)(run + ex);
        __syncthreads();
        if (threadIdx.x == 0) run_s = run + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const unsigned long long tot = (unsigned long long)run_s;
        const int64_t m = (int64_t)(tot >> 32), bytes = (int64_t)(tot & 0xffffffffull);
        sc->total = tot;
        sc->leading = (sc->first_seq != FA_NONE && sc->first_seq < sc->first_hdr) ? 1ull : 0ull;
        sum->reserved = 0;
        if (bytes + 2 > norm_cap || m + 2 > cap_records) {
            sum->status = RD_FQ_LINES;          // the caller's buffers are too small for this batch: framed again with full-size ones
            sum->n_records = 0;
        } else if (m == 0 && sc->first_seq != FA_NONE) {
            // sequence lines and no header: carried until one arrives; at the end of the stream they are ONE record with an empty header
            // line (the reference's `yield header, seq` with header == ''): rd_fa_emit_kernel writes the lines from position 1 on
            if (final) {
                norm[0] = '\n';
                norm[bytes + 1] = '\n';
                rec_start[0] = 0;
                rec_start[1] = bytes + 2;
                hdr_len[0] = 0;
                sum->n_records = 1;
                sum->consumed = sum->end;
                sum->reserved = bytes + 2;
            } else {
                sum->n_records = 0;
                sum->consumed = sum->begin;
            }
        } else if (m == 0) {                    // nothing but blank lines
            sum->n_records = 0;
            sum->consumed = final ? sum->end : (L ? (int64_t)line_end[L - 1] + 1 : sum->begin);
        }
        // (m > 0: the thread that meets the last header line in rd_fa_emit_kernel closes the batch)
    }
}

__global__ __launch_bounds__(FQ_THREADS) void rd_fa_emit_kernel(const uint8_t *__restrict__ text, const int32_t *__restrict__ info_a, const uint32_t *__restrict__ info_k,
                                                               const unsigned long long *__restrict__ blk, FqSummary *__restrict__ sum,
                                                               const FaScratch *__restrict__ sc, int final, uint8_t *__restrict__ norm,
                                                               int64_t *__restrict__ rec_start, int32_t *__restrict__ hdr_len) {
    __shared__ int64_t sh[4];
    __shared__ int n_long;                               // lines of FA_LONG bytes or more are copied by the whole workgroup (long reads on
    __shared__ int64_t long_dst[FA_BLOCK];               // one line: a thread of its own would walk megabytes byte by byte)
    __shared__ int32_t long_src[FA_BLOCK], long_len[FA_BLOCK];
    if (sum->status != RD_FQ_OK) return;
    const int64_t L = sum->n_lines;
    if ((int64_t)blockIdx.x * FA_BLOCK >= L) return;
    if (threadIdx.x == 0) n_long = 0;
    const int64_t j0 = (int64_t)blockIdx.x * FA_BLOCK + (int64_t)threadIdx.x * FA_LINES;
    const int64_t m = (int64_t)(sc->total >> 32), T = (int64_t)(sc->total & 0xffffffffull);
    // sequence lines in front of the first header: that header's bytes go first (position -1 like every first header), the lines behind
    // them; no header at all: the lines start at position 1, behind the empty header line rd_fa_base_kernel wrote
    const bool leading = sc->leading != 0 && m > 0;
    const int64_t fh = leading ? (int64_t)sc->first_hdr : -1;
    const int64_t len_h0 = leading ? (int64_t)(info_k[fh] & 0x3fffffffu) : 0;
    const int64_t virt = m == 0 ? 2 : 0;
    int a[FA_LINES];
    uint32_t lk[FA_LINES];
    unsigned long long s = 0;
#pragma unroll
    for (int k = 0; k < FA_LINES; ++k) {
        const int64_t j = j0 + k;
        a[k] = 0;
        lk[k] = 0;
        if (j < L) {
            a[k] = info_a[j];
            lk[k] = info_k[j];
            const uint32_t len = lk[k] & 0x3fffffffu, kind = lk[k] >> 30;
            s += kind == 1 ? (1ull << 32) + len + 2ull : kind == 2 ? (unsigned long long)len : 0ull;
        }
    }
    int64_t total;
    unsigned long long E = blk[blockIdx.x] + (unsigned long long)gz_block_scan((int64_t)s, sh, total);
#pragma unroll 1
    for (int k = 0; k < FA_LINES; ++k) {
        const int64_t j = j0 + k;
        if (j >= L) break;
        const int len = (int)(lk[k] & 0x3fffffffu), kind = (int)(lk[k] >> 30);
        int64_t P = (int64_t)(E & 0xffffffffull) - 1 + virt;
        const int64_t hidx = (int64_t)(E >> 32);
        if (leading && j <= fh) P = j == fh ? -1 : P + len_h0 + 2;
        const uint8_t *src = text + a[k];
        if (kind == 1) {
            if (P >= 0) norm[P] = '\n';
            uint8_t *d = norm + P + 1;
            if (len >= FA_LONG) {
                const int at = atomicAdd(&n_long, 1);
                long_dst[at] = P + 1; long_src[at] = a[k]; long_len[at] = len;                 // (a header: copied as it is)
            } else {
                for (int q = 0; q < len; ++q) d[q] = src[q];
            }
            d[len] = '\n';
            rec_start[hidx] = P + 1;
            hdr_len[hidx] = len;
            if (hidx == m - 1) {                 // the last header of the batch: its record is complete only when the stream ends here
                if (final) {
                    const int64_t seq_bytes = T - 1 - (P + 1 + len + 1);
                    rec_start[m] = T;
                    norm[T - 1] = '\n';
                    const bool keep = seq_bytes > 0 || final == 2;   // (the reference drops a last record without a sequence at the end of the
                    sum->n_records = keep ? m : m - 1;               // FILE; final == 2: the end of a rank's share, the file goes on)
                    sum->consumed = sum->end;
                    sum->reserved = keep ? T : P + 1;                  // bytes of `norm` that belong to records
                } else {
                    sum->n_records = m - 1;
                    sum->consumed = leading && hidx == 0 ? sum->begin : a[k];     // (the lines in front of the first header travel with it)
                    sum->reserved = P + 1;
                }
            }
            E += (1ull << 32) + (unsigned long long)len + 2ull;
        } else if (kind == 2) {
            uint8_t *d = norm + P;
            if (len >= FA_LONG) {
                const int at = atomicAdd(&n_long, 1);
                long_dst[at] = P; long_src[at] = a[k]; long_len[at] = -len;                    // (negative: a sequence line, upper-cased)
            } else {
                for (int q = 0; q < len; ++q) {
                    const unsigned c = src[q];
                    d[q] = (uint8_t)((c >= 'a' && c <= 'z') ? c - 32 : c);
                }
            }
            E += (unsigned long long)len;
        }
    }
    __syncthreads();
    const int nl = n_long;
    for (int i = 0; i < nl; ++i) {
        const bool up = long_len[i] < 0;
        const int len = up ? -long_len[i] : long_len[i];
        const uint8_t *src = text + long_src[i];
        uint8_t *d = norm + long_dst[i];
        for (int q = threadIdx.x; q < len; q += FQ_THREADS) {
            const unsigned c = src[q];
            d[q] = (uint8_t)((up && c >= 'a' && c <= 'z') ? c - 32 : c);
        }
    }
}

__global__ __launch_bounds__(FQ_THREADS) void rd_fa_gather_kernel(const uint8_t *__restrict__ norm, const int64_t *__restrict__ rec_tab, const int32_t *__restrict__ hdr_tab,
                                                                 const FqSummary *__restrict__ sum, int64_t lo, int64_t hi, uint8_t *__restrict__ out_text,
                                                                 int64_t out_cap, const int64_t *__restrict__ cursor_in, int64_t *__restrict__ cursor_out,
                                                                 int64_t *__restrict__ rec_start, int64_t *__restrict__ seq_off, int32_t *__restrict__ seq_len) {
    const int64_t d0 = *cursor_in;
    const bool ok = sum->status == RD_FQ_OK && hi <= sum->n_records && lo <= hi && d0 >= 0;
    const int64_t src0 = ok ? rec_tab[lo] : 0, src1 = ok ? rec_tab[hi] : 0;
    const int64_t nbytes = src1 - src0;
    const int64_t gtid = (int64_t)blockIdx.x * FQ_THREADS + threadIdx.x, stride = (int64_t)gridDim.x * FQ_THREADS;
    if (!ok || d0 + nbytes > out_cap) {
        if (gtid == 0) *cursor_out = -1;
        return;
    }
    const uint8_t *src = norm + src0;
    const int64_t d1 = d0 + nbytes;
    for (int64_t o = (d0 & ~(int64_t)15) + 16 * gtid; o < d1; o += 16 * stride) {
        const int64_t a = o < d0 ? d0 : o, e = o + 16 < d1 ? o + 16 : d1;
        if (a == o && e == o + 16) {
            u32x4 v;
            __builtin_memcpy(&v, src + (o - d0), 16);
            *reinterpret_cast<u32x4 *>(out_text + o) = v;
        } else {
            for (int64_t q = a; q < e; ++q) out_text[q] = src[q - d0];
        }
    }
    const int64_t shift = d0 - src0;
    for (int64_t i = gtid; i < hi - lo; i += stride) {
        const int64_t r = lo + i;
        const int64_t rs = rec_tab[r], so = rs + hdr_tab[r] + 1;
        rec_start[i] = rs + shift;
        seq_off[i] = so + shift;
        seq_len[i] = (int32_t)(rec_tab[r + 1] - 1 - so);
    }
    if (gtid == 0) {
        rec_start[hi - lo] = d1;
        *cursor_out = d1;
    }
}

// samples[k] = offset in `norm` where record k * every starts (-1 beyond the batch's records)
__global__ __launch_bounds__(FQ_THREADS) void rd_fa_sample_kernel(const int64_t *__restrict__ rec_tab, const FqSummary *__restrict__ sum, int64_t every,
                                                                 int32_t *__restrict__ samples, int64_t cap) {
    const int64_t n = sum->status == RD_FQ_OK ? sum->n_records : -1;
    const int64_t stride = (int64_t)gridDim.x * FQ_THREADS;
    for (int64_t k = (int64_t)blockIdx.x * FQ_THREADS + threadIdx.x; k < cap; k += stride) samples[k] = k * every <= n ? (int32_t)rec_tab[k * every] : -1;
}

struct FaPlan {
    FqPlan fq;
    int nblk;
    size_t a_off, k_off, blk_off, sc_off, total;
};
FaPlan fa_plan(int64_t text_end, int64_t cap_lines) {
    FaPlan p;
    p.fq = fq_plan(text_end);
    p.nblk = (int)((cap_lines + FA_BLOCK - 1) / FA_BLOCK);
    if (p.nblk < 1) p.nblk = 1;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    p.a_off = up(p.fq.total);
    p.k_off = p.a_off + up((size_t)cap_lines * 4);
    p.blk_off = p.k_off + up((size_t)cap_lines * 4);
    p.sc_off = p.blk_off + up((size_t)p.nblk * 8);
    p.total = p.sc_off + 256;
    return p;
}

}  // namespace
