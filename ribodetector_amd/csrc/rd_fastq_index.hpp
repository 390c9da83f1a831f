// rd_fastq_index.hpp - the FASTQ record index of a text buffer in HBM (rd_fq_* kernels), and rd_select_pack
// Part of the single translation unit rd_kernels.hip (included from there, in order); DESIGN.md §3.12 has the numbers.
//
// Replaces, for text that is already on the device (BGZF members inflated there, rd_inflate_dev.hpp; plain files copied there), the
// host parser's FASTQ state machine - reference data_loader/fastx_parser.py:15-37: four lines per record, every line rstrip()-ed,
// header / '+' line / quality kept verbatim, the sequence NOT upper-cased - with the rules of this build's host reader
// (csrc/rd_host.cpp rd_reader_next): a header line must start with '@' (the reference mis-frames silently there), the last line may
// lack its '\n', fewer than four trailing lines are tolerated when they are blank.
//
// The text arrives in BATCHES (one launch of the inflate kernel, or one piece of a plain file). A batch's buffer has a PAD in front:
//   rd_fq_begin_kernel   copies the CARRY of the batch before - the bytes behind its last complete record, read from that batch's
//                        summary ON THE DEVICE - right-aligned in front of the new bytes, so that successive batches chain on the
//                        stream without a host round trip; appends the '\n' a final batch may lack
//   rd_fq_count_kernel   newlines per 16 KiB tile (64 contiguous bytes per thread: a 64-bit newline mask from byte-parallel compares)
//   rd_fq_scan_kernel    one workgroup: tile counts -> tile bases; lines, records
//   rd_fq_fill_kernel    line_end[j] = offset of the '\n' that ends line j (int32: a batch buffer is < 2 GiB)
//   rd_fq_check_kernel   per record: header starts with '@'; any line with trailing whitespace (CR LF files) marks the batch DIRTY - the
//                        host then strips such a batch (rd_fastq_strip_mark + a compaction) and indexes the result: records are
//                        verbatim ranges of the text in the common case and the writer's bytes are the reference's
//                        ('\n'.join(stripped lines) + '\n', detect.py:489-492) in both
//   rd_fq_gather_kernel  records [lo, hi) of a batch -> a chunk: text copied behind a device-side cursor (16-byte pieces: one unaligned
//                        load, one aligned store), rec_start / seq_off / seq_len of the chunk written - chunks of exactly N records are
//                        assembled from consecutive batches without the host ever seeing an offset
// All HBM-bound and small next to the recurrence: a 218-byte record costs 218 B read twice (count, fill) + 16 B of index, then the
// gather's copy; measured numbers in DESIGN.md.
#pragma once
#include "rd_common.hpp"
#include "rd_deflate.hpp"

namespace {

constexpr int FQ_THREADS = 256;
constexpr int FQ_BYTES = 64;                        // contiguous bytes per thread
constexpr int FQ_TILE = FQ_THREADS * FQ_BYTES;      // 16 KiB per workgroup

struct FqSummary {       // = rd_fq_summary of the C ABI (64 bytes)
    int64_t begin, end, n_lines, n_records, consumed;
    unsigned long long bad_record;      // ~0 = none
    int32_t status, dirty;
    int64_t reserved;
};
static_assert(sizeof(FqSummary) == 64 && sizeof(rd_fq_summary) == 64, "rd_fq_summary layout");

// Python str.rstrip() whitespace for the byte values that can occur (= csrc/rd_host.cpp is_ws)
__device__ __forceinline__ bool fq_is_ws(unsigned c) {
    return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f) || c == 0x85 || c == 0xa0;
}

// 64-bit mask of the '\n' bytes among the 64 bytes at text + o (o is a multiple of 64; the buffer is readable up to the next multiple
// of 64 behind `end`), restricted to [begin, end)
__device__ __forceinline__ uint64_t fq_newline_mask(const uint8_t *__restrict__ text, int64_t o, int64_t begin, int64_t end) {
    if (o >= end || o + FQ_BYTES <= begin) return 0;
    const u32x4 *p = reinterpret_cast<const u32x4 *>(text + o);
    uint64_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const u32x4 v = p[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t x = v[k] ^ 0x0a0a0a0au;
            const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);      // 0x80 in every byte that was '\n'
            const uint64_t nib = (((z >> 7) * 0x204081u) >> 21) & 0xfu;                    // those four flags as four adjacent bits
            m |= nib << (16 * q + 4 * k);
        }
    }
    const int lo = begin > o ? (int)(begin - o) : 0, hi = end - o < FQ_BYTES ? (int)(end - o) : FQ_BYTES;
    if (lo > 0) m &= ~0ull << lo;
    if (hi < 64) m &= ~(~0ull << hi);
    return m;
}

__device__ __forceinline__ uint32_t fq_block_scan(uint32_t v, uint32_t *sh, uint32_t &total) {   // exclusive scan over 256 threads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += sh[w];
    total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(FQ_THREADS) void rd_fq_begin_kernel(uint8_t *__restrict__ text, int64_t pad, int64_t end, const uint8_t *__restrict__ prev_text,
                                                                const FqSummary *__restrict__ prev, int final, FqSummary *__restrict__ sum) {
    __shared__ int64_t s_begin;
    __shared__ int s_status;
    if (threadIdx.x == 0) {
        int64_t begin = pad;
        int status = RD_FQ_OK;
        if (prev) {
            const int64_t carry = prev->end - prev->consumed;
            if (prev->status != RD_FQ_OK) status = RD_FQ_CHAIN;      // the batch before failed: this one cannot be framed
            else if (carry < 0 || carry > pad) status = RD_FQ_CARRY;                   // a record longer than the pad
            else begin = pad - carry;
        }
        s_begin = begin;
        s_status = status;
    }
    __syncthreads();
    const int64_t begin = s_begin;
    if (prev && s_status == RD_FQ_OK) {
        const uint8_t *src = prev_text + prev->consumed;
        for (int64_t i = threadIdx.x; i < pad - begin; i += FQ_THREADS) text[begin + i] = src[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t e = end;
        if (final && e > begin && text[e - 1] != '\n') text[e++] = '\n';      // the last line of a file may lack its terminator
        sum->begin = begin;
        sum->end = e;
        sum->n_lines = sum->n_records = 0;
        sum->consumed = begin;
        sum->bad_record = ~0ull;
        sum->status = s_status;
        sum->dirty = 0;
        sum->reserved = 0;
    }
}

__global__ __launch_bounds__(FQ_THREADS) void rd_fq_count_kernel(const uint8_t *__restrict__ text, const FqSummary *__restrict__ sum,
                                                                uint32_t *__restrict__ tile_count) {
    __shared__ uint32_t sh[4];
    const int64_t o = (int64_t)blockIdx.x * FQ_TILE + (int64_t)threadIdx.x * FQ_BYTES;
    const uint64_t m = sum->status == RD_FQ_OK ? fq_newline_mask(text, o, sum->begin, sum->end) : 0;
    uint32_t total;
    fq_block_scan((uint32_t)__popcll(m), sh, total);
    if (threadIdx.x == 0) tile_count[blockIdx.x] = total;
}

// one workgroup: tile_count[] -> exclusive bases in place; lines and records of the window
__global__ __launch_bounds__(FQ_THREADS) void rd_fq_scan_kernel(uint32_t *__restrict__ tile_count, int ntiles, FqSummary *__restrict__ sum, int64_t cap_lines) {
    __shared__ uint32_t sh[4];
    __shared__ uint32_t run_s;
    if (threadIdx.x == 0) run_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < ntiles; b0 += FQ_THREADS) {
        const int b = b0 + threadIdx.x;
        const uint32_t v = b < ntiles ? tile_count[b] : 0;
        uint32_t total;
        const uint32_t ex = fq_block_scan(v, sh, total);
        const uint32_t run = run_s;
        if (b < ntiles) tile_count[b] = run + ex;
        __syncthreads();
        if (threadIdx.x == 0) run_s = run + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int64_t lines = run_s;
        if (lines > cap_lines) {            // (cannot happen with cap_lines >= window bytes; a smaller table is the caller's choice)
            sum->status = RD_FQ_LINES;
            lines = 0;
        }
        sum->n_lines = lines;
        sum->n_records = lines / 4;
    }
}

__global__ __launch_bounds__(FQ_THREADS) void rd_fq_fill_kernel(const uint8_t *__restrict__ text, const FqSummary *__restrict__ sum,
                                                               const uint32_t *__restrict__ tile_base, int32_t *__restrict__ line_end) {
    __shared__ uint32_t sh[4];
    if (sum->status != RD_FQ_OK) return;
    const int64_t o = (int64_t)blockIdx.x * FQ_TILE + (int64_t)threadIdx.x * FQ_BYTES;
    uint64_t m = fq_newline_mask(text, o, sum->begin, sum->end);
    uint32_t total;
    uint32_t j = tile_base[blockIdx.x] + fq_block_scan((uint32_t)__popcll(m), sh, total);
    while (m) {
        const int b = __ffsll((long long)m) - 1;
        line_end[j++] = (int32_t)(o + b);
        m &= m - 1;
    }
}

__device__ __forceinline__ int64_t fq_line_start(const int32_t *__restrict__ line_end, int64_t j, int64_t begin) {
    return j ? (int64_t)line_end[j - 1] + 1 : begin;
}

__global__ __launch_bounds__(FQ_THREADS) void rd_fq_check_kernel(const uint8_t *__restrict__ text, const int32_t *__restrict__ line_end,
                                                                FqSummary *__restrict__ sum, int final) {
    // (TRUNCATED is this kernel's own verdict, written by workgroup 0 while others may still be starting: it must not stop them)
    if (sum->status != RD_FQ_OK && sum->status != RD_FQ_TRUNCATED) return;
    const int64_t n = sum->n_records, begin = sum->begin, L = sum->n_lines;
    const int64_t stride = (int64_t)gridDim.x * FQ_THREADS;
    bool dirty = false;
    for (int64_t r = (int64_t)blockIdx.x * FQ_THREADS + threadIdx.x; r < n; r += stride) {
        int64_t ls = fq_line_start(line_end, 4 * r, begin);
        if (text[ls] != '@') atomicMin(&sum->bad_record, (unsigned long long)r);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t p = line_end[4 * r + k];
            dirty |= p > ls && fq_is_ws(text[p - 1]);
            ls = p + 1;
        }
    }
    if (dirty) sum->dirty = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int64_t consumed = n ? (int64_t)line_end[4 * n - 1] + 1 : begin;
        if (final) {
            // fewer than four lines behind the last record: tolerated when blank (csrc/rd_host.cpp rd_reader_next; a line of
            // whitespace only marks the batch dirty and is judged after the strip)
            bool trunc = false, ws = false;
            for (int64_t j = 4 * n; j < L; ++j) {
                const int64_t ls = fq_line_start(line_end, j, begin), p = line_end[j];
                if (p > ls) {
                    if (fq_is_ws(text[p - 1])) ws = true; else trunc = true;
                }
            }
            if (ws) sum->dirty = 1;
            else if (trunc) sum->status = RD_FQ_TRUNCATED;
            consumed = sum->end;
        }
        sum->consumed = consumed;
    }
}

// (after rd_fq_check_kernel: a bad header is a status of its own once every workgroup has voted)
__global__ void rd_fq_verdict_kernel(FqSummary *__restrict__ sum) {
    // (a bad header in front of a truncated tail is the first damage of the stream: it is what the host reader meets first, too)
    if ((sum->status == RD_FQ_OK || sum->status == RD_FQ_TRUNCATED) && sum->bad_record != ~0ull && !sum->dirty) sum->status = RD_FQ_HEADER;
}

__global__ __launch_bounds__(FQ_THREADS) void rd_fq_gather_kernel(const uint8_t *__restrict__ text, const int32_t *__restrict__ line_end,
                                                                 const FqSummary *__restrict__ sum, int64_t lo, int64_t hi,
                                                                 uint8_t *__restrict__ out_text, int64_t out_cap, const int64_t *__restrict__ cursor_in,
                                                                 int64_t *__restrict__ cursor_out, int64_t *__restrict__ rec_start,
                                                                 int64_t *__restrict__ seq_off, int32_t *__restrict__ seq_len) {
    const int64_t begin = sum->begin;
    const int64_t d0 = *cursor_in;
    // (a batch that holds a malformed record - HEADER, TRUNCATED - is framed up to it: the records in front of it can be gathered)
    const bool framed = sum->status == RD_FQ_OK || sum->status == RD_FQ_HEADER || sum->status == RD_FQ_TRUNCATED;
    const bool ok = framed && hi <= sum->n_records && lo <= hi && d0 >= 0;
    const int64_t src0 = ok ? fq_line_start(line_end, 4 * lo, begin) : 0, src1 = ok ? fq_line_start(line_end, 4 * hi, begin) : 0;
    const int64_t nbytes = src1 - src0;
    const int64_t gtid = (int64_t)blockIdx.x * FQ_THREADS + threadIdx.x, stride = (int64_t)gridDim.x * FQ_THREADS;
    if (!ok || d0 + nbytes > out_cap) {           // a cursor of -1 poisons every later piece of the chunk: the host sees it in the chunk's total
        if (gtid == 0) *cursor_out = -1;
        return;
    }
    const uint8_t *src = text + src0;
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
        const int64_t rs = fq_line_start(line_end, 4 * r, begin), so = (int64_t)line_end[4 * r] + 1;
        rec_start[i] = rs + shift;
        seq_off[i] = so + shift;
        seq_len[i] = (int32_t)((int64_t)line_end[4 * r + 1] - so);
    }
    if (gtid == 0) {
        rec_start[hi - lo] = d1;
        *cursor_out = d1;
    }
}

// del[q] = 1 for every byte of the trailing whitespace run of lines [0, n_lines) - what rstrip() removes (one thread per line)
__global__ __launch_bounds__(FQ_THREADS) void rd_fq_strip_mark_kernel(const uint8_t *__restrict__ text, const int32_t *__restrict__ line_end,
                                                                     const FqSummary *__restrict__ sum, uint8_t *__restrict__ del) {
    const int64_t L = sum->n_lines, begin = sum->begin;
    const int64_t stride = (int64_t)gridDim.x * FQ_THREADS;
    for (int64_t j = (int64_t)blockIdx.x * FQ_THREADS + threadIdx.x; j < L; j += stride) {
        const int64_t ls = fq_line_start(line_end, j, begin);
        for (int64_t q = line_end[j]; q > ls && fq_is_ws(text[q - 1]); --q) del[q - 1] = 1;
    }
}

// samples[k] = offset where record k * every starts (k * every <= n_records): with the summary's `consumed` they bound the bytes of
// any record range from the host without a round trip per range
__global__ __launch_bounds__(FQ_THREADS) void rd_fq_sample_kernel(const int32_t *__restrict__ line_end, const FqSummary *__restrict__ sum, int64_t every,
                                                                 int32_t *__restrict__ samples, int64_t cap) {
    const bool framed = sum->status == RD_FQ_OK || sum->status == RD_FQ_HEADER || sum->status == RD_FQ_TRUNCATED;
    const int64_t n = framed ? sum->n_records : -1, begin = sum->begin;
    const int64_t stride = (int64_t)gridDim.x * FQ_THREADS;
    for (int64_t k = (int64_t)blockIdx.x * FQ_THREADS + threadIdx.x; k < cap; k += stride)
        samples[k] = k * every <= n ? (int32_t)fq_line_start(line_end, 4 * k * every, begin) : -1;
}

// bytes moved by a kernel instead of a DMA engine: dst / src are device memory or pinned host memory (mapped into the device's address
// space). hipMemcpyAsync between pinned memory and HBM goes to an SDMA queue that other streams' copies share IN ORDER - a 96 MB H2D of
// the feeder was seen waiting 60-100 ms behind a label D2H that itself waited for two recurrence launches; a kernel on the feeder's own
// stream waits for nothing but a free workgroup slot (~1 ms). 64 bytes per thread and trip keep ~55 GB/s of PCIe busy from 8 workgroups.
constexpr int COPY_THREADS = 1024;   // FEW, fat workgroups: a workgroup that copies holds its CU from the recurrence kernel for as long as the copy runs (that kernel's
                                     // waves take whole SIMDs); 8 x 1,024 threads keep as many bytes in flight as 32 x 256 did, on 8 CUs instead of 32
__global__ __launch_bounds__(COPY_THREADS) void rd_copy_kernel(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, int64_t n) {
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t head = (16 - ((uintptr_t)dst & 15)) & 15;                 // bytes before dst's first 16-byte boundary
    if (gtid < head && gtid < n) dst[gtid] = src[gtid];
    const int64_t body = n > head ? (n - head) / 16 : 0;                    // 16-byte pieces, aligned at dst
    u32x4 *d = reinterpret_cast<u32x4 *>(dst + head);
    const uint8_t *s0 = src + head;
    int64_t i = gtid;
    for (; i + 3 * stride < body; i += 4 * stride) {
        u32x4 v0, v1, v2, v3;
        __builtin_memcpy(&v0, s0 + 16 * i, 16);
        __builtin_memcpy(&v1, s0 + 16 * (i + stride), 16);
        __builtin_memcpy(&v2, s0 + 16 * (i + 2 * stride), 16);
        __builtin_memcpy(&v3, s0 + 16 * (i + 3 * stride), 16);
        d[i] = v0; d[i + stride] = v1; d[i + 2 * stride] = v2; d[i + 3 * stride] = v3;
    }
    for (; i < body; i += stride) {
        u32x4 v;
        __builtin_memcpy(&v, s0 + 16 * i, 16);
        d[i] = v;
    }
    const int64_t tail0 = head + 16 * body;
    if (tail0 + gtid < n && gtid < 16) dst[tail0 + gtid] = src[tail0 + gtid];
}

struct FqPlan {
    int ntiles;
    size_t count_bytes, total;
};
FqPlan fq_plan(int64_t text_end) {
    FqPlan p;
    p.ntiles = (int)((text_end + 1 + FQ_TILE - 1) / FQ_TILE);
    if (p.ntiles < 1) p.ntiles = 1;
    p.count_bytes = ((size_t)p.ntiles * 4 + 255) / 256 * 256;
    p.total = p.count_bytes;
    return p;
}

}  // namespace
