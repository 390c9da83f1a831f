// rd_pgzip.h - parallel decoding of one gzip member of TEXT (FASTQ/FASTA), for the reader of librd_host.so.
//
// A .gz FASTQ is one DEFLATE stream: a block can only be decoded after everything before it, because matches copy from the
// previous 32 KiB of output. rd_inflate.h decodes that stream at ~1.1 GB/s on one core, i.e. ~4 M reads/s - the slowest stage
// of the whole command by a factor of seven. This decoder applies the two-pass scheme of pugz (Kerbiriou & Chikhi, "Parallel
// decompression of gzip-compressed files and random access to DNA sequences", 2019) to the compressed bytes of one member:
//   1. the compressed bytes are cut into sections; a thread per section SEARCHES the first bit position in its section at
//      which a dynamic-Huffman block starts (complete code-length, literal/length and distance codes, every literal of the
//      block a text byte, a plausible block header behind it);
//   2. every section is decoded from its block start with an UNKNOWN window: the output is 16-bit symbols, a byte or a marker
//      "byte i of the 32 KiB before this section"; it stops at the block start the next section found;
//   3. in order, the markers of a section are replaced with the bytes of the window its predecessor left (the last 32 KiB of
//      each section first, sequentially; then whole sections in parallel), CRC-32s are computed per section and combined.
// Nothing speculative reaches the caller: section j+1 is used only if section j - itself confirmed - ended at exactly the bit
// where j+1 started. Anything else (no block start found, a decode error, a marker into nothing, an input that is not text)
// hands the rest of the file to the sequential decoder (GzipStream::resume), which also produces every error message.
// Members after the first (multi-member files, zero padding) are decoded sequentially as well.
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <thread>

#include "rd_inflate.h"

namespace rdz {

constexpr uint16_t PG_MARK = 0x8000;          // symbol >= PG_MARK: byte (symbol - PG_MARK) of the unknown window
constexpr uint64_t PG_NONE = ~0ull;
constexpr size_t PG_PAD = 64;                  // readable zero bytes behind the compressed bytes of a batch

inline uint64_t pg_load64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}
// the 57+ bits from bit position `bit` on
inline uint64_t pg_peek(const uint8_t *z, uint64_t bit) { return pg_load64(z + (bit >> 3)) >> (bit & 7); }

struct PgTables {
    std::vector<uint32_t> lit, dist, pre;
};

inline bool pg_kraft_complete(const uint8_t *lens, int n) {   // sum 2^-l == 1 over the non-zero lengths
    uint32_t s = 0;
    for (int i = 0; i < n; ++i)
        if (lens[i]) s += 1u << (15 - lens[i]);
    return s == (1u << 15);
}

// Dynamic block header at bit `pos` (the three header bits already consumed): code lengths -> tables. strict = what a block START
// candidate must satisfy (complete codes, as every deflate encoder writes them); otherwise the acceptance of the sequential decoder.
// Returns the bit position of the first symbol, or PG_NONE.
inline uint64_t pg_dynamic_header(const uint8_t *z, uint64_t zbits, uint64_t pos, bool strict, PgTables &t) {
    if (pos + 14 > zbits) return PG_NONE;
    uint64_t v = pg_peek(z, pos);
    const unsigned hlit = (v & 31) + 257, hdist = ((v >> 5) & 31) + 1, hclen = ((v >> 10) & 15) + 4;
    if (hlit > 286 || hdist > 30) return PG_NONE;
    pos += 14;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t pl[19] = {0};
    if (pos + 3 * hclen > zbits) return PG_NONE;
    for (unsigned i = 0; i < hclen; ++i) {
        pl[order[i]] = (uint8_t)(pg_peek(z, pos) & 7);
        pos += 3;
    }
    if (strict && !pg_kraft_complete(pl, 19)) return PG_NONE;
    if (!build_table(pl, 19, PRE_BITS, K_PRE, t.pre)) return PG_NONE;
    uint8_t lens[320 + 140];
    unsigned i = 0;
    const unsigned total = hlit + hdist;
    while (i < total) {
        if (pos + 16 > zbits) return PG_NONE;
        v = pg_peek(z, pos);
        const uint32_t e = t.pre[v & ((1u << PRE_BITS) - 1)];
        if (e & F_BAD) return PG_NONE;
        pos += e_len(e);
        v >>= e_len(e);
        const unsigned sym = e_val(e);
        if (sym < 16) {
            lens[i++] = (uint8_t)sym;
            continue;
        }
        unsigned rep, val = 0;
        if (sym == 16) {
            if (i == 0) return PG_NONE;
            val = lens[i - 1];
            rep = 3 + (v & 3);
            pos += 2;
        } else if (sym == 17) {
            rep = 3 + (v & 7);
            pos += 3;
        } else {
            rep = 11 + (v & 127);
            pos += 7;
        }
        if (i + rep > total) return PG_NONE;
        memset(lens + i, (int)val, rep);
        i += rep;
    }
    if (lens[256] == 0) return PG_NONE;
    if (strict) {
        if (!pg_kraft_complete(lens, (int)hlit)) return PG_NONE;
        int nd = 0;
        for (unsigned k = 0; k < hdist; ++k) nd += lens[hlit + k] != 0;
        if (nd > 1 && !pg_kraft_complete(lens + hlit, (int)hdist)) return PG_NONE;   // a single distance code may be incomplete
    }
    if (!build_table(lens, (int)hlit, LIT_BITS, K_LITLEN, t.lit)) return PG_NONE;
    if (!build_table(lens + hlit, (int)hdist, DIST_BITS, K_DIST, t.dist)) return PG_NONE;
    return pos;
}

inline bool pg_is_text(unsigned c) { return (c >= 32 && c < 127) || c == '\n' || c == '\r' || c == '\t'; }

// One section: 16-bit symbols; sym[0 .. 32768) is the unknown window (markers), the decoded data follows.
struct PgSymBuf {                 // uninitialised, reused from batch to batch (a std::vector would zero 20 MB per section and batch)
    uint16_t *p = nullptr;
    size_t cap = 0;
    PgSymBuf() = default;
    PgSymBuf(const PgSymBuf &) = delete;
    PgSymBuf &operator=(const PgSymBuf &) = delete;
    ~PgSymBuf() { free(p); }
    bool reserve(size_t n) {
        if (n <= cap) return true;
        void *q = realloc(p, n * sizeof(uint16_t));
        if (!q) return false;
        p = (uint16_t *)q;
        cap = n;
        return true;
    }
    uint16_t *data() { return p; }
    const uint16_t *data() const { return p; }
    size_t size() const { return cap; }
};
struct PgBytes {                  // resolved bytes of a section: uninitialised, recycled once handed out
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    PgBytes() = default;
    PgBytes(const PgBytes &) = delete;
    PgBytes &operator=(const PgBytes &) = delete;
    PgBytes(PgBytes &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
    PgBytes &operator=(PgBytes &&o) noexcept {
        std::swap(p, o.p);
        std::swap(n, o.n);
        std::swap(cap, o.cap);
        return *this;
    }
    ~PgBytes() { free(p); }
    bool resize(size_t k) {
        if (k > cap) {
            free(p);
            p = (uint8_t *)malloc(k + 64);
            cap = p ? k : 0;
            if (!p) { n = 0; return false; }
        }
        n = k;
        return true;
    }
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
};
struct PgSection {
    PgSymBuf sym;
    size_t n = 0;                 // symbols decoded (behind the 32768 markers)
    uint64_t start = PG_NONE;     // bit position of its first block header
    uint64_t end = 0;             // bit position behind its last block
    int stop_index = -1;          // index into the stop list of the block start it ended at
    bool member_end = false;      // its last block was the member's final block
    bool ok = false;
    PgBytes bytes;                // after resolution
    uint32_t crc = 0;
};

struct PgFixed {
    std::vector<uint32_t> lit, dist;
    PgFixed() {
        uint8_t l[288], d[32];
        for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
        memset(d, 5, 32);
        build_table(l, 288, LIT_BITS, K_LITLEN, lit);
        build_table(d, 32, DIST_BITS, K_DIST, dist);
    }
};

// Decode blocks from the header at bit `start`. Stops (ok) when a block header sits exactly on one of the ascending `stops`
// (stop_index = which), or behind the member's final block (member_end). text_only: fail on a literal that is not a text byte and
// decode at most max_blocks blocks (block-start search). z must be readable for PG_PAD bytes beyond zbits/8.
inline bool pg_decode(const uint8_t *z, uint64_t zbits, uint64_t start, const uint64_t *stops, int nstops, size_t max_out, bool text_only,
                      int max_blocks, PgSection &o, size_t size_hint = 0) {
    static const PgFixed fixed;
    PgTables t;
    if (!o.sym.reserve(32768 + std::max<size_t>(1u << 20, size_hint))) return false;
    for (unsigned i = 0; i < 32768; ++i) o.sym.data()[i] = (uint16_t)(PG_MARK + i);
    uint16_t *out = o.sym.data();
    size_t op = 32768, cap = o.sym.size();
    uint64_t pos = start;
    int si = 0, blocks = 0;
    o.ok = false;
    o.member_end = false;
    o.stop_index = -1;
    o.start = start;
    for (;;) {
        while (si < nstops && stops[si] < pos) ++si;   // a block start that was passed over was not one
        if (si < nstops && stops[si] == pos && pos != start) {
            o.stop_index = si;
            break;
        }
        if (max_blocks && blocks == max_blocks) break;
        if (pos + 3 > zbits) return false;
        const unsigned hdr = (unsigned)(pg_peek(z, pos) & 7);
        const bool final_block = hdr & 1;
        const unsigned type = hdr >> 1;
        pos += 3;
        ++blocks;
        if (type == 3) return false;
        if (type == 0) {
            pos = (pos + 7) & ~7ull;
            if (pos + 32 > zbits) return false;
            const uint8_t *b = z + (pos >> 3);
            const unsigned len = b[0] | (b[1] << 8), nlen = b[2] | (b[3] << 8);
            if ((len ^ nlen) != 0xffff) return false;
            pos += 32;
            if (pos + 8ull * len > zbits) return false;
            if (op + len + 600 > cap) {
                if (op + len > max_out) return false;
                if (!o.sym.reserve(cap = std::max(cap * 2, op + len + 600))) return false;
                out = o.sym.data();
            }
            b += 4;
            for (unsigned i = 0; i < len; ++i) {
                if (text_only && !pg_is_text(b[i])) return false;
                out[op + i] = b[i];
            }
            op += len;
            pos += 8ull * len;
        } else {
            const uint32_t *lt, *dt;
            if (type == 1) {
                lt = fixed.lit.data();
                dt = fixed.dist.data();
            } else {
                pos = pg_dynamic_header(z, zbits, pos, false, t);
                if (pos == PG_NONE) return false;
                lt = t.lit.data();
                dt = t.dist.data();
            }
            // symbols: the loop of GzipStream::huff on 16-bit output
            uint64_t bb = 0;
            unsigned bc = 0;
            size_t p = pos >> 3;
            {
                const unsigned skip = (unsigned)(pos & 7);
                bb = z[p] >> skip;
                bc = 8 - skip;
                ++p;
            }
            const size_t plimit = (size_t)(zbits >> 3) + 8;   // decoding may look at (zero) padding, never beyond it
            constexpr uint32_t LM = (1u << LIT_BITS) - 1, DM = (1u << DIST_BITS) - 1;
#define PG_REFILL()                          \
    do {                                     \
        bb |= pg_load64(z + p) << bc;        \
        p += (63 - bc) >> 3;                 \
        bc |= 56;                            \
    } while (0)
#define PG_CONSUME(n) \
    do {              \
        bb >>= (n);   \
        bc -= (n);    \
    } while (0)
#define PG_LOOKUP(e)                                                                                \
    do {                                                                                            \
        e = lt[bb & LM];                                                                            \
        if (e & F_SUB) e = lt[e_val(e) + ((uint32_t)(bb >> LIT_BITS) & ((1u << e_extra(e)) - 1))];  \
    } while (0)
            for (;;) {
                if (op + 600 > cap) {
                    if (op > max_out) return false;
                    if (!o.sym.reserve(cap = cap * 2)) return false;
                    out = o.sym.data();
                }
                if (p > plimit) return false;
                PG_REFILL();
                uint32_t e;
                PG_LOOKUP(e);
                if (e & F_LIT) {
                    PG_CONSUME(e_len(e));
                    const unsigned c0 = e_val(e);
                    out[op++] = (uint16_t)c0;
                    if (text_only && !pg_is_text(c0)) return false;
                    PG_LOOKUP(e);
                    if (e & F_LIT) {
                        PG_CONSUME(e_len(e));
                        const unsigned c1 = e_val(e);
                        out[op++] = (uint16_t)c1;
                        if (text_only && !pg_is_text(c1)) return false;
                        PG_LOOKUP(e);
                        if (e & F_LIT) {
                            PG_CONSUME(e_len(e));
                            const unsigned c2 = e_val(e);
                            out[op++] = (uint16_t)c2;
                            if (text_only && !pg_is_text(c2)) return false;
                            continue;
                        }
                    }
                    PG_REFILL();
                }
                if (e & (F_EOB | F_BAD)) {
                    if (e & F_BAD) return false;
                    PG_CONSUME(e_len(e));
                    break;
                }
                const unsigned ll = e_len(e), le = e_extra(e);
                size_t len = e_val(e) + ((uint32_t)(bb >> ll) & ((1u << le) - 1));
                PG_CONSUME(ll + le);
                uint32_t d = dt[bb & DM];
                if (d & F_SUB) d = dt[e_val(d) + ((uint32_t)(bb >> DIST_BITS) & ((1u << e_extra(d)) - 1))];
                if (d & F_BAD) return false;
                const unsigned dl = e_len(d), de = e_extra(d);
                const size_t dist = e_val(d) + ((uint32_t)(bb >> dl) & ((1u << de) - 1));
                PG_CONSUME(dl + de);
                if (dist > op) return false;   // beyond even the unknown window
                uint16_t *dst = out + op;
                const uint16_t *src = dst - dist;
                op += len;
                if (dist >= 8) {              // 16-byte copies = 8 symbols (most matches in FASTQ text are 4-16 long); the slack absorbs the overshoot
                    long left = (long)len;
                    do {
                        memcpy(dst, src, 16);
                        dst += 8;
                        src += 8;
                        left -= 8;
                    } while (left > 0);
                } else if (dist >= 4) {       // 8-byte copies = 4 symbols; in order, so a distance of 4..7 reads what was just written
                    long left = (long)len;
                    do {
                        memcpy(dst, src, 8);
                        dst += 4;
                        src += 4;
                        left -= 4;
                    } while (left > 0);
                } else {
                    for (size_t i = 0; i < len; ++i) dst[i] = src[i];
                }
            }
#undef PG_LOOKUP
#undef PG_REFILL
#undef PG_CONSUME
            pos = (uint64_t)p * 8 - bc;
            if (pos > zbits) return false;
        }
        if (final_block) {
            o.member_end = true;
            break;
        }
    }
    o.n = op - 32768;
    o.end = pos;
    o.ok = true;
    return true;
}

// First bit position in [from, to) where a block START candidate passes every check (see the header of this file), or PG_NONE.
inline uint64_t pg_find_block(const uint8_t *z, uint64_t zbits, uint64_t from, uint64_t to) {
    PgTables t;
    PgSection trial;
    if (to + 64 > zbits) to = zbits > 64 ? zbits - 64 : 0;
    for (uint64_t pos = from; pos < to; ++pos) {
        const uint64_t v = pg_peek(z, pos);
        if ((v & 7) != 4) continue;                               // not final, dynamic Huffman
        if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) continue;
        const uint64_t first = pg_dynamic_header(z, zbits, pos + 3, true, t);
        if (first == PG_NONE) continue;
        if (!pg_decode(z, zbits, pos, nullptr, 0, 8u << 20, true, 1, trial) || trial.member_end || trial.n == 0) continue;
        // behind it: another dynamic block with complete codes, or a stored block whose two length fields agree
        const uint64_t nx = trial.end;
        if (nx + 64 > zbits) continue;
        const unsigned h = (unsigned)(pg_peek(z, nx) & 7);
        const unsigned type = h >> 1;
        bool plausible = false;
        if (type == 2) plausible = pg_dynamic_header(z, zbits, nx + 3, true, t) != PG_NONE;
        else if (type == 0) {
            const uint64_t a = (nx + 3 + 7) & ~7ull;
            if (a + 32 <= zbits) {
                const uint8_t *b = z + (a >> 3);
                plausible = (((b[0] | (b[1] << 8)) ^ (b[2] | (b[3] << 8))) == 0xffff);
            }
        }
        if (plausible) return pos;
    }
    return PG_NONE;
}

class ParallelGzip {
  public:
    std::string err;
    // statistics (tests, tools/host_bench.py)
    uint64_t sections_used = 0, sections_dropped = 0, batches = 0;
    bool fell_back = false;

    // `fp` stays owned by the caller and is positioned at the start of the file
    ParallelGzip(FILE *fp, int threads, size_t section_bytes) : fp_(fp), T_(threads < 2 ? 2 : threads), sec_(section_bytes < 4096 ? 4096 : section_bytes) {
        struct stat st;
        fsize_ = fstat(fileno(fp), &st) == 0 ? (uint64_t)st.st_size : 0;
    }
    ~ParallelGzip() {
        if (bg_.joinable()) bg_.join();
        if (getenv("RD_PGZ_TIMES"))   // where the batches spent their time (tools/host_bench.py)
            fprintf(stderr, "pgzip: read %.3f search %.3f decode %.3f resolve %.3f s\n", t_read, t_search, t_decode, t_resolve);
        delete seq_;
    }

    // up to `cap` decompressed bytes into dst; 0 = end of the stream, -1 = error (see err). Same contract as GzipStream::read.
    long read(uint8_t *dst, size_t cap) {
        size_t done = 0;
        while (done < cap) {
            if (mode_ == M_SEQ) {
                const long got = seq_->read(dst + done, cap - done);
                if (got < 0) {
                    err = seq_->err;
                    if (done) return (long)done;   // what precedes the error first (the next call reports it again)
                    return -1;
                }
                if (got == 0) break;
                done += (size_t)got;
                continue;
            }
            if (mode_ == M_DONE) break;
            if (mode_ == M_FAIL) return done ? (long)done : -1;
            if (ri_ == ready_.size()) {
                if (!indexed_) {   // recycle the section buffers of the batch just handed out (a bounded pool; the rest is freed)
                    std::lock_guard<std::mutex> lk(spare_m_);
                    for (auto &b : ready_)
                        if (spare_.size() < 2 * ((size_t)T_ + 1)) spare_.push_back(std::move(b));
                }
                ready_.clear();
                ri_ = roff_ = 0;
                if (!started_) {
                    started_ = true;
                    if (!member_header()) continue;   // mode_ says what happens next
                }
                next_step();
                continue;
            }
            const PgBytes &b = ready_[ri_];
            const size_t n = std::min(cap - done, b.size() - roff_);
            memcpy(dst + done, b.data() + roff_, n);
            roff_ += n;
            done += n;
            if (roff_ == b.size()) {
                ++ri_;
                roff_ = 0;
            }
        }
        return (long)done;
    }

  private:
    enum Mode { M_PAR, M_SEQ, M_DONE, M_FAIL };
    FILE *fp_;
    int T_;
    size_t sec_;
    uint64_t fsize_ = 0;
    Mode mode_ = M_PAR;
    bool started_ = false;
    uint64_t next_bit_ = 0;            // confirmed position (bits from the start of the file) of the next block header
    std::vector<uint8_t> window_;      // the last <= 32 KiB of the member's output
    uint32_t crc_ = 0;
    uint64_t total_ = 0;               // bytes of the member produced so far
    std::vector<PgBytes> ready_, staged_, spare_;   // being handed out; produced by the batch running in the background; recycled
    std::mutex spare_m_;
    std::thread bg_;
    size_t ri_ = 0, roff_ = 0;
    GzipStream *seq_ = nullptr;
    std::vector<uint8_t> blob_;
    std::vector<PgSection> pool_;
    double ratio_hint_ = 4.0;

    void fail(const char *m) {
        err = m;
        mode_ = M_FAIL;
    }
    // everything from the confirmed position on is decoded by the sequential decoder (which reports any error in its own words)
    void sequential_from_here() {
        fell_back = true;
        seq_ = new GzipStream(fp_, nullptr, 0);
        if (!seq_->resume(next_bit_ >> 3, (unsigned)(next_bit_ & 7), window_.data(), window_.size(), crc_, total_)) {
            err = seq_->err;
            mode_ = M_FAIL;
            return;
        }
        mode_ = M_SEQ;
    }
    void sequential_member_at(uint64_t byte_off) {   // a further member (or garbage): the sequential decoder from its header on
        seq_ = new GzipStream(fp_, nullptr, 0);
        if (!seq_->restart_at(byte_off)) {
            err = seq_->err;
            mode_ = M_FAIL;
            return;
        }
        mode_ = M_SEQ;
    }

    bool pread_all(uint8_t *dst, size_t n, uint64_t off) {
        size_t done = 0;
        while (done < n) {
            const ssize_t k = pread(fileno(fp_), dst + done, n - done, (off_t)(off + done));
            if (k <= 0) return false;
            done += (size_t)k;
        }
        return true;
    }

    // RFC 1952 header at `off`: its length, and the member's total size when an extra subfield gives it (BGZF 'B','C': 16 bits; this
    // build's writer 'R','D': 32 bits), else 0
    bool parse_header(uint64_t off, size_t &hlen, uint64_t &msize) {
        uint8_t h[1 << 12];
        if (off + 18 > fsize_) return false;
        const size_t n = (size_t)std::min<uint64_t>(sizeof(h), fsize_ - off);
        if (!pread_all(h, n, off) || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8) return false;
        const unsigned flg = h[3];
        size_t p = 10;
        msize = 0;
        if (flg & 4) {
            if (p + 2 > n) return false;
            const size_t xlen = h[p] | ((size_t)h[p + 1] << 8);
            p += 2;
            if (p + xlen > n) return false;
            for (size_t q = p; q + 4 <= p + xlen;) {
                const size_t sl = h[q + 2] | ((size_t)h[q + 3] << 8);
                if (q + 4 + sl > p + xlen) break;
                if (h[q] == 'B' && h[q + 1] == 'C' && sl == 2) msize = (uint64_t)(h[q + 4] | (h[q + 5] << 8)) + 1;
                if (h[q] == 'R' && h[q + 1] == 'D' && sl == 4)
                    msize = (uint64_t)(h[q + 4] | (h[q + 5] << 8) | (h[q + 6] << 16) | ((uint64_t)h[q + 7] << 24)) + 1;
                q += 4 + sl;
            }
            p += xlen;
        }
        for (unsigned bit = 8; bit <= 16; bit <<= 1) {
            if (!(flg & bit)) continue;
            while (p < n && h[p]) ++p;
            ++p;
        }
        if (flg & 2) p += 2;
        if (p + 8 > n) return false;
        hlen = p;
        return true;
    }

    // ---- indexed mode: every member says how long it is, so the members are walked without decoding and decoded in parallel ----
    bool indexed_ = false;
    uint64_t member_off_ = 0;
    struct Member { uint64_t off; size_t hlen; uint64_t size; };
    void run_batch_indexed() {
        ++batches;
        std::vector<Member> ms;
        uint64_t bytes = 0;
        const uint64_t budget = (uint64_t)T_ * sec_;
        uint64_t off = member_off_;
        Then after = TH_BATCH;
        while (bytes < budget && ms.size() < 16384) {
            while (off < fsize_) {   // zero padding between members
                uint8_t z;
                if (!pread_all(&z, 1, off) || z != 0) break;
                ++off;
            }
            if (off >= fsize_) {
                after = TH_DONE;
                break;
            }
            Member m{off, 0, 0};
            if (!parse_header(off, m.hlen, m.size) || m.size == 0 || m.size < m.hlen + 8 || off + m.size > fsize_) {
                after = TH_MEMBER;   // a member without its size, damage, garbage: the sequential decoder from here on
                next_member_ = off;
                break;
            }
            ms.push_back(m);
            bytes += m.size;
            off += m.size;
        }
        std::vector<PgBytes> outs(ms.size());
        std::vector<char> ok(ms.size(), 0);
        parallel_for((int)ms.size(), [&](int i) {
            const Member &m = ms[(size_t)i];
            std::vector<uint8_t> z((size_t)m.size);
            if (!pread_all(z.data(), z.size(), m.off)) return;
            const uint8_t *t = z.data() + m.size - 8;
            const uint32_t want_crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
            const uint32_t isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
            if ((uint64_t)isize > 1032ull * m.size + 64 || !outs[(size_t)i].resize(isize)) return;   // deflate cannot expand more than 1032:1
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) return;
            zs.next_in = z.data() + m.hlen;
            zs.avail_in = (uInt)(m.size - m.hlen - 8);
            uint8_t dummy;
            zs.next_out = isize ? outs[(size_t)i].data() : &dummy;
            zs.avail_out = isize;
            const int rc = inflate(&zs, Z_FINISH);
            const bool fine = rc == Z_STREAM_END && zs.avail_in == 0 && zs.total_out == isize;
            inflateEnd(&zs);
            if (fine && crc32_update(0, outs[(size_t)i].data(), isize) == want_crc) ok[(size_t)i] = 1;
        });
        for (size_t i = 0; i < ms.size(); ++i) {
            if (!ok[i]) {   // the sequential decoder finds the words for what is wrong with this member
                after = TH_MEMBER;
                next_member_ = ms[i].off;
                off = ms[i].off;
                break;
            }
            staged_.push_back(std::move(outs[i]));
            ++sections_used;
        }
        member_off_ = off;
        then_ = after;
    }

    bool member_header() {   // RFC 1952 header at offset 0 -> next_bit_
        {
            size_t hl;
            uint64_t ms;
            if (parse_header(0, hl, ms) && ms) {
                indexed_ = true;
                member_off_ = 0;
                return true;
            }
        }
        uint8_t h[1 << 16];
        const size_t n = (size_t)std::min<uint64_t>(sizeof(h), fsize_);
        if (n < 18 || !pread_all(h, n, 0) || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8) {
            sequential_member_at(0);   // too short, unreadable or not gzip: the sequential decoder says which
            return false;
        }
        const unsigned flg = h[3];
        size_t p = 10;
        bool ok = true;
        if (flg & 4) {
            if (p + 2 > n) ok = false;
            else p += 2 + (h[p] | ((size_t)h[p + 1] << 8));
        }
        for (unsigned bit = 8; ok && bit <= 16; bit <<= 1) {
            if (!(flg & bit)) continue;
            while (p < n && h[p]) ++p;
            ++p;
        }
        if (flg & 2) p += 2;
        if (!ok || p + 8 > n) {
            sequential_member_at(0);
            return false;
        }
        next_bit_ = (uint64_t)p * 8;
        return true;
    }

    template <typename F>
    void parallel_for(int n, F f) {
        std::atomic<int> next(0);
        const int nt = std::min(T_, n);
        std::vector<std::thread> th;
        for (int k = 1; k < nt; ++k)
            th.emplace_back([&]() {
                for (int i; (i = next.fetch_add(1)) < n;) f(i);
            });
        for (int i; (i = next.fetch_add(1)) < n;) f(i);
        for (auto &t : th) t.join();
    }

    double t_search = 0, t_decode = 0, t_resolve = 0, t_read = 0;   // seconds per phase, summed over the batches (RD_PGZ_TIMES)
    static double now() {
        struct timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
    }
    void run_batch() {
        if (indexed_) return run_batch_indexed();
        ++batches;
        double t0 = now();
        const uint64_t b0 = next_bit_ >> 3;                       // first byte of the batch
        const uint64_t want = (uint64_t)(T_ + 1) * sec_;           // T sections + the one the last of them may run into
        const uint64_t have = std::min<uint64_t>(want, fsize_ - b0);
        if (blob_.size() < (size_t)have + PG_PAD) blob_.resize((size_t)want + PG_PAD);
        memset(blob_.data() + have, 0, PG_PAD);
        if (!pread_all(blob_.data(), (size_t)have, b0)) {
            then_ = TH_RESUME;   // (decided here, done by the reading thread once it has handed out what is ready)
            return;
        }
        t_read += now() - t0;
        t0 = now();
        const uint8_t *z = blob_.data();
        const uint64_t zbits = have * 8;
        const bool to_eof = b0 + have == fsize_;
        // 1. block starts: section 0 starts at the confirmed position; sections 1..T at the first block start found in their range
        const int ns = (int)std::min<uint64_t>((uint64_t)T_ + 1, (have + sec_ - 1) / sec_);
        std::vector<uint64_t> S((size_t)ns, PG_NONE);
        S[0] = next_bit_ - b0 * 8;
        parallel_for(ns - 1, [&](int i) {
            const int j = i + 1;
            S[(size_t)j] = pg_find_block(z, zbits, (uint64_t)j * sec_ * 8, std::min<uint64_t>((uint64_t)(j + 1) * sec_ * 8, zbits));
        });
        t_search += now() - t0;
        t0 = now();
        std::vector<uint64_t> stops;
        std::vector<int> stop_section;
        for (int j = 1; j < ns; ++j)
            if (S[(size_t)j] != PG_NONE) {
                stops.push_back(S[(size_t)j]);
                stop_section.push_back(j);
            }
        // 2. decode: every section that has a start, up to the next start that turns out to be real. The section at the LAST start found
        //    is decoded only when the file ends inside this batch: otherwise nothing says where it ends, and the next batch begins there.
        int last = 0;
        for (int j = 1; j < ns; ++j)
            if (S[(size_t)j] != PG_NONE) last = j;
        if (last == 0 && !to_eof) {   // no block start in T sections: not the kind of stream this decoder is for
            then_ = TH_RESUME;   // (decided here, done by the reading thread once it has handed out what is ready)
            return;
        }
        if (pool_.size() < (size_t)T_ + 1) pool_ = std::vector<PgSection>((size_t)T_ + 1);
        std::vector<PgSection> &sec = pool_;
        for (auto &s : sec) {
            s.ok = false;
            s.bytes.n = 0;
        }
        const size_t max_out = std::max<size_t>(16u << 20, sec_ * 32);   // symbols per section; beyond that (a 32:1 text) the sequential decoder takes over
        parallel_for(ns, [&](int j) {
            if (S[(size_t)j] == PG_NONE) return;
            if (j == last && !to_eof) return;
            pg_decode(z, zbits, S[(size_t)j], stops.data(), (int)stops.size(), max_out, false, 0, sec[(size_t)j], (size_t)(sec_ * ratio_hint_) + (1u << 20));
        });
        t_decode += now() - t0;
        t0 = now();
        // 3. the chain of confirmed sections
        std::vector<int> chain;
        int cur = 0;
        bool member_end = false, broken = false;
        for (;;) {
            PgSection &s = sec[(size_t)cur];
            if (!s.ok) {
                broken = true;
                break;
            }
            chain.push_back(cur);
            if (s.member_end) {
                member_end = true;
                break;
            }
            const int nx = stop_section[(size_t)s.stop_index];
            sections_dropped += (uint64_t)(nx - cur - 1);
            cur = nx;
            if (cur == last && !to_eof) break;   // the next batch starts there
        }
        if (chain.empty()) {   // not even the section at the confirmed position decoded: damaged data, binary payload, or a block longer than the batch
            then_ = TH_RESUME;   // (decided here, done by the reading thread once it has handed out what is ready)
            return;
        }
        // 4. resolve the markers: windows first (sequential, 32 KiB per section), then whole sections in parallel
        const size_t nc = chain.size();
        std::vector<std::vector<uint8_t>> win(nc + 1);
        win[0] = window_;
        bool bad_marker = false;
        // symbol -> byte through a 64 K-entry table per section (bytes map to themselves, marker i to byte i of the window the predecessor
        // left; 0x100 + .. and markers before the start of the member to "invalid"): branch-free, the table stays in the L2 cache
        auto make_lut = [&](const std::vector<uint8_t> &w, std::vector<uint16_t> &lut) {
            lut.assign(65536, 0x100);
            for (unsigned v = 0; v < 256; ++v) lut[v] = (uint16_t)v;
            const size_t wn = w.size();
            for (size_t k = 0; k < wn; ++k) lut[PG_MARK + 32768 - wn + k] = w[k];
        };
        auto resolve = [&](const PgSection &s, const std::vector<uint16_t> &lut, size_t from, size_t to, uint8_t *dst) -> bool {
            const uint16_t *sy = s.sym.data() + 32768;
            const uint16_t *lt = lut.data();
            unsigned bad = 0;
            size_t i = from;
            for (; i + 16 <= to; i += 16) {   // sixteen plain bytes at a time when no marker is among them
                const __m128i a = _mm_loadu_si128((const __m128i *)(sy + i)), b = _mm_loadu_si128((const __m128i *)(sy + i + 8));
                if (_mm_movemask_epi8(_mm_or_si128(a, b)) & 0xaaaa) {
                    for (size_t k = i; k < i + 16; ++k) {
                        const unsigned v = lt[sy[k]];
                        bad |= v;
                        dst[k - from] = (uint8_t)v;
                    }
                } else {
                    _mm_storeu_si128((__m128i *)(dst + (i - from)), _mm_packus_epi16(a, b));
                }
            }
            for (; i < to; ++i) {
                const unsigned v = lt[sy[i]];
                bad |= v;
                dst[i - from] = (uint8_t)v;
            }
            return !(bad & 0x100);
        };
        std::vector<std::vector<uint16_t>> luts(nc);
        for (size_t c = 0; c < nc; ++c) {
            const PgSection &s = sec[(size_t)chain[c]];
            std::vector<uint8_t> &nw = win[c + 1];
            if (s.n >= 32768) {
                nw.resize(32768);
                make_lut(win[c], luts[c]);
                if (!resolve(s, luts[c], s.n - 32768, s.n, nw.data())) bad_marker = true;
            } else {
                const size_t keep = std::min(win[c].size(), 32768 - s.n);
                nw.assign(win[c].end() - (long)keep, win[c].end());
                nw.resize(keep + s.n);
                make_lut(win[c], luts[c]);
                if (!resolve(s, luts[c], 0, s.n, nw.data() + keep)) bad_marker = true;
            }
        }
        {   // recycled output buffers (no page faults)
            std::lock_guard<std::mutex> lk(spare_m_);
            for (size_t c = 0; c < nc && !spare_.empty(); ++c) {
                PgSection &s = sec[(size_t)chain[c]];
                if (s.bytes.cap < s.n) {
                    s.bytes = std::move(spare_.back());
                    spare_.pop_back();
                }
            }
        }
        std::atomic<bool> bad(bad_marker);
        parallel_for((int)nc, [&](int c) {
            PgSection &s = sec[(size_t)chain[(size_t)c]];
            if (!s.bytes.resize(s.n) || !resolve(s, luts[(size_t)c], 0, s.n, s.bytes.data())) bad = true;
            s.crc = crc32_update(0, s.bytes.data(), s.bytes.size());
        });
        if (bad) {   // a match reaching before the start of the member: let the sequential decoder say so
            then_ = TH_RESUME;   // (decided here, done by the reading thread once it has handed out what is ready)
            return;
        }
        for (size_t c = 0; c < nc; ++c) {
            PgSection &s = sec[(size_t)chain[c]];
            crc_ = (uint32_t)crc32_combine(crc_, s.crc, (z_off_t)s.bytes.size());
            total_ += s.bytes.size();
            staged_.push_back(std::move(s.bytes));
            ++sections_used;
        }
        t_resolve += now() - t0;
        {   // output per compressed byte of this batch: sizes the symbol buffers of the next one
            uint64_t outb = 0;
            for (size_t c = 0; c < nc; ++c) outb += sec[(size_t)chain[c]].n;
            const uint64_t inb = (sec[(size_t)chain[nc - 1]].end - S[0]) / 8 + 1;
            ratio_hint_ = std::min(24.0, std::max(ratio_hint_, 1.15 * (double)outb / (double)inb));
        }
        window_ = win[nc];
        const PgSection &tail = sec[(size_t)chain[nc - 1]];
        next_bit_ = b0 * 8 + tail.end;
        if (member_end) {
            const uint64_t tb = (next_bit_ + 7) >> 3;   // trailer: CRC-32, ISIZE
            uint8_t t[8];
            if (tb + 8 > fsize_ || !pread_all(t, 8, tb)) return then_error("Compressed file ended before the end-of-stream marker was reached");
            const uint32_t want_crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
            const uint32_t want_size = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
            if (crc_ != want_crc) return then_error("CRC check failed");
            if ((uint32_t)total_ != want_size) return then_error("Incorrect length of data produced");
            uint64_t q = tb + 8;   // zero padding, then the end of the file or another member
            uint8_t buf[4096];
            for (;;) {
                if (q >= fsize_) {
                    then_ = TH_DONE;
                    return;
                }
                const size_t k = (size_t)std::min<uint64_t>(sizeof(buf), fsize_ - q);
                if (!pread_all(buf, k, q)) return then_error("read error");
                size_t i = 0;
                while (i < k && buf[i] == 0) ++i;
                q += i;
                if (i < k) {
                    next_member_ = q;
                    then_ = TH_MEMBER;
                    return;
                }
            }
        }
        if (broken) then_ = TH_RESUME;   // the chain ended at a section that did not decode: sequentially from the end of the last good one
    }

    // what happens once the sections resolved by the last batch have been handed out
    enum Then { TH_BATCH, TH_DONE, TH_MEMBER, TH_RESUME, TH_ERROR };
    Then then_ = TH_BATCH;
    uint64_t next_member_ = 0;
    std::string then_err_;
    void then_error(const char *m) {
        then_err_ = m;
        then_ = TH_ERROR;
    }
    // The next batch is decoded in the background while the reader hands out (and the parser consumes) the current one: run_batch
    // touches only the decoder state and staged_, never mode_ or ready_.
    void start_background() {
        if (then_ == TH_BATCH) bg_ = std::thread([this]() { run_batch(); });
    }
    void next_step() {   // ready_ is drained
        if (bg_.joinable()) {
            bg_.join();
            ready_.swap(staged_);
            if (!ready_.empty()) {
                start_background();
                return;
            }
        }
        const Then t = then_;
        then_ = TH_BATCH;
        switch (t) {
        case TH_BATCH:
            run_batch();
            ready_.swap(staged_);
            if (!ready_.empty()) start_background();
            break;
        case TH_DONE: mode_ = M_DONE; break;
        case TH_MEMBER: sequential_member_at(next_member_); break;
        case TH_RESUME: sequential_from_here(); break;
        case TH_ERROR: fail(then_err_.c_str()); break;
        }
    }
};

}  // namespace rdz
