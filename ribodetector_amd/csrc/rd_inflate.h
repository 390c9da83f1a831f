// rd_inflate.h - streaming gzip (RFC 1952) / DEFLATE (RFC 1951) decoder behind the FASTQ/FASTA reader of librd_host.so.
//
// The reference reads .gz input through Python's gzip module (data_loader/fastx_parser.py:15-17 and
// data_loader/seq_encoder.py:75-92 open the file by extension); zlib's inflate decodes ~0.3 GB/s here, i.e. ~1.3 M reads/s,
// an order of magnitude below what the kernels consume. This decoder is built for long runs of literals and short matches
// (FASTQ text): a 64-bit bit buffer refilled once per length/distance pair, packed 32-bit table entries with an 11-bit
// (literal/length) and 10-bit (distance) first level, up to three literals per refill, 8-byte match copies into a buffer
// with slack. The member CRC-32 and ISIZE are verified like gzip.GzipFile does (a mismatch is an error, zero padding
// between/after members is skipped); the CRC is folded 64 bytes at a time with carry-less multiplies (PCLMULQDQ,
// ~10 GB/s, against ~1 GB/s for zlib 1.2.11's table-driven crc32, which would cost more than the decoding itself).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <immintrin.h>
#include <zlib.h>   // crc32_z only (tails shorter than 64 bytes, CPUs without PCLMULQDQ)

#include <algorithm>
#include <string>
#include <vector>

namespace rdz {

constexpr uint32_t F_LIT = 1u << 31, F_EOB = 1u << 30, F_SUB = 1u << 29, F_BAD = 1u << 28;
constexpr int LIT_BITS = 11, DIST_BITS = 10, PRE_BITS = 7;   // FASTQ matches reach far back: long distance codes are common
constexpr size_t WIN = 32768, CHUNK = 2u << 20, OSLACK = 258 + 64, OSAFE = WIN + CHUNK;
constexpr size_t IN_CAP = 1u << 20, IN_PAD = 128;

inline uint32_t e_len(uint32_t e) { return e & 0xff; }
inline uint32_t e_extra(uint32_t e) { return (e >> 8) & 0x1f; }
inline uint32_t e_val(uint32_t e) { return (e >> 13) & 0x7fff; }
inline uint32_t mk(uint32_t flags, uint32_t val, uint32_t extra, uint32_t len) { return flags | (val << 13) | (extra << 8) | len; }

// CRC-32 (IEEE 802.3, reflected) of a multiple of 16 bytes, >= 64, by folding with carry-less multiplication (Gopal et
// al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009). crc = the raw register
// (the complemented running value). Checked against zlib's crc32 in tests/test_host.py.
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_fold(const uint8_t *buf, size_t len, uint32_t crc) {
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};   // x^(512+32), x^(512-32) mod P
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};   // x^(128+32), x^(128-32) mod P
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0};                 // x^64 mod P
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};   // P and floor(x^64 / P)
#define RDZ_FOLD(acc, k, next)                                          \
    do {                                                                \
        const __m128i lo_ = _mm_clmulepi64_si128(acc, k, 0x00);         \
        acc = _mm_clmulepi64_si128(acc, k, 0x11);                       \
        acc = _mm_xor_si128(_mm_xor_si128(acc, lo_), next);             \
    } while (0)
    const __m128i *p = (const __m128i *)buf;
    __m128i x1 = _mm_loadu_si128(p), x2 = _mm_loadu_si128(p + 1), x3 = _mm_loadu_si128(p + 2), x4 = _mm_loadu_si128(p + 3);
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    __m128i k = _mm_load_si128((const __m128i *)k1k2);
    p += 4;
    len -= 64;
    for (; len >= 64; len -= 64, p += 4) {
        RDZ_FOLD(x1, k, _mm_loadu_si128(p));
        RDZ_FOLD(x2, k, _mm_loadu_si128(p + 1));
        RDZ_FOLD(x3, k, _mm_loadu_si128(p + 2));
        RDZ_FOLD(x4, k, _mm_loadu_si128(p + 3));
    }
    k = _mm_load_si128((const __m128i *)k3k4);
    RDZ_FOLD(x1, k, x2);
    RDZ_FOLD(x1, k, x3);
    RDZ_FOLD(x1, k, x4);
    for (; len >= 16; len -= 16, ++p) RDZ_FOLD(x1, k, _mm_loadu_si128(p));
#undef RDZ_FOLD
    const __m128i m32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k, 0x10);       // 128 -> 96 bits
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    k = _mm_loadl_epi64((const __m128i *)k5k0);          // 96 -> 64 bits
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, m32), k, 0x00), t);
    k = _mm_load_si128((const __m128i *)poly);           // Barrett reduction 64 -> 32 bits
    t = _mm_and_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, m32), k, 0x10), m32);
    x1 = _mm_xor_si128(x1, _mm_clmulepi64_si128(t, k, 0x00));
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

// same contract as zlib's crc32_z
inline uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {
    static const bool fast = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (fast && n >= 64) {
        const size_t m = n & ~(size_t)15;
        crc = ~crc32_fold(p, m, ~crc);
        p += m;
        n -= m;
    }
    return (uint32_t)crc32_z(crc, p, n);
}

enum Kind { K_LITLEN, K_DIST, K_PRE };

// table entry for symbol `sym` of an alphabet, consuming `len` code bits
inline uint32_t sym_entry(Kind kind, int sym, int len) {
    if (kind == K_PRE) return mk(0, (uint32_t)sym, 0, (uint32_t)len);
    if (kind == K_LITLEN) {
        if (sym < 256) return mk(F_LIT, (uint32_t)sym, 0, (uint32_t)len);
        if (sym == 256) return mk(F_EOB, 0, 0, (uint32_t)len);
        const int k = sym - 257;
        if (k < 8) return mk(0, (uint32_t)(3 + k), 0, (uint32_t)len);
        if (k < 28) {
            const int ex = (k >> 2) - 1;
            return mk(0, (uint32_t)(((4 + (k & 3)) << ex) + 3), (uint32_t)ex, (uint32_t)len);
        }
        if (k == 28) return mk(0, 258, 0, (uint32_t)len);
        return mk(F_BAD, 0, 0, (uint32_t)len);   // 286, 287 never occur in valid data
    }
    if (sym < 2) return mk(0, (uint32_t)(sym + 1), 0, (uint32_t)len);
    if (sym < 30) {
        const int ex = (sym >> 1) - 1;
        return mk(0, (uint32_t)(((2 + (sym & 1)) << ex) + 1), (uint32_t)ex, (uint32_t)len);
    }
    return mk(F_BAD, 0, 0, (uint32_t)len);       // 30, 31
}

// Canonical Huffman decoding table from code lengths: tab[0 .. 1<<tbits) is indexed by the next tbits input bits (LSB
// first); codes longer than tbits go through F_SUB entries (value = offset of the second-level table, extra = its index
// bits). Entries no code reaches are F_BAD. Returns false for an over-subscribed set of lengths.
inline bool build_table(const uint8_t *lens, int n, int tbits, Kind kind, std::vector<uint32_t> &tab) {
    int count[16] = {0};
    for (int s = 0; s < n; ++s) ++count[lens[s]];
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;
    }
    uint32_t next[16];
    uint32_t code = 0;
    for (int l = 1; l <= 15; ++l) {
        code = (code + (uint32_t)count[l - 1]) << 1;
        next[l] = code;
    }
    const uint32_t tsize = 1u << tbits, tmask = tsize - 1;
    tab.assign(tsize, mk(F_BAD, 0, 0, 1));
    uint16_t rev[320];
    uint8_t submax[1 << LIT_BITS];
    bool any_long = false;
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        uint32_t c = next[l]++, r = 0;
        for (int b = 0; b < l; ++b) r |= ((c >> b) & 1u) << (l - 1 - b);
        rev[s] = (uint16_t)r;
        if (l <= tbits) {
            const uint32_t e = sym_entry(kind, s, l);
            for (uint32_t i = r; i < tsize; i += 1u << l) tab[i] = e;
        } else {
            if (!any_long) {
                memset(submax, 0, tsize);
                any_long = true;
            }
            submax[r & tmask] = std::max<uint8_t>(submax[r & tmask], (uint8_t)l);
        }
    }
    if (!any_long) return true;
    for (uint32_t p = 0; p < tsize; ++p) {
        if (!submax[p]) continue;
        const uint32_t sb = (uint32_t)submax[p] - (uint32_t)tbits, off = (uint32_t)tab.size();
        tab[p] = mk(F_SUB, off, sb, 0);
        tab.resize(off + (1u << sb), mk(F_BAD, 0, 0, (uint32_t)tbits + 1));
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l <= tbits) continue;
        const uint32_t r = rev[s], p = r & tmask, sb = e_extra(tab[p]), off = e_val(tab[p]);
        const uint32_t e = sym_entry(kind, s, l);
        for (uint32_t i = r >> tbits; i < (1u << sb); i += 1u << (l - tbits)) tab[off + i] = e;
    }
    return true;
}

class GzipStream {
  public:
    std::string err;

    // `fp` stays owned by the caller; `prefix` = bytes already read from it (the magic the caller peeked at)
    GzipStream(FILE *fp, const uint8_t *prefix, size_t nprefix) : fp_(fp) {
        ib_.assign(IN_CAP + IN_PAD, 0);
        if (nprefix) memcpy(ib_.data(), prefix, nprefix);
        iend_ = nprefix;
        ob_.assign(WIN + CHUNK + OSLACK, 0);
    }

    // up to `cap` decompressed bytes into dst; 0 = end of the stream, -1 = error (see err)
    long read(uint8_t *dst, size_t cap) {
        size_t done = 0;
        while (done < cap) {
            if (rp_ == op_) {
                if (failed_) return -1;
                if (finished_) break;
                if (!produce()) {
                    failed_ = true;
                    if (done) break;   // hand over what precedes the error first; the next call reports it
                    return -1;
                }
                continue;
            }
            const size_t n = std::min(cap - done, op_ - rp_);
            memcpy(dst + done, ob_.data() + rp_, n);
            rp_ += n;
            done += n;
        }
        return (long)done;
    }

    // Continue a member in its middle (rd_pgzip.h hands over here): the next block header is `bit_off` bits into byte `byte_off`
    // of the file; `window` = the last <= 32 KiB produced so far; crc / produced = CRC-32 and size of everything produced so far.
    bool resume(uint64_t byte_off, unsigned bit_off, const uint8_t *window, size_t wlen, uint32_t crc, uint64_t produced) {
        if (!seek_input(byte_off)) return false;
        if (iend_ == 0) return fail("Compressed file ended before the end-of-stream marker was reached");
        bitbuf_ = (uint64_t)(ib_[0] >> bit_off);
        bitcnt_ = 8 - bit_off;
        ip_ = 1;
        const size_t keep = std::min(wlen, WIN);
        if (keep) memcpy(ob_.data() + WIN - keep, window + (wlen - keep), keep);
        hist0_ = WIN - keep;
        op_ = rp_ = crc_from_ = WIN;
        crc_ = crc;
        base_abs_ = produced;
        member_abs_ = 0;
        state_ = ST_BLOCK;
        final_block_ = finished_ = failed_ = false;
        return true;
    }
    // Start over at a member header at byte `byte_off` of the file.
    bool restart_at(uint64_t byte_off) {
        if (!seek_input(byte_off)) return false;
        bitbuf_ = 0;
        bitcnt_ = 0;
        op_ = rp_ = hist0_ = crc_from_ = WIN;
        state_ = ST_MEMBER;
        final_block_ = finished_ = failed_ = false;
        return true;
    }

  private:
    enum State { ST_MEMBER, ST_BLOCK, ST_STORED, ST_HUFF, ST_TRAILER };

    FILE *fp_;
    std::vector<uint8_t> ib_;
    size_t ip_ = 0, iend_ = 0;
    bool in_eof_ = false;
    uint64_t bitbuf_ = 0;
    unsigned bitcnt_ = 0;

    std::vector<uint8_t> ob_;                  // [0, WIN) = the last 32 KiB already handed out, [WIN, op_) = this block
    size_t op_ = WIN, rp_ = WIN, hist0_ = WIN;
    uint64_t base_abs_ = 0, member_abs_ = 0;   // stream offset of ob_[WIN]; offset where the current member began
    uint32_t crc_ = 0;
    size_t crc_from_ = WIN;                    // ob_[crc_from_, op_) belongs to the current member and is not summed yet

    State state_ = ST_MEMBER;
    bool final_block_ = false, finished_ = false, failed_ = false;
    size_t stored_left_ = 0;
    std::vector<uint32_t> lit_, dist_, pre_, fixed_lit_, fixed_dist_;
    const uint32_t *lt_ = nullptr, *dt_ = nullptr;

    bool fail(const char *msg) {
        err = msg;
        return false;
    }

    bool seek_input(uint64_t byte_off) {
        if (fseeko(fp_, (off_t)byte_off, SEEK_SET) != 0) return fail("seek failed");
        ip_ = iend_ = 0;
        in_eof_ = false;
        refill_input();
        return true;
    }
    void refill_input() {
        if (in_eof_) return;
        // Whole bytes waiting in the bit buffer go back into the byte buffer first: what is moved to the front below starts at ip_, and
        // align_byte() - a stored block, i.e. every sync flush of a pigz-made stream - hands unread bytes back by moving ip_ DOWN. (Round
        // 6: a block header within the last KiB of an input buffer, followed by a stored block, moved ip_ below zero: "Compressed file
        // ended ..." in the middle of a good file, about once per GB of pigz output.)
        {
            const unsigned back = bitcnt_ >> 3;
            if (back && back <= ip_) {
                ip_ -= back;
                bitcnt_ &= 7;
                bitbuf_ &= (1ull << bitcnt_) - 1;
            }
        }
        if (ip_ > 0) {
            memmove(ib_.data(), ib_.data() + ip_, iend_ - ip_);
            iend_ -= ip_;
            ip_ = 0;
        }
        while (iend_ < IN_CAP) {
            const size_t got = fread(ib_.data() + iend_, 1, IN_CAP - iend_, fp_);
            if (got == 0) {
                in_eof_ = true;
                break;
            }
            iend_ += got;
        }
        memset(ib_.data() + iend_, 0, IN_PAD);
    }
    bool need(size_t n) {   // n real bytes at ip_
        while (iend_ - ip_ < n) {
            if (in_eof_) return false;
            refill_input();
        }
        return true;
    }
    void soft_need(size_t n) {
        if (iend_ - ip_ < n && !in_eof_) refill_input();
    }
    // A decode step may start while ip_ <= isafe(): it does at most a handful of 8-byte loads, each advancing <= 7 bytes.
    // Before the end of the file that keeps every load inside real data; at the end the loads run into the zero padding
    // (IN_PAD bytes, half of it headroom) and overrun() then reports the truncation.
    size_t isafe() const {
        if (in_eof_) return iend_ + IN_PAD / 2;
        return iend_ >= 24 ? iend_ - 24 : 0;
    }
    bool overrun() const { return (uint64_t)ip_ * 8 > (uint64_t)iend_ * 8 + bitcnt_; }
    // a structural error met while the decoder was (possibly) looking at the zero padding behind a truncated file is
    // reported as the truncation it is
    const char *classify(const char *msg, size_t p, unsigned bc) const {
        if (in_eof_ && (uint64_t)p * 8 + 64 > (uint64_t)iend_ * 8 + bc) return "Compressed file ended before the end-of-stream marker was reached";
        return msg;
    }
    void align_byte() {   // drop the rest of the current byte and hand whole unread bytes back to the byte reader
        const unsigned drop = bitcnt_ & 7;
        bitbuf_ >>= drop;
        bitcnt_ -= drop;
        ip_ -= bitcnt_ >> 3;
        bitbuf_ = 0;
        bitcnt_ = 0;
    }
    static uint64_t load64(const uint8_t *p) {
        uint64_t v;
        memcpy(&v, p, 8);
        return v;   // little-endian host (x86-64)
    }

#define RDZ_REFILL()                                  \
    do {                                              \
        bb |= load64(in + p) << bc;                   \
        p += (63 - bc) >> 3;                          \
        bc |= 56;                                     \
    } while (0)
#define RDZ_CONSUME(n) \
    do {               \
        bb >>= (n);    \
        bc -= (n);     \
    } while (0)

    void slide() {   // everything in [WIN, op_) has been handed out: keep the last 32 KiB as history
        crc_ = crc32_update(crc_, ob_.data() + crc_from_, op_ - crc_from_);
        const size_t keep = std::min(op_ - hist0_, WIN);
        memmove(ob_.data() + WIN - keep, ob_.data() + op_ - keep, keep);
        base_abs_ += op_ - WIN;
        hist0_ = WIN - keep;
        op_ = rp_ = crc_from_ = WIN;
    }

    bool produce() {
        slide();
        for (;;) {
            switch (state_) {
            case ST_MEMBER:
                if (!member_header()) return false;
                state_ = ST_BLOCK;
                break;
            case ST_BLOCK:
                if (!block_header()) return false;
                break;
            case ST_STORED: {
                if (op_ >= OSAFE) return true;
                if (stored_left_ == 0) {
                    state_ = final_block_ ? ST_TRAILER : ST_BLOCK;
                    break;
                }
                if (ip_ == iend_) {
                    if (in_eof_) return fail("Compressed file ended before the end-of-stream marker was reached");
                    refill_input();
                    break;
                }
                const size_t n = std::min(std::min(stored_left_, OSAFE - op_), iend_ - ip_);
                memcpy(ob_.data() + op_, ib_.data() + ip_, n);
                op_ += n;
                ip_ += n;
                stored_left_ -= n;
                break;
            }
            case ST_HUFF: {
                const int rc = huff();
                if (rc < 0) return false;
                if (rc == 1) return true;
                state_ = final_block_ ? ST_TRAILER : ST_BLOCK;
                break;
            }
            case ST_TRAILER:
                if (!trailer()) return false;
                if (finished_) return true;
                state_ = ST_MEMBER;
                break;
            }
        }
    }

    bool member_header() {
        if (!need(10)) return fail("Compressed file ended before the end-of-stream marker was reached");
        const uint8_t *h = ib_.data() + ip_;
        if (h[0] != 0x1f || h[1] != 0x8b) return fail("Not a gzipped file");
        if (h[2] != 8) return fail("Unknown compression method");
        const unsigned flg = h[3];
        ip_ += 10;
        if (flg & 4) {   // FEXTRA
            if (!need(2)) return fail("Compressed file ended before the end-of-stream marker was reached");
            const size_t xlen = ib_[ip_] | ((size_t)ib_[ip_ + 1] << 8);
            ip_ += 2;
            if (!need(xlen)) return fail("Compressed file ended before the end-of-stream marker was reached");
            ip_ += xlen;
        }
        for (unsigned bit = 8; bit <= 16; bit <<= 1) {   // FNAME, FCOMMENT: zero-terminated
            if (!(flg & bit)) continue;
            for (;;) {
                if (!need(1)) return fail("Compressed file ended before the end-of-stream marker was reached");
                if (ib_[ip_++] == 0) break;
            }
        }
        if (flg & 2) {   // FHCRC
            if (!need(2)) return fail("Compressed file ended before the end-of-stream marker was reached");
            ip_ += 2;
        }
        member_abs_ = base_abs_ + (op_ - WIN);
        hist0_ = op_;     // a member's window starts empty: a distance reaching into the previous member is invalid (as in zlib)
        crc_ = 0;
        crc_from_ = op_;
        bitbuf_ = 0;
        bitcnt_ = 0;
        return true;
    }

    bool trailer() {
        align_byte();
        if (ip_ > iend_ || !need(8)) return fail("Compressed file ended before the end-of-stream marker was reached");
        const uint8_t *t = ib_.data() + ip_;
        const uint32_t want_crc = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        const uint32_t want_size = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
        ip_ += 8;
        crc_ = crc32_update(crc_, ob_.data() + crc_from_, op_ - crc_from_);
        crc_from_ = op_;
        if (crc_ != want_crc) return fail("CRC check failed");
        if ((uint32_t)((base_abs_ + (op_ - WIN)) - member_abs_) != want_size) return fail("Incorrect length of data produced");
        for (;;) {   // zero padding, then either the end of the file or another member
            if (ip_ == iend_) {
                if (in_eof_) {
                    finished_ = true;
                    return true;
                }
                refill_input();
                continue;
            }
            if (ib_[ip_] != 0) return true;
            ++ip_;
        }
    }

    bool block_header() {
        soft_need(1024);
        const uint8_t *in = ib_.data();
        uint64_t bb = bitbuf_;
        unsigned bc = bitcnt_;
        size_t p = ip_;
        if (p > isafe()) return fail("Compressed file ended before the end-of-stream marker was reached");
        RDZ_REFILL();
        final_block_ = bb & 1;
        const unsigned type = (bb >> 1) & 3;
        RDZ_CONSUME(3);
        if (type == 3) return fail(classify("Error -3 while decompressing data: invalid block type", p, bc));
        if (type == 0) {
            bitbuf_ = bb;
            bitcnt_ = bc;
            ip_ = p;
            align_byte();
            if (ip_ > iend_ || !need(4)) return fail("Compressed file ended before the end-of-stream marker was reached");
            const unsigned len = ib_[ip_] | (ib_[ip_ + 1] << 8), nlen = ib_[ip_ + 2] | (ib_[ip_ + 3] << 8);
            if ((len ^ nlen) != 0xffff) return fail(classify("Error -3 while decompressing data: invalid stored block lengths", p, bc));
            ip_ += 4;
            stored_left_ = len;
            state_ = ST_STORED;
            return true;
        }
        if (type == 1) {
            if (fixed_lit_.empty()) {
                uint8_t l[288], d[32];
                for (int i = 0; i < 288; ++i) l[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
                memset(d, 5, 32);
                build_table(l, 288, LIT_BITS, K_LITLEN, fixed_lit_);
                build_table(d, 32, DIST_BITS, K_DIST, fixed_dist_);
            }
            lt_ = fixed_lit_.data();
            dt_ = fixed_dist_.data();
        } else {
            const unsigned hlit = (bb & 31) + 257, hdist = ((bb >> 5) & 31) + 1, hclen = ((bb >> 10) & 15) + 4;
            RDZ_CONSUME(14);
            if (hlit > 286 || hdist > 30) return fail(classify("Error -3 while decompressing data: too many length or distance symbols", p, bc));
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t pl[19] = {0};
            RDZ_REFILL();
            for (unsigned i = 0; i < hclen; ++i) {
                if (bc < 3) RDZ_REFILL();
                pl[order[i]] = bb & 7;
                RDZ_CONSUME(3);
                if ((i & 7) == 7) RDZ_REFILL();
            }
            if (!build_table(pl, 19, PRE_BITS, K_PRE, pre_)) return fail(classify("Error -3 while decompressing data: invalid code lengths set", p, bc));
            uint8_t lens[320 + 140];
            unsigned i = 0;
            const unsigned total = hlit + hdist;
            while (i < total) {
                if (p > isafe()) return fail("Compressed file ended before the end-of-stream marker was reached");
                RDZ_REFILL();
                const uint32_t e = pre_[bb & ((1u << PRE_BITS) - 1)];
                if (e & F_BAD) return fail(classify("Error -3 while decompressing data: invalid code lengths set", p, bc));
                RDZ_CONSUME(e_len(e));
                const unsigned sym = e_val(e);
                if (sym < 16) {
                    lens[i++] = (uint8_t)sym;
                    continue;
                }
                unsigned rep, val = 0;
                if (sym == 16) {
                    if (i == 0) return fail(classify("Error -3 while decompressing data: invalid bit length repeat", p, bc));
                    val = lens[i - 1];
                    rep = 3 + (bb & 3);
                    RDZ_CONSUME(2);
                } else if (sym == 17) {
                    rep = 3 + (bb & 7);
                    RDZ_CONSUME(3);
                } else {
                    rep = 11 + (bb & 127);
                    RDZ_CONSUME(7);
                }
                if (i + rep > total) return fail(classify("Error -3 while decompressing data: invalid bit length repeat", p, bc));
                memset(lens + i, (int)val, rep);
                i += rep;
            }
            if (lens[256] == 0) return fail(classify("Error -3 while decompressing data: invalid code -- missing end-of-block", p, bc));
            if (!build_table(lens, (int)hlit, LIT_BITS, K_LITLEN, lit_)) return fail(classify("Error -3 while decompressing data: invalid literal/lengths set", p, bc));
            if (!build_table(lens + hlit, (int)hdist, DIST_BITS, K_DIST, dist_)) return fail(classify("Error -3 while decompressing data: invalid distances set", p, bc));
            lt_ = lit_.data();
            dt_ = dist_.data();
        }
        bitbuf_ = bb;
        bitcnt_ = bc;
        ip_ = p;
        state_ = ST_HUFF;
        return true;
    }

    // decode literal/length + distance symbols of the current block: 0 = end of block, 1 = output buffer full, -1 = error
    int huff() {
        uint8_t *out = ob_.data();
        const uint32_t *lt = lt_, *dt = dt_;
        const uint8_t *in = ib_.data();
        uint64_t bb = bitbuf_;
        unsigned bc = bitcnt_;
        size_t p = ip_, op = op_, safe = isafe();
        const size_t hist0 = hist0_;
        constexpr uint32_t LM = (1u << LIT_BITS) - 1, DM = (1u << DIST_BITS) - 1;
        int rc;
#define RDZ_LOOKUP_LIT(e)                                                                              \
    do {                                                                                               \
        e = lt[bb & LM];                                                                               \
        if (e & F_SUB) e = lt[e_val(e) + ((uint32_t)(bb >> LIT_BITS) & ((1u << e_extra(e)) - 1))];     \
    } while (0)
        for (;;) {
            if (op >= OSAFE) {
                rc = 1;
                break;
            }
            if (p > safe) {
                if (in_eof_) {
                    err = "Compressed file ended before the end-of-stream marker was reached";
                    return -1;
                }
                ip_ = p;
                bitbuf_ = bb;          // (refill_input hands the bit buffer's whole bytes back: it must see the live one)
                bitcnt_ = bc;
                refill_input();
                p = ip_;
                bb = bitbuf_;
                bc = bitcnt_;
                safe = isafe();
                if (p > safe && !in_eof_) {   // cannot happen with IN_CAP >> 24
                    err = "input buffer too small";
                    return -1;
                }
                continue;
            }
            RDZ_REFILL();
            uint32_t e;
            RDZ_LOOKUP_LIT(e);
            if (e & F_LIT) {
                RDZ_CONSUME(e_len(e));
                out[op++] = (uint8_t)e_val(e);
                RDZ_LOOKUP_LIT(e);
                if (e & F_LIT) {
                    RDZ_CONSUME(e_len(e));
                    out[op++] = (uint8_t)e_val(e);
                    RDZ_LOOKUP_LIT(e);
                    if (e & F_LIT) {
                        RDZ_CONSUME(e_len(e));
                        out[op++] = (uint8_t)e_val(e);
                        continue;
                    }
                }
                RDZ_REFILL();   // the low bits (already looked up) are unchanged by a refill
            }
            if (e & (F_EOB | F_BAD)) {
                if (e & F_BAD) {
                    err = classify("Error -3 while decompressing data: invalid literal/length code", p, bc);
                    return -1;
                }
                RDZ_CONSUME(e_len(e));
                rc = 0;
                break;
            }
            const unsigned ll = e_len(e), le = e_extra(e);
            size_t len = e_val(e) + ((uint32_t)(bb >> ll) & ((1u << le) - 1));
            RDZ_CONSUME(ll + le);
            uint32_t d = dt[bb & DM];
            if (d & F_SUB) d = dt[e_val(d) + ((uint32_t)(bb >> DIST_BITS) & ((1u << e_extra(d)) - 1))];
            if (d & F_BAD) {
                err = classify("Error -3 while decompressing data: invalid distance code", p, bc);
                return -1;
            }
            const unsigned dl = e_len(d), de = e_extra(d);
            const size_t dist = e_val(d) + ((uint32_t)(bb >> dl) & ((1u << de) - 1));
            RDZ_CONSUME(dl + de);
            if (dist > op - hist0) {
                err = classify("Error -3 while decompressing data: invalid distance too far back", p, bc);
                return -1;
            }
            uint8_t *dst = out + op;
            const uint8_t *src = dst - dist;
            op += len;
            if (dist >= 8) {
                // most matches in FASTQ text are 4-16 bytes: two unconditional 8-byte copies (in order, so a distance of
                // 8..15 still reads what the first one wrote), a loop only beyond that; the slack behind OSAFE absorbs the overshoot
                memcpy(dst, src, 8);
                memcpy(dst + 8, src + 8, 8);
                if (len > 16) {
                    long left = (long)len - 16;
                    dst += 16;
                    src += 16;
                    do {
                        memcpy(dst, src, 8);
                        dst += 8;
                        src += 8;
                        left -= 8;
                    } while (left > 0);
                }
            } else if (dist == 1) {
                memset(dst, *src, len);
            } else {
                for (size_t i = 0; i < len; ++i) dst[i] = src[i];
            }
        }
#undef RDZ_LOOKUP_LIT
        bitbuf_ = bb;
        bitcnt_ = bc;
        ip_ = p;
        op_ = op;
        if (overrun()) {
            err = "Compressed file ended before the end-of-stream marker was reached";
            return -1;
        }
        return rc;
    }
#undef RDZ_REFILL
#undef RDZ_CONSUME
};

}  // namespace rdz
