// rd_inflate_dev.hpp - gzip members inflated on the device (rd_gz_inflate_kernel): the input side of csrc/rd_deflate.hpp
// Part of the single translation unit rd_kernels.hip (included from there, in order); DESIGN.md §3.11 has the numbers.
//
// What it is for: a .gz whose members say how long they are - BGZF (bgzip, htslib, and every .gz this build's CLI writes: one member
// per 65,280 bytes, 'B','C' subfield) - is a list of independent DEFLATE streams. The reference reads such a file through Python's
// gzip module like any other (reference data_loader/seq_encoder.py:21-39, fastx_parser.py:15-55); round 2 gave the host a parallel
// member decoder (csrc/rd_pgzip.h). Here the members are decoded where the reads are classified: the compressed bytes travel to HBM
// (a fifth of the text), ONE WAVE decodes one member, thousands of members at a time.
//
// How a wave decodes a serial format. Bit position, output position and the current symbol are wave-uniform (SGPRs); the lanes do
// what a CPU decoder does with tables, loops and a 64-bit bit buffer:
//   * THE INPUT lives in registers: lane l holds dwords win + l and win + 64 + l of the stream; the bits at any position are two or
//     three v_readlane away, and the next 256 bytes are requested one register ahead of their use (no load on the symbol path);
//   * SIXTY-FOUR SYMBOLS ARE TRIED AT ONCE: lane i looks the 10 bits that start i bits behind the current position up in the block's
//     table (LDS, 2 KiB) - the symbol that WOULD start there. The wave then walks the chain of real starts (position += code length:
//     a v_readlane, a bit set and an add per literal) until it meets something that is not a short literal; the lanes on the chain
//     store their literals with one instruction (rank among the starts = output offset). DNA and quality lines are runs of 2-6 bit
//     literals: 10-25 symbols per table access;
//   * what stops the walk - a length code, the end of the block, a code longer than 10 bits - is handled with uniform bit
//     arithmetic; a code longer than the table is decoded by COMPARISON: lane L holds first / count / offset of the canonical code's
//     length-L codewords and tests first <= code < first + count, a ballot finds the one lane that says yes. After a match the walk
//     continues in the same 64 positions;
//   * a match is copied by all lanes at once (out[p + k] = out[p - dist + k mod dist]: the same formula for overlapping runs);
//     stores and loads of one wave reach the L1 in program order, so the bytes just written are the bytes read, without a wait;
//   * CRC-32 of the member by the 64 lanes (1 KiB per lane, table in LDS, pieces combined with x^(8 n) mod P as in the deflate
//     kernel) and checked against the member's trailer, like ISIZE: a damaged member is reported, never silently accepted.
// Every DEFLATE block type (stored, fixed, dynamic), any number of blocks per member.
#pragma once
#include "rd_deflate.hpp"

namespace {

struct GzMemberIn {       // one gzip member, described by the host (which walks the headers: it needs no decoding for that)
    int64_t in_off;       // first byte of the member's raw DEFLATE data in the compressed buffer
    int64_t out_off;      // where its ISIZE bytes go in the text buffer
    int32_t in_len;       // bytes of raw DEFLATE data (the 8-byte trailer follows them)
    int32_t out_len;      // ISIZE
};

enum { GZI_OK = 0, GZI_BAD_BLOCK = 1, GZI_BAD_CODE = 2, GZI_BAD_LENGTHS = 3, GZI_OVERRUN = 4, GZI_BAD_DISTANCE = 5, GZI_TRUNCATED = 6, GZI_SIZE = 7,
       GZI_CRC = 8, GZI_STORED = 9, GZI_MEMBER = 10 };

constexpr int GZI_WAVES = 4;        // members per workgroup (one per wave; the waves share nothing but the CRC table)
constexpr int GZI_LBITS = 10;       // literal/length table: the codes of up to 10 bits
constexpr int GZI_DBITS = 9;        // distance table
constexpr int GZI_STEP = 56;        // dwords between the two input registers of a wave

struct __attribute__((aligned(16))) GziWave {
    uint16_t llut[1 << GZI_LBITS];  // symbol | code length << 9 of the codeword that starts the 10 bits (0: none that short)
    uint16_t dlut[1 << GZI_DBITS];
    uint16_t lsym[288];     // literal/length symbols ordered by (code length, symbol)
    uint16_t dsym[32];      // distance symbols likewise
    uint16_t loff[32];      // first position of every code length in the orderings (literal/length: [1..15], distance: [17..31])
    uint8_t len[320];       // code lengths being read
};
struct __attribute__((aligned(16))) GziSmem {
    uint32_t crc_tab[4][256];       // crc_tab[k][b]: the CRC of byte b followed by k zero bytes (slicing-by-4)
    GziWave w[GZI_WAVES];
};

// per-lane view of a canonical Huffman code: lane L in [base + 1, base + 15] describes the codewords of length L - base
struct GziCode { uint32_t first, count, offs; };

__device__ __forceinline__ uint32_t gzi_rl(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// Build the two orderings and the per-lane triples from S.len[0 .. nl) (literal/length) and S.len[nl .. nl + nd) (distance).
// Returns false when a length set is over-subscribed (or incomplete in a way zlib's inflate also rejects).
__device__ __forceinline__ bool gzi_build(GziWave &S, int nl, int nd, int lane, GziCode &lit, GziCode &dst) {
    // counts per length: lane L counts the symbols whose length is L (lit: lanes 1..15; dist: lanes 17..31)
    const bool isd = lane >= 16;
    const int L = lane & 15;
    const int n0 = isd ? nl : 0, n1 = isd ? nl + nd : nl;
    uint32_t cnt = 0;
    if (L >= 1 && lane < 32)
#pragma unroll 4
        for (int s = n0; s < n1; ++s) cnt += S.len[s] == L ? 1u : 0u;
    // first code and offset of every length: a prefix scan over the 15 lengths, done by each lane for itself (15 steps, uniform)
    uint32_t first = 0, offs = 0, code = 0, off = 0;
    int left = 1;
    bool over = false;
    for (int l = 1; l <= 15; ++l) {
        const uint32_t cl = gzi_rl(cnt, l), cd = gzi_rl(cnt, 16 + l);
        const uint32_t c = isd ? cd : cl;
        if (L == l) { first = code; offs = off; }
        code = (code + c) << 1;
        off += c;
        left = (left << 1) - (int)c;
        over = over || left < 0;
    }
    // over-subscribed: invalid. Incomplete codes are let through: a bit string that is no codeword is an error where it is met
    // (zlib accepts an incomplete distance code of one symbol; an all-zero distance code is legal for a block without matches)
    const uint64_t bad = __ballot((lane == 1 || lane == 17) && over);
    if (bad) return false;
    lit = GziCode{first, cnt, offs};
    dst = lit;   // (same registers: a lane is either a literal-code lane or a distance-code lane)
    if (lane < 32) S.loff[lane] = (uint16_t)offs;
    // the orderings: symbol s goes to loff[its length] + (number of earlier symbols of the same length) - lane-parallel over the symbols
#pragma unroll 1
    for (int s0 = 0; s0 < nl; s0 += 64) {
        const int s = s0 + lane;
        const int l = s < nl ? S.len[s] : 0;
        uint32_t before = 0;
#pragma unroll 4
        for (int t = 0; t < s0 + 64 && t < nl; ++t) before += (t < s && S.len[t] == l) ? 1u : 0u;
        if (l) S.lsym[S.loff[l] + before] = (uint16_t)s;
    }
    {
        const int s = lane;
        const int l = s < nd ? S.len[nl + s] : 0;
        uint32_t before = 0;
#pragma unroll 2
        for (int t = 0; t < nd; ++t) before += (t < s && S.len[nl + t] == l) ? 1u : 0u;
        if (l) S.dsym[S.loff[16 + l] + before] = (uint16_t)s;
    }
    // the tables: entry e describes the codeword that starts the bit string e (first bit read = bit 0 of e = the code's MSB).
    // Lane-parallel over the entries; the length is found by comparison against the uniform (first, count) of every length.
    {
        uint32_t f[GZI_LBITS + 1], c[GZI_LBITS + 1], o[GZI_LBITS + 1];
#pragma unroll
        for (int l = 1; l <= GZI_LBITS; ++l) { f[l] = gzi_rl(first, l); c[l] = gzi_rl(cnt, l); o[l] = gzi_rl(offs, l); }
#pragma unroll 1
        for (int e0 = 0; e0 < (1 << GZI_LBITS); e0 += 64) {
            const uint32_t e = (uint32_t)(e0 + lane), rev = __brev(e) >> (32 - GZI_LBITS);
            uint32_t ent = 0;
#pragma unroll
            for (int l = 1; l <= GZI_LBITS; ++l) {
                const uint32_t cd = rev >> (GZI_LBITS - l);
                if (cd - f[l] < c[l]) ent = (uint32_t)S.lsym[o[l] + cd - f[l]] | ((uint32_t)l << 9);
            }
            S.llut[e] = (uint16_t)ent;
        }
    }
    {
        uint32_t f[GZI_DBITS + 1], c[GZI_DBITS + 1], o[GZI_DBITS + 1];
#pragma unroll
        for (int l = 1; l <= GZI_DBITS; ++l) { f[l] = gzi_rl(first, 16 + l); c[l] = gzi_rl(cnt, 16 + l); o[l] = gzi_rl(offs, 16 + l); }
#pragma unroll 1
        for (int e0 = 0; e0 < (1 << GZI_DBITS); e0 += 64) {
            const uint32_t e = (uint32_t)(e0 + lane), rev = __brev(e) >> (32 - GZI_DBITS);
            uint32_t ent = 0;
#pragma unroll
            for (int l = 1; l <= GZI_DBITS; ++l) {
                const uint32_t cd = rev >> (GZI_DBITS - l);
                if (cd - f[l] < c[l]) ent = (uint32_t)S.dsym[o[l] + cd - f[l]] | ((uint32_t)l << 9);
            }
            S.dlut[e] = (uint16_t)ent;
        }
    }
    return true;
}

// the symbol whose codeword starts the bit string v (LSB first), by the lanes [base + 1, base + 15]; -1: no codeword matches.
// nbits receives its length.
__device__ __forceinline__ int gzi_decode(uint32_t v, const GziCode &c, int lane, int base, const uint16_t *syms, int &nbits) {
    const int L = lane - base;
    const bool mine = L >= 1 && L <= 15;
    const uint32_t code = mine ? (__brev(v) >> (32 - L)) : 0u;
    const bool hit = mine && code - c.first < c.count;      // (unsigned: code >= first && code < first + count)
    const uint64_t m = __ballot(hit);
    if (!m) return -1;
    const int f = __builtin_ctzll(m);
    nbits = f - base;
    const uint32_t idx = gzi_rl(c.offs + code - c.first, f);
    return __builtin_amdgcn_readfirstlane((int)syms[idx]);
}

#ifdef RD_DIAG
// diagnostic build only: cycles per stage of a wave's member loop (tools/gz_bench.py --stages): g_gz_prof[16 ...]
#define GZI_T(k)                                       \
    do {                                               \
        const unsigned long long n_ = clock64();       \
        acc_[k] += n_ - t_;                            \
        t_ = n_;                                       \
    } while (0)
#else
#define GZI_T(k) do { } while (0)
#endif

__global__ __launch_bounds__(64 * GZI_WAVES) __attribute__((amdgpu_waves_per_eu(6, 8))) void rd_gz_inflate_kernel(const uint8_t *__restrict__ comp, int64_t comp_bytes, const GzMemberIn *__restrict__ mem,
                                                                       int64_t nmem, uint8_t *__restrict__ text, int64_t text_bytes,
                                                                       uint32_t *__restrict__ status) {
    __shared__ GziSmem SM;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    GziWave &S = SM.w[wave];
    {
        uint32_t c = threadIdx.x;
        for (int b = 0; b < 8; ++b) c = (c & 1) ? (c >> 1) ^ 0xedb88320u : c >> 1;
        SM.crc_tab[0][threadIdx.x] = c;
        __syncthreads();
        for (int k = 1; k < 4; ++k) {
            c = (c >> 8) ^ SM.crc_tab[0][c & 0xffu];
            SM.crc_tab[k][threadIdx.x] = c;
        }
    }
    __syncthreads();      // (the last barrier: from here on every wave is on its own)
    for (int64_t m = (int64_t)blockIdx.x * GZI_WAVES + wave; m < nmem; m += (int64_t)gridDim.x * GZI_WAVES) {
        const GzMemberIn me = mem[m];
#ifdef RD_DIAG
        unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_ = clock64();
#endif
        const int in_len = me.in_len, out_len = me.out_len;
        if (me.in_off < 0 || in_len < 0 || in_len > (1 << 28) || me.in_off + in_len + 8 > comp_bytes || me.out_off < 0 || out_len < 0 || me.out_off + out_len > text_bytes) {
            if (lane == 0) status[m] = GZI_MEMBER;
            continue;
        }
        uint8_t *out = text + me.out_off;
        // the stream as dwords from the 4-byte boundary at or before its first byte; bit positions count from there
        const int64_t a0 = me.in_off & ~(int64_t)3;
        const uint8_t *inb = comp + a0;
        const int64_t limit = comp_bytes - a0;                          // readable bytes from inb
        const uint32_t end_bits = (uint32_t)((me.in_off - a0 + in_len) * 8);
        auto load_dw = [&](int k) -> uint32_t {                         // dword k of the stream; zero when it is not wholly inside the buffer
            const int64_t b = (int64_t)k * 4;                           // (the 8-byte trailer follows the data: such a dword holds no data bit)
            return b + 4 <= limit ? *reinterpret_cast<const uint32_t *>(inb + b) : 0u;
        };
        // cur = dwords [win, win + 64), nxt = dwords [win + GZI_STEP, win + GZI_STEP + 64): the registers overlap, so that everything a
        // round touches (up to 7 dwords behind the position's own) is in cur and one v_readlane away; nxt is requested when cur
        // is replaced and not looked at until it replaces cur in turn (no load on the symbol path)
        int win = 0;
        uint32_t cur = load_dw(lane), nxt = load_dw(GZI_STEP + lane);
        uint32_t p = (uint32_t)(me.in_off - a0) * 8;                    // the bit position
        int op = 0;
        int err = GZI_OK;
        auto ensure = [&](uint32_t q) {                                 // dword (q >> 5) among the first GZI_STEP of cur
            while ((int)(q >> 5) - win >= GZI_STEP) {
                if ((int)(q >> 5) - win >= 2 * GZI_STEP) {              // (behind a stored block: far ahead)
                    win = (int)(q >> 5);
                    cur = load_dw(win + lane);
                } else {
                    win += GZI_STEP;
                    cur = nxt;
                }
                nxt = load_dw(win + GZI_STEP + lane);
            }
        };
        auto peek32 = [&](uint32_t q) -> uint32_t {                     // the 32 bits at position q
            const int rel = (int)(q >> 5) - win;
            const uint32_t d0 = gzi_rl(cur, rel), d1 = gzi_rl(cur, rel + 1);
            return (uint32_t)(((((uint64_t)d1) << 32) | d0) >> (q & 31));
        };
        auto peek64 = [&](uint32_t q) -> uint64_t {
            const int rel = (int)(q >> 5) - win;
            const uint32_t d0 = gzi_rl(cur, rel), d1 = gzi_rl(cur, rel + 1), d2 = gzi_rl(cur, rel + 2);
            const uint32_t s = q & 31;
            const uint32_t lo = (uint32_t)((((uint64_t)d1 << 32) | d0) >> s), hi = (uint32_t)((((uint64_t)d2 << 32) | d1) >> s);
            return ((uint64_t)hi << 32) | lo;
        };
        bool last = false;
        while (!last && err == GZI_OK) {
            ensure(p);
            const uint64_t H = peek64(p);                               // the block header: 3 bits, and 14 more of a dynamic block
            last = (H & 1) != 0;
            const uint32_t type = (uint32_t)(H >> 1) & 3u;
            p += 3;
            if (type == 0) {                                            // stored
                p = (p + 7u) & ~7u;
                ensure(p);
                const uint32_t w = peek32(p);
                const uint32_t ln = w & 0xffffu, nl = w >> 16;
                p += 32;
                if ((ln ^ nl) != 0xffffu) { err = GZI_STORED; break; }
                const int64_t ib = (int64_t)(p >> 3);
                if (p + ln * 8 > end_bits || op + (int)ln > out_len) { err = GZI_OVERRUN; break; }
                for (int k = lane; k < (int)ln; k += 64) out[op + k] = inb[ib + k];
                p += ln * 8; op += (int)ln;
                continue;
            }
            if (type == 3) { err = GZI_BAD_BLOCK; break; }
            int nl, nd;
            if (type == 1) {                                            // fixed codes
                nl = 288; nd = 30;
                for (int s = lane; s < 288; s += 64) S.len[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                if (lane < 30) S.len[288 + lane] = 5;
            } else {                                                    // dynamic codes: the code-length code first
                nl = (int)((uint32_t)(H >> 3) & 31u) + 257; nd = (int)((uint32_t)(H >> 8) & 31u) + 1;
                const int nc = (int)((uint32_t)(H >> 13) & 15u) + 4;
                p += 14;
                if (nl > 286 || nd > 30) { err = GZI_BAD_LENGTHS; break; }
                for (int s = lane; s < 320; s += 64) S.len[s] = 0;
                // (19 code-length symbols of 3 bits; their lengths go to S.len[300 + symbol] for the build below)
                ensure(p);
                const uint64_t C = peek64(p);
                if (lane < nc) S.len[300 + GZ_CLORD[lane]] = (uint8_t)((uint32_t)(C >> (3 * lane)) & 7u);
                p += 3u * (uint32_t)nc;
                GziCode cl;
                {   // the code-length code as a "literal" code of 19 symbols at S.len[300..318]: build its per-lane triple by hand
                    const int L = lane & 15;
                    uint32_t cnt = 0;
                    if (L >= 1 && lane < 16)
#pragma unroll 1
                        for (int s = 0; s < 19; ++s) cnt += S.len[300 + s] == L ? 1u : 0u;
                    uint32_t first = 0, offs = 0, code = 0, off = 0;
                    int left = 1;
                    bool over = false;
                    for (int l = 1; l <= 7; ++l) {
                        const uint32_t c = gzi_rl(cnt, l);
                        if (L == l) { first = code; offs = off; }
                        code = (code + c) << 1;
                        off += c;
                        left = (left << 1) - (int)c;
                        over = over || left < 0;
                    }
                    if (__ballot(lane == 1 && over)) { err = GZI_BAD_LENGTHS; break; }
                    cl = GziCode{first, lane < 16 ? cnt : 0u, offs};
                    if (lane < 19) {
                        const int l = S.len[300 + lane];
                        if (l) {
                            uint32_t before = 0, o = 0;
#pragma unroll 1
                            for (int t = 0; t < 19; ++t) {
                                const int lt = S.len[300 + t];
                                before += (t < lane && lt == l) ? 1u : 0u;
                                o += (lt != 0 && lt < l) ? 1u : 0u;
                            }
                            S.dsym[o + before] = (uint16_t)lane;        // (dsym doubles as the code-length code's ordering until the real build)
                        }
                    }
                }
                // the nl + nd code lengths, run-length coded
                int i = 0, prev = 0;
                while (i < nl + nd) {
                    ensure(p);
                    int nb = 0;
                    const uint32_t v = peek32(p);                        // the code (<= 7 bits) and its extra bits (<= 7)
                    const int sym = gzi_decode(v, cl, lane, 0, S.dsym, nb);
                    if (sym < 0) { err = GZI_BAD_CODE; break; }
                    const uint32_t x = v >> nb;
                    int rep = 1, val = sym;
                    if (sym == 16) { if (i == 0) { err = GZI_BAD_LENGTHS; break; } rep = 3 + (int)(x & 3u); val = prev; nb += 2; }
                    else if (sym == 17) { rep = 3 + (int)(x & 7u); val = 0; nb += 3; }
                    else if (sym == 18) { rep = 11 + (int)(x & 127u); val = 0; nb += 7; }
                    p += (uint32_t)nb;
                    if (i + rep > nl + nd) { err = GZI_BAD_LENGTHS; break; }
                    if (lane < rep) S.len[i + lane] = (uint8_t)val;      // (rep <= 138: up to three rounds)
                    if (lane + 64 < rep) S.len[i + lane + 64] = (uint8_t)val;
                    if (lane + 128 < rep) S.len[i + lane + 128] = (uint8_t)val;
                    i += rep;
                    prev = val;
                }
                if (err != GZI_OK) break;
                if (__builtin_amdgcn_readfirstlane((int)S.len[256]) == 0) { err = GZI_BAD_LENGTHS; break; }   // no end-of-block code
            }
            GziCode lit, dst;
            GZI_T(0);   // block header, code lengths
            if (!gzi_build(S, nl, nd, lane, lit, dst)) { err = GZI_BAD_LENGTHS; break; }
            GZI_T(1);   // tables
            // ---- the block's symbols: 64 bit positions per round --------------------------------------------------------------------
            bool eob = false;
            while (!eob && err == GZI_OK) {
                if (p > end_bits + 64u) { err = GZI_TRUNCATED; break; }
                ensure(p);
                // lane i: the 64 bits that start at p + i (aligned dwords A0 .. A3 = bits p .. p + 127), the table entry of the codeword
                // that would start there and - should it be a length code - the whole match it would begin
                uint32_t ent, adv, mres;
                {
                    const int rel = (int)(p >> 5) - win;
                    const uint32_t d0 = gzi_rl(cur, rel), d1 = gzi_rl(cur, rel + 1), d2 = gzi_rl(cur, rel + 2), d3 = gzi_rl(cur, rel + 3),
                                   d4 = gzi_rl(cur, rel + 4);
                    const uint32_t s = p & 31;
                    const uint32_t A0 = (uint32_t)((((uint64_t)d1 << 32) | d0) >> s), A1 = (uint32_t)((((uint64_t)d2 << 32) | d1) >> s),
                                   A2 = (uint32_t)((((uint64_t)d3 << 32) | d2) >> s), A3 = (uint32_t)((((uint64_t)d4 << 32) | d3) >> s);
                    const bool lowh = lane < 32;
                    const uint32_t x0 = lowh ? A0 : A1, x1 = lowh ? A1 : A2, x2 = lowh ? A2 : A3;
                    const uint32_t sh = (uint32_t)(lane & 31);
                    const uint32_t lo = __builtin_amdgcn_alignbit(x1, x0, sh), hi = __builtin_amdgcn_alignbit(x2, x1, sh);
                    ent = S.llut[lo & ((1u << GZI_LBITS) - 1u)];
                    const uint32_t sym = ent & 511u, cl = ent >> 9;
                    // a short literal advances the walk by its code length, unless it would leave the 64 positions (the next round
                    // starts with it); everything else stops the walk
                    adv = ((ent & 0x100u) == 0 && lane + (int)cl <= 63) ? cl : 0u;
                    // the match a length code here would begin: length, distance, bits - each lane for itself (s <= 24 < 32 bits in)
                    const uint32_t ls = sym - 257u, l5 = ls & 31u;       // (l5, d5: shift counts stay in range where the lane holds no match)
                    const uint32_t le = (l5 < 8u || l5 >= 28u) ? 0u : (l5 >> 2) - 1u;
                    const uint32_t lb = l5 < 8u ? 3u + l5 : l5 >= 28u ? 258u : ((4u + (l5 & 3u)) << le) + 3u;
                    const uint32_t w1 = __builtin_amdgcn_alignbit(hi, lo, cl);
                    const uint32_t len = lb + (w1 & ((1u << le) - 1u));
                    const uint32_t w2 = __builtin_amdgcn_alignbit(hi, lo, cl + le);
                    const uint32_t de = S.dlut[w2 & ((1u << GZI_DBITS) - 1u)];
                    const uint32_t ds = de & 511u, dl = de >> 9;
                    const uint32_t d5 = ds & 31u;
                    const uint32_t dx = d5 < 4u ? 0u : (d5 >> 1) - 1u;
                    const uint32_t db = d5 < 4u ? 1u + d5 : ((2u + (d5 & 1u)) << dx) + 1u;
                    const uint32_t w3 = __builtin_amdgcn_alignbit(hi, lo, cl + le + dl);
                    const uint32_t dist = db + (w3 & ((1u << dx) - 1u));
                    const bool okm = ls < 29u && cl != 0 && dl != 0 && ds < 30u;
                    mres = okm ? (len | (dist << 9) | ((cl + le + dl + dx) << 25)) : 0u;     // 9 + 16 + 6 bits
                }
                GZI_T(2);   // round: the 64 positions' table entries
                uint32_t pos = 0;                                        // bits behind p
                for (;;) {
                    // the chain of literal starts from pos: four hops per test (a stop is sticky: its advance is 0)
                    uint64_t M = 0;
                    uint32_t a;
                    do {
#pragma unroll
                        for (int h = 0; h < 4; ++h) {
                            a = gzi_rl(adv, (int)pos);
                            asm("s_bitset1_b64 %0, %1" : "+s"(M) : "s"(pos));
                            pos += a;
                        }
                    } while (a != 0);
                    asm("s_bitset0_b64 %0, %1" : "+s"(M) : "s"(pos));  // (where it stopped is not a start of the chain)
                    if (M) {
                        const int n = __builtin_popcountll(M);
                        if (op + n > out_len) { err = GZI_OVERRUN; break; }
                        const int r = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0u));
                        if ((M >> lane) & 1) out[op + r] = (uint8_t)ent;
                        op += n;
                    }
                    GZI_T(3);   // walk + literal stores
                    // ---- what stopped the walk at p + pos ---------------------------------------------------------------------------
                    const uint32_t e = gzi_rl(ent, (int)pos);
                    int sym = (int)(e & 511u), cl = (int)(e >> 9);
                    if (cl != 0 && sym < 256) break;                     // a literal that reaches past the 64 positions: next round
                    int len, dist;
                    const uint32_t mr = gzi_rl(mres, (int)pos);
                    if (mr != 0) {
                        len = (int)(mr & 511u); dist = (int)((mr >> 9) & 0xffffu);
                        pos += mr >> 25;
                    } else {
                        if (cl == 0) {                                   // no codeword of up to 10 bits: by comparison
                            sym = gzi_decode(peek32(p + pos), lit, lane, 0, S.lsym, cl);
                            if (sym < 0) { err = GZI_BAD_CODE; break; }
                        }
                        if (sym < 256) {                                 // (a literal with a long code)
                            if (op >= out_len) { err = GZI_OVERRUN; break; }
                            if (lane == 0) out[op] = (uint8_t)sym;
                            ++op;
                            pos += (uint32_t)cl;
                            if (pos > 53u) break;
                            continue;
                        }
                        if (sym == 256) { pos += (uint32_t)cl; eob = true; break; }
                        if (sym > 285) { err = GZI_BAD_CODE; break; }
                        uint64_t B = peek64(p + pos + (uint32_t)cl);     // extra bits, distance code, extra bits: at most 5 + 15 + 13
                        const int ls = sym - 257;
                        const int le = ls < 8 || ls == 28 ? 0 : (ls >> 2) - 1;
                        len = (ls < 8 ? 3 + ls : ls == 28 ? 258 : ((4 + (ls & 3)) << le) + 3) + (int)((uint32_t)B & ((1u << le) - 1u));
                        B >>= le;
                        int dl = 0;
                        const int ds = gzi_decode((uint32_t)B, dst, lane, 16, S.dsym, dl);
                        if (ds < 0 || ds > 29) { err = GZI_BAD_CODE; break; }
                        B >>= dl;
                        const int dx = ds < 4 ? 0 : (ds >> 1) - 1;
                        dist = (ds < 4 ? 1 + ds : ((2 + (ds & 1)) << dx) + 1) + (int)((uint32_t)B & ((1u << dx) - 1u));
                        pos += (uint32_t)(cl + le + dl + dx);
                    }
                    if (dist > op) { err = GZI_BAD_DISTANCE; break; }
                    if (op + len > out_len) { err = GZI_OVERRUN; break; }
                    // the copy: the wave's earlier stores reach the L1 before these loads (program order within a wave)
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const uint8_t *src = out + op - dist;
                    if (dist >= len) {
                        for (int k = lane; k < len; k += 64) out[op + k] = src[k];
                    } else if (dist == 1) {
                        const uint8_t b = src[0];
                        for (int k = lane; k < len; k += 64) out[op + k] = b;
                    } else {
                        for (int k = lane; k < len; k += 64) out[op + k] = src[k % dist];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    op += len;
                    GZI_T(4);   // match
                    if (pos > 53u) break;                                // (what is left of the 64 positions is not worth a walk)
                }
                p += pos;
            }
        }
        GZI_T(5);   // (what the stamps above left out)
        if (err == GZI_OK && p > end_bits) err = GZI_TRUNCATED;
        if (err == GZI_OK && op != out_len) err = GZI_SIZE;
        if (err == GZI_OK) {   // CRC-32 of the member (trailer: CRC-32, ISIZE little-endian right behind the DEFLATE data)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const int per = (((out_len + 63) >> 6) + 3) & ~3;            // bytes per lane: a multiple of 4 (dword loads)
            const int b0 = lane * per < out_len ? lane * per : out_len, b1 = b0 + per < out_len ? b0 + per : out_len;
            uint32_t c = 0xffffffffu;
            int b = b0;
            // bytes up to a 16-byte boundary, then 16 bytes per load and four table reads per dword that do not wait for each other
            for (; b < b1 && ((reinterpret_cast<uintptr_t>(out) + (uintptr_t)b) & 15) != 0; ++b) c = SM.crc_tab[0][(c ^ out[b]) & 0xffu] ^ (c >> 8);
            for (; b + 16 <= b1; b += 16) {
                const u32x4 v = *reinterpret_cast<const u32x4 *>(out + b);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t x = c ^ v[j];
                    c = SM.crc_tab[3][x & 0xffu] ^ SM.crc_tab[2][(x >> 8) & 0xffu] ^ SM.crc_tab[1][(x >> 16) & 0xffu] ^ SM.crc_tab[0][x >> 24];
                }
            }
            for (; b < b1; ++b) c = SM.crc_tab[0][(c ^ out[b]) & 0xffu] ^ (c >> 8);
            c = ~c;
            if (b1 == b0) c = 0;
            c = gz_multmodp(gz_x8n((uint32_t)(out_len - b1)), c);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c ^= (uint32_t)__shfl_xor((int)c, o);
            const uint8_t *tr = comp + me.in_off + in_len;
            uint32_t want = 0;
            for (int q = 0; q < 4; ++q) want |= (uint32_t)tr[q] << (8 * q);
            if (c != want) err = GZI_CRC;
        }
        GZI_T(6);   // CRC
#ifdef RD_DIAG
        if (g_gz_prof && lane == 0)
            for (int k = 0; k < 7; ++k) atomicAdd(&g_gz_prof[16 + k], acc_[k]);
#endif
        if (lane == 0) status[m] = (uint32_t)err;
    }
}

}  // namespace
