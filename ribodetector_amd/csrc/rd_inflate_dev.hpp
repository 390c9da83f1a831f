// rd_inflate_dev.hpp - gzip members inflated on the device (rd_gz_inflate_kernel): the input side of csrc/rd_deflate.hpp
// Part of the single translation unit rd_kernels.hip (included from there, in order); DESIGN.md §3.11 has the numbers.
//
// What it is for: a .gz whose members say how long they are - BGZF (bgzip, htslib, and every .gz this build's CLI writes: one member
// per 65,280 bytes, 'B','C' subfield) - is a list of independent DEFLATE streams. The reference reads such a file through Python's
// gzip module like any other (reference data_loader/seq_encoder.py:21-39, fastx_parser.py:15-55); round 2 gave the host a parallel
// member decoder (csrc/rd_pgzip.h). Here the members are decoded where the reads are classified: the compressed bytes travel to HBM
// (a fifth of the text), ONE WAVE decodes one member, thousands of members at a time.
//
// How a wave decodes a serial format: the control flow is wave-uniform (bit buffer, positions and the current symbol are the same in
// every lane), and the lanes are used where a CPU decoder uses tables and loops:
//   * a Huffman symbol is decoded by COMPARISON, not by table: lane L (1..15) holds first[L] / count[L] / offset[L] of the canonical
//     code's length-L codewords, reverses the next L bits of the stream and tests first[L] <= code < first[L] + count[L]; exactly one
//     lane says yes (prefix code), a ballot finds it, and the symbol is one LDS read away. Lanes 17..31 do the same for the distance
//     code. No 2^15-entry table per member (LDS would allow two members per CU), nothing to build but three 16-entry arrays;
//   * literals collect in a 64-byte register window (lane = position mod 64) and are stored 64 at a time;
//   * a match is copied by all lanes at once (out[p + k] = out[p - dist + k mod dist]: the same formula for overlapping runs);
//   * CRC-32 of the member by the 64 lanes (1 KiB per lane per round, table in LDS, pieces combined with x^(8 n) mod P as in the
//     deflate kernel) and checked against the member's trailer, like ISIZE: a damaged member is reported, never silently accepted.
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
       GZI_CRC = 8, GZI_STORED = 9 };

struct __attribute__((aligned(16))) GziSmem {
    uint16_t lsym[288];     // literal/length symbols ordered by (code length, symbol)
    uint16_t dsym[32];      // distance symbols likewise
    uint8_t len[320];       // code lengths being read
    uint16_t loff[32];      // first position of every code length in the orderings (literal/length: [1..15], distance: [17..31])
    uint32_t crc_tab[256];
};

// per-lane view of a canonical Huffman code: lane L in [base + 1, base + 15] describes the codewords of length L - base
struct GziCode { uint32_t first, count, offs; };

// Build the two orderings and the per-lane triples from S.len[0 .. nl) (literal/length) and S.len[nl .. nl + nd) (distance).
// Returns false when a length set is over-subscribed (or incomplete in a way zlib's inflate also rejects).
__device__ __forceinline__ bool gzi_build(GziSmem &S, int nl, int nd, int lane, GziCode &lit, GziCode &dst) {
    // counts per length: lane L counts the symbols whose length is L (lit: lanes 1..15; dist: lanes 17..31)
    const bool isd = lane >= 16;
    const int L = lane & 15;
    const int n0 = isd ? nl : 0, n1 = isd ? nl + nd : nl;
    uint32_t cnt = 0;
    if (L >= 1 && lane < 32)
        for (int s = n0; s < n1; ++s) cnt += S.len[s] == L ? 1u : 0u;
    // first code and offset of every length: a prefix scan over the 15 lengths, done by each lane for itself (15 steps, uniform)
    uint32_t first = 0, offs = 0, code = 0, off = 0;
    int left = 1;
    bool over = false;
    for (int l = 1; l <= 15; ++l) {
        const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane((int)cnt, l), cd = (uint32_t)__builtin_amdgcn_readlane((int)cnt, 16 + l);
        const uint32_t c = isd ? cd : cl;
        if (L == l) { first = code; offs = off; }
        code = (code + c) << 1;
        off += c;
        left = (left << 1) - (int)c;
        over = over || left < 0;
    }
    // over-subscribed: invalid. Incomplete: only allowed for a code with a single codeword (zlib: distance code of one symbol);
    // an all-zero distance code is legal for a block without matches
    const uint32_t bad = __ballot((lane == 1 || lane == 17) && over);
    if (bad) return false;
    lit = GziCode{first, cnt, offs};
    dst = lit;   // (same registers: a lane is either a literal-code lane or a distance-code lane)
    if (lane < 32) S.loff[lane] = (uint16_t)offs;
    // the orderings: symbol s goes to loff[its length] + (number of earlier symbols of the same length) - lane-parallel over the symbols
    for (int s0 = 0; s0 < nl; s0 += 64) {
        const int s = s0 + lane;
        const int l = s < nl ? S.len[s] : 0;
        if (l) {
            uint32_t before = 0;
            for (int t = 0; t < s; ++t) before += S.len[t] == l ? 1u : 0u;
            S.lsym[S.loff[l] + before] = (uint16_t)s;
        }
    }
    {
        const int s = lane;
        const int l = s < nd ? S.len[nl + s] : 0;
        if (l) {
            uint32_t before = 0;
            for (int t = 0; t < s; ++t) before += S.len[nl + t] == l ? 1u : 0u;
            S.dsym[S.loff[16 + l] + before] = (uint16_t)s;
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
    const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)(c.offs + code - c.first), f);
    return __builtin_amdgcn_readfirstlane((int)syms[idx]);
}

__global__ __launch_bounds__(64) void rd_gz_inflate_kernel(const uint8_t *__restrict__ comp, const GzMemberIn *__restrict__ mem, int64_t nmem,
                                                          uint8_t *__restrict__ text, uint32_t *__restrict__ status) {
    __shared__ GziSmem S;
    const int lane = threadIdx.x;
    for (int k = lane; k < 256; k += 64) {
        uint32_t c = (uint32_t)k;
        for (int b = 0; b < 8; ++b) c = (c & 1) ? (c >> 1) ^ 0xedb88320u : c >> 1;
        S.crc_tab[k] = c;
    }
    for (int64_t m = blockIdx.x; m < nmem; m += gridDim.x) {
        const GzMemberIn me = mem[m];
        const uint8_t *in = comp + me.in_off;
        uint8_t *out = text + me.out_off;
        const int in_len = me.in_len, out_len = me.out_len;
        uint64_t bitbuf = 0;
        int bitcnt = 0, ip = 0, op = 0;      // bits in the buffer, next input byte, next output byte
        int flushed = 0;                     // bytes of out[] that are in memory; [flushed, op) wait in `pend` (lane = position & 63)
        uint32_t pend = 0;
        int err = GZI_OK;
        auto refill = [&]() {                // at least 32 bits in the buffer (zeros past the member's end: the decoder notices by position)
            if (bitcnt < 32) {
                uint32_t w = 0;
                if (ip + 4 <= in_len) __builtin_memcpy(&w, in + ip, 4);
                else for (int b = 0; b < 4; ++b) w |= (ip + b < in_len ? (uint32_t)in[ip + b] : 0u) << (8 * b);
                w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
                bitbuf |= (uint64_t)w << bitcnt;
                bitcnt += 32;
                ip += 4;
            }
        };
        auto take = [&](int n) -> uint32_t {   // n <= 16 bits
            refill();
            const uint32_t v = (uint32_t)bitbuf & ((1u << n) - 1u);
            bitbuf >>= n;
            bitcnt -= n;
            return v;
        };
        auto flush = [&]() {                 // the waiting literals go to memory (each lane its own byte)
            const int pos = flushed + ((lane - flushed) & 63);   // the position in [flushed, flushed + 64) this lane holds
            if (pos < op) out[pos] = (uint8_t)pend;
            flushed = op;
        };
        bool last = false;
        while (!last && err == GZI_OK) {
            last = take(1) != 0;
            const uint32_t type = take(2);
            if (type == 0) {                                            // stored
                bitbuf >>= bitcnt & 7;                                  // to the byte boundary
                bitcnt -= bitcnt & 7;
                refill();
                const uint32_t ln = take(16), nl = take(16);
                if ((ln ^ nl) != 0xffffu) { err = GZI_STORED; break; }
                // bytes still in the bit buffer belong to the stored data: step the input position back
                ip -= bitcnt >> 3;
                bitbuf = 0; bitcnt = 0;
                if (ip + (int)ln > in_len || op + (int)ln > out_len) { err = GZI_OVERRUN; break; }
                flush();
                for (int k = lane; k < (int)ln; k += 64) out[op + k] = in[ip + k];
                ip += (int)ln; op += (int)ln; flushed = op;
                continue;
            }
            if (type == 3) { err = GZI_BAD_BLOCK; break; }
            int nl, nd;
            if (type == 1) {                                            // fixed codes
                nl = 288; nd = 30;
                for (int s = lane; s < 288; s += 64) S.len[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                if (lane < 30) S.len[288 + lane] = 5;
            } else {                                                    // dynamic codes: the code-length code first
                nl = (int)take(5) + 257; nd = (int)take(5) + 1;
                const int nc = (int)take(4) + 4;
                if (nl > 286 || nd > 30) { err = GZI_BAD_LENGTHS; break; }
                for (int s = lane; s < 320; s += 64) S.len[s] = 0;
                // (19 code-length symbols; their lengths go to S.len[300 + symbol] for the build below)
                for (int i = 0; i < nc; ++i) {
                    const uint32_t l3 = take(3);
                    if (lane == 0) S.len[300 + GZ_CLORD[i]] = (uint8_t)l3;
                }
                GziCode cl, unused;
                {   // the code-length code as a "literal" code of 19 symbols at S.len[300..318]: build its per-lane triple by hand
                    const int L = lane & 15;
                    uint32_t cnt = 0;
                    if (L >= 1 && lane < 16)
                        for (int s = 0; s < 19; ++s) cnt += S.len[300 + s] == L ? 1u : 0u;
                    uint32_t first = 0, offs = 0, code = 0, off = 0;
                    int left = 1;
                    bool over = false;
                    for (int l = 1; l <= 7; ++l) {
                        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cnt, l);
                        if (L == l) { first = code; offs = off; }
                        code = (code + c) << 1;
                        off += c;
                        left = (left << 1) - (int)c;
                        over = over || left < 0;
                    }
                    if (__ballot(lane == 1 && over)) { err = GZI_BAD_LENGTHS; break; }
                    cl = GziCode{first, lane < 16 ? cnt : 0u, offs};
                    (void)unused;
                    if (lane < 19) {
                        const int l = S.len[300 + lane];
                        if (l) {
                            uint32_t before = 0, o = 0;
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
                    refill();
                    int nb = 0;
                    const int sym = gzi_decode((uint32_t)bitbuf, cl, lane, 0, S.dsym, nb);
                    if (sym < 0) { err = GZI_BAD_CODE; break; }
                    bitbuf >>= nb; bitcnt -= nb;
                    int rep = 1, val = sym;
                    if (sym == 16) { if (i == 0) { err = GZI_BAD_LENGTHS; break; } rep = 3 + (int)take(2); val = prev; }
                    else if (sym == 17) { rep = 3 + (int)take(3); val = 0; }
                    else if (sym == 18) { rep = 11 + (int)take(7); val = 0; }
                    if (i + rep > nl + nd) { err = GZI_BAD_LENGTHS; break; }
                    if (lane < rep) S.len[i + lane] = (uint8_t)val;      // (rep <= 138: up to three rounds)
                    if (lane + 64 < rep) S.len[i + lane + 64] = (uint8_t)val;
                    if (lane + 128 < rep) S.len[i + lane + 128] = (uint8_t)val;
                    i += rep;
                    prev = val;
                }
                if (err != GZI_OK) break;
                if (S.len[256] == 0) { err = GZI_BAD_LENGTHS; break; }   // no end-of-block code
            }
            GziCode lit, dst;
            if (!gzi_build(S, nl, nd, lane, lit, dst)) { err = GZI_BAD_LENGTHS; break; }
            // ---- the block's symbols -------------------------------------------------------------------------------------------
            for (;;) {
                refill();
                int nb = 0;
                const int sym = gzi_decode((uint32_t)bitbuf, lit, lane, 0, S.lsym, nb);
                if (sym < 0) { err = GZI_BAD_CODE; break; }
                bitbuf >>= nb; bitcnt -= nb;
                if (sym < 256) {                                         // literal: into the register window
                    if (op >= out_len) { err = GZI_OVERRUN; break; }
                    if (lane == (op & 63)) pend = (uint32_t)sym;
                    ++op;
                    if (op - flushed == 64) flush();
                    continue;
                }
                if (sym == 256) break;                                   // end of block
                if (sym > 285) { err = GZI_BAD_CODE; break; }
                const int ls = sym - 257;
                const int len = GZ_LBASE[ls] + (int)take(GZ_LEXTRA[ls]);
                refill();
                const int ds = gzi_decode((uint32_t)bitbuf, dst, lane, 16, S.dsym, nb);
                if (ds < 0 || ds > 29) { err = GZI_BAD_CODE; break; }
                bitbuf >>= nb; bitcnt -= nb;
                const int dist = GZ_DBASE[ds] + (int)take(GZ_DEXTRA[ds]);
                if (dist > op) { err = GZI_BAD_DISTANCE; break; }
                if (op + len > out_len) { err = GZI_OVERRUN; break; }
                flush();
                __threadfence_block();                                   // the bytes just stored are the copy's source
                const uint8_t *src = out + op - dist;
                for (int k = lane; k < len; k += 64) out[op + k] = src[dist >= len ? k : k % dist];
                op += len; flushed = op;
                __threadfence_block();
            }
        }
        flush();
        if (err == GZI_OK && ip - (bitcnt >> 3) > in_len) err = GZI_TRUNCATED;
        if (err == GZI_OK && op != out_len) err = GZI_SIZE;
        if (err == GZI_OK) {   // CRC-32 of the member (trailer: CRC-32, ISIZE little-endian right behind the DEFLATE data)
            __threadfence_block();
            const int per = (((out_len + 63) >> 6) + 3) & ~3;            // bytes per lane: a multiple of 4 (dword loads)
            const int b0 = lane * per < out_len ? lane * per : out_len, b1 = b0 + per < out_len ? b0 + per : out_len;
            uint32_t c = 0xffffffffu;
            int b = b0;
            for (; b + 4 <= b1; b += 4) {
                uint32_t w;
                __builtin_memcpy(&w, out + b, 4);
                c = S.crc_tab[(c ^ w) & 0xffu] ^ (c >> 8);
                c = S.crc_tab[(c ^ (w >> 8)) & 0xffu] ^ (c >> 8);
                c = S.crc_tab[(c ^ (w >> 16)) & 0xffu] ^ (c >> 8);
                c = S.crc_tab[(c ^ (w >> 24)) & 0xffu] ^ (c >> 8);
            }
            for (; b < b1; ++b) c = S.crc_tab[(c ^ out[b]) & 0xffu] ^ (c >> 8);
            c = ~c;
            if (b1 == b0) c = 0;
            c = gz_multmodp(gz_x8n((uint32_t)(out_len - b1)), c);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c ^= (uint32_t)__shfl_xor((int)c, o);
            uint32_t want = 0;
            for (int b = 0; b < 4; ++b) want |= (uint32_t)in[in_len + b] << (8 * b);
            if (c != want) err = GZI_CRC;
        }
        if (lane == 0) status[m] = (uint32_t)err;
        __syncthreads();
    }
}

}  // namespace
