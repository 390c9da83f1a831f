// rd_inflate_stream.hpp - ONE DEFLATE stream (a plain .gz: what sequencers write) inflated on the device (rd_gzs_* kernels)
// Part of the single translation unit rd_kernels.hip (included from there, in order); DESIGN.md §3.14 has the numbers.
//
// A .gz FASTQ is one DEFLATE stream: a block can be decoded only after everything before it, because matches copy from the previous
// 32 KiB of output (reference: gzip.open(path, 'rt'), data_loader/seq_encoder.py:21-39). The host reader of this build decodes it with
// the two-pass scheme of pugz (csrc/rd_pgzip.h; Kerbiriou & Chikhi 2019) on up to 16 cores; here the same scheme runs on the GPU, one
// WAVE per section of the compressed bytes, with the round-based decoder of rd_inflate_dev.hpp (64 bit positions tried per round):
//   rd_gzs_search_kernel   section k looks for the first bit position in [k S, (k + 1) S) at which a dynamic-Huffman block starts:
//                          every lane tests one position per round (BFINAL = 0, BTYPE = 2, HLIT / HDIST in range, a COMPLETE code-length
//                          code - 1 position in ~2,000 passes), the survivors are validated by the whole wave: complete literal/length
//                          and distance codes, the block decoded to its end with every literal a text byte, a plausible header behind it;
//   rd_gzs_decode_kernel   section k is decoded from its start to the start the next section found, with an UNKNOWN window: the
//                          output is 16-bit symbols, a byte or a marker "byte i of the 32 KiB before this section"; a section that
//                          does not end exactly where the next one starts poisons the batch (nothing speculative is accepted);
//   rd_gzs_scan_kernel     one wave: symbols per section -> text offsets; the member's end (BFINAL) closes the list;
//   rd_gzs_symwin_kernel   the window chain on SYMBOLS (round 6): one workgroup per group of 32 sections, all groups at once, each from the
//   rd_gzs_chain_kernel    identity map; then one workgroup applies the group maps in order to the window in front of the batch;
//   rd_gzs_resolve_kernel  all sections in parallel: symbols -> bytes at their offsets in the batch's text (range mode: -> 16-bit symbols
//                          whose markers point into the window in front of the RANGE; rd_gzs_symtext_kernel makes bytes of them later);
//   rd_gzs_crc_kernel / rd_gzs_fold_kernel   CRC-32 per 64 KiB of text, folded into the stream's running CRC (x^(8 n) mod P).
// The stream's state (window, CRC, length, where the next batch starts) lives in HBM and is carried from batch to batch on the stream.
#pragma once
#include "rd_inflate_dev.hpp"

namespace {

constexpr uint32_t GZS_NONE = 0xffffffffu;
constexpr uint32_t GZS_SEARCH = 0xfffffffeu;     // first_start_bit of a RANGE that begins inside the stream: section 0 looks for its start like the others
constexpr int GZS_WIN = 32768;
constexpr uint32_t GZS_MARK = 0x8000u;
#ifndef RD_GZS_WAVES
#define RD_GZS_WAVES 4
#endif
constexpr int GZS_WAVES = RD_GZS_WAVES;     // sections per workgroup of the search / decode kernels (one per wave). More of them per workgroup = the
                                            // same waves on fewer compute units: a CU that holds a section's wave holds no recurrence workgroup
struct __attribute__((aligned(16))) GzsSmem { GziWave w[GZS_WAVES]; };
enum { GZS_OK = 0, GZS_DECODE = 1, GZS_MISMATCH = 2, GZS_OVERFLOW = 3, GZS_NOSTOP = 4, GZS_WINDOW = 5, GZS_TEXTCAP = 6, GZS_NOSTART = 7 };

struct GzsSec {            // what the decode of one section left
    uint32_t n_syms, end_bit, status, final;
};

struct GzsState {          // = rd_gzs_state of the C ABI (64 bytes), device resident, carried from batch to batch
    uint64_t total_len;    // bytes of text of the member so far
    int64_t n_text;        // bytes of text this batch produced
    uint32_t crc;          // CRC-32 of the member so far
    uint32_t status;       // GZS_* of this batch (sticky: a failed batch fails the following ones)
    uint32_t final;        // the member's last block was decoded in this batch
    uint32_t end_bit;      // ... and ended at this bit of the batch buffer (the 8-byte trailer follows at the next byte boundary)
    uint32_t next_start;   // where the next batch's first section starts, as a bit position of THIS batch's buffer
    uint32_t win_valid;    // bytes of the carried window that are text (the member's first 32 KiB have less behind them)
    uint32_t bad_section;  // first section that failed
    uint32_t n_sections;   // sections that produced text
    uint64_t reserved[2];
};
static_assert(sizeof(GzsState) == 64 && sizeof(rd_gzs_state) == 64, "rd_gzs_state layout");

__device__ __forceinline__ bool gzs_is_text(uint32_t c) { return (c >= 32u && c < 127u) || c == '\n' || c == '\r' || c == '\t'; }

// DEFLATE blocks from bit position p0 of the stream at inb (4-byte aligned; `limit` readable bytes, `end_bits` valid bits), by one wave.
//   WRITE: 16-bit symbols to out16[0 .. cap) with an unknown window (markers), until the block boundary `stop_bit` is met exactly
//          (GZS_MISMATCH when a boundary lies behind it), or the final block ends (fin = true);
//   !WRITE: validation of a block-start candidate: ONE block, dynamic, complete codes, every literal a text byte.
// Returns GZS_*; p_end = the bit behind the last block decoded, n_out = symbols.
constexpr int GZS_VALIDATE_SYMS = 3072;     // a candidate that decodes this many text symbols with complete codes IS a block start
template <bool WRITE>
__device__ __forceinline__ int gzs_blocks(GziWave &S, int lane, const uint8_t *__restrict__ inb, int64_t limit, uint32_t end_bits, uint32_t p0, uint32_t stop_bit,
                          uint16_t *__restrict__ out16, int cap, uint32_t &p_end, int &n_out, bool &fin) {
    auto load_dw = [&](int k) -> uint32_t {
        const int64_t b = (int64_t)k * 4;
        return b + 4 <= limit ? *reinterpret_cast<const uint32_t *>(inb + b) : 0u;
    };
    int win = (int)(p0 >> 5);
    uint32_t cur = load_dw(win + lane), nxt = load_dw(win + GZI_STEP + lane);
    uint32_t p = p0;
    int op = 0;
    int err = GZS_OK;
    auto ensure = [&](uint32_t q) {
        while ((int)(q >> 5) - win >= GZI_STEP) {
            if ((int)(q >> 5) - win >= 2 * GZI_STEP) {
                win = (int)(q >> 5);
                cur = load_dw(win + lane);
            } else {
                win += GZI_STEP;
                cur = nxt;
            }
            nxt = load_dw(win + GZI_STEP + lane);
        }
    };
    auto peek32 = [&](uint32_t q) -> uint32_t {
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
    fin = false;
    int blocks = 0;
    while (err == GZS_OK) {
        if (WRITE) {
            if (p == stop_bit) break;                                   // exactly where the next section starts: done
            if (p > stop_bit) { err = GZS_MISMATCH; break; }            // its start was not a block boundary of this stream
        } else if (blocks == 1) {
            break;
        }
        if (p + 3 > end_bits) { err = WRITE ? GZS_NOSTOP : GZS_DECODE; break; }
        ensure(p);
        const uint64_t H = peek64(p);
        last = (H & 1) != 0;
        const uint32_t type = (uint32_t)(H >> 1) & 3u;
        p += 3;
        ++blocks;
        if (type == 0) {                                                // stored
            if (!WRITE) { err = GZS_DECODE; break; }
            p = (p + 7u) & ~7u;
            ensure(p);
            const uint32_t w = peek32(p);
            const uint32_t ln = w & 0xffffu, nl = w >> 16;
            p += 32;
            if ((ln ^ nl) != 0xffffu) { err = GZS_DECODE; break; }
            const int64_t ib = (int64_t)(p >> 3);
            if (p + ln * 8 > end_bits) { err = GZS_NOSTOP; break; }
            if (op + (int)ln > cap) { err = GZS_OVERFLOW; break; }
            for (int k = lane; k < (int)ln; k += 64) out16[op + k] = inb[ib + k];
            p += ln * 8; op += (int)ln;
            if (last) { fin = true; break; }
            continue;
        }
        if (type == 3) { err = GZS_DECODE; break; }
        int nl, nd;
        if (type == 1) {                                                // fixed codes
            if (!WRITE) { err = GZS_DECODE; break; }
            nl = 288; nd = 30;
            for (int s = lane; s < 288; s += 64) S.len[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 30) S.len[288 + lane] = 5;
        } else {
            nl = (int)((uint32_t)(H >> 3) & 31u) + 257; nd = (int)((uint32_t)(H >> 8) & 31u) + 1;
            const int nc = (int)((uint32_t)(H >> 13) & 15u) + 4;
            p += 14;
            if (nl > 286 || nd > 30) { err = GZS_DECODE; break; }
            for (int s = lane; s < 320; s += 64) S.len[s] = 0;
            ensure(p);
            const uint64_t C = peek64(p);
            if (lane < nc) S.len[300 + GZ_CLORD[lane]] = (uint8_t)((uint32_t)(C >> (3 * lane)) & 7u);
            p += 3u * (uint32_t)nc;
            GziCode cl;
            {
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
                if (__ballot(lane == 1 && (over || (!WRITE && left != 0)))) { err = GZS_DECODE; break; }
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
                        S.dsym[o + before] = (uint16_t)lane;
                    }
                }
            }
            int i = 0, prev = 0;
            // what is left of the two codes' space (units of 2^-15): a length set that over-subscribes its code is rejected where it does
            // - a bit position that is NOT a block start (the search tries ~60 per section that pass the filter of the code-length code)
            // gives itself away within a few symbols instead of after 316 of them and the table build's counting
            int left_l = 1 << 15, left_d = 1 << 15;
            while (i < nl + nd) {
                if (p + 14 > end_bits + 64u) { err = GZS_DECODE; break; }
                ensure(p);
                int nb = 0;
                const uint32_t v = peek32(p);
                const int sym = gzi_decode(v, cl, lane, 0, S.dsym, nb);
                if (sym < 0) { err = GZS_DECODE; break; }
                const uint32_t x = v >> nb;
                int rep = 1, val = sym;
                if (sym == 16) { if (i == 0) { err = GZS_DECODE; break; } rep = 3 + (int)(x & 3u); val = prev; nb += 2; }
                else if (sym == 17) { rep = 3 + (int)(x & 7u); val = 0; nb += 3; }
                else if (sym == 18) { rep = 11 + (int)(x & 127u); val = 0; nb += 7; }
                p += (uint32_t)nb;
                if (i + rep > nl + nd) { err = GZS_DECODE; break; }
                if (val) {
                    const int in_l = i >= nl ? 0 : (i + rep <= nl ? rep : nl - i);      // symbols of the run in the literal/length code
                    left_l -= in_l << (15 - val);
                    left_d -= (rep - in_l) << (15 - val);
                    if ((left_l | left_d) < 0) { err = GZS_DECODE; break; }
                }
                if (lane < rep) S.len[i + lane] = (uint8_t)val;
                if (lane + 64 < rep) S.len[i + lane + 64] = (uint8_t)val;
                if (lane + 128 < rep) S.len[i + lane + 128] = (uint8_t)val;
                i += rep;
                prev = val;
            }
            if (err != GZS_OK) break;
            if (__builtin_amdgcn_readfirstlane((int)S.len[256]) == 0) { err = GZS_DECODE; break; }
            if (!WRITE) {
                // a block START candidate must carry complete codes, as every deflate encoder writes them (a single distance code may
                // be incomplete; none at all is legal for a block without matches): sum 2^-len == 1
                uint32_t sl = 0, sd = 0, ndist = 0;
                for (int s = lane; s < nl; s += 64) { const int l = S.len[s]; sl += l ? (1u << (15 - l)) : 0u; }
                if (lane < nd) { const int l = S.len[nl + lane]; sd = l ? (1u << (15 - l)) : 0u; ndist = l ? 1u : 0u; }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { sl += (uint32_t)__shfl_xor((int)sl, o); sd += (uint32_t)__shfl_xor((int)sd, o); ndist += (uint32_t)__shfl_xor((int)ndist, o); }
                sl = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl);      // (the same in every lane: say so, or the branch - and with it
                sd = (uint32_t)__builtin_amdgcn_readfirstlane((int)sd);      // every wave-uniform value of the loop - counts as divergent)
                ndist = (uint32_t)__builtin_amdgcn_readfirstlane((int)ndist);
                if (sl != (1u << 15) || (ndist > 1 && sd != (1u << 15))) { err = GZS_DECODE; break; }
            }
        }
        GziCode lit, dst;
        if (!gzi_build(S, nl, nd, lane, lit, dst)) { err = GZS_DECODE; break; }
        bool eob = false;
        bool nontext = false;
        while (!eob && err == GZS_OK) {
            if (p > end_bits + 64u) { err = WRITE ? GZS_NOSTOP : GZS_DECODE; break; }
            if (!WRITE && op >= GZS_VALIDATE_SYMS) { fin = true; break; }   // (validation: enough of the block has been seen; fin = "not at its end")
            ensure(p);
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
                adv = ((ent & 0x100u) == 0 && lane + (int)cl <= 63) ? cl : 0u;
                const uint32_t ls = sym - 257u, l5 = ls & 31u;
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
                mres = okm ? (len | (dist << 9) | ((cl + le + dl + dx) << 25)) : 0u;
            }
            uint32_t pos = 0;
            for (;;) {
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
                asm("s_bitset0_b64 %0, %1" : "+s"(M) : "s"(pos));
                if (M) {
                    const int n = __builtin_popcountll(M);
                    if (op + n > cap) { err = GZS_OVERFLOW; break; }
                    if (WRITE) {
                        const int r = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0u));
                        if ((M >> lane) & 1) out16[op + r] = (uint16_t)(ent & 0xffu);
                    } else {
                        nontext = nontext || (((M >> lane) & 1) && !gzs_is_text(ent & 0xffu));
                    }
                    op += n;
                }
                const uint32_t e = gzi_rl(ent, (int)pos);
                int sym = (int)(e & 511u), cl = (int)(e >> 9);
                if (cl != 0 && sym < 256) break;
                int len, dist;
                const uint32_t mr = gzi_rl(mres, (int)pos);
                if (mr != 0) {
                    len = (int)(mr & 511u); dist = (int)((mr >> 9) & 0xffffu);
                    pos += mr >> 25;
                } else {
                    if (cl == 0) {
                        sym = gzi_decode(peek32(p + pos), lit, lane, 0, S.lsym, cl);
                        if (sym < 0) { err = GZS_DECODE; break; }
                    }
                    if (sym < 256) {
                        if (op >= cap) { err = GZS_OVERFLOW; break; }
                        if (WRITE) { if (lane == 0) out16[op] = (uint16_t)sym; }
                        else nontext = nontext || !gzs_is_text((uint32_t)sym);
                        ++op;
                        pos += (uint32_t)cl;
                        if (pos > 53u) break;
                        continue;
                    }
                    if (sym == 256) { pos += (uint32_t)cl; eob = true; break; }
                    if (sym > 285) { err = GZS_DECODE; break; }
                    uint64_t B = peek64(p + pos + (uint32_t)cl);
                    const int ls = sym - 257;
                    const int le = ls < 8 || ls == 28 ? 0 : (ls >> 2) - 1;
                    len = (ls < 8 ? 3 + ls : ls == 28 ? 258 : ((4 + (ls & 3)) << le) + 3) + (int)((uint32_t)B & ((1u << le) - 1u));
                    B >>= le;
                    int dl = 0;
                    const int ds = gzi_decode((uint32_t)B, dst, lane, 16, S.dsym, dl);
                    if (ds < 0 || ds > 29) { err = GZS_DECODE; break; }
                    B >>= dl;
                    const int dx = ds < 4 ? 0 : (ds >> 1) - 1;
                    dist = (ds < 4 ? 1 + ds : ((2 + (ds & 1)) << dx) + 1) + (int)((uint32_t)B & ((1u << dx) - 1u));
                    pos += (uint32_t)(cl + le + dl + dx);
                }
                if (dist > op + GZS_WIN) { err = GZS_DECODE; break; }    // behind the 32 KiB window: not DEFLATE
                if (op + len > cap) { err = GZS_OVERFLOW; break; }
                if (WRITE) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const int j0 = op - dist;                         // (negative: into the unknown window = a marker)
                    if (dist >= len) {
                        for (int k = lane; k < len; k += 64) {
                            const int j = j0 + k;
                            out16[op + k] = j < 0 ? (uint16_t)(GZS_MARK | (uint32_t)(GZS_WIN + j)) : out16[j];
                        }
                    } else if (dist == 1) {
                        const uint16_t b = j0 < 0 ? (uint16_t)(GZS_MARK | (uint32_t)(GZS_WIN + j0)) : out16[j0];
                        for (int k = lane; k < len; k += 64) out16[op + k] = b;
                    } else {
                        for (int k = lane; k < len; k += 64) {
                            const int j = j0 + k % dist;
                            out16[op + k] = j < 0 ? (uint16_t)(GZS_MARK | (uint32_t)(GZS_WIN + j)) : out16[j];
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
                op += len;
                if (pos > 53u) break;
            }
            p += pos;
        }
        if (err != GZS_OK) break;
        if (!WRITE && __ballot(nontext)) { err = GZS_DECODE; break; }
        if (last) { fin = true; break; }
    }
    p_end = p;
    n_out = op;
    return err;
}

// found[k] = the first block start in section k's bits (k = 0: given - the member's first block, or what the batch before found)
__global__ __launch_bounds__(64 * GZS_WAVES) __attribute__((amdgpu_waves_per_eu(6, 8))) void rd_gzs_search_kernel(
    const uint8_t *__restrict__ comp, int64_t limit, uint32_t end_bits, uint32_t sec_bits, int nsec, uint32_t first_start, const GzsState *__restrict__ carry,
    int64_t carry_delta_bits, uint32_t *__restrict__ found) {
    __shared__ GzsSmem SM;
    __shared__ uint8_t KT[512];       // three code-length-code lengths (3 bits each) -> the sum of their 2^(7 - length), 0 for length 0
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    GziWave &S = SM.w[wave];
    for (int e = threadIdx.x; e < 512; e += 64 * GZS_WAVES) {
        uint32_t t = 0;
        for (int j = 0; j < 3; ++j) { const uint32_t l = ((uint32_t)e >> (3 * j)) & 7u; t += l ? (128u >> l) : 0u; }
        KT[e] = (uint8_t)t;
    }
    __syncthreads();
    for (int k = blockIdx.x * GZS_WAVES + wave; k <= nsec; k += gridDim.x * GZS_WAVES) {
        if (k == 0 && !(carry == nullptr && first_start == GZS_SEARCH)) {
            uint32_t st = first_start;
            if (carry) {
                const int64_t v = (int64_t)carry->next_start - carry_delta_bits;
                st = (carry->status != GZS_OK || carry->next_start == GZS_NONE || v < 0) ? GZS_NONE : (uint32_t)v;
            }
            if (lane == 0) found[0] = st;
            continue;
        }
        // (the extra section behind the last one looks for the next batch's start in everything the caller sent along)
        const uint64_t lo64 = (uint64_t)k * sec_bits, hi64 = k == nsec ? (uint64_t)end_bits : lo64 + sec_bits;
        uint32_t res = GZS_NONE;
        if (lo64 + 80 < end_bits) {
            const uint32_t q_end = hi64 + 80 < end_bits ? (uint32_t)hi64 : end_bits - 80u;
            auto load_dw = [&](int d) -> uint32_t {
                const int64_t b = (int64_t)d * 4;
                return b + 4 <= limit ? *reinterpret_cast<const uint32_t *>(comp + b) : 0u;
            };
            // the section's bits pass through one register, 64 dwords at a time (one coalesced load per 57 dwords = 28 trips; a load of
            // eight dwords per trip put a memory round trip into every one of a wave's 2,048 trips)
            int wbase = (int)((uint32_t)lo64 >> 5);
            uint32_t wcur = load_dw(wbase + lane);
            for (uint32_t q = (uint32_t)lo64; q < q_end && res == GZS_NONE; q += 64) {
                // lane i: the 96 bits that start at q + i
                const int d = (int)(q >> 5);
                if (d - wbase > 64 - 7) {
                    wbase = d;
                    wcur = load_dw(wbase + lane);
                }
                const int rel = d - wbase;
                const uint32_t d0 = gzi_rl(wcur, rel), d1 = gzi_rl(wcur, rel + 1), d2 = gzi_rl(wcur, rel + 2), d3 = gzi_rl(wcur, rel + 3),
                               d4 = gzi_rl(wcur, rel + 4), d5 = gzi_rl(wcur, rel + 5), d6 = gzi_rl(wcur, rel + 6);
                const uint32_t s = q & 31;
                const uint32_t A0 = (uint32_t)((((uint64_t)d1 << 32) | d0) >> s), A1 = (uint32_t)((((uint64_t)d2 << 32) | d1) >> s),
                               A2 = (uint32_t)((((uint64_t)d3 << 32) | d2) >> s), A3 = (uint32_t)((((uint64_t)d4 << 32) | d3) >> s),
                               A4 = (uint32_t)((((uint64_t)d5 << 32) | d4) >> s), A5 = (uint32_t)((((uint64_t)d6 << 32) | d5) >> s);
                const bool lowh = lane < 32;
                const uint32_t x0 = lowh ? A0 : A1, x1 = lowh ? A1 : A2, x2 = lowh ? A2 : A3, x3 = lowh ? A3 : A4;
                (void)A5;
                const uint32_t sh = (uint32_t)(lane & 31);
                const uint32_t w0 = __builtin_amdgcn_alignbit(x1, x0, sh), w1 = __builtin_amdgcn_alignbit(x2, x1, sh),
                               w2 = __builtin_amdgcn_alignbit(x3, x2, sh);
                // BFINAL = 0, BTYPE = 2 (bits 1-2 = 0, 1), HLIT <= 29, HDIST <= 29, and a complete code-length code
                bool ok = (w0 & 7u) == 4u && ((w0 >> 3) & 31u) <= 29u && ((w0 >> 8) & 31u) <= 29u && q + (uint32_t)lane < q_end;
                const uint32_t nc = ((w0 >> 13) & 15u) + 4u;
                // (the nc fields of 3 bits, three at a time through a 512-entry table of their 2^(7 - length) sums: 7 LDS reads instead
                // of 19 shift / mask / compare / select / add steps - the trip is VALU-bound when four waves share a SIMD)
                const uint64_t P = (((uint64_t)(w0 >> 17)) | ((uint64_t)w1 << 15) | ((uint64_t)w2 << 47)) & ((1ull << (3u * nc)) - 1ull);
                const uint32_t Plo = (uint32_t)P, Pmi = (uint32_t)(P >> 27), Phi = (uint32_t)(P >> 54);
                const uint32_t kraft = (uint32_t)KT[Plo & 511u] + KT[(Plo >> 9) & 511u] + KT[(Plo >> 18) & 511u] + KT[Pmi & 511u] + KT[(Pmi >> 9) & 511u] +
                                       KT[(Pmi >> 18) & 511u] + KT[Phi & 7u];
                ok = ok && kraft == 128u;
                uint64_t cand = __ballot(ok);
                while (cand && res == GZS_NONE) {
                    const int b = __builtin_ctzll(cand);
                    cand &= cand - 1;
                    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(q + (uint32_t)b));
                    uint32_t pe;
                    int n;
                    bool fin;
                    if (gzs_blocks<false>(S, lane, comp, limit, end_bits, c0, GZS_NONE, nullptr, 1 << 30, pe, n, fin) != GZS_OK) continue;
                    // a plausible header behind it (when the block was seen to its end): not the reserved block type; a dynamic one
                    // with counts in range
                    if (!fin && pe + 17 <= end_bits) {
                        const int dd = (int)(pe >> 5);
                        const uint32_t e0 = load_dw(dd), e1 = load_dw(dd + 1);
                        const uint32_t hv = (uint32_t)(((((uint64_t)e1) << 32) | e0) >> (pe & 31));
                        const uint32_t bt = (hv >> 1) & 3u;
                        if (bt == 3u || (bt == 2u && (((hv >> 3) & 31u) > 29u || ((hv >> 8) & 31u) > 29u))) continue;
                    }
                    res = c0;
                }
            }
        }
        if (lane == 0) found[k] = res;
    }
}

// (8 waves per SIMD = 64 VGPRs: the same time alone - a wave is a latency chain -, and beside the recurrence kernel the sections' workgroups
// pack eight to a CU instead of seven: a CU that holds one of them holds no recurrence workgroup)
__global__ __launch_bounds__(64 * GZS_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void rd_gzs_decode_kernel(
    const uint8_t *__restrict__ comp, int64_t limit, uint32_t end_bits, int nsec, const uint32_t *__restrict__ found, uint16_t *__restrict__ syms, int cap,
    GzsSec *__restrict__ sec) {
    __shared__ GzsSmem SM;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    GziWave &S = SM.w[wave];
    for (int k = blockIdx.x * GZS_WAVES + wave; k < nsec; k += gridDim.x * GZS_WAVES) {
        const uint32_t st = (uint32_t)__builtin_amdgcn_readfirstlane((int)found[k]);      // (wave-uniform: bit positions live in SGPRs)
        GzsSec r{0u, 0u, (uint32_t)GZS_OK, 0u};
        if (st != GZS_NONE) {
            uint32_t stop = GZS_NONE;
            for (int j = k + 1; j <= nsec && stop == GZS_NONE; ++j) stop = (uint32_t)__builtin_amdgcn_readfirstlane((int)found[j]);
            uint32_t pe = st;
            int n = 0;
            bool fin = false;
            const int e = gzs_blocks<true>(S, lane, comp, limit, end_bits, st, stop, syms + (size_t)k * (size_t)cap, cap, pe, n, fin);
            r = GzsSec{(uint32_t)n, pe, (uint32_t)e, fin ? 1u : 0u};
        }
        if (lane == 0) sec[k] = r;
    }
}

// one WAVE: the sections that make up the batch's text, in order - up to the member's end (or the first section that failed); offsets; the
// batch's verdict. 64 sections per trip: found[] / sec[] arrive in one coalesced load, the lanes vote (ballots), the running offset is a
// wave scan. (Round 5: one thread walked the sections, a memory round trip each - 0.44 ms per batch of 1,256.)
__global__ __launch_bounds__(64) void rd_gzs_scan_kernel(const GzsSec *__restrict__ sec, const uint32_t *__restrict__ found, int nsec, int at_eof, int64_t text_cap,
                                                        int64_t *__restrict__ off, int32_t *__restrict__ wslot, int32_t *__restrict__ plist, GzsState *__restrict__ st,
                                                        const GzsState *__restrict__ carry, int search0) {
    const int lane = threadIdx.x;
    GzsState s;
    s.total_len = carry ? carry->total_len : 0;
    s.crc = carry ? carry->crc : 0;
    s.win_valid = carry ? carry->win_valid : 0;
    s.status = carry ? carry->status : (uint32_t)GZS_OK;
    s.final = 0; s.end_bit = 0; s.next_start = GZS_NONE; s.bad_section = GZS_NONE; s.n_sections = 0; s.n_text = 0; s.reserved[0] = s.reserved[1] = 0;
    int64_t o = 0;
    int slot = 0;
    bool done = false;
    // (search0: the batch opens a RANGE of the stream - the sections in front of its first block start belong to the range before)
    if (!search0 && found[0] == GZS_NONE && s.status == GZS_OK) { s.status = GZS_NOSTART; s.bad_section = 0; }
    uint32_t first = GZS_NONE;
    for (int base = 0; base < nsec; base += 64) {
        const int k = base + lane;
        const bool in = k < nsec;
        const uint32_t f = in ? found[k] : GZS_NONE;
        GzsSec r{0u, 0u, (uint32_t)GZS_OK, 0u};
        if (in && f != GZS_NONE) r = sec[k];
        const bool live = !done && s.status == GZS_OK;                      // (uniform)
        const bool cand = live && in && f != GZS_NONE;
        const uint64_t stopm = __ballot(cand && (r.status != GZS_OK || r.final != 0));
        const int X = stopm ? __builtin_ctzll(stopm) : 64;                  // the first section of this trip that ends the list
        const uint32_t xs = (uint32_t)__builtin_amdgcn_readlane((int)r.status, X & 63), xe = (uint32_t)__builtin_amdgcn_readlane((int)r.end_bit, X & 63);
        const bool x_err = stopm && xs != GZS_OK;
        const bool inc = cand && (lane < X || (lane == X && !x_err));       // part of the text
        const uint64_t incm = __ballot(inc);
        // exclusive scan of the included sections' symbols
        uint32_t v = inc ? r.n_syms : 0u, sc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)sc, d);
            if (lane >= d) sc += t;
        }
        const int before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(incm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)incm, 0u));
        if (in) {
            off[k] = o + (int64_t)(sc - v);
            wslot[k] = slot + before;
        }
        if (inc) plist[slot + before] = k;
        if (first == GZS_NONE && incm) first = (uint32_t)__builtin_amdgcn_readlane((int)f, __builtin_ctzll(incm));
        o += (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)sc, 63);
        const int n_inc = __builtin_popcountll(incm);
        slot += n_inc;
        s.n_sections += (uint32_t)n_inc;
        if (stopm) {
            if (x_err) { s.status = xs; s.bad_section = (uint32_t)(base + X); }
            else { done = true; s.final = 1; s.end_bit = xe; }
            // (the sections behind the stop in this trip got offsets as if nothing followed: overwrite them with the final values)
            if (in && lane > X) { off[k] = o; wslot[k] = slot; }
        }
    }
    if (lane == 0) {
        off[nsec] = o;
        wslot[nsec] = slot;
        if (search0 && first == GZS_NONE && s.status == GZS_OK) { s.status = GZS_NOSTART; s.bad_section = 0; }
        s.reserved[0] = first;                      // the bit this batch's text starts at (a range's first batch: what its search found)
        if (s.status == GZS_OK && o > text_cap) s.status = GZS_TEXTCAP;
        if (s.status == GZS_OK && !done) {
            // the stream goes on: the next batch starts where this batch's last section stopped = the start the extra section found
            s.next_start = found[nsec];
            if (s.next_start == GZS_NONE) { s.status = at_eof ? GZS_NOSTOP : GZS_NOSTART; s.bad_section = (uint32_t)nsec; }   // at EOF: the member never ended
        }
        s.n_text = s.status == GZS_OK ? o : 0;
        *st = s;
    }
}

// ---- the window chain, on SYMBOLS and in parallel (round 6) -----------------------------------------------------------------------------
// The 32 KiB behind a section are a function of the 32 KiB in front of it: entry i is a byte the section produced, or "entry j of the
// window before" (a marker that survived). Such maps compose - a marker into the previous window is replaced by that window's entry,
// itself a byte or a marker - so the chain need not run in order on one workgroup (round 5: 1,256 sections x 1.6 us, 4.9 ms per batch)
// and need not know the text in front of the batch at all:
//   rd_gzs_symwin_kernel   one workgroup per GROUP of GZS_GROUP consecutive sections, all groups at once: the group's sections in order,
//                          starting from the identity map; windows16[slot] = the window in front of section `slot` RELATIVE to the window
//                          in front of its group, gmaps[g] = the window behind group g likewise;
//   rd_gzs_chain_kernel    one workgroup, the groups in order (a gather of 32 Ki entries each): gwin[g] = the window in front of group g,
//                          from the window in front of the batch - bytes (a stream decoded from its first byte: rd_gz_stream_inflate) or
//                          itself a map (a RANGE of the stream whose predecessor is still being decoded: rd_gz_range_decode);
//   rd_gzs_resolve_kernel  all sections in parallel: a marker goes through windows16[slot], what is still a marker through gwin[group].
// A marker that survives all of it points in front of the member's first byte (streaming: GZS_WINDOW) or into the window in front of
// the range (range mode: it stays in the 16-bit text until the ranks have exchanged their maps).
constexpr int GZS_GROUP = 32;

// thread t of the workgroup owns window entries [32 t, 32 t + 32); W = the current window in LDS (64 KiB); `pre` = the thread's 32 symbols
// of the section about to be processed, fetched one section ahead (they do not depend on the window)
struct GzsWinStep {
    const uint16_t *syms;
    int cap;
    const GzsSec *sec;
    const int32_t *plist;
    int tid;
    u32x4 pre[4];
    __device__ __forceinline__ void fetch(int slot, int end) {
        if (slot >= end) return;
        const int k = plist[slot];
        const int n = (int)sec[k].n_syms;
        if (n < GZS_WIN) return;                    // (short section: the slow path reads its symbols itself)
        const uint8_t *src = reinterpret_cast<const uint8_t *>(syms + (size_t)k * (size_t)cap + (size_t)(n - GZS_WIN) + (size_t)tid * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) __builtin_memcpy(&pre[q], src + 16 * q, 16);
    }
    // the window behind section `slot` into out[16] (two entries per dword); reads W; the caller barriers, stores, barriers
    __device__ __forceinline__ void step(int slot, int end, const uint16_t *W, uint32_t (&out)[16]) {
        const int k = plist[slot];
        const int n = (int)sec[k].n_syms;
        if (n >= GZS_WIN) {
            u32x4 now[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) now[q] = pre[q];
            fetch(slot + 1, end);                   // the next section's symbols travel while this one is resolved
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t two = now[q][j];
                    uint32_t lo = two & 0xffffu, hi = two >> 16;
                    if (lo & GZS_MARK) lo = W[lo & 0x7fffu];
                    if (hi & GZS_MARK) hi = W[hi & 0x7fffu];
                    out[q * 4 + j] = lo | (hi << 16);
                }
            }
        } else {
            const uint16_t *sy = syms + (size_t)k * (size_t)cap;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int i = tid * 32 + j;
                uint32_t b;
                if (i >= GZS_WIN - n) {
                    b = sy[i - (GZS_WIN - n)];
                    if (b & GZS_MARK) b = W[b & 0x7fffu];
                } else {
                    b = W[i + n];
                }
                out[j >> 1] = (j & 1) ? (out[j >> 1] | (b << 16)) : b;
            }
            fetch(slot + 1, end);
        }
    }
};

__global__ __launch_bounds__(1024) void rd_gzs_symwin_kernel(const uint16_t *__restrict__ syms, int cap, const GzsSec *__restrict__ sec, const int32_t *__restrict__ plist,
                                                             const int32_t *__restrict__ wslot, int nsec, const GzsState *__restrict__ st,
                                                             uint16_t *__restrict__ windows16, uint16_t *__restrict__ gmaps) {
    __shared__ __attribute__((aligned(16))) uint16_t W[GZS_WIN];
    const int tid = threadIdx.x;
    if (st->status != GZS_OK) return;
    const int nslots = wslot[nsec];
    const int s0 = (int)blockIdx.x * GZS_GROUP;
    if (s0 >= nslots) return;
    const int s1 = s0 + GZS_GROUP < nslots ? s0 + GZS_GROUP : nslots;
    {   // the identity: entry i = "entry i of the window in front of the group"
        uint32_t id[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) id[j] = (GZS_MARK | (uint32_t)(tid * 32 + 2 * j)) | ((GZS_MARK | (uint32_t)(tid * 32 + 2 * j + 1)) << 16);
        uint16_t *g0 = windows16 + (size_t)s0 * GZS_WIN + tid * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4 o = {id[4 * q], id[4 * q + 1], id[4 * q + 2], id[4 * q + 3]};
            *reinterpret_cast<u32x4 *>(W + tid * 32 + 8 * q) = o;
            *reinterpret_cast<u32x4 *>(g0 + 8 * q) = o;
        }
    }
    GzsWinStep ws{syms, cap, sec, plist, tid, {}};
    ws.fetch(s0, s1);
    __syncthreads();
    for (int slot = s0; slot < s1; ++slot) {
        uint32_t out[16];
        ws.step(slot, s1, W, out);
        __syncthreads();                            // every read of the old window is over
        uint16_t *gdst = (slot + 1 < s1 ? windows16 + (size_t)(slot + 1) * GZS_WIN : gmaps + (size_t)blockIdx.x * GZS_WIN) + tid * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4 o = {out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]};
            *reinterpret_cast<u32x4 *>(W + tid * 32 + 8 * q) = o;
            *reinterpret_cast<u32x4 *>(gdst + 8 * q) = o;
        }
        __syncthreads();
    }
}

// one workgroup: gwin[0] = the window in front of the batch - win8 (bytes, of which the LAST st->win_valid are text: the entries in front of
// them stay markers and any use of them is a distance too far back), or map_in (16-bit entries: range mode), or the identity (a range's first
// batch) -, gwin[g + 1] = gmaps[g] applied to gwin[g]. The last one goes to map_out (16-bit) and / or win_out (bytes).
__global__ __launch_bounds__(1024) void rd_gzs_chain_kernel(const uint16_t *__restrict__ gmaps, const int32_t *__restrict__ wslot, int nsec, GzsState *__restrict__ st,
                                                            const uint8_t *__restrict__ win8, const uint16_t *__restrict__ map_in, int streaming,
                                                            uint16_t *__restrict__ gwin, uint16_t *__restrict__ map_out, uint8_t *__restrict__ win_out) {
    __shared__ __attribute__((aligned(16))) uint16_t W[GZS_WIN];
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    if (st->status != GZS_OK) return;
    const uint32_t valid = streaming ? (st->win_valid > (uint32_t)GZS_WIN ? (uint32_t)GZS_WIN : st->win_valid) : 0u;
    for (int i = tid; i < GZS_WIN; i += 1024) {
        uint16_t b = (uint16_t)(GZS_MARK | (uint32_t)i);
        if (map_in) b = map_in[i];
        else if (win8 && (uint32_t)i >= (uint32_t)GZS_WIN - valid) b = win8[i];
        W[i] = b;
        gwin[i] = b;
    }
    if (tid == 0) s_bad = 0;
    const int nslots = wslot[nsec];
    const int ngroups = (nslots + GZS_GROUP - 1) / GZS_GROUP;
    u32x4 pre[4];
    auto fetch = [&](int g) {
        if (g >= ngroups) return;
        const uint8_t *src = reinterpret_cast<const uint8_t *>(gmaps + (size_t)g * GZS_WIN + (size_t)tid * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) __builtin_memcpy(&pre[q], src + 16 * q, 16);
    };
    fetch(0);
    __syncthreads();
    for (int g = 0; g < ngroups; ++g) {
        u32x4 now[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) now[q] = pre[q];
        fetch(g + 1);
        uint32_t out[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t two = now[q][j];
                uint32_t lo = two & 0xffffu, hi = two >> 16;
                if (lo & GZS_MARK) lo = W[lo & 0x7fffu];
                if (hi & GZS_MARK) hi = W[hi & 0x7fffu];
                out[q * 4 + j] = lo | (hi << 16);
            }
        }
        __syncthreads();
        uint16_t *gdst = gwin + (size_t)(g + 1) * GZS_WIN + tid * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4 o = {out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]};
            *reinterpret_cast<u32x4 *>(W + tid * 32 + 8 * q) = o;
            *reinterpret_cast<u32x4 *>(gdst + 8 * q) = o;
        }
        __syncthreads();
    }
    for (int i = tid; i < GZS_WIN; i += 1024) {
        const uint16_t b = W[i];
        if (map_out) map_out[i] = b;
        if (win_out) win_out[i] = (uint8_t)b;
    }
    if (streaming) {
        // how much of the last window is text: what was valid before + what the batch produced. A marker inside that part points in
        // front of the member's first byte (the markers of the LAST 32 KiB only; the member's CRC covers the rest)
        __shared__ uint32_t s_nv;
        if (tid == 0) {
            const uint64_t v = (uint64_t)valid + (uint64_t)st->n_text;
            s_nv = v > (uint64_t)GZS_WIN ? (uint32_t)GZS_WIN : (uint32_t)v;
        }
        __syncthreads();
        const uint32_t nv = s_nv;
        bool bad = false;
        for (int i = tid; i < GZS_WIN; i += 1024) bad = bad || ((W[i] & GZS_MARK) && (uint32_t)i >= (uint32_t)GZS_WIN - nv);
        if (bad) s_bad = 1;
        __syncthreads();
        if (tid == 0) {
            st->win_valid = nv;
            if (s_bad) st->status = GZS_WINDOW;
        }
    }
}

constexpr int GZS_RTILE = 8192;     // symbols per workgroup of the resolve pass
// all sections in parallel: symbols -> bytes (OUT16 = false; a marker that survives - in front of the member - becomes 0 and is the CRC's
// to report) or -> 16-bit symbols whose markers point into the window in front of the RANGE (OUT16 = true), at their offsets in the text
template <bool OUT16>
__global__ __launch_bounds__(256) void rd_gzs_resolve_kernel(const uint16_t *__restrict__ syms, int cap, const GzsSec *__restrict__ sec, const uint32_t *__restrict__ found,
                                                            const int64_t *__restrict__ off, const int32_t *__restrict__ wslot, int tiles_per_sec,
                                                            const uint16_t *__restrict__ windows16, const uint16_t *__restrict__ gwin, const GzsState *__restrict__ st,
                                                            uint8_t *__restrict__ text, uint16_t *__restrict__ sym_text) {
    const int k = blockIdx.x / tiles_per_sec, t = blockIdx.x % tiles_per_sec;
    if (st->status != GZS_OK && st->status != GZS_WINDOW) return;
    if (found[k] == GZS_NONE || wslot[k + 1] == wslot[k]) return;       // a section that is not part of the text
    const int n = (int)sec[k].n_syms;
    const int i0 = t * GZS_RTILE;
    if (i0 >= n) return;
    const uint16_t *sy = syms + (size_t)k * (size_t)cap;
    const int slot = wslot[k];
    const uint16_t *W1 = windows16 + (size_t)slot * GZS_WIN, *W0 = gwin + (size_t)(slot / GZS_GROUP) * GZS_WIN;
    const int i1 = i0 + GZS_RTILE < n ? i0 + GZS_RTILE : n;
    const int64_t o = off[k];
    auto res = [&](uint32_t s) -> uint32_t {
        if (s & GZS_MARK) {
            s = W1[s & 0x7fffu];
            if (s & GZS_MARK) s = W0[s & 0x7fffu];
        }
        return s;
    };
    auto one = [&](int i) {
        const uint32_t s = res(sy[i]);
        if (OUT16) sym_text[o + i] = (uint16_t)s;
        else text[o + i] = (uint8_t)s;
    };
    // eight symbols per thread and step: one 16-byte load (the slot is aligned, the tile's first symbol need not be), one 8-byte store (16 bytes
    // in range mode) at an aligned place of the text; the few symbols in front of the first aligned place and behind the last whole eight go one by one
    const int tid = (int)threadIdx.x;
    int lead = (int)((8 - ((o + i0) & 7)) & 7);
    if (lead > i1 - i0) lead = i1 - i0;
    if (tid < lead) one(i0 + tid);
    const int v0 = i0 + lead, nvec = (i1 - v0) >> 3;
    for (int g = tid; g < nvec; g += 256) {
        const int i = v0 + 8 * g;
        u32x4 v;
        __builtin_memcpy(&v, sy + i, 16);
        uint32_t r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = res((v[j >> 1] >> (16 * (j & 1))) & 0xffffu);
        if (OUT16) {
            const u32x4 w = {r[0] | (r[1] << 16), r[2] | (r[3] << 16), r[4] | (r[5] << 16), r[6] | (r[7] << 16)};
            *reinterpret_cast<u32x4 *>(sym_text + o + i) = w;
        } else {
            const uint2 w = make_uint2((r[0] & 0xffu) | ((r[1] & 0xffu) << 8) | ((r[2] & 0xffu) << 16) | (r[3] << 24),
                                       (r[4] & 0xffu) | ((r[5] & 0xffu) << 8) | ((r[6] & 0xffu) << 16) | (r[7] << 24));
            *reinterpret_cast<uint2 *>(text + o + i) = w;
        }
    }
    const int t0 = v0 + 8 * nvec;
    if (tid < i1 - t0) one(t0 + tid);
}

// the range's symbols -> bytes, once the window in front of the range is known (`valid` = how many of its LAST bytes are text of the member:
// a marker in front of them is a distance too far back); st: n_text = n for the CRC kernels behind this one, status GZS_WINDOW on such a marker
__global__ __launch_bounds__(256) void rd_gzs_symtext_kernel(const uint16_t *__restrict__ sym_text, int64_t n, const uint8_t *__restrict__ win, uint32_t valid,
                                                            uint8_t *__restrict__ text, GzsState *__restrict__ st) {
    if (blockIdx.x == 0 && threadIdx.x == 0) st->n_text = n;
    bool bad = false;
    const uint32_t lowest = (uint32_t)GZS_WIN - (valid > (uint32_t)GZS_WIN ? (uint32_t)GZS_WIN : valid);
    const int64_t n8 = n / 8;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n8; g += (int64_t)gridDim.x * 256) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(sym_text + g * 8);
        uint32_t o[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t s = (v[j >> 1] >> (16 * (j & 1))) & 0xffffu;
            if (s & GZS_MARK) {
                const uint32_t w = s & 0x7fffu;
                bad = bad || w < lowest;
                s = w < lowest ? 0u : win[w];
            }
            o[j >> 2] |= (s & 0xffu) << (8 * (j & 3));
        }
        *reinterpret_cast<uint2 *>(text + g * 8) = make_uint2(o[0], o[1]);
    }
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < n - n8 * 8) {
        const int64_t i = n8 * 8 + threadIdx.x;
        uint32_t s = sym_text[i];
        if (s & GZS_MARK) {
            const uint32_t w = s & 0x7fffu;
            bad = bad || w < lowest;
            s = w < lowest ? 0u : win[w];
        }
        text[i] = (uint8_t)s;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicMax(&st->status, (uint32_t)GZS_WINDOW);
}

// CRC-32 of every 64 KiB tile of the batch's text (one wave per tile, 1 KiB per lane): 16 bytes per load, four bytes per step
// (slicing-by-4: four independent table reads per dword instead of four dependent ones). Round 5: the byte-at-a-time form - a
// global byte load per lane and step, the lanes 1 KiB apart - took 2.1 ms per 300 MB batch.
constexpr int GZS_CTILE = 65536;
__global__ __launch_bounds__(256) void rd_gzs_crc_kernel(const uint8_t *__restrict__ text, const GzsState *__restrict__ st, uint32_t *__restrict__ tile_crc) {
    __shared__ uint32_t tab[4][256];      // tab[k][b]: the CRC of byte b followed by k zero bytes
    {
        uint32_t c = threadIdx.x;
        for (int b = 0; b < 8; ++b) c = (c & 1) ? (c >> 1) ^ 0xedb88320u : c >> 1;
        tab[0][threadIdx.x] = c;
        __syncthreads();
        uint32_t v = c;
        for (int k = 1; k < 4; ++k) {
            v = (v >> 8) ^ tab[0][v & 0xffu];
            tab[k][threadIdx.x] = v;
        }
    }
    __syncthreads();
    const int64_t n = st->status == GZS_OK ? st->n_text : 0;
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t t0 = tile * GZS_CTILE;
    if (t0 >= n) return;
    const int len = (int)(n - t0 < GZS_CTILE ? n - t0 : GZS_CTILE);
    const uint8_t *p = text + t0;
    const int per = 1024;
    const int b0 = lane * per < len ? lane * per : len, b1 = b0 + per < len ? b0 + per : len;
    uint32_t c = 0xffffffffu;
    int b = b0;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {                   // (b0 is a multiple of 1,024)
        for (; b + 16 <= b1; b += 16) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(p + b);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t x = c ^ v[j];
                c = tab[3][x & 0xffu] ^ tab[2][(x >> 8) & 0xffu] ^ tab[1][(x >> 16) & 0xffu] ^ tab[0][x >> 24];
            }
        }
    }
    for (; b < b1; ++b) c = tab[0][(c ^ p[b]) & 0xffu] ^ (c >> 8);
    c = ~c;
    if (b1 == b0) c = 0;
    c = gz_multmodp(gz_x8n((uint32_t)(len - b1)), c);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c ^= (uint32_t)__shfl_xor((int)c, o);
    if (lane == 0) tile_crc[tile] = c;
}

// the tile CRCs folded into the stream's running CRC: crc(A || B) = crc(A) x^(8 |B|) + crc(B) (mod P). One wave: lane l folds its
// run of tiles, then the 64 partial results are folded in order.
__global__ __launch_bounds__(64) void rd_gzs_fold_kernel(const uint32_t *__restrict__ tile_crc, GzsState *__restrict__ st) {
    if (st->status != GZS_OK) return;
    const int lane = threadIdx.x;
    const int64_t n = st->n_text;
    const int64_t ntiles = (n + GZS_CTILE - 1) / GZS_CTILE;
    const int64_t per = (ntiles + 63) / 64;
    const int64_t t0 = lane * per < ntiles ? lane * per : ntiles, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    const uint32_t xfull = gz_x8n(GZS_CTILE);
    uint32_t c = 0;
    uint64_t bytes = 0;
    for (int64_t t = t0; t < t1; ++t) {
        const int64_t len = n - t * GZS_CTILE < GZS_CTILE ? n - t * GZS_CTILE : GZS_CTILE;
        c = gz_multmodp(len == GZS_CTILE ? xfull : gz_x8n((uint32_t)len), c) ^ tile_crc[t];
        bytes += (uint64_t)len;
    }
    // x^(8 bytes) of this lane's run, by the bits of `bytes` (it can exceed 2^32 in principle; a batch is far below)
    const uint32_t xr = gz_x8n((uint32_t)bytes);
    uint32_t crc = st->crc;
    for (int l = 0; l < 64; ++l) {
        const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane((int)c, l), xl = (uint32_t)__builtin_amdgcn_readlane((int)xr, l);
        const uint32_t has = (uint32_t)__builtin_amdgcn_readlane((int)(t1 > t0 ? 1 : 0), l);
        if (has) crc = gz_multmodp(xl, crc) ^ cl;
    }
    if (lane == 0) {
        st->crc = crc;
        st->total_len += (uint64_t)n;
    }
}

struct GzsPlan {
    int nsec, tiles_per_sec, ctiles, ngroups;
    size_t found_bytes, sec_bytes, off_bytes, wslot_bytes, plist_bytes, crc_bytes, windows_bytes, gmaps_bytes, gwin_bytes, syms_bytes, total;
};
GzsPlan gzs_plan(int64_t data_bytes, int32_t section_bytes, int32_t cap_syms, int64_t text_cap) {
    GzsPlan p;
    p.nsec = (int)((data_bytes + section_bytes - 1) / section_bytes);
    if (p.nsec < 1) p.nsec = 1;
    p.tiles_per_sec = (cap_syms + GZS_RTILE - 1) / GZS_RTILE;
    p.ctiles = (int)((text_cap + GZS_CTILE - 1) / GZS_CTILE) + 1;
    p.ngroups = (p.nsec + GZS_GROUP - 1) / GZS_GROUP;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    p.found_bytes = al((size_t)(p.nsec + 1) * 4);
    p.sec_bytes = al((size_t)p.nsec * sizeof(GzsSec));
    p.off_bytes = al((size_t)(p.nsec + 1) * 8);
    p.wslot_bytes = al((size_t)(p.nsec + 1) * 4);
    p.plist_bytes = al((size_t)(p.nsec + 1) * 4);
    p.crc_bytes = al((size_t)p.ctiles * 4);
    p.windows_bytes = al((size_t)(p.nsec + 1) * GZS_WIN * 2);      // (16-bit entries: a window relative to its group's)
    p.gmaps_bytes = al((size_t)p.ngroups * GZS_WIN * 2);
    p.gwin_bytes = al((size_t)(p.ngroups + 1) * GZS_WIN * 2);
    p.syms_bytes = al((size_t)p.nsec * (size_t)cap_syms * 2);
    p.total = p.found_bytes + p.sec_bytes + p.off_bytes + p.wslot_bytes + p.plist_bytes + p.crc_bytes + p.windows_bytes + p.gmaps_bytes + p.gwin_bytes + p.syms_bytes;
    return p;
}

}  // namespace
