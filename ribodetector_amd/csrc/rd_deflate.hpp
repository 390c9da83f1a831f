// rd_deflate.hpp - label-partitioned records of a chunk -> gzip (BGZF) members, on the device (rd_gz_* kernels)
// Part of the single translation unit rd_kernels.hip (included from there, in order); DESIGN.md §3.10 has the numbers.
//
// Replaces, for the GPU path, the reference's output side: `gzip.open(out, 'wt', compresslevel=5)` fed with the records of one label
// in input order (reference detect.py:485-492,729-741). The records of a chunk are already in HBM (the read bytes travel as the
// chunk's text) and so are the labels; round 3 shipped the labels to the host, which gathered and deflated the records on its 16
// cores - 13-20 % of the device rate, flat in the GPU count. Here the chunk's records of one label become gzip members without
// leaving the device; the host appends the compressed bytes to the file.
//
//   rd_gz_sel_* / rd_gz_pack_kernel   selected records -> one contiguous byte stream (scan of the selected lengths + coalesced copy)
//   rd_gz_deflate_kernel              one workgroup per member of 65,280 input bytes (BGZF's block size: the file is valid BGZF -
//                                     bgzip / htslib index it, this build's reader inflates its members in parallel):
//        wave w owns part w of the member (4,080 bytes: a sixteenth) and its own hash table in LDS (seeded with the last 512 bytes of the
//        part before; 256 buckets of the 8 nearest earlier positions with
//        the same 8-byte hash); a STRIP of 64 consecutive positions is handled at once, one per lane: hash, the bucket's 8 candidates
//        (positions before the strip) and distance 1 (runs) compared over the first 16 bytes; then the 64 positions are inserted; then
//        the strip's parse is resolved from the ballot of match lanes (lazy rule: a match shorter than 16 yields to a longer one at
//        the next position), and only the CHOSEN matches are measured exactly - by the whole wave, 8 lanes per candidate, 32 bytes
//        per candidate and round; tokens go to a scratch in HBM, symbol counts to per-wave LDS histograms;
//        ONE dynamic-Huffman block per member: lengths by two-queue merge over the rank-sorted used symbols, limited to 15 bits the
//        way zlib's gen_bitlen does it, canonical codes; code lengths sent without the run-length symbols (+0.2 %); every wave emits
//        its part's tokens with a prefix sum of bit lengths and LDS atomic-or; stored block if that is smaller; CRC-32 of the
//        member from 255 per-thread CRCs combined with x^n mod P multiplications (zlib's crc32_combine identity).
//   rd_gz_moff_kernel / rd_gz_compact_kernel   member sizes -> offsets, members -> one contiguous stream for the D2H copy
// Matches of 8+ bytes only (runs: 6+): on FASTQ shorter matches cost more bits than the 2-bit literals they replace; measured with
// the lane-for-lane CPU model tools/gzdev_model.c against zlib level 5: 0.98 / 1.00 / 0.94 of its size on three FASTQ profiles (1.02-1.06 on files of reads that are all copies of ONE template).
#pragma once
#include "rd_common.hpp"

namespace {

constexpr int GZ_MEMBER = 65280;                       // input bytes per member (BGZF_BLOCK_SIZE 0xff00)
constexpr int GZ_NQ = 16, GZ_PART = GZ_MEMBER / GZ_NQ;      // waves per member; each parses its own part 
constexpr int GZ_THREADS = 64 * GZ_NQ;
constexpr int GZ_SEED = 512;                                   // bytes of the part before that a wave's table starts with
constexpr int GZ_CRCB = 68;                                    // bytes per thread in the CRC pass: the threads of waves 1 .. 15 (wave 0 builds the tree meanwhile)
static_assert(GZ_CRCB * (GZ_THREADS - 64) >= GZ_MEMBER && GZ_CRCB % 4 == 0 && GZ_MEMBER % (4 * GZ_NQ) == 0 && GZ_PART + GZ_SEED < (1 << 13) && GZ_SEED % 64 == 0, "deflate kernel geometry");
constexpr int GZ_HBITS = 9, GZ_WAYS = 4;                // 512 buckets of the 4 nearest earlier positions per wave
// a table entry, 16 bits: 2 more hash bits | 1 bit that only an EMPTY way has set | 13 bits of position in the part (+ 1)
constexpr uint32_t GZ_EMPTY = 0x2000u, GZ_EMPTY2 = GZ_EMPTY | (GZ_EMPTY << 16), GZ_POSM = 0x1fffu;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int GZ_MINM = 8, GZ_MINRUN = 6, GZ_MAXM = 258, GZ_CAP = 16;
constexpr int GZ_SLOT = 65536;                         // output bytes reserved per member (BGZF: total block size <= 65536)
constexpr int GZ_HDR = 18, GZ_TRL = 8;
constexpr int GZ_NSYM = 320;                           // 0..285 literal/length symbols, 286..315 distance symbols
constexpr int GZ_MAX_GRID = 512;

__constant__ uint8_t GZ_CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
// x^(2^i) mod P of the (reflected) CRC-32 polynomial: zlib's x2n_table
__constant__ uint32_t GZ_X2N[32] = {0x40000000u, 0x20000000u, 0x08000000u, 0x00800000u, 0x00008000u, 0xedb88320u, 0xb1e6b092u, 0xa06a2517u,
                                    0xed627daeu, 0x88d14467u, 0xd7bbfe6au, 0xec447f11u, 0x8e7ea170u, 0x6427800eu, 0x4d47bae0u, 0x09fe548fu,
                                    0x83852d0fu, 0x30362f1au, 0x7b5a9cc3u, 0x31fec169u, 0x9fec022au, 0x6c8dedc4u, 0x15d6874du, 0x5fde7a4eu,
                                    0xbad90e37u, 0x2e4e5eefu, 0x4eaba214u, 0xa8a472c0u, 0x429a969eu, 0x148d302au, 0xc40ba6d0u, 0xc4e22c3cu};

// ------------------------------------------------------------------------------------------------
// selection: out_off[i] = bytes of the selected records before record i (record i is selected when labels[i] == label)
// ------------------------------------------------------------------------------------------------
constexpr int GZ_SCAN_ITEMS = 2048;   // records per workgroup (256 threads x 8)

__device__ __forceinline__ int64_t gz_sel_len(const int64_t *rec_start, const int8_t *labels, int64_t i, int64_t n, int label) {
    return (i < n && labels[i] == (int8_t)label) ? rec_start[i + 1] - rec_start[i] : 0;
}

__device__ __forceinline__ int64_t gz_block_scan(int64_t v, int64_t *sh, int64_t &total) {   // exclusive scan over the 256 threads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    int64_t base = 0;
    for (int w = 0; w < wave; ++w) base += sh[w];
    total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(256) void rd_gz_sel_sum_kernel(const int64_t *__restrict__ rec_start, const int8_t *__restrict__ labels,
                                                           int64_t n, int label, int64_t *__restrict__ bsum) {
    __shared__ int64_t sh[4];
    const int64_t i0 = (int64_t)blockIdx.x * GZ_SCAN_ITEMS + threadIdx.x * 8;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += gz_sel_len(rec_start, labels, i0 + k, n, label);
    int64_t total;
    gz_block_scan(s, sh, total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// one workgroup: bsum[] -> exclusive bases in place; info = {compressed bytes (filled later), plain bytes, members}
__global__ __launch_bounds__(256) void rd_gz_sel_base_kernel(int64_t *__restrict__ bsum, int nb, int64_t *__restrict__ info, int64_t text_bytes) {
    __shared__ int64_t sh[4];
    __shared__ int64_t run_s;
    if (threadIdx.x == 0) run_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int b = b0 + threadIdx.x;
        const int64_t v = b < nb ? bsum[b] : 0;
        int64_t total;
        const int64_t ex = gz_block_scan(v, sh, total);
        const int64_t run = run_s;
        if (b < nb) bsum[b] = run + ex;
        __syncthreads();
        if (threadIdx.x == 0) run_s = run + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // info[3] != 0: the record table does not describe the text (selected bytes > text bytes: overlapping or stray records) - the
        // workspace is sized by the text, so nothing is packed or compressed and the caller is told
        const bool bad = run_s > text_bytes || run_s < 0;
        info[0] = 0;
        info[1] = bad ? 0 : run_s;
        info[2] = bad ? 0 : (run_s + GZ_MEMBER - 1) / GZ_MEMBER;
        info[3] = bad ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void rd_gz_sel_off_kernel(const int64_t *__restrict__ rec_start, const int8_t *__restrict__ labels,
                                                           int64_t n, int label, const int64_t *__restrict__ bbase,
                                                           int64_t *__restrict__ out_off) {
    __shared__ int64_t sh[4];
    const int64_t i0 = (int64_t)blockIdx.x * GZ_SCAN_ITEMS + threadIdx.x * 8;
    int64_t v[8], s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = gz_sel_len(rec_start, labels, i0 + k, n, label); s += v[k]; }
    int64_t total;
    int64_t run = bbase[blockIdx.x] + gz_block_scan(s, sh, total);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (i0 + k <= n) out_off[i0 + k] = run;   // (entry n = the total)
        run += v[k];
    }
}

// selected records -> plain[out_off[i] ...): a workgroup takes 256 consecutive records, whose output range is contiguous, and its
// threads take 16-byte pieces of that range (a piece finds its record by bisection over the 257 staged offsets)
constexpr int GZ_PACK_RECS = 256;
__global__ __launch_bounds__(256) void rd_gz_pack_kernel(const uint8_t *__restrict__ text, const int64_t *__restrict__ rec_start,
                                                        const int64_t *__restrict__ out_off, int64_t n, uint8_t *__restrict__ plain,
                                                        const int64_t *__restrict__ info) {
    __shared__ int64_t offs[GZ_PACK_RECS + 1];
    __shared__ int64_t srcs[GZ_PACK_RECS];
    if (info[3]) return;
    const int64_t r0 = (int64_t)blockIdx.x * GZ_PACK_RECS;
    const int nr = (int)(n - r0 < GZ_PACK_RECS ? n - r0 : GZ_PACK_RECS);
    for (int k = threadIdx.x; k <= nr; k += 256) offs[k] = out_off[r0 + k];
    for (int k = threadIdx.x; k < nr; k += 256) srcs[k] = rec_start[r0 + k];
    __syncthreads();
    const int64_t ob = offs[0], oe = offs[nr];
    for (int64_t o = (ob & ~(int64_t)15) + 16 * (int64_t)threadIdx.x; o < oe; o += 16 * 256) {
        int64_t a = o < ob ? ob : o;                    // the piece is [a, e) (its neighbours in other workgroups write the rest)
        const int64_t e = o + 16 < oe ? o + 16 : oe;
        int lo = 0, hi = nr;                            // largest r with offs[r] <= a (records of zero length share an offset:
        while (hi - lo > 1) {                           // the LAST of them is the one that holds byte a)
            const int mid = (lo + hi) >> 1;
            if (offs[mid] <= a) lo = mid; else hi = mid;
        }
        int r = lo;
        if (a == o && e == o + 16 && offs[r + 1] >= e) {   // a whole piece from one record: one unaligned 16-byte load, one aligned store
            u32x4 v;
            __builtin_memcpy(&v, text + srcs[r] + (a - offs[r]), 16);
            *reinterpret_cast<u32x4 *>(plain + o) = v;
            continue;
        }
        while (a < e) {
            while (offs[r + 1] <= a) ++r;
            const int64_t stop = offs[r + 1] < e ? offs[r + 1] : e;
            const uint8_t *src = text + srcs[r] + (a - offs[r]);
            for (; a < stop; ++a) plain[a] = *src++;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// DEFLATE of one member
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) GzSmem {
    uint32_t text[(GZ_MEMBER + 16) / 4];      // the member's bytes (+ zero pad); after the parse: the output (header, deflate data, trailer)
    u32x2 tab[GZ_NQ][1 << GZ_HBITS];          // per wave: hash -> the 4 nearest earlier positions in its part (+ 1), 16 bits each, nearest first
    uint32_t hist[GZ_NQ][GZ_NSYM / 2];        // per wave: symbol counts of its part, two 16-bit counters per word (a part has < 2^16 tokens)
    uint32_t freq[GZ_NSYM];
    uint8_t lens[GZ_NSYM];
    uint16_t codes[GZ_NSYM];
    uint32_t clfreq[19];
    uint8_t cllen[19];
    uint16_t clcode[19];
    alignas(16) uint32_t key[288];                        // tree scratch: (frequency << 9) | symbol of the used symbols, 0xffffffff for the others (16-byte aligned: see below)
    uint32_t cw[GZ_NSYM];                     // code | length << 16 (what the emission looks up: one LDS read per code)
    uint16_t order[288];                      // tree scratch: used symbols by (frequency, symbol)
    uint32_t w[576];
    uint16_t parent[576];
    uint8_t depth[576];
    uint32_t crc_tab[256];
    uint64_t litmask[GZ_NQ][(GZ_PART + 63) / 64];   // per strip: which positions became literals
    uint32_t scan[GZ_NQ];
    uint32_t blc[16];                         // leaves per code length (tree scratch)
    uint32_t wcnt[5][16];                     // symbols per code length in each wave of symbols (canonical codes)
    uint32_t qbits[GZ_NQ], qtok[GZ_NQ];
    uint32_t crc;
    int used, hlit, hdist, hclen;
};

// Cross-lane steps as DPP modifiers of VALU instructions (a few cycles each) instead of ds_bpermute round trips (__shfl_*: the LDS
// crossbar, ~100 cycles when the next step depends on it - a strip resolves several matches, each with two reductions).
template <int CTRL>
__device__ __forceinline__ int gz_dpp(int v) {   // every lane has a valid source for the controls used with this form
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ uint32_t gz_dpp0(uint32_t v) {   // lanes without a source read 0 (row_shr / wave_shl shift zeros in)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
constexpr int DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_ROW_MIRROR = 0x140, DPP_ROW_HALF_MIRROR = 0x141, DPP_WAVE_SHL1 = 0x130;
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
__device__ __forceinline__ int gz_min8(int v) {    // minimum over each group of 8 consecutive lanes, in all of them
    v = min(v, gz_dpp<DPP_QUAD_1032>(v));
    v = min(v, gz_dpp<DPP_QUAD_2301>(v));
    return min(v, gz_dpp<DPP_ROW_HALF_MIRROR>(v));
}
__device__ __forceinline__ int gz_min64(int v) {   // wave minimum (uniform)
    v = gz_min8(v);
    v = min(v, gz_dpp<DPP_ROW_MIRROR>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int gz_min16(int v) {   // minimum over each row of 16 lanes, in all of them
    v = gz_min8(v);
    return min(v, gz_dpp<DPP_ROW_MIRROR>(v));
}
__device__ __forceinline__ int gz_max64_of_rows(int v) {   // wave maximum of a value that is already uniform within rows of 16
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ uint32_t gz_wave_scan(uint32_t x) {   // inclusive prefix sum over the 64 lanes
    uint32_t t = x;
    t += gz_dpp0<DPP_ROW_SHR1>(t);
    t += gz_dpp0<DPP_ROW_SHR2>(t);
    t += gz_dpp0<DPP_ROW_SHR4>(t);
    t += gz_dpp0<DPP_ROW_SHR8>(t);                              // inclusive within each row of 16
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)t, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)t, 31),
                   r2 = (uint32_t)__builtin_amdgcn_readlane((int)t, 47);
    const int row = (int)(threadIdx.x & 63) >> 4;
    return t + (row == 0 ? 0u : row == 1 ? r0 : row == 2 ? r0 + r1 : r0 + r1 + r2);
}

// hash of the 8 bytes at a position: ONE 32-bit multiply (quarter rate on the vector unit: the three of rounds 4-5 were a twelfth of a
// strip's compare stage); bucket = the top GZ_HBITS bits, the entry's tag = the two bits below them. tools/gzdev_model.c: the size is
// the same or a hair smaller than with the three-multiply mix.
__device__ __forceinline__ uint32_t gz_hash(uint32_t w0, uint32_t w1) { return (w0 ^ __builtin_amdgcn_alignbit(w1, w1, 17)) * 0x9E3779B1u; }
// position of the lowest set bit; 0xffffffff for 0 (what v_ffbl_b32 returns: the C builtins make a select around it)
__device__ __forceinline__ uint32_t gz_ffbl(uint32_t x) {
    uint32_t r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ uint32_t gz_ld32(const uint32_t *T, int i) {   // 4 bytes at byte offset i (any alignment)
    const uint32_t *q = T + (i >> 2);
    return __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)(i & 3));
}

// 16 bytes at byte offset i (any alignment): five dwords requested together (one LDS round trip), shifted into place
__device__ __forceinline__ void gz_ld128(const uint32_t *T, int i, uint32_t &w0, uint32_t &w1, uint32_t &w2, uint32_t &w3) {
    const uint32_t *q = T + (i >> 2);
    const uint32_t a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4], sh = (uint32_t)(i & 3);
    w0 = __builtin_amdgcn_alignbyte(a1, a0, sh);
    w1 = __builtin_amdgcn_alignbyte(a2, a1, sh);
    w2 = __builtin_amdgcn_alignbyte(a3, a2, sh);
    w3 = __builtin_amdgcn_alignbyte(a4, a3, sh);
}

// length 3..258 -> length symbol - 257; distance 1..32768 -> distance symbol (RFC 1951 3.2.5: groups of four / two codes per extra-bit
// count, so the symbol is the position of the leading bit and the bits behind it; a few VALU instructions instead of a table in LDS -
// a dependent round trip per token)
__device__ __forceinline__ int gz_len_sym(int L) {
    const int l = L - 3;
    const int e = 29 - __clz(l | 8);
    const int s = 4 * (e + 1) + ((l >> e) & 3);
    return l < 8 ? l : l == 255 ? 28 : s;
}
__device__ __forceinline__ int gz_dist_sym(int D) {
    const int d = D - 1;
    const int e = 30 - __clz(d | 4);
    const int s = 2 * (e + 1) + ((d >> e) & 1);
    return d < 4 ? d : s;
}

// extra bits behind a length / distance symbol (RFC 1951 3.2.5 as arithmetic: a __constant__ table indexed per lane is a vector
// load from memory); the extra VALUE is the low bits of (length - 3) / (distance - 1): every symbol's base is a
// multiple of its range there
__device__ __forceinline__ int gz_len_extra(int ls) { return (ls < 8 || ls == 28) ? 0 : (ls >> 2) - 1; }
__device__ __forceinline__ int gz_dist_extra(int ds) { return ds < 4 ? 0 : (ds >> 1) - 1; }

__device__ __forceinline__ uint32_t gz_multmodp(uint32_t a, uint32_t b) {   // zlib's multmodp: a(x) b(x) mod P, reflected
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ 0xedb88320u : b >> 1;
    }
    return p;
}
__device__ __forceinline__ uint32_t gz_x8n(uint32_t nbytes) {   // x^(8 nbytes) mod P
    uint32_t p = 1u << 31;
    for (int k = 3; nbytes; nbytes >>= 1, ++k)
        if (nbytes & 1) p = gz_multmodp(GZ_X2N[k & 31], p);
    return p;
}

// per-wave symbol counters: two 16-bit counters in a 32-bit word (LDS atomics are 32 bits wide; a part holds < 2^16 tokens)
__device__ __forceinline__ void gz_count(uint32_t *hist, int sym) { atomicAdd(&hist[sym >> 1], 1u << (16 * (sym & 1))); }
__device__ __forceinline__ uint32_t gz_counted(const uint32_t *hist, int sym) { return (hist[sym >> 1] >> (16 * (sym & 1))) & 0xffffu; }

__device__ __forceinline__ uint32_t gz_scan_wg(uint32_t v, uint32_t *sh, uint32_t &total) {   // exclusive, the deflate kernel's GZ_THREADS threads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t inc = gz_wave_scan(v);
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += sh[w];
    total = 0;
#pragma unroll
    for (int w = 0; w < GZ_NQ; ++w) total += sh[w];
    __syncthreads();
    return base + inc - v;
}

__device__ __forceinline__ void gz_or_bits(uint32_t *out, uint32_t bitpos, uint64_t bits, int nb) {   // nb <= 64: up to three words
    if (nb <= 0) return;
    const uint32_t sh = bitpos & 31u;
    uint32_t *w = out + (bitpos >> 5);
    const uint32_t w0 = (uint32_t)(bits << sh);
    const uint64_t rest = sh ? bits >> (32 - sh) : bits >> 32;
    if (w0) atomicOr(w, w0);
    if ((uint32_t)rest) atomicOr(w + 1, (uint32_t)rest);
    if (rest >> 32) atomicOr(w + 2, (uint32_t)(rest >> 32));
}

#ifdef RD_DIAG
// diagnostic build only: cycles per stage of the member loop, summed per workgroup (tools/gz_bench.py --stages)
__device__ unsigned long long *g_gz_prof = nullptr;
#define GZ_STAMP(k)                                                                                         \
    do {                                                                                                    \
        if (g_gz_prof && threadIdx.x == 0) {                                                                \
            const unsigned long long now_ = clock64();                                                      \
            atomicAdd(&g_gz_prof[k], now_ - stamp_);                                                        \
            stamp_ = now_;                                                                                  \
        }                                                                                                   \
    } while (0)
#define GZ_STAMP_ARG , unsigned long long &stamp_
#define GZ_STAMP_PASS , stamp_
#else
#define GZ_STAMP(k) do { } while (0)
#define GZ_STAMP_ARG
#define GZ_STAMP_PASS
#endif

// Code lengths of an alphabet of N symbols (frequencies in LDS), at most MAXB bits: rank sort of the used symbols (all threads), then
// one thread: two-queue merge, depths, zlib's repair of the lengths beyond MAXB, the rarest leaves get the longest codes; canonical
// codes (bit-reversed: DEFLATE sends Huffman codes MSB first) by all threads. Called by all GZ_THREADS threads.
template <int N, int MAXB, class Side>
__device__ void gz_huff(GzSmem &S, uint32_t *freq, uint8_t *lens, uint16_t *codes, Side side GZ_STAMP_ARG) {
    static_assert(N <= 288 && N <= GZ_THREADS, "gz_huff: one symbol per thread");
    const int tid = threadIdx.x;
    GZ_STAMP(4);
    int used = __syncthreads_count(tid < N && freq[tid] != 0);
    if (used < 2) {   // at least two codes (zlib build_tree): a decoder never sees a 0-bit code
        if (tid == 0)
            for (int i = 0; used < 2 && i < N; ++i)
                if (!freq[i]) { freq[i] = 1; ++used; }
        used = 2;
        __syncthreads();
    }
    // rank sort of the used symbols by (frequency, symbol): one key per symbol ((f << 9) | symbol: a member holds < 2^17 tokens),
    // every thread counts the keys below its own - four keys per LDS read, a compare and an add-with-carry per key
    const uint32_t f = tid < N ? freq[tid] : 0u;
    const uint32_t mykey = f ? (f << 9) | (uint32_t)tid : 0xffffffffu;
    if (tid < 288) S.key[tid] = mykey;
    if (tid < N) lens[tid] = 0;
    __syncthreads();
    if (f) {
        const u32x4 *k4 = reinterpret_cast<const u32x4 *>(S.key);
        int r = 0;
#pragma unroll 6
        for (int j = 0; j < (N + 3) / 4; ++j) {
            const u32x4 k = k4[j];
            r += (k.x < mykey ? 1 : 0) + (k.y < mykey ? 1 : 0) + (k.z < mykey ? 1 : 0) + (k.w < mykey ? 1 : 0);
        }
        S.order[r] = (uint16_t)tid;
        S.w[r] = f;
    }
    __syncthreads();
    GZ_STAMP(11);   // rank sort
    if (tid == 0) {
        const int ns = used;
        // two-queue merge (leaves in rank order, internal nodes in the order they are made). The next four of either queue wait in
        // registers: a pick costs no LDS round trip (it was three per node: heads, heads again, the two weights), the refill is
        // requested four picks ahead. 0xffffffff = nothing there (weights stay below 2^18).
        constexpr uint32_t NONE = 0xffffffffu;
        int a = 0, b = ns, nn = ns;
        uint32_t A0 = S.w[0], A1 = S.w[1], A2 = 2 < ns ? S.w[2] : NONE, A3 = 3 < ns ? S.w[3] : NONE;   // leaves a .. a + 3 (ns >= 2)
        uint32_t B0 = NONE, B1 = NONE, B2 = NONE, B3 = NONE;                                              // internal nodes b .. b + 3
        while (nn < 2 * ns - 1) {
            uint32_t sum = 0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                int pick;
                if (A0 <= B0) {      // (a tie goes to the leaf, as in zlib's order of equal weights; both empty cannot happen)
                    pick = a++;
                    sum += A0;
                    A0 = A1; A1 = A2; A2 = A3;
                    A3 = a + 3 < ns ? S.w[a + 3] : NONE;
                } else {
                    pick = b++;
                    sum += B0;
                    B0 = B1; B1 = B2; B2 = B3;
                    B3 = b + 3 < nn ? S.w[b + 3] : NONE;
                }
                S.parent[pick] = (uint16_t)nn;
            }
            S.w[nn] = sum;
            const int d = nn - b;    // the new node's place in the window of internal nodes (beyond it: read back when the window gets there)
            B0 = d == 0 ? sum : B0; B1 = d == 1 ? sum : B1; B2 = d == 2 ? sum : B2; B3 = d == 3 ? sum : B3;
            ++nn;
        }
        // depths of the INTERNAL nodes only (each hangs below a node made later): the leaves - most of the nodes - take theirs from
        // their parents in parallel below, and are counted and given their lengths in parallel too
        GZ_STAMP(12);   // merge
        S.depth[nn - 1] = 0;
        for (int i = nn - 2; i >= ns; --i) S.depth[i] = (uint8_t)(S.depth[S.parent[i]] + 1);   // (<= 287: fits)
        GZ_STAMP(13);   // depths of the internal nodes
    } else if (tid >= 64) {
        side();   // (the waves that only wait for the merge: work that depends on nothing here - the member's CRC)
    }
    if (tid <= MAXB) S.blc[tid] = 0;
    __syncthreads();
    const int ns = used;
    if (tid < ns) {
        int d = S.depth[S.parent[tid]] + 1;
        if (d > MAXB) d = MAXB;
        atomicAdd(&S.blc[d], 1u);
    }
    __syncthreads();
    if (tid == 0) {
        int bl[MAXB + 1];
        long K = 0;
#pragma unroll
        for (int k = 0; k <= MAXB; ++k) {
            bl[k] = k ? (int)S.blc[k] : 0;
            if (k) K += (long)bl[k] << (MAXB - k);
        }
        bool moved = false;
        while (K > (1L << MAXB)) {   // zlib gen_bitlen: a leaf moves one level down and takes an overflowed leaf as its brother
            int bits = MAXB - 1;
            for (;;) {
                int c = 0;
#pragma unroll
                for (int k = 1; k <= MAXB; ++k) c = (k == bits) ? bl[k] : c;
                if (c) break;
                --bits;
            }
#pragma unroll
            for (int k = 1; k <= MAXB; ++k) {
                if (k == bits) bl[k] -= 1;
                if (k == bits + 1) bl[k] += 2;
            }
            bl[MAXB] -= 1;
            K -= 1;
            moved = true;
        }
        if (moved) {
#pragma unroll
            for (int k = 1; k <= MAXB; ++k) S.blc[k] = (uint32_t)bl[k];
        }
    }
    __syncthreads();
    if (tid < ns) {   // leaf tid of the (frequency, symbol) order: the rarest leaves get the longest codes
        int len = 0, cum = 0;
        for (int k = MAXB; k >= 1; --k) {
            const int c = (int)S.blc[k];
            if (len == 0 && tid < cum + c) len = k;
            cum += c;
        }
        lens[S.order[tid]] = (uint8_t)len;
    }
    __syncthreads();
    // canonical codes (N <= GZ_THREADS: symbol = thread): code = sum over shorter lengths l of count(l) 2^(L - l) + rank among the
    // symbols of the same length - the rank from one ballot per length (this wave) and the earlier waves' counts, the counts per
    // length are S.blc (a loop over all symbols per symbol was 286 dependent LDS reads)
    static_assert(N <= GZ_THREADS && (N + 63) / 64 <= 5 && MAXB <= 15, "gz_huff: one symbol per thread, five waves of symbols");
    {
        const int lane = tid & 63, wv = tid >> 6;
        const int L = tid < N ? (int)lens[tid] : 0;
        uint32_t same_before = 0, mine = 0;
        if (wv * 64 < N) {
#pragma unroll
            for (int l = 1; l <= MAXB; ++l) {
                const uint64_t b = __ballot(L == l);
                same_before = L == l ? (uint32_t)__popcll(b & ((1ull << lane) - 1ull)) : same_before;
                mine = lane == l ? (uint32_t)__popcll(b) : mine;
            }
            if (lane >= 1 && lane <= MAXB) S.wcnt[wv][lane] = mine;
        }
        __syncthreads();
        if (tid < N) {
            uint32_t code = 0;
            if (L) {
                uint32_t cnt_shorter = 0;
#pragma unroll
                for (int l = 1; l < MAXB; ++l) cnt_shorter += l < L ? S.blc[l] << (L - l) : 0u;
                for (int v = 0; v < wv; ++v) same_before += S.wcnt[v][L];
                code = cnt_shorter + same_before;
                code = __brev(code) >> (32 - L);
            }
            codes[tid] = (uint16_t)code;
        }
    }
    __syncthreads();
    GZ_STAMP(14);   // leaf lengths, repair, canonical codes
}


// plain: the selected records of the chunk as one stream (info[1] bytes); member m = bytes [65280 m, ...). toks: 65,280 words of
// scratch per workgroup. slots: GZ_SLOT bytes per member; msize[m] = the member's size.
__global__ __launch_bounds__(GZ_THREADS) void rd_gz_deflate_kernel(const uint8_t *__restrict__ plain, const int64_t *__restrict__ info,
                                                           uint32_t *__restrict__ toks, uint8_t *__restrict__ slots,
                                                           uint32_t *__restrict__ msize) {
    __shared__ GzSmem S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t total = info[1];
    const int64_t nmem = (total + GZ_MEMBER - 1) / GZ_MEMBER;
    {   // tables, once per workgroup
        if (tid < 256) {
            uint32_t c = (uint32_t)tid;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0xedb88320u : c >> 1;
            S.crc_tab[tid] = c;
        }
    }
    // x^(8 (bytes of a FULL member behind this thread's piece)) mod P, once: every member but a file's last is full, and the
    // exponentiation (up to eleven 32-step multiplications) was most of the CRC stage's time
    const int crc_b0 = GZ_CRCB * (tid - 64);      // (this thread's piece of a member: threads 64 ..)
    const uint32_t x8n_full = tid >= 64 && crc_b0 < GZ_MEMBER ? gz_x8n((uint32_t)(GZ_MEMBER - (crc_b0 + GZ_CRCB < GZ_MEMBER ? crc_b0 + GZ_CRCB : GZ_MEMBER))) : 0u;
    uint32_t *mytoks = toks + (size_t)blockIdx.x * GZ_MEMBER;
    for (int64_t m = blockIdx.x; m < nmem; m += gridDim.x) {
        const int len = (int)(total - m * GZ_MEMBER < GZ_MEMBER ? total - m * GZ_MEMBER : GZ_MEMBER);
        __syncthreads();
#ifdef RD_DIAG
        unsigned long long stamp_ = clock64();
#endif
        {   // member -> LDS (16-byte loads: plain is 256-byte aligned and GZ_MEMBER a multiple of 16), zero pad; tables cleared
            const u32x4 *src4 = reinterpret_cast<const u32x4 *>(plain + m * GZ_MEMBER);
            u32x4 *t4 = reinterpret_cast<u32x4 *>(S.text);
            for (int k = tid; k < (GZ_MEMBER + 16) / 16; k += GZ_THREADS) {
                u32x4 v = {0u, 0u, 0u, 0u};
                const int b = 16 * k;
                if (b + 16 <= len) {
                    v = src4[k];
                } else if (b < len) {            // (the last vector of a file's last member: dwords, the bytes behind the end masked)
                    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src4 + k);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (b + 4 * j < len) {
                            uint32_t w = s32[j];
                            if (b + 4 * j + 4 > len) w &= 0xffffffffu >> (8 * (b + 4 * j + 4 - len));
                            v[j] = w;
                        }
                }
                t4[k] = v;
            }
            static_assert(sizeof(S.tab) % 16 == 0 && sizeof(S.hist) % 16 == 0, "tables are cleared 16 bytes at a time");
            u32x4 *tab4 = reinterpret_cast<u32x4 *>(&S.tab[0][0]), *hist4 = reinterpret_cast<u32x4 *>(&S.hist[0][0]);
            for (int k = tid; k < (int)(sizeof(S.tab) / 16); k += GZ_THREADS) tab4[k] = u32x4{GZ_EMPTY2, GZ_EMPTY2, GZ_EMPTY2, GZ_EMPTY2};
            for (int k = tid; k < (int)(sizeof(S.hist) / 16); k += GZ_THREADS) hist4[k] = u32x4{0u, 0u, 0u, 0u};
            if (tid == 0) S.crc = 0;
        }
        __syncthreads();
        GZ_STAMP(0);   // load
        // ---- parse: wave w, part w ---------------------------------------------------------------------------------------------------
        const int q0 = wave * GZ_PART, q1 = len < q0 + GZ_PART ? len : q0 + GZ_PART;
        uint32_t *qt = mytoks + q0;
        int ntok = 0, carry = 0;
        const uint8_t *tb = reinterpret_cast<const uint8_t *>(S.text);
        // the last GZ_SEED bytes of the part before are INSERTED into this wave's table first (hashed, not parsed): the first records of a
        // part find their header and quality matches in the record before them like any other record (-3 % of size on sequencer-like
        // FASTQ for 8 more strips of hashing per part of 64; positions in the table count from qb)
        const int qb = q0 >= GZ_SEED ? q0 - GZ_SEED : 0;
        if (q0 < len)
            for (int s0 = qb; s0 < q0; s0 += 64) {
                const int p = s0 + lane;
                if (p < q0) {
                    const uint32_t w0 = gz_ld32(S.text, p), w1 = gz_ld32(S.text, p + 4);
                    const uint32_t hh = gz_hash(w0, w1);
                    const uint32_t h = hh >> (32 - GZ_HBITS);
                    const uint32_t tag = ((hh >> (30 - GZ_HBITS)) & 3u) << 14;
                    const u32x2 ent = S.tab[wave][h];
                    S.tab[wave][h] = u32x2{(ent.x << 16) | tag | (uint32_t)(p - qb + 1), (ent.y << 16) | (ent.x >> 16)};
                }
            }
        for (int s0 = q0; s0 < q1; s0 += 64) {
            const int n = q1 - s0 < 64 ? q1 - s0 : 64;
            const int p = s0 + lane;
            const bool in = lane < n;
            const int lim = q1 - p < GZ_MAXM ? q1 - p : GZ_MAXM;
            const bool hv = in && p + 8 <= q1;
            const int pl = in ? p : s0;                 // (lanes past the part's end load somewhere harmless)
            uint32_t w0, w1, w2, w3;                    // the 16 bytes at the position
            gz_ld128(S.text, pl, w0, w1, w2, w3);
            const uint32_t hh = gz_hash(w0, w1);
            const uint32_t h = hh >> (32 - GZ_HBITS);
            const uint32_t tag = ((hh >> (30 - GZ_HBITS)) & 3u) << 14;   // two more hash bits ride in the entry (a position needs 13 bits):
                                                                        // three of four hash collisions are rejected without touching the text
            // Per lane: the bucket's 4 candidates (the 4 nearest earlier positions with this hash, nearest first) compared over their
            // first 16 bytes: a length CAPPED at 16 per lane - all the lazy rule needs (zlib level 5 does not look for a better match
            // behind one of 16+ either). Exact lengths are found later, for the CHOSEN matches only.
            // (Round 5: 512 x 4 instead of 256 x 8 - the compares are VALU-bound, not latency-bound, and this stage was 47 % of the
            // kernel: half the compares for +0.5 % of size on sequencer-like FASTQ, nothing on the bench's; tools/gzdev_model.c.)
            const u32x2 ent = hv ? S.tab[wave][h] : u32x2{GZ_EMPTY2, GZ_EMPTY2};
            int Lc = 0, Dc = 0;        // capped length and distance of the lane's best candidate
            uint32_t full = 0;         // ways whose first 16 bytes agree (bit 8: the run candidate): their exact length is still open
            if (carry < 64) {          // (else every position of the strip lies inside a match: nothing to find, only to insert)
                // All loads BEFORE any compare: every way requests 16 bytes (five dwords) from its candidate - or, when the way is empty
                // or its tag differs, from the lane's own position (consecutive lanes, consecutive addresses: no bank conflicts) - and
                // the empty asm keeps the compiler from sinking the loads into per-way branches (dependent LDS round trips).
                const uint32_t tagw = tag | (tag << 16);
                const uint32_t y[2] = {ent.x ^ tagw, ent.y ^ tagw};      // a way passes when its tag bits and its empty bit are all 0 here
                uint32_t d[GZ_WAYS][5];
                int cs[GZ_WAYS];
                bool oks[GZ_WAYS];
#pragma unroll
                for (int u = 0; u < GZ_WAYS; ++u) {
                    oks[u] = (y[u >> 1] & (0xe000u << (16 * (u & 1)))) == 0;
                    const int c13 = (int)((ent[u >> 1] >> (16 * (u & 1))) & GZ_POSM);
                    cs[u] = oks[u] ? qb - 1 + c13 : pl;
                    const uint32_t *cq = S.text + (cs[u] >> 2);
#pragma unroll
                    for (int j = 0; j < 5; ++j) d[u][j] = cq[j];
                }
                asm volatile("" ::"v"(d[0][0]), "v"(d[0][1]), "v"(d[0][2]), "v"(d[0][3]), "v"(d[0][4]), "v"(d[1][0]), "v"(d[1][1]), "v"(d[1][2]),
                             "v"(d[1][3]), "v"(d[1][4]), "v"(d[2][0]), "v"(d[2][1]), "v"(d[2][2]), "v"(d[2][3]), "v"(d[2][4]), "v"(d[3][0]),
                             "v"(d[3][1]), "v"(d[3][2]), "v"(d[3][3]), "v"(d[3][4]));
                uint32_t key = 0;      // (capped length << 2) | (3 - way): the longest, the nearest on ties
#pragma unroll
                for (int u = 0; u < GZ_WAYS; ++u) {
                    const uint32_t sh = (uint32_t)(cs[u] & 3);
                    const uint32_t x0 = __builtin_amdgcn_alignbyte(d[u][1], d[u][0], sh) ^ w0, x1 = __builtin_amdgcn_alignbyte(d[u][2], d[u][1], sh) ^ w1;
                    const uint32_t x2 = __builtin_amdgcn_alignbyte(d[u][3], d[u][2], sh) ^ w2, x3 = __builtin_amdgcn_alignbyte(d[u][4], d[u][3], sh) ^ w3;
                    // the first byte of 8..15 that differs: lowest set bit of x3:x2 (no set bit: 0xffffffff from both, capped below)
                    const uint32_t fb = min(gz_ffbl(x2), gz_ffbl(x3) | 32u);
                    const uint32_t kt = 8u + min(fb >> 3, 8u);
                    const uint32_t k = (oks[u] && !(x0 | x1)) ? kt : 0u;
                    full |= (k >> 4) << u;
                    key = max(key, (k << 2) | (uint32_t)(3 - u));
                }
                Lc = (int)(key >> 2);
                const uint32_t bw = 3u - (key & 3u);
                Dc = p - (qb - 1 + (int)(((bw & 2u ? ent.y : ent.x) >> (16 * (bw & 1u))) & GZ_POSM));
                const uint32_t splat = (uint32_t)tb[pl > q0 ? pl - 1 : pl] * 0x01010101u;
                if (in && p > q0 && lim >= GZ_MINRUN && w0 == splat && (w1 & 0xffffu) == (splat & 0xffffu)) {   // a run of 6+
                    const uint32_t y1 = w1 ^ splat, y2 = w2 ^ splat, y3 = w3 ^ splat;
                    const int k = y1 ? 4 + (__builtin_ctz(y1) >> 3) : y2 ? 8 + (__builtin_ctz(y2) >> 3) : y3 ? 12 + (__builtin_ctz(y3) >> 3) : GZ_CAP;
                    if (k == GZ_CAP) full |= 1u << 8;
                    if (k >= Lc) { Lc = k; Dc = 1; }
                }
                if (Lc > lim) Lc = lim;
            }
            // insert: the occupants from before this strip move one way down (lanes of this strip with the same hash differ only in the
            // low half of .x: whichever of them wins the store leaves a valid bucket)
            if (hv) S.tab[wave][h] = u32x2{(ent.x << 16) | tag | (uint32_t)(p - qb + 1), (ent.y << 16) | (ent.x >> 16)};
            GZ_STAMP(8);    // strip: per-lane candidates + insert
            if (carry >= n) {
                if (lane == 0) S.litmask[wave][(s0 - q0) >> 6] = 0;
                carry -= n;
                continue;
            }
            const int Ln = (int)gz_dpp0<DPP_WAVE_SHL1>((uint32_t)Lc);   // lane + 1's (lane 63: 0)
            const bool defer = Lc > 0 && Lc < GZ_CAP && lane + 1 < n && Ln > Lc;
            const bool eff = in && Lc > 0 && !defer;
            const uint64_t mm = __ballot(eff);
            const uint64_t all = n == 64 ? ~0ull : ((1ull << n) - 1);
            uint64_t sel = 0;
            int pos = carry, L = 0, D = 0;
            while (pos < n) {   // from chosen match to chosen match (uniform: everything below is wave-wide)
                const uint64_t from = ~0ull << pos;
                const uint64_t m2 = mm & from;
                if (!m2) { sel |= all & from; pos = n; break; }
                const int f = __builtin_ctzll(m2);
                sel |= (f == 63 ? ~0ull : ((2ull << f) - 1)) & from;
                int bestL = __builtin_amdgcn_readlane(Lc, f), bestD = __builtin_amdgcn_readlane(Dc, f);
                const uint32_t fullf = (uint32_t)__builtin_amdgcn_readlane((int)full, f);
                if (fullf) {
                    // the exact length of lane f's match, by the whole wave
                    const int pf = s0 + f, limf = q1 - pf < GZ_MAXM ? q1 - pf : GZ_MAXM;
                    const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)ent.x, f), e1 = (uint32_t)__builtin_amdgcn_readlane((int)ent.y, f);
                    bestL = 0;
                    if (fullf & 0xffu) {
                        // 16 lanes per way, each comparing 4 bytes - 64 bytes per way and round (a match of 258 in four rounds)
                        const int way = lane >> 4, sub = lane & 15;
                        const uint32_t ew = way < 2 ? e0 : e1;
                        const bool act = (fullf >> way) & 1u;
                        const int c = act ? qb + (int)((ew >> (16 * (way & 1))) & GZ_POSM) - 1 : pf;
                        int glen = act ? limf : 0;
                        bool open = act;
                        for (int base = GZ_CAP; base < limf; base += 64) {
                            if (!__ballot(open)) break;
                            const int o = base + 4 * sub;
                            uint32_t x = 0;
                            if (open && o < limf) x = gz_ld32(S.text, pf + o) ^ gz_ld32(S.text, c + o);
                            const int cand = gz_min16(x ? o + (__builtin_ctz(x) >> 3) : 0x7fffffff);
                            if (open && cand != 0x7fffffff) { glen = cand < limf ? cand : limf; open = false; }
                        }
                        const int key = gz_max64_of_rows((glen << 2) | (3 - way));   // longest; the nearest on ties
                        const int bw = 3 - (key & 3);
                        const uint32_t eb = bw < 2 ? e0 : e1;
                        bestL = key >> 2;
                        bestD = pf - (qb + (int)((eb >> (16 * (bw & 1))) & GZ_POSM) - 1);
                    }
                    if (fullf >> 8) {   // the run: the first byte from pf + 16 on that differs from the byte before pf
                        const uint32_t sp = (uint32_t)tb[pf - 1] * 0x01010101u;
                        int r = limf;
                        for (int base = GZ_CAP; base < limf; base += 256) {
                            const int o = base + 4 * lane;
                            const uint32_t x = o < limf ? gz_ld32(S.text, pf + o) ^ sp : 0u;
                            const int cand = gz_min64(x ? o + (__builtin_ctz(x) >> 3) : 0x7fffffff);
                            if (cand != 0x7fffffff) { r = cand < limf ? cand : limf; break; }
                        }
                        if (r >= bestL) { bestL = r; bestD = 1; }
                    }
                    // (a lane's capped candidates of fewer than 16 bytes cannot beat one of 16+)
                }
                if (lane == f) { L = bestL; D = bestD; }
                pos = f + bestL;
            }
            carry = pos - n;
            GZ_STAMP(9);    // strip: walk over the chosen matches
            const bool tk = (sel >> lane) & 1ull;
            const bool ismatch = tk && eff;
            const uint32_t byte = w0 & 0xffu;
            if (tk) {
                const int idx = __popcll(sel & ((1ull << lane) - 1));
                qt[ntok + idx] = ismatch ? ((uint32_t)L << 16) | (uint32_t)D : byte;
                if (ismatch) {
                    gz_count(S.hist[wave], 257 + gz_len_sym(L));
                    gz_count(S.hist[wave], 286 + gz_dist_sym(D));
                }
            }
            // literal counts: not here (64 lanes adding to the same few counters - A, C, G, T ... - every strip); the strip leaves the mask
            // of its literal positions and the part is counted in one pass after the loop
            const uint64_t lm = __ballot(tk && !ismatch);
            if (lane == 0) S.litmask[wave][(s0 - q0) >> 6] = lm;
            ntok += __popcll(sel);
            GZ_STAMP(10);   // strip: tokens + counts
        }
        {   // literal counts of the part: lane l walks the literal positions of strips l, l + 64, ... (ds_add without return)
            const int nstrip = q1 > q0 ? (q1 - q0 + 63) >> 6 : 0;
            for (int st = lane; st < nstrip; st += 64) {
                uint64_t lm = S.litmask[wave][st];
                const uint8_t *sb = tb + q0 + 64 * st;
                while (lm) {
                    const int b = __builtin_ctzll(lm);
                    lm &= lm - 1;
                    gz_count(S.hist[wave], sb[b]);
                }
            }
        }
        if (lane == 0) S.qtok[wave] = (uint32_t)ntok;
        GZ_STAMP(2);   // parse of wave 0
        __syncthreads();
        GZ_STAMP(3);   // waiting for the slowest wave
        // ---- codes --------------------------------------------------------------------------------------------------------------------
        for (int s = tid; s < GZ_NSYM; s += GZ_THREADS) {
            uint32_t f = 0;
#pragma unroll
            for (int w = 0; w < GZ_NQ; ++w) f += gz_counted(S.hist[w], s);
            if (s == 256) f += 1;   // end of block
            S.freq[s] = f;
        }
        __syncthreads();
        // ---- CRC-32: thread 64 + t takes bytes [GZ_CRCB t, GZ_CRCB (t + 1)), the pieces are combined with x^(8 bytes after) mod P; run by
        // waves 1 .. 15 while wave 0's first thread merges the literal/length tree (gz_huff's side work)
        auto crc_side = [&]() {
            if (crc_b0 < len) {
                const int b0 = crc_b0, b1 = b0 + GZ_CRCB < len ? b0 + GZ_CRCB : len;
                uint32_t c = 0xffffffffu;
                for (int b = b0; b < b1; b += 4) {
                    const uint32_t v = S.text[b >> 2];
                    const int nb = b1 - b < 4 ? b1 - b : 4;
                    for (int k = 0; k < nb; ++k) c = S.crc_tab[(c ^ (v >> (8 * k))) & 0xffu] ^ (c >> 8);
                }
                c = ~c;
                c = gz_multmodp(len == GZ_MEMBER ? x8n_full : gz_x8n((uint32_t)(len - b1)), c);
                atomicXor(&S.crc, c);
            }
        };
        auto no_side = []() {};
        gz_huff<286, 15>(S, S.freq, S.lens, S.codes, crc_side GZ_STAMP_PASS);
        gz_huff<30, 15>(S, S.freq + 286, S.lens + 286, S.codes + 286, no_side GZ_STAMP_PASS);
        for (int k = tid; k < 316; k += GZ_THREADS) S.cw[k] = (uint32_t)S.codes[k] | ((uint32_t)S.lens[k] << 16);
        if (tid == 0) {
            int hlit = 286, hdist = 30;
            while (hlit > 257 && S.lens[hlit - 1] == 0) --hlit;
            while (hdist > 1 && S.lens[286 + hdist - 1] == 0) --hdist;
            S.hlit = hlit; S.hdist = hdist;
        }
        if (tid < 19) S.clfreq[tid] = 0;
        __syncthreads();
        const int hlit = S.hlit, hdist = S.hdist, nseq = hlit + hdist;
        // the code lengths are sent one by one (no run-length symbols 16-18: about 30 bytes more per member, nothing sequential)
        int sq[2];
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int i = tid + GZ_THREADS * rep;
            sq[rep] = i < nseq ? (i < hlit ? S.lens[i] : S.lens[286 + i - hlit]) : -1;
            if (sq[rep] >= 0) atomicAdd(&S.clfreq[sq[rep]], 1u);
        }
        __syncthreads();
        gz_huff<19, 7>(S, S.clfreq, S.cllen, S.clcode, no_side GZ_STAMP_PASS);
        if (tid == 0) {
            int hclen = 19;
            while (hclen > 4 && S.cllen[GZ_CLORD[hclen - 1]] == 0) --hclen;
            S.hclen = hclen;
        }
        // bits of every part's tokens
        {
            uint32_t b = 0;
            for (int s = lane; s < GZ_NSYM; s += 64) {
                const uint32_t extra = s >= 286 ? (uint32_t)gz_dist_extra(s - 286 < 30 ? s - 286 : 0) : s >= 257 ? (uint32_t)gz_len_extra(s - 257) : 0u;
                if (s < 316) b += gz_counted(S.hist[wave], s) * (S.lens[s] + extra);
            }
            b = gz_wave_scan(b);
            if (lane == 63) S.qbits[wave] = b;
        }
        __syncthreads();
        GZ_STAMP(4);   // codes
        const int hclen = S.hclen;
        const uint32_t c0 = sq[0] >= 0 ? S.cllen[sq[0]] : 0, c1 = sq[1] >= 0 ? S.cllen[sq[1]] : 0;
        uint32_t seqbits_lo, seqbits_total;
        // (thread t holds entries t and t + GZ_THREADS: two scans, so that the entries stay in order)
        const uint32_t ex0 = gz_scan_wg(c0, S.scan, seqbits_lo);
        const uint32_t ex1 = seqbits_lo + gz_scan_wg(c1, S.scan, seqbits_total);
        seqbits_total += seqbits_lo;
        const uint32_t hdr_bits = 3 + 5 + 5 + 4 + 3 * (uint32_t)hclen + seqbits_total;
        uint32_t tok_bits = S.lens[256];
#pragma unroll
        for (int w = 0; w < GZ_NQ; ++w) tok_bits += S.qbits[w];
        const uint32_t cbytes_dyn = (hdr_bits + tok_bits + 7) >> 3;
        const uint32_t cbytes_sto = 5 + (uint32_t)len;
        const uint32_t crc = S.crc;
        uint8_t *slot = slots + (size_t)m * GZ_SLOT;
        __syncthreads();
        // (the dynamic block is assembled in S.text - GZ_MEMBER + 16 bytes: header + block + trailer, and the 8 bytes gz_or_bits may touch
        // past its last bit, must fit there; a member that compresses by less than that margin is stored too)
        if (cbytes_dyn >= cbytes_sto || GZ_HDR + cbytes_dyn + GZ_TRL + 8 > (uint32_t)sizeof(S.text)) {
            // ---- stored block (text that does not compress): header, LEN, NLEN, the bytes as they are -----------------------------------
            const uint32_t tot = GZ_HDR + cbytes_sto + GZ_TRL;
            if (tid == 0) {
                uint32_t *sw = reinterpret_cast<uint32_t *>(slot);   // 18 header bytes, then 01 | LEN | NLEN: 23 bytes in six words
                const uint32_t ul = (uint32_t)len, nl = ~ul & 0xffffu;
                sw[0] = 0x04088b1fu; sw[1] = 0; sw[2] = 0x0006ff00u; sw[3] = 0x00024342u;
                sw[4] = ((tot - 1) & 0xffffu) | (1u << 16) | ((ul & 0xffu) << 24);
                sw[5] = (ul >> 8) | (nl << 8);                       // (byte 23, the first data byte, is written below)
                uint8_t *t = slot + GZ_HDR + cbytes_sto;
                for (int k = 0; k < 4; ++k) { t[k] = (uint8_t)(crc >> (8 * k)); t[4 + k] = (uint8_t)((uint32_t)len >> (8 * k)); }
                msize[m] = tot;
            }
            for (int k = tid; k < len; k += GZ_THREADS) slot[GZ_HDR + 5 + k] = tb[k];
            continue;
        }
        // ---- output buffer = the text's LDS: header | dynamic block | trailer -----------------------------------------------------------
        for (int k = tid; k < (GZ_MEMBER + 16) / 4; k += GZ_THREADS) S.text[k] = 0;
        __syncthreads();
        uint32_t *out = S.text;
        const uint32_t tot = GZ_HDR + cbytes_dyn + GZ_TRL;
        constexpr uint32_t B0 = GZ_HDR * 8;   // first bit of the deflate data
        if (tid == 0) {
            out[0] = 0x04088b1fu; out[1] = 0; out[2] = 0x0006ff00u; out[3] = 0x00024342u;   // 1f 8b 08 04 | mtime | xfl os xlen | 'B' 'C' 2 0
            out[4] = (tot - 1) & 0xffffu;                                                   // BSIZE; the deflate data follows in the same word
            gz_or_bits(out, B0, 1u | (2u << 1) | ((uint32_t)(hlit - 257) << 3) | ((uint32_t)(hdist - 1) << 8) | ((uint32_t)(hclen - 4) << 13), 17);
            gz_or_bits(out, B0 + hdr_bits + tok_bits - S.lens[256], S.codes[256], S.lens[256]);   // end of block
        }
        if (tid < hclen) gz_or_bits(out, B0 + 17 + 3 * tid, S.cllen[GZ_CLORD[tid]], 3);
        if (sq[0] >= 0) gz_or_bits(out, B0 + 17 + 3 * hclen + ex0, S.clcode[sq[0]], (int)c0);
        if (sq[1] >= 0) gz_or_bits(out, B0 + 17 + 3 * hclen + ex1, S.clcode[sq[1]], (int)c1);
        {   // tokens of this wave's part: FOUR per lane and trip (one 16-byte load from the scratch, requested a trip ahead; one prefix sum per
            // 256 tokens; the four code words of a lane go out as one bit string when they fit). A trip is VALU-bound (sixteen waves code
            // at once): requesting the tokens four trips ahead, or right behind the parse, changes nothing (measured, round 5).
            uint32_t bit = B0 + hdr_bits;
            for (int v = 0; v < wave; ++v) bit += S.qbits[v];
            const int nt = ntok;
            auto trip = [&](const u32x4 tv4, const int t) {
                uint64_t bits[4];
                int nb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bits[j] = 0;
                    nb[j] = 0;
                    if (t + j < nt) {
                        const uint32_t tv = tv4[j];
                        const uint32_t L = tv >> 16;
                        if (L) {
                            const uint32_t D = tv & 0xffffu;
                            const int ls = gz_len_sym((int)L), ds = gz_dist_sym((int)D);
                            const uint32_t lw = S.cw[257 + ls], dw = S.cw[286 + ds];
                            const int le = gz_len_extra(ls), de = gz_dist_extra(ds);
                            uint64_t b = lw & 0xffffu;
                            int n = (int)(lw >> 16);
                            b |= (uint64_t)((L - 3u) & ((1u << le) - 1u)) << n;
                            n += le;
                            b |= (uint64_t)(dw & 0xffffu) << n;
                            n += (int)(dw >> 16);
                            b |= (uint64_t)((D - 1u) & ((1u << de) - 1u)) << n;
                            n += de;
                            bits[j] = b;
                            nb[j] = n;
                        } else {
                            const uint32_t cw = S.cw[tv];
                            bits[j] = cw & 0xffffu;
                            nb[j] = (int)(cw >> 16);
                        }
                    }
                }
                const int sum = nb[0] + nb[1] + nb[2] + nb[3];
                const uint32_t inc = gz_wave_scan((uint32_t)sum);
                const uint32_t at = bit + inc - (uint32_t)sum;
                if (sum <= 64) {                     // (four literals: 8-36 bits)
                    const uint64_t all = bits[0] | (nb[0] < 64 ? bits[1] << nb[0] : 0ull) | (nb[0] + nb[1] < 64 ? bits[2] << (nb[0] + nb[1]) : 0ull) |
                                         (nb[0] + nb[1] + nb[2] < 64 ? bits[3] << (nb[0] + nb[1] + nb[2]) : 0ull);
                    gz_or_bits(out, at, all, sum);
                } else {
                    gz_or_bits(out, at, bits[0], nb[0]);
                    gz_or_bits(out, at + (uint32_t)nb[0], bits[1], nb[1]);
                    gz_or_bits(out, at + (uint32_t)(nb[0] + nb[1]), bits[2], nb[2]);
                    gz_or_bits(out, at + (uint32_t)(nb[0] + nb[1] + nb[2]), bits[3], nb[3]);
                }
                bit += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            };
            const u32x4 *qt4 = reinterpret_cast<const u32x4 *>(qt);      // (16-byte aligned: GZ_PART and GZ_MEMBER are multiples of 4 words)
            u32x4 tv_next = 4 * lane < nt ? qt4[lane] : u32x4{0u, 0u, 0u, 0u};
            for (int t0 = 0; t0 < nt; t0 += 256) {
                const int t = t0 + 4 * lane;
                const u32x4 tv4 = tv_next;
                if (t + 256 < nt) tv_next = qt4[(t + 256) >> 2];
                trip(tv4, t);
            }
        }
        __syncthreads();
        GZ_STAMP(5);   // emission
        if (tid == 0) {   // trailer: CRC-32, ISIZE (byte-granular position)
            uint8_t *ob = reinterpret_cast<uint8_t *>(out) + GZ_HDR + cbytes_dyn;
            for (int k = 0; k < 4; ++k) { ob[k] = (uint8_t)(crc >> (8 * k)); ob[4 + k] = (uint8_t)((uint32_t)len >> (8 * k)); }
            msize[m] = tot;
        }
        __syncthreads();
        uint32_t *dst = reinterpret_cast<uint32_t *>(slot);
        for (int k = tid; k < (int)((tot + 3) >> 2); k += GZ_THREADS) dst[k] = out[k];
        GZ_STAMP(6);   // copy to the slot
    }
}

// member sizes -> offsets (one workgroup), info[0] = bytes of the compressed stream
__global__ __launch_bounds__(256) void rd_gz_moff_kernel(const uint32_t *__restrict__ msize, int64_t *__restrict__ moff, int64_t *__restrict__ info) {
    __shared__ int64_t sh[4];
    __shared__ int64_t run_s;
    const int64_t nmem = info[2];
    if (threadIdx.x == 0) run_s = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < nmem; b0 += 256) {
        const int64_t b = b0 + threadIdx.x;
        const int64_t v = b < nmem ? (int64_t)msize[b] : 0;
        int64_t total;
        const int64_t ex = gz_block_scan(v, sh, total);
        const int64_t run = run_s;
        if (b < nmem) moff[b] = run + ex;
        __syncthreads();
        if (threadIdx.x == 0) run_s = run + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) info[0] = run_s;
}

// members -> one contiguous stream (byte-granular destinations: dword loads from the slot, byte stores)
__global__ __launch_bounds__(256) void rd_gz_compact_kernel(const uint8_t *__restrict__ slots, const uint32_t *__restrict__ msize,
                                                           const int64_t *__restrict__ moff, const int64_t *__restrict__ info,
                                                           uint8_t *__restrict__ out, int64_t out_cap) {
    const int64_t nmem = info[2];
    for (int64_t m = blockIdx.x; m < nmem; m += gridDim.x) {
        const uint32_t sz = msize[m];
        const int64_t o = moff[m];
        if (o + sz > out_cap) continue;   // (the host sees info[0] > out_cap and reports it)
        const uint32_t *src = reinterpret_cast<const uint32_t *>(slots + (size_t)m * GZ_SLOT);
        uint8_t *dst = out + o;
        const int head = (int)((4 - (o & 3)) & 3);   // bytes up to the first dword boundary of the destination
        for (int k = threadIdx.x; k < head && k < (int)sz; k += 256) dst[k] = (uint8_t)(src[0] >> (8 * k));
        const int nw = sz > (uint32_t)head ? (int)((sz - head) >> 2) : 0;
        uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + head);
        for (int k = threadIdx.x; k < nw; k += 256) {
            const int sb = head + 4 * k;              // source byte offset
            const uint32_t lo = src[sb >> 2], hi = src[(sb >> 2) + 1];
            d32[k] = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(sb & 3));
        }
        for (int k = head + 4 * nw + threadIdx.x; k < (int)sz; k += 256) dst[k] = (uint8_t)(src[k >> 2] >> (8 * (k & 3)));
    }
}

}  // namespace
