// gzdev_model.c - CPU model of the device DEFLATE encoder (ribodetector_amd/csrc/rd_deflate.hpp), lane for lane.
//
// A development tool, not a product path and not the oracle: it exists to (a) measure the compression ratio of the algorithm the
// kernel implements against zlib level 5 (the reference's writer: gzip.open(..., compresslevel=5), reference detect.py:729-741) on
// FASTQ before and while the kernel is written, and (b) pin down the format logic (length-limited Huffman codes, the dynamic block
// header, BGZF framing) in a form that zlib's inflate checks on this machine. The kernel follows the same steps:
//
//   member  = 65,280 input bytes (BGZF's block size: the output is a valid BGZF file), one workgroup;
//   part    = 4,080 bytes (a sixteenth), one wave: its own hash table (seeded with the last 512 bytes of the part before) of 512 buckets x the 4 nearest earlier positions (8-byte hashes; matches
//             of 8+ bytes only: on FASTQ the shorter ones cost more bits than the 2-bit literals they replace - measured here);
//   strip   = 64 consecutive positions, one per lane: every lane hashes its position, looks its candidate up (positions before the
//             strip), also tries distance 1 (runs), measures the match; then all 64 positions are inserted (the highest lane wins a
//             slot); then the parse of the strip is resolved left to right: lazy rule (a match shorter than 16 is dropped for a
//             literal when the next position holds a longer one), matches may run over the following strips;
//   one dynamic-Huffman block per member (stored blocks if that is not smaller), codes limited to 15 bits the way zlib does it.
//
// build: gcc -O2 -o /tmp/gzdev_model tools/gzdev_model.c -lz        run: /tmp/gzdev_model file.fastq [more files]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#ifndef HBITS
#define HBITS 9
#endif
#ifndef HBYTES
#define HBYTES 8
#endif
#ifndef MINM
#define MINM 8
#endif
#ifndef NQ
#define NQ 16
#endif
#ifndef NWAYS
#define NWAYS 4
#endif
#ifndef SEED
#define SEED 512   /* bytes of the part before that a part's table starts with */
#endif
#ifndef LAZY_MAX
#define LAZY_MAX 16
#endif
enum { MEMBER = 65280, PART = MEMBER / NQ, MAXM = 258 };

typedef struct { uint16_t len, dist; } Tok;   // len == 0: literal byte in dist

static const uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DEXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static int len_sym(int len) {   // 3..258 -> 0..28
    int s = 28;
    while (LBASE[s] > len) --s;
    return s;
}
static int dist_sym(int d) {
    int s = 29;
    while (DBASE[s] > d) --s;
    return s;
}

static uint32_t load32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static int hash4(uint32_t v) { return (int)((v * 2654435761u) >> (32 - HBITS)); }
static int hashn(const uint8_t *p) {   // the kernel's hash (gz_hash): two dwords, one 32-bit multiply
    if (HBYTES == 8) { const uint32_t a = load32(p), b = load32(p + 4); return (int)(((a ^ ((b << 15) | (b >> 17))) * 0x9E3779B1u) >> (32 - HBITS)); }
    uint64_t v = 0; memcpy(&v, p, HBYTES); return (int)((v * 0x9E3779B97F4A7C15ull) >> (64 - HBITS));
}

static int mlen(const uint8_t *m, int p, int c, int lim) {
    int l = 0;
    while (l < lim && m[p + l] == m[c + l]) ++l;
    return l;
}

static int opt_dist1_min = 6, opt_lazy = 1, opt_two = 1, opt_rep = 0;

// one part [q0, q1) of the member m; returns the number of tokens
static int parse_part(const uint8_t *m, int q0, int q1, Tok *out) {
    static uint32_t tab[1 << HBITS][NWAYS];
    memset(tab, 0, sizeof(tab));
    int nt = 0, carry = 0, dlast = 0;
    for (int s0 = q0 - SEED < 0 ? 0 : q0 - SEED; s0 < q0; s0 += 64) {     // the end of the part before: inserted, not parsed
        const int n = q0 - s0 < 64 ? q0 - s0 : 64;
        for (int l = 0; l < n; ++l) {
            const int h = hashn(m + s0 + l);
            if (tab[h][0] && (int)tab[h][0] - 1 < s0)
                for (int wy = NWAYS - 1; wy > 0; --wy) tab[h][wy] = tab[h][wy - 1];
            tab[h][0] = (uint32_t)(s0 + l + 1);
        }
    }
    for (int s0 = q0; s0 < q1; s0 += 64) {
        const int n = q1 - s0 < 64 ? q1 - s0 : 64;
        int L[64], D[64], H[64];
        for (int l = 0; l < n; ++l) {
            const int p = s0 + l;
            L[l] = 0; D[l] = 0; H[l] = -1;
            const int lim = q1 - p < MAXM ? q1 - p : MAXM;
            if (p + HBYTES <= q1) {
                H[l] = HBYTES == 4 ? hash4(load32(m + p)) : hashn(m + p);
                if (carry < 64) {
                    for (int wy = 0; wy < (opt_two ? NWAYS : 1); ++wy) {
                        const int c = (int)tab[H[l]][wy] - 1;
                        if (c >= 0) {
                            const int k = mlen(m, p, c, lim);
                            if (k >= MINM && k > L[l]) { L[l] = k; D[l] = p - c; }
                        }
                    }
                }
            }
            if (opt_rep && carry < 64 && dlast > 1 && p - dlast >= q0) {   // the distance of the last match again (a substitution inside a repeat)
                const int k = mlen(m, p, p - dlast, lim);
                if (k >= opt_rep && k > L[l]) { L[l] = k; D[l] = dlast; }
            }
            if (carry < 64 && p > q0) {
                const int k = mlen(m, p, p - 1, lim);
                if (k >= opt_dist1_min && k >= L[l]) { L[l] = k; D[l] = 1; }
            }
        }
        for (int l = 0; l < n; ++l)
            if (H[l] >= 0) {
                if (tab[H[l]][0] && (int)tab[H[l]][0] - 1 < s0)      // the occupants from before the strip move one way down
                    for (int wy = NWAYS - 1; wy > 0; --wy) tab[H[l]][wy] = tab[H[l]][wy - 1];
                tab[H[l]][0] = (uint32_t)(s0 + l + 1);
            }
        if (carry >= n) { carry -= n; continue; }
        int pos = carry;
        while (pos < n) {
            const int k = L[pos];
            const int defer = opt_lazy && k > 0 && k < LAZY_MAX && pos + 1 < n && L[pos + 1] > k;
            if (k > 0 && !defer) {
                out[nt].len = (uint16_t)k; out[nt].dist = (uint16_t)D[pos]; ++nt;
                if (D[pos] > 1) dlast = D[pos];
                pos += k;
            } else {
                out[nt].len = 0; out[nt].dist = m[s0 + pos]; ++nt;
                pos += 1;
            }
        }
        carry = pos - n;
    }
    return nt;
}

// ---- length-limited Huffman code lengths (two-queue merge on sorted leaves, zlib's bl_count repair) -----------------------------
static void huff_lengths(const uint32_t *freq_in, int n, int maxbits, uint8_t *lens) {
    uint32_t freq[288];
    int sym[288], ns = 0;
    memcpy(freq, freq_in, sizeof(uint32_t) * (size_t)n);
    memset(lens, 0, (size_t)n);
    int used = 0;
    for (int i = 0; i < n; ++i) used += freq[i] != 0;
    for (int i = 0; used < 2 && i < n; ++i)   // at least two codes (zlib build_tree): a decoder never sees a 0-bit code
        if (!freq[i]) { freq[i] = 1; ++used; }
    for (int i = 0; i < n; ++i)
        if (freq[i]) sym[ns++] = i;
    // sort leaves by (freq, symbol) ascending: rank sort, as the kernel does it
    int order[288];
    for (int a = 0; a < ns; ++a) {
        int r = 0;
        for (int b = 0; b < ns; ++b)
            r += freq[sym[b]] < freq[sym[a]] || (freq[sym[b]] == freq[sym[a]] && b < a);
        order[r] = sym[a];
    }
    // two-queue merge: nodes 0..ns-1 = leaves (sorted), ns.. = internal, created in non-decreasing weight
    uint64_t w[576];
    int parent[576];
    for (int i = 0; i < ns; ++i) w[i] = freq[order[i]];
    int a = 0, b = ns, nn = ns;
    while (nn < 2 * ns - 1) {
        int pick[2];
        for (int k = 0; k < 2; ++k) {
            if (a < ns && (b >= nn || w[a] <= w[b])) pick[k] = a++;
            else pick[k] = b++;
        }
        w[nn] = w[pick[0]] + w[pick[1]];
        parent[pick[0]] = parent[pick[1]] = nn;
        ++nn;
    }
    int depth[576];
    depth[nn - 1] = 0;
    for (int i = nn - 2; i >= 0; --i) depth[i] = depth[parent[i]] + 1;
    int bl[32];
    memset(bl, 0, sizeof(bl));
    for (int i = 0; i < ns; ++i) {
        if (depth[i] > maxbits) depth[i] = maxbits;
        bl[depth[i]]++;
    }
    long K = 0;
    for (int bts = 1; bts <= maxbits; ++bts) K += (long)bl[bts] << (maxbits - bts);
    while (K > (1L << maxbits)) {   // zlib gen_bitlen: one leaf moves down to become the brother of an overflowed one: K -= 1
        int bts = maxbits - 1;
        while (bl[bts] == 0) --bts;
        bl[bts]--; bl[bts + 1] += 2; bl[maxbits]--;
        K -= 1;
    }
    int i = 0;   // the rarest leaves get the longest codes
    for (int bts = maxbits; bts >= 1; --bts)
        for (int k = 0; k < bl[bts]; ++k) lens[order[i++]] = (uint8_t)bts;
}

static void huff_codes(const uint8_t *lens, int n, uint16_t *codes) {   // canonical codes, bit-reversed (DEFLATE packs them MSB first)
    int bl[16] = {0}, next[16];
    for (int i = 0; i < n; ++i) bl[lens[i]]++;
    bl[0] = 0;
    int code = 0;
    for (int b = 1; b <= 15; ++b) { code = (code + bl[b - 1]) << 1; next[b] = code; }
    for (int i = 0; i < n; ++i) {
        if (!lens[i]) { codes[i] = 0; continue; }
        int c = next[lens[i]]++, r = 0;
        for (int b = 0; b < lens[i]; ++b) r |= ((c >> b) & 1) << (lens[i] - 1 - b);
        codes[i] = (uint16_t)r;
    }
}

typedef struct { uint8_t *p; uint64_t acc; int nb; size_t n; } BitW;
static void put(BitW *w, uint32_t v, int nbits) {
    w->acc |= (uint64_t)v << w->nb;
    w->nb += nbits;
    while (w->nb >= 8) { w->p[w->n++] = (uint8_t)w->acc; w->acc >>= 8; w->nb -= 8; }
}
static void flush_bits(BitW *w) { if (w->nb) { w->p[w->n++] = (uint8_t)w->acc; w->acc = 0; w->nb = 0; } }

// raw DEFLATE of one member (final block); returns bytes written to out (capacity >= len + 64)
static size_t deflate_member(const uint8_t *m, int len, uint8_t *out, long *ntok_out, long *nmatch_out) {
    static Tok toks[MEMBER];
    int nt = 0;
    for (int q = 0; q < NQ; ++q) {
        const int q0 = q * PART, q1 = (q + 1) * PART < len ? (q + 1) * PART : len;
        if (q0 < q1) nt += parse_part(m, q0, q1, toks + nt);
    }
    uint32_t fl[286] = {0}, fd[30] = {0};
    long nmatch = 0;
    for (int i = 0; i < nt; ++i) {
        if (toks[i].len) { fl[257 + len_sym(toks[i].len)]++; fd[dist_sym(toks[i].dist)]++; ++nmatch; }
        else fl[toks[i].dist]++;
    }
    fl[256] = 1;
    *ntok_out += nt; *nmatch_out += nmatch;
    uint8_t ll[286], dl[30];
    uint16_t lc[286], dc[30];
    huff_lengths(fl, 286, 15, ll);
    huff_lengths(fd, 30, 15, dl);
    huff_codes(ll, 286, lc);
    huff_codes(dl, 30, dc);
    int hlit = 286, hdist = 30;
    while (hlit > 257 && ll[hlit - 1] == 0) --hlit;
    while (hdist > 1 && dl[hdist - 1] == 0) --hdist;
    // run-length code of the hlit + hdist lengths
    uint8_t seq[316], rl_sym[316], rl_extra[316];
    int nseq = 0, nrl = 0;
    for (int i = 0; i < hlit; ++i) seq[nseq++] = ll[i];
    for (int i = 0; i < hdist; ++i) seq[nseq++] = dl[i];
    for (int i = 0; i < nseq;) {
        int j = i;
        while (j < nseq && seq[j] == seq[i]) ++j;
        int run = j - i;
        if (seq[i] == 0) {
            while (run >= 11) { int r = run > 138 ? 138 : run; rl_sym[nrl] = 18; rl_extra[nrl++] = (uint8_t)(r - 11); run -= r; }
            if (run >= 3) { rl_sym[nrl] = 17; rl_extra[nrl++] = (uint8_t)(run - 3); run = 0; }
            while (run-- > 0) { rl_sym[nrl] = 0; rl_extra[nrl++] = 0; }
        } else {
            rl_sym[nrl] = seq[i]; rl_extra[nrl++] = 0; --run;
            while (run >= 3) { int r = run > 6 ? 6 : run; rl_sym[nrl] = 16; rl_extra[nrl++] = (uint8_t)(r - 3); run -= r; }
            while (run-- > 0) { rl_sym[nrl] = seq[i]; rl_extra[nrl++] = 0; }
        }
        i = j;
    }
    uint32_t fc[19] = {0};
    for (int i = 0; i < nrl; ++i) fc[rl_sym[i]]++;
    uint8_t cl[19];
    uint16_t cc[19];
    huff_lengths(fc, 19, 7, cl);
    huff_codes(cl, 19, cc);
    static const uint8_t ORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cl[ORD[hclen - 1]] == 0) --hclen;
    // size of the dynamic block in bits
    uint64_t bits = 3 + 5 + 5 + 4 + 3 * (uint64_t)hclen;
    for (int i = 0; i < nrl; ++i) bits += cl[rl_sym[i]] + (rl_sym[i] == 16 ? 2 : rl_sym[i] == 17 ? 3 : rl_sym[i] == 18 ? 7 : 0);
    for (int s = 0; s < 286; ++s) bits += (uint64_t)fl[s] * (ll[s] + (s >= 257 ? LEXTRA[s - 257] : 0));
    for (int s = 0; s < 30; ++s) bits += (uint64_t)fd[s] * (dl[s] + DEXTRA[s]);
    BitW w = {out, 0, 0, 0};
    if ((bits + 7) / 8 >= (uint64_t)len + 5 * ((len + 65534) / 65535)) {   // stored blocks
        int off = 0;
        do {
            const int k = len - off > 65535 ? 65535 : len - off;
            put(&w, off + k == len ? 1 : 0, 1); put(&w, 0, 2); flush_bits(&w);
            put(&w, (uint32_t)k, 16); put(&w, (uint32_t)(~k & 0xffff), 16);
            memcpy(w.p + w.n, m + off, (size_t)k);
            w.n += (size_t)k;
            off += k;
        } while (off < len);
        return w.n;
    }
    put(&w, 1, 1); put(&w, 2, 2);
    put(&w, (uint32_t)(hlit - 257), 5); put(&w, (uint32_t)(hdist - 1), 5); put(&w, (uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i) put(&w, cl[ORD[i]], 3);
    for (int i = 0; i < nrl; ++i) {
        put(&w, cc[rl_sym[i]], cl[rl_sym[i]]);
        if (rl_sym[i] == 16) put(&w, rl_extra[i], 2);
        else if (rl_sym[i] == 17) put(&w, rl_extra[i], 3);
        else if (rl_sym[i] == 18) put(&w, rl_extra[i], 7);
    }
    for (int i = 0; i < nt; ++i) {
        if (toks[i].len) {
            const int ls = len_sym(toks[i].len), ds = dist_sym(toks[i].dist);
            put(&w, lc[257 + ls], ll[257 + ls]);
            put(&w, (uint32_t)(toks[i].len - LBASE[ls]), LEXTRA[ls]);
            put(&w, dc[ds], dl[ds]);
            put(&w, (uint32_t)(toks[i].dist - DBASE[ds]), DEXTRA[ds]);
        } else {
            put(&w, lc[toks[i].dist], ll[toks[i].dist]);
        }
    }
    put(&w, lc[256], ll[256]);
    flush_bits(&w);
    return w.n;
}

static size_t zlib_size(const uint8_t *src, size_t n, int level, size_t member) {   // raw deflate, in members of `member` bytes (0 = one stream)
    size_t total = 0;
    for (size_t off = 0; off < n;) {
        const size_t k = member && n - off > member ? member : n - off;
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        size_t cap = deflateBound(&zs, (uLong)k) + 64;
        uint8_t *buf = malloc(cap);
        zs.next_in = (Bytef *)(src + off); zs.avail_in = (uInt)k; zs.next_out = buf; zs.avail_out = (uInt)cap;
        deflate(&zs, Z_FINISH);
        total += cap - zs.avail_out + (member ? 26 : 0);
        deflateEnd(&zs);
        free(buf);
        off += k;
    }
    return total;
}

int main(int argc, char **argv) {
    for (int a = 1; a < argc; ++a) {
        if (!strncmp(argv[a], "--d1=", 5)) { opt_dist1_min = atoi(argv[a] + 5); continue; }
        if (!strcmp(argv[a], "--nolazy")) { opt_lazy = 0; continue; }
        if (!strcmp(argv[a], "--one")) { opt_two = 0; continue; }
        if (!strncmp(argv[a], "--rep=", 6)) { opt_rep = atoi(argv[a] + 6); continue; }
        FILE *f = fopen(argv[a], "rb");
        if (!f) { perror(argv[a]); return 1; }
        fseek(f, 0, SEEK_END);
        size_t n = (size_t)ftell(f);
        fseek(f, 0, SEEK_SET);
        uint8_t *src = malloc(n + 8), *out = malloc(MEMBER + 1024), *chk = malloc(MEMBER + 8);
        if (fread(src, 1, n, f) != n) return 1;
        fclose(f);
        size_t total = 0;
        long ntok = 0, nmatch = 0, members = 0;
        for (size_t off = 0; off < n; off += MEMBER) {
            const int len = (int)(n - off > MEMBER ? MEMBER : n - off);
            const size_t k = deflate_member(src + off, len, out, &ntok, &nmatch);
            z_stream zs;   // zlib's inflate is the judge of the stream
            memset(&zs, 0, sizeof(zs));
            inflateInit2(&zs, -15);
            zs.next_in = out; zs.avail_in = (uInt)k; zs.next_out = chk; zs.avail_out = MEMBER + 8;
            const int rc = inflate(&zs, Z_FINISH);
            if (rc != Z_STREAM_END || zs.total_out != (uLong)len || memcmp(chk, src + off, (size_t)len)) {
                fprintf(stderr, "member at %zu: inflate rc %d, %lu of %d bytes\n", off, rc, zs.total_out, len);
                return 2;
            }
            inflateEnd(&zs);
            total += k + 26;   // + BGZF header (18) and trailer (8)
            ++members;
        }
        const size_t z1 = zlib_size(src, n, 1, 0), z5 = zlib_size(src, n, 5, 0), z6 = zlib_size(src, n, 6, 0), z5m = zlib_size(src, n, 5, MEMBER);
        printf("%s: %zu bytes, %ld members, tokens %ld (matches %ld)\n  model %zu (x%.3f)  zlib-5 %zu (x%.3f)  model/zlib-5 = %.4f\n"
               "  zlib-1 %zu (x%.3f)  zlib-6 %zu (x%.3f)  zlib-5 in 65280-byte members %zu (model/that = %.4f)\n",
               argv[a], n, members, ntok, nmatch, total, (double)n / total, z5, (double)n / z5, (double)total / z5, z1, (double)n / z1, z6,
               (double)n / z6, z5m, (double)total / z5m);
        free(src); free(out); free(chk);
    }
    return 0;
}
