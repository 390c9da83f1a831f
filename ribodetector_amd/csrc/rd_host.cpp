// rd_host.cpp - host-side FASTQ/FASTA ingest and label-partitioned output (librd_host.so). See
// include/ribodetector_amd_host.h for the reference interfaces this replaces.
#include <ctype.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ribodetector_amd_host.h"
#include "rd_inflate.h"
#include "rd_pgzip.h"

namespace {

thread_local char g_err[512] = "";
#define RDH_FAIL(...)                                  \
    do {                                               \
        snprintf(g_err, sizeof(g_err), __VA_ARGS__);   \
        return -1;                                     \
    } while (0)

bool ends_with(const std::string &s, const char *suf) {
    size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// Python str.rstrip()/strip() whitespace for the byte values that can occur in these files
inline bool is_ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || c == 0x1c || c == 0x1d || c == 0x1e || c == 0x1f || c == 0x85 || c == 0xa0; }

}  // namespace

// File reading (plain) or decompression (gzip) runs ahead of the parser in its own thread: blocks of text wait in a short
// queue. For plain files this takes the page-cache copy (~40 % of the reader's time) off the parsing thread.
struct rd_prefetch {
    static constexpr size_t BLOCK = 4u << 20, DEPTH = 3;
    std::function<long(uint8_t *, size_t)> source;   // bytes produced (0 = end, < 0 = error), message via source_err
    std::function<std::string()> source_err;
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::vector<uint8_t>> ready, spare;
    bool done = false, stop = false, failed = false;
    std::string err;

    rd_prefetch(std::function<long(uint8_t *, size_t)> src, std::function<std::string()> src_err)
        : source(std::move(src)), source_err(std::move(src_err)) {
        th = std::thread([this]() { run(); });
    }
    ~rd_prefetch() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
    void run() {
        for (;;) {
            std::vector<uint8_t> blk;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [this]() { return stop || ready.size() < DEPTH; });
                if (stop) return;
                if (!spare.empty()) {
                    blk.swap(spare.front());
                    spare.pop_front();
                }
            }
            blk.resize(BLOCK);
            const long got = source(blk.data(), BLOCK);
            std::lock_guard<std::mutex> lk(m);
            if (got <= 0) {
                failed = got < 0;
                if (failed) err = source_err();
                done = true;
                cv.notify_all();
                return;
            }
            blk.resize((size_t)got);
            ready.push_back(std::move(blk));
            cv.notify_all();
        }
    }
    // next block (empty = end of stream or error, see failed); `used` = the previous block, recycled
    std::vector<uint8_t> next(std::vector<uint8_t> &&used) {
        std::unique_lock<std::mutex> lk(m);
        if (used.capacity()) spare.push_back(std::move(used));
        cv.wait(lk, [this]() { return done || !ready.empty(); });
        std::vector<uint8_t> blk;
        if (!ready.empty()) {
            blk.swap(ready.front());
            ready.pop_front();
            cv.notify_all();
        }
        return blk;
    }
};

// a reader whose bytes are FED by the caller (rd_reader_open_feed): the decompressed text of gzip members inflated elsewhere - on the
// GPU, librd_hip.so rd_gz_inflate_members. rd_reader_feed hands a span over and returns when the reader (the thread in rd_reader_next)
// has copied all of it into its window (a synchronous hand-over: the caller may reuse the buffer at once, no queue of buffers to own).
struct rd_feed {
    std::mutex m;
    std::condition_variable cv;
    const uint8_t *p = nullptr;
    size_t left = 0;
    bool eof = false, aborted = false;
    std::string err;
};

struct rd_reader {
    FILE *fp = nullptr;
    rd_feed *feed = nullptr;
    rdz::GzipStream *gz = nullptr;   // set when the file starts with the gzip magic; plain bytes otherwise
    rdz::ParallelGzip *pgz = nullptr;   // ... and is large: sections of the DEFLATE stream decoded in parallel (rd_pgzip.h)
    rd_prefetch *pf = nullptr;
    std::vector<uint8_t> blk;        // block being copied into the window, from blk_off
    size_t blk_off = 0;
    bool failed = false;             // decompression error: message in err
    std::string err;
    int fasta;
    std::vector<uint8_t> in;     // raw input window
    const uint8_t *map = nullptr;   // plain regular file: the window IS the mapped file (no page-cache copy, no window copy, no
    size_t map_len = 0;             // prefetch thread); pos / end are file offsets then and nothing is ever filled in
    const uint8_t *wdata() const { return map ? map : in.data(); }
    size_t pos, end;             // unconsumed bytes are window[pos, end)
    bool eof;
    bool flush_empty_tail = false;   // byte-range reader whose range ends before the file does: see rd_reader_open_range
    std::string pending_header;  // FASTA: header of the record being assembled ('' until the first '>' line, like the reference)
    std::string pending_seq;
    size_t scan_next;            // window offset just past the lines returned by scan_lines

    // Offsets [lb[k], le[k]) of the next `want` lines (terminators excluded) without consuming them; the offsets are
    // relative to wdata() and stay valid until the next fill(). Returns how many lines exist (fewer only at end of file;
    // the last line of a file may lack its '\n').
    int scan_lines(int want, size_t *lb, size_t *le) {
        for (;;) {
            size_t p = pos;
            int k = 0;
            while (k < want) {
                const uint8_t *nl = (const uint8_t *)memchr(wdata() + p, '\n', end - p);
                if (!nl) break;
                lb[k] = p;
                le[k] = (size_t)(nl - wdata());
                p = le[k] + 1;
                ++k;
            }
            if (k == want) { scan_next = p; return k; }
            if (eof) {   // nothing more will arrive: a last line without terminator still counts
                if (p < end) { lb[k] = p; le[k] = end; p = end; ++k; }
                scan_next = p;
                return k;
            }
            fill();      // may compact the window: rescan from pos either way
        }
    }

    bool fill() {   // make room and read more; false when nothing more arrives
        if (eof) return false;
        if (pos > 0) {
            memmove(in.data(), in.data() + pos, end - pos);
            end -= pos;
            pos = 0;
        }
        if (end == in.size()) in.resize(in.size() * 2);
        long got;
        if (feed && !pf) {
            // fed text goes straight from the feeder's buffer into the window (one copy, no thread in between); rd_reader_feed
            // returns when its span has been taken. The stream's end - and a feeder's error - come after everything that was fed.
            rd_feed *f = feed;
            std::unique_lock<std::mutex> lk(f->m);
            f->cv.wait(lk, [f]() { return f->left > 0 || f->eof; });
            if (f->left == 0) {
                if (!f->err.empty()) {
                    failed = true;
                    err = f->err;
                }
                got = 0;
            } else {
                const size_t k = std::min(in.size() - end, f->left);
                memcpy(in.data() + end, f->p, k);
                f->p += k;
                f->left -= k;
                if (f->left == 0) f->cv.notify_all();
                got = (long)k;
            }
        } else if (pf) {
            if (blk_off == blk.size()) {
                blk = pf->next(std::move(blk));
                blk_off = 0;
            }
            got = (long)std::min(in.size() - end, blk.size() - blk_off);
            if (got > 0) memcpy(in.data() + end, blk.data() + blk_off, (size_t)got);
            blk_off += (size_t)got;
            if (got == 0 && pf->failed) {   // set before the empty block was handed over
                failed = true;
                err = pf->err;
            }
        } else {
            got = (long)fread(in.data() + end, 1, in.size() - end, fp);
        }
        if (got <= 0) {
            eof = true;
            return false;
        }
        end += (size_t)got;
        return true;
    }
};

// gzip output is written as a sequence of independent gzip members (RFC 1952 allows concatenation; zcat, Python's gzip and
// the reference's own readers accept it), each compressed at level 5 by its own thread: the reference's single-threaded
// gzip.open(..., compresslevel=5) is the slowest stage of its pipeline ("2 times slower to write gz files").
struct rd_writer {
    int fd;
    int64_t off;                    // bytes written so far (= file offset: the file is only appended to)
    bool gz;
    int threads;
    std::vector<uint8_t> pending;   // gz only: selected record bytes not yet compressed
    std::vector<void *> comp;       // gz only: one libdeflate compressor per worker slot (empty: zlib)
    bool bgzf = false;              // members written by rd_writer_write_members (BGZF blocks made on the GPU)
    bool eof_marker = true;         // ... then the file ends with BGZF's empty block (rd_writer_set_eof_marker)
};

// libdeflate (a system library of this image, libdeflate0 1.10) compresses level 5 ~2.5x faster than zlib at the same
// ratio; it is bound at run time so that the library still loads - and falls back to zlib - where it is absent.
// RD_HOST_ZLIB=1 forces the zlib path.
struct LibDeflate {
    void *(*alloc)(int) = nullptr;
    void (*release)(void *) = nullptr;
    size_t (*gzip_compress)(void *, const void *, size_t, void *, size_t) = nullptr;
    size_t (*gzip_bound)(void *, size_t) = nullptr;
    size_t (*deflate_compress)(void *, const void *, size_t, void *, size_t) = nullptr;
    size_t (*deflate_bound)(void *, size_t) = nullptr;
    bool ok = false;
};

const LibDeflate &libdeflate() {
    static const LibDeflate L = []() {
        LibDeflate l;
        const char *force = getenv("RD_HOST_ZLIB");
        if (force && force[0] == '1') return l;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return l;
        l.alloc = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor");
        l.release = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
        l.gzip_compress = (size_t(*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_gzip_compress");
        l.gzip_bound = (size_t(*)(void *, size_t))dlsym(h, "libdeflate_gzip_compress_bound");
        l.deflate_compress = (size_t(*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_deflate_compress");
        l.deflate_bound = (size_t(*)(void *, size_t))dlsym(h, "libdeflate_deflate_compress_bound");
        l.ok = l.alloc && l.release && l.gzip_compress && l.gzip_bound && l.deflate_compress && l.deflate_bound;
        return l;
    }();
    return L;
}

bool pwrite_all(int fd, const uint8_t *p, size_t len, int64_t off) {
    while (len) {
        const ssize_t k = pwrite(fd, p, len, (off_t)off);
        if (k <= 0) return false;
        p += k;
        len -= (size_t)k;
        off += k;
    }
    return true;
}

int g_threads = 0;                  // 0 = auto
int g_gz_threads = -1;              // decoder threads per .gz input: -1 = auto, 0 = sequential decoder

int usable_threads() {
    if (g_threads > 0) return g_threads;
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");   // containers: quota / period
    if (f) {
        char q[64];
        long long period = 0;
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            long long quota = atoll(q);
            int c = (int)std::max<long long>(1, quota / period);
            n = std::min(n, c);
        }
        fclose(f);
    }
    return std::min(n, 32);
}

constexpr size_t GZ_BLOCK = 4u << 20;   // uncompressed bytes per gzip member

// One gzip member: header with an extra field that holds the member's own size ('R','D', 4 bytes: size - 1, the 32-bit analogue of
// BGZF's 'B','C' subfield - readers that do not know it skip it, RFC 1952 2.3.1.1), raw deflate at level 5, CRC-32 + ISIZE. With the
// sizes in the headers a reader can walk the members without decoding them and decode them in parallel (rd_pgzip.h, indexed mode).
constexpr size_t GZ_HDR = 20;
bool gz_member(const uint8_t *src, size_t len, std::vector<uint8_t> &out, void *comp) {
    size_t body = 0;
    if (comp) {
        const LibDeflate &L = libdeflate();
        out.resize(GZ_HDR + L.deflate_bound(comp, len) + 8);
        body = L.deflate_compress(comp, src, len, out.data() + GZ_HDR, out.size() - GZ_HDR - 8);
    }
    if (!body) {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, 5, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        out.resize(GZ_HDR + deflateBound(&zs, (uLong)len) + 64 + 8);
        zs.next_in = const_cast<Bytef *>(src);
        zs.avail_in = (uInt)len;
        zs.next_out = out.data() + GZ_HDR;
        zs.avail_out = (uInt)(out.size() - GZ_HDR - 8);
        const int rc = deflate(&zs, Z_FINISH);
        body = (out.size() - GZ_HDR - 8) - zs.avail_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) return false;
    }
    const size_t total = GZ_HDR + body + 8;
    out.resize(total);
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 8, 0, 'R', 'D', 4, 0};
    memcpy(out.data(), hdr, 16);
    const uint32_t sz = (uint32_t)(total - 1), crc = rdz::crc32_update(0, src, len), isize = (uint32_t)len;
    memcpy(out.data() + 16, &sz, 4);                    // little-endian host
    memcpy(out.data() + GZ_HDR + body, &crc, 4);
    memcpy(out.data() + GZ_HDR + body + 4, &isize, 4);
    return true;
}

// compress w->pending[0, upto) as members of GZ_BLOCK bytes and write them in order: the workers take blocks from a shared
// counter (no barrier between batches), the calling thread writes each member as soon as it and its predecessors are done
int gz_flush(rd_writer *w, size_t upto) {
    const size_t nblk = (upto + GZ_BLOCK - 1) / GZ_BLOCK;
    std::vector<std::vector<uint8_t>> outs(nblk);
    std::vector<signed char> state(nblk, 0);   // 0 = pending, 1 = compressed, -1 = failed (guarded by m)
    std::mutex m;
    std::condition_variable cv;
    size_t next = 0;
    const size_t nthreads = std::min<size_t>((size_t)std::max(1, w->threads), nblk);
    std::vector<std::thread> th;
    for (size_t t = 0; t < nthreads; ++t) {
        void *comp = t < w->comp.size() ? w->comp[t] : nullptr;
        th.emplace_back([&, comp]() {
            for (;;) {
                size_t b;
                {
                    std::lock_guard<std::mutex> lk(m);
                    if (next >= nblk) return;
                    b = next++;
                }
                const size_t off = b * GZ_BLOCK, len = std::min(GZ_BLOCK, upto - off);
                const bool ok = gz_member(w->pending.data() + off, len, outs[b], comp);
                {
                    std::lock_guard<std::mutex> lk(m);
                    state[b] = ok ? 1 : -1;
                }
                cv.notify_all();
            }
        });
    }
    int rc = 0;
    for (size_t b = 0; b < nblk; ++b) {
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&]() { return state[b] != 0; });
            if (state[b] < 0) rc = -1;
        }
        if (rc == 0) {
            if (pwrite_all(w->fd, outs[b].data(), outs[b].size(), w->off)) w->off += (int64_t)outs[b].size();
            else rc = -1;
        }
        std::vector<uint8_t>().swap(outs[b]);
    }
    for (auto &t : th) t.join();
    w->pending.erase(w->pending.begin(), w->pending.begin() + (ptrdiff_t)upto);
    return rc;
}

extern "C" {

const char *rd_host_last_error(void) { return g_err; }

}  // extern "C"
namespace {
// A plain regular file is parsed where the page cache holds it: mapped read-only, lines scanned and records copied out of the
// mapping - instead of fread into blocks (one copy), blocks into the window (another), and a thread for the first of the two.
// RD_READER_MMAP=0 keeps the buffered reader (a file that is truncated while it is read ends a mapped reader with SIGBUS).
bool map_plain_file(FILE *fp, rd_reader *r, int64_t start, int64_t end_or_neg) {
    const char *e = getenv("RD_READER_MMAP");
    if (e && e[0] == '0') return false;
    struct stat st;
    if (fstat(fileno(fp), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) return false;
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(fp), 0);
    if (m == MAP_FAILED) return false;
    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    r->map = (const uint8_t *)m;
    r->map_len = (size_t)st.st_size;
    const int64_t b = std::min<int64_t>(std::max<int64_t>(start, 0), st.st_size);
    const int64_t t = end_or_neg < 0 ? (int64_t)st.st_size : std::min<int64_t>(std::max<int64_t>(end_or_neg, b), st.st_size);
    r->pos = (size_t)b;
    r->end = (size_t)t;
    r->eof = true;              // everything there will ever be is in the window
    return true;
}
}  // namespace
extern "C" {

int rd_reader_open(const char *path, int format, rd_reader **out) {
    if (!path || !out) RDH_FAIL("rd_reader_open: null argument");
    std::string p(path), stem = p;
    if (ends_with(p, ".gz")) stem = p.substr(0, p.size() - 3);
    else if (ends_with(p, ".gzip")) stem = p.substr(0, p.size() - 5);
    if (format < 0) {
        if (ends_with(stem, ".fq") || ends_with(stem, ".fastq")) format = 0;
        else if (ends_with(stem, ".fasta") || ends_with(stem, ".fa") || ends_with(stem, ".fna") || ends_with(stem, ".fas")) format = 1;
        else RDH_FAIL("Unknown extension of %s. Only fastq and fasta sequence formats are supported.", path);
    }
    FILE *fp = fopen(path, "rb");
    if (!fp) RDH_FAIL("cannot open %s", path);
    setvbuf(fp, nullptr, _IONBF, 0);   // both paths read in MiB-sized pieces themselves
    rd_reader *r = new rd_reader();
    r->fp = fp;
    r->fasta = format;
    r->in.resize(8 << 20);
    r->pos = r->end = 0;
    // gzip is recognised by its magic, like zlib's gzopen (a superset of the reference's by-extension rule)
    uint8_t magic[2];
    const size_t got = fread(magic, 1, 2, fp);
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        // large files: the parallel decoder (falls back to the sequential one by itself for anything it is not made for).
        // Threads per file: rd_host_set_gz_threads (the CLI divides -t among its input files), else half of the usable cores - 1,
        // <= 8 (paired input decodes two files at once); RD_GZ_THREADS overrides both, 0 = the sequential decoder.
        int pt = g_gz_threads >= 0 ? g_gz_threads : std::min(8, std::max(2, usable_threads() / 2 - 1));
        if (const char *e = getenv("RD_GZ_THREADS")) pt = atoi(e);
        struct stat st;
        long long min_size = 16ll << 20, section = 2ll << 20;   // (RD_GZ_PARALLEL_MIN / RD_GZ_SECTION: knobs of the tests)
        if (const char *e = getenv("RD_GZ_PARALLEL_MIN")) min_size = atoll(e);
        if (const char *e = getenv("RD_GZ_SECTION")) section = atoll(e);
        const bool large = fstat(fileno(fp), &st) == 0 && (long long)st.st_size >= min_size;
        if (pt >= 2 && large && (usable_threads() >= 4 || getenv("RD_GZ_THREADS"))) {
            rdz::ParallelGzip *pg = r->pgz = new rdz::ParallelGzip(fp, pt, (size_t)std::max(4096ll, section));
            r->pf = new rd_prefetch([pg](uint8_t *dst, size_t cap) { return pg->read(dst, cap); }, [pg]() { return pg->err; });
        } else {
            rdz::GzipStream *gz = r->gz = new rdz::GzipStream(fp, magic, 2);
            r->pf = new rd_prefetch([gz](uint8_t *dst, size_t cap) { return gz->read(dst, cap); }, [gz]() { return gz->err; });
        }
    } else if (map_plain_file(fp, r, 0, -1)) {
        r->scan_next = 0;
        *out = r;
        return 0;
    } else {
        memcpy(r->in.data(), magic, got);
        r->end = got;
        r->pf = new rd_prefetch(
            [fp](uint8_t *dst, size_t cap) {
                const size_t k = fread(dst, 1, cap, fp);
                return (k == 0 && ferror(fp)) ? -1L : (long)k;
            },
            []() { return std::string("read error"); });
    }
    r->eof = false;
    r->scan_next = 0;
    *out = r;
    return 0;
}

// ---- byte-range readers for the multi-rank CLI: every rank parses only its own part of a PLAIN file -------------------------
namespace {

struct RangeFile {
    int fd = -1;
    int64_t size = 0;
    std::vector<uint8_t> buf;
    int64_t base = 0, len = 0;   // buf holds file bytes [base, base + len)
    bool open_path(const char *path) {
        fd = open(path, O_RDONLY);
        if (fd < 0) return false;
        size = (int64_t)lseek(fd, 0, SEEK_END);
        buf.resize(4u << 20);
        return true;
    }
    ~RangeFile() { if (fd >= 0) close(fd); }
    bool load(int64_t at) {   // buf <- file bytes from `at`
        base = at;
        len = 0;
        while (len < (int64_t)buf.size() && base + len < size) {
            const ssize_t k = pread(fd, buf.data() + len, buf.size() - (size_t)len, (off_t)(base + len));
            if (k < 0) return false;
            if (k == 0) break;
            len += k;
        }
        return true;
    }
    // offset of the first byte after the next '\n' at or after `at` (= start of the next line), or size when there is none
    int64_t next_line(int64_t at) {
        for (;;) {
            if (at >= size) return size;
            if (at < base || at >= base + len) { if (!load(at)) return -1; }
            const uint8_t *p = buf.data() + (at - base);
            const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(base + len - at));
            if (nl) return base + (nl - buf.data()) + 1;
            at = base + len;
        }
    }
    int byte_at(int64_t at) {   // -1 past the end
        if (at >= size) return -1;
        if (at < base || at >= base + len) { if (!load(at)) return -2; }
        return buf[(size_t)(at - base)];
    }
};

int range_format(const char *path, int format) {
    std::string p(path);
    if (format >= 0) return format;
    if (ends_with(p, ".fq") || ends_with(p, ".fastq")) return 0;
    if (ends_with(p, ".fasta") || ends_with(p, ".fa") || ends_with(p, ".fna") || ends_with(p, ".fas")) return 1;
    return -1;
}

}  // namespace

int rd_host_file_info(const char *path, int64_t *size, int32_t *is_gzip) {
    if (!path || !size || !is_gzip) RDH_FAIL("rd_host_file_info: null argument");
    RangeFile f;
    if (!f.open_path(path)) RDH_FAIL("cannot open %s", path);
    *size = f.size;
    *is_gzip = (f.byte_at(0) == 0x1f && f.byte_at(1) == 0x8b) ? 1 : 0;
    return 0;
}

// First record boundary at or after byte `pos` of a plain file. FASTQ (4-line records, fastx_parser.py:18-37): a line that
// starts with '@' whose second-next line starts with '+' - a quality line may start with '@' too, but then the second-next
// line is a sequence line, which never starts with '+'. FASTA: a line that starts with '>'. Returns the file size when no
// record starts after pos.
int rd_host_find_record_start(const char *path, int format, int64_t pos, int64_t *out) {
    if (!path || !out) RDH_FAIL("rd_host_find_record_start: null argument");
    format = range_format(path, format);
    if (format < 0) RDH_FAIL("Unknown extension of %s. Only fastq and fasta sequence formats are supported.", path);
    RangeFile f;
    if (!f.open_path(path)) RDH_FAIL("cannot open %s", path);
    if (pos <= 0) { *out = 0; return 0; }
    if (pos >= f.size) { *out = f.size; return 0; }
    int64_t line = (f.byte_at(pos - 1) == '\n') ? pos : f.next_line(pos);
    while (line >= 0 && line < f.size) {
        const int c = f.byte_at(line);
        if (format == 1) {
            if (c == '>') { *out = line; return 0; }
        } else if (c == '@') {
            const int64_t l1 = f.next_line(line), l2 = l1 >= 0 ? f.next_line(l1) : -1;
            if (l2 < 0) break;
            if (l2 < f.size && f.byte_at(l2) == '+') { *out = line; return 0; }
        }
        line = f.next_line(line);
    }
    if (line < 0) RDH_FAIL("read error in %s", path);
    *out = f.size;
    return 0;
}

// Records that start in [start, end) (start = a record boundary): FASTQ = lines / 4, FASTA = lines starting with '>'.
int rd_host_count_records(const char *path, int format, int64_t start, int64_t end, int64_t *n_out) {
    if (!path || !n_out) RDH_FAIL("rd_host_count_records: null argument");
    format = range_format(path, format);
    if (format < 0) RDH_FAIL("Unknown extension of %s. Only fastq and fasta sequence formats are supported.", path);
    RangeFile f;
    if (!f.open_path(path)) RDH_FAIL("cannot open %s", path);
    if (end > f.size) end = f.size;
    // FASTQ: the reader takes four lines at a time and tolerates a blank remainder of fewer than four, so the records are
    // floor(lines / 4) of ALL the lines - a last record whose sequence and quality lines are empty ("@b\n\n+\n\n") is a record
    // (round 2 stripped trailing whitespace first and lost it: advisor finding)
    int64_t lines = 0, headers = 0, at = start;
    bool line_start = true;
    while (at < end) {
        if (!f.load(at)) RDH_FAIL("read error in %s", path);
        const int64_t m = std::min(f.len, end - at);
        if (m <= 0) break;
        const uint8_t *p = f.buf.data(), *e = p + m;
        while (p < e) {
            if (line_start && *p == '>') ++headers;
            const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(e - p));
            if (!nl) { line_start = false; break; }
            ++lines;
            line_start = true;
            p = nl + 1;
        }
        at += m;
    }
    if (!line_start) ++lines;   // last line without terminator
    *n_out = format == 1 ? headers : lines / 4;
    return 0;
}

// Byte offset of the record boundary `k` records after the boundary `start` (the file size if fewer records follow).
int rd_host_skip_records(const char *path, int format, int64_t start, int64_t k, int64_t *out) {
    if (!path || !out || k < 0) RDH_FAIL("rd_host_skip_records: bad argument");
    format = range_format(path, format);
    if (format < 0) RDH_FAIL("Unknown extension of %s. Only fastq and fasta sequence formats are supported.", path);
    RangeFile f;
    if (!f.open_path(path)) RDH_FAIL("cannot open %s", path);
    int64_t at = start;
    if (format == 0) {
        for (int64_t i = 0; i < 4 * k && at < f.size; ++i) {
            at = f.next_line(at);
            if (at < 0) RDH_FAIL("read error in %s", path);
        }
    } else {
        for (int64_t i = 0; i < k && at < f.size; ++i) {   // from one header to the next
            do {
                at = f.next_line(at);
                if (at < 0) RDH_FAIL("read error in %s", path);
            } while (at < f.size && f.byte_at(at) != '>');
        }
    }
    *out = at;
    return 0;
}

// Reader over the bytes [start, end) of a plain (not gzip) file; both offsets must be record boundaries
// (rd_host_find_record_start / rd_host_skip_records), so the range parses exactly like a file of its own.
int rd_reader_open_range(const char *path, int format, int64_t start, int64_t end, rd_reader **out) {
    if (!path || !out || start < 0 || end < start) RDH_FAIL("rd_reader_open_range: bad argument");
    format = range_format(path, format);
    if (format < 0) RDH_FAIL("Unknown extension of %s. Only fastq and fasta sequence formats are supported.", path);
    FILE *fp = fopen(path, "rb");
    if (!fp) RDH_FAIL("cannot open %s", path);
    setvbuf(fp, nullptr, _IONBF, 0);
    uint8_t magic[2];
    if (fread(magic, 1, 2, fp) == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        fclose(fp);
        RDH_FAIL("rd_reader_open_range: %s is gzip-compressed; byte ranges need a plain file", path);
    }
    if (fseeko(fp, (off_t)start, SEEK_SET) != 0) {
        fclose(fp);
        RDH_FAIL("cannot seek in %s", path);
    }
    rd_reader *r = new rd_reader();
    r->fp = fp;
    r->fasta = format;
    r->in.resize(8 << 20);
    r->pos = r->end = 0;
    {
        struct stat sb;
        r->flush_empty_tail = fstat(fileno(fp), &sb) == 0 && end < (int64_t)sb.st_size;
    }
    if (map_plain_file(fp, r, start, end)) {
        r->scan_next = 0;
        *out = r;
        return 0;
    }
    auto left = std::make_shared<int64_t>(end - start);
    r->pf = new rd_prefetch(
        [fp, left](uint8_t *dst, size_t cap) {
            if (*left <= 0) return 0L;
            const size_t k = fread(dst, 1, (size_t)std::min<int64_t>((int64_t)cap, *left), fp);
            *left -= (int64_t)k;
            return (k == 0 && ferror(fp)) ? -1L : (long)k;
        },
        []() { return std::string("read error"); });
    r->eof = false;
    r->scan_next = 0;
    // FASTA: the reference yields a record at the NEXT header, and at the end of the FILE only if its sequence is not empty
    // (fastx_parser.py:39-55). A range that ends before the file does is followed by a header (the next rank's first record), so
    // its last record is yielded even with an empty sequence - only the true end of the file drops it (advisor finding, round 2)
    {
        struct stat sb;
        r->flush_empty_tail = fstat(fileno(fp), &sb) == 0 && end < (int64_t)sb.st_size;
    }
    *out = r;
    return 0;
}

void rd_reader_close(rd_reader *r) {
    if (!r) return;
    if (r->feed) {   // a feeder that still waits must not wait forever, and the prefetch thread must not wait for a feeder that is gone
        std::lock_guard<std::mutex> lk(r->feed->m);
        r->feed->aborted = true;
        r->feed->eof = true;
        r->feed->cv.notify_all();
    }
    delete r->pf;   // joins the decompression thread first
    delete r->gz;
    delete r->pgz;
    if (r->map) munmap(const_cast<uint8_t *>(r->map), r->map_len);
    if (r->fp) fclose(r->fp);
    delete r->feed;
    delete r;
}

int rd_reader_open_feed(int format, rd_reader **out) {
    if (!out || (format != 0 && format != 1)) RDH_FAIL("rd_reader_open_feed: format must be 0 (FASTQ) or 1 (FASTA)");
    rd_reader *r = new rd_reader();
    r->fasta = format;
    r->in.resize(8 << 20);
    r->pos = r->end = 0;
    r->feed = new rd_feed();
    r->eof = false;
    r->scan_next = 0;
    *out = r;
    return 0;
}

// FASTA, a feed reader over a SHARE of a stream that goes on behind it (a rank's members of a BGZF file): the share's last record is
// followed by another rank's header, so it is yielded even with an empty sequence (rd_reader_open_range does the same for byte ranges)
int rd_reader_set_flush_empty_tail(rd_reader *r, int on) {
    if (!r) RDH_FAIL("rd_reader_set_flush_empty_tail: null reader");
    r->flush_empty_tail = on != 0;
    return 0;
}

int rd_reader_feed(rd_reader *r, const uint8_t *bytes, int64_t len) {
    if (!r || !r->feed || len < 0 || (!bytes && len)) RDH_FAIL("rd_reader_feed: not a feed reader, or bad argument");
    rd_feed *f = r->feed;
    std::unique_lock<std::mutex> lk(f->m);
    if (f->eof) RDH_FAIL("rd_reader_feed: the stream was closed");
    if (len == 0) return 0;
    f->p = bytes;
    f->left = (size_t)len;
    f->cv.notify_all();
    f->cv.wait(lk, [f]() { return f->left == 0 || f->aborted; });
    if (f->aborted) RDH_FAIL("rd_reader_feed: the reader was closed");
    return 0;
}

// the reader is about to be closed before its stream ended: wake a feeder that waits in rd_reader_feed (it returns -1) and make every
// later feed call fail, WITHOUT freeing anything - the caller joins its feeder threads, then calls rd_reader_close
int rd_reader_feed_abort(rd_reader *r) {
    if (!r || !r->feed) RDH_FAIL("rd_reader_feed_abort: not a feed reader");
    rd_feed *f = r->feed;
    std::lock_guard<std::mutex> lk(f->m);
    f->aborted = true;
    f->eof = true;
    f->cv.notify_all();
    return 0;
}

int rd_reader_feed_end(rd_reader *r, const char *error) {
    if (!r || !r->feed) RDH_FAIL("rd_reader_feed_end: not a feed reader");
    rd_feed *f = r->feed;
    std::lock_guard<std::mutex> lk(f->m);
    if (error && error[0]) f->err = error;    // the records parsed so far are delivered, then rd_reader_next fails with this text
    f->eof = true;
    f->cv.notify_all();
    return 0;
}

// Walk gzip members that say how long they are: BGZF ('B','C' subfield: 16-bit size - 1) and this build's host writer ('R','D':
// 32-bit size - 1). No decoding: header, size subfield, trailer. Fills one entry per member; stops at the end of the bytes, at an
// incomplete member (the caller reads more and walks on from *consumed), at `cap` entries, or at a member without a size subfield
// (*consumed then points at it and the return value is 1: the caller hands the rest to the streaming decoder). Empty members (BGZF's
// end-of-file marker) are skipped.
int rd_host_gz_index(const uint8_t *buf, int64_t len, int64_t in_base, int64_t out_base, rd_host_gz_member *out, int64_t cap, int64_t *n_out,
                     int64_t *consumed, int64_t *out_bytes) {
    if (!buf || !out || !n_out || !consumed || !out_bytes || len < 0 || cap < 0) RDH_FAIL("rd_host_gz_index: bad argument");
    int64_t p = 0, n = 0, ob = 0;
    int rc = 0;
    while (p < len && n < cap) {
        if (len - p < 18) break;
        const uint8_t *h = buf + p;
        if (h[0] != 0x1f || h[1] != 0x8b) { snprintf(g_err, sizeof(g_err), "rd_host_gz_index: not a gzip member at byte %lld", (long long)(in_base + p)); return -1; }
        if (h[2] != 8 || !(h[3] & 4) || (h[3] & ~4)) { rc = 1; break; }   // only FEXTRA: anything else is for the streaming decoder
        const size_t xlen = h[10] | ((size_t)h[11] << 8);
        if ((int64_t)(12 + xlen) > len - p) break;
        int64_t msize = 0;
        for (size_t q = 12; q + 4 <= 12 + xlen;) {
            const size_t sl = h[q + 2] | ((size_t)h[q + 3] << 8);
            if (q + 4 + sl > 12 + xlen) break;
            if (h[q] == 'B' && h[q + 1] == 'C' && sl == 2) msize = (int64_t)(h[q + 4] | ((uint32_t)h[q + 5] << 8)) + 1;
            if (h[q] == 'R' && h[q + 1] == 'D' && sl == 4)
                msize = (int64_t)(h[q + 4] | ((uint32_t)h[q + 5] << 8) | ((uint32_t)h[q + 6] << 16) | ((uint32_t)h[q + 7] << 24)) + 1;
            q += 4 + sl;
        }
        if (msize == 0) { rc = 1; break; }
        if (msize < (int64_t)(12 + xlen + 8)) { snprintf(g_err, sizeof(g_err), "rd_host_gz_index: member size %lld too small at byte %lld", (long long)msize, (long long)(in_base + p)); return -1; }
        if (msize > len - p) break;                                        // incomplete: more bytes needed
        const uint8_t *t = h + msize - 8;
        const uint32_t isize = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
        // a member of more than 64 MiB (either side) is not for the member-per-wave decoder: the caller's batch buffers are sized for
        // BGZF blocks / this writer's 4 MiB members, and a crafted size field must not drive its allocations - streaming decoder
        if (isize > (64u << 20) || msize - 12 - (int64_t)xlen - 8 > (64 << 20)) { rc = 1; break; }
        if (isize) {
            out[n].in_off = in_base + p + 12 + (int64_t)xlen;
            out[n].out_off = out_base + ob;
            out[n].in_len = (int32_t)(msize - 12 - (int64_t)xlen - 8);
            out[n].out_len = (int32_t)isize;
            ++n;
            ob += isize;
        }
        p += msize;
    }
    *n_out = n;
    *consumed = p;
    *out_bytes = ob;
    return rc;
}

int rd_reader_next(rd_reader *r, int64_t max_records, uint8_t *buf, int64_t buf_cap, int64_t *rec_start, int64_t *seq_off,
                   int32_t *seq_len, int64_t *n_out, int64_t *nbytes_out) {
    if (!r || !buf || !rec_start || !seq_off || !seq_len || !n_out || !nbytes_out) RDH_FAIL("rd_reader_next: null argument");
    int64_t n = 0, w = 0, need_hint = 0;
    bool at_eof = false;
    if (!r->fasta) {
        // FASTQ: 4 lines per record, each rstrip()-ed (fastx_parser.py:18-37)
        while (n < max_records) {
            size_t lb[4], le[4];
            const int got = r->scan_lines(4, lb, le);
            if (r->failed) {                 // a damaged stream: what was parsed before the damage is delivered, the next call fails
                if (n > 0) break;
                RDH_FAIL("%s", r->err.c_str());
            }
            if (got == 0) { at_eof = true; break; }   // clean end of file
            if (got < 4) {
                bool blank = true;   // tolerate trailing blank lines only
                for (int k = 0; k < got; ++k)
                    for (size_t x = lb[k]; x < le[k]; ++x) blank = blank && is_ws(r->wdata()[x]);
                if (blank) { r->pos = r->scan_next; at_eof = true; break; }
                RDH_FAIL("truncated FASTQ record at end of file (number of lines is not a multiple of 4)");
            }
            const uint8_t *base = r->wdata();
            for (int k = 0; k < 4; ++k)
                while (le[k] > lb[k] && is_ws(base[le[k] - 1])) --le[k];
            if (le[0] == lb[0] || base[lb[0]] != '@') RDH_FAIL("FASTQ record does not start with '@'");
            int64_t need = 4;
            for (int k = 0; k < 4; ++k) need += (int64_t)(le[k] - lb[k]);
            if (w + need > buf_cap) {        // does not fit: the record stays in the window for the next call
                if (n == 0) need_hint = need;   // not even alone: tell the caller how much room this record takes
                break;
            }
            rec_start[n] = w;
            for (int k = 0; k < 4; ++k) {
                const size_t len = le[k] - lb[k];
                if (k == 1) { seq_off[n] = w; seq_len[n] = (int32_t)len; }
                memcpy(buf + w, base + lb[k], len);
                w += (int64_t)len;
                buf[w++] = '\n';
            }
            r->pos = r->scan_next;
            ++n;
        }
    } else {
        // FASTA: strip() every line, skip blanks, '>' starts a record, sequence lines joined + upper-cased
        // (fastx_parser.py:39-55; like the reference, a record is yielded at the next header if header != '', and at end of
        // file if seq != '')
        auto emit = [&](void) -> int {   // 1 = emitted, 0 = does not fit, -1 = error
            int64_t need = (int64_t)r->pending_header.size() + 1 + (int64_t)r->pending_seq.size() + 1;
            if (n >= max_records) return 0;
            if (w + need > buf_cap) {
                if (n == 0) need_hint = need;
                return 0;
            }
            rec_start[n] = w;
            memcpy(buf + w, r->pending_header.data(), r->pending_header.size());
            w += (int64_t)r->pending_header.size();
            buf[w++] = '\n';
            seq_off[n] = w;
            seq_len[n] = (int32_t)r->pending_seq.size();
            memcpy(buf + w, r->pending_seq.data(), r->pending_seq.size());
            w += (int64_t)r->pending_seq.size();
            buf[w++] = '\n';
            ++n;
            return 1;
        };
        while (n < max_records) {
            size_t lb, le;
            const int got = r->scan_lines(1, &lb, &le);
            if (r->failed) {                 // a damaged stream: what was parsed before the damage is delivered, the next call fails
                if (n > 0) break;
                RDH_FAIL("%s", r->err.c_str());
            }
            if (got == 0) {   // end of input
                if (!r->pending_seq.empty() || (r->flush_empty_tail && !r->pending_header.empty())) {
                    int rc = emit();
                    if (rc < 0) return -1;
                    if (rc == 0) break;
                    r->pending_seq.clear();
                    r->pending_header.clear();
                }
                at_eof = true;
                break;
            }
            const uint8_t *base = r->wdata();
            while (le > lb && is_ws(base[le - 1])) --le;
            while (lb < le && is_ws(base[lb])) ++lb;
            if (lb == le) { r->pos = r->scan_next; continue; }
            if (base[lb] == '>') {
                if (!r->pending_header.empty()) {
                    int rc = emit();
                    if (rc < 0) return -1;
                    if (rc == 0) break;            // header line stays unconsumed for the next call
                    r->pending_seq.clear();
                }
                r->pending_header.assign((const char *)base + lb, le - lb);
            } else {
                const size_t o = r->pending_seq.size();
                r->pending_seq.append((const char *)base + lb, le - lb);
                for (size_t i = o; i < r->pending_seq.size(); ++i) r->pending_seq[i] = (char)toupper((unsigned char)r->pending_seq[i]);
            }
            r->pos = r->scan_next;
        }
    }
    rec_start[n] = w;
    *n_out = n;
    *nbytes_out = (n == 0 && !at_eof) ? need_hint : w;   // nothing delivered: bytes the next record needs (the reference has no record-size limit)
    return at_eof ? 1 : 0;
}

int rd_host_gunzip(const char *path, uint8_t *out, int64_t cap, int64_t *n_out) {
    if (!path || !out || !n_out) RDH_FAIL("rd_host_gunzip: null argument");
    FILE *fp = fopen(path, "rb");
    if (!fp) RDH_FAIL("cannot open %s", path);
    setvbuf(fp, nullptr, _IONBF, 0);
    int64_t n = 0;
    int rc = 0;
    {
        rdz::GzipStream gz(fp, nullptr, 0);
        for (;;) {
            uint8_t extra;
            const long got = n < cap ? gz.read(out + n, (size_t)(cap - n)) : gz.read(&extra, 1);
            if (got < 0) {
                snprintf(g_err, sizeof(g_err), "%s", gz.err.c_str());
                rc = -1;
                break;
            }
            if (got == 0) break;
            if (n >= cap) {
                snprintf(g_err, sizeof(g_err), "rd_host_gunzip: output exceeds the buffer (%lld bytes)", (long long)cap);
                rc = -1;
                break;
            }
            n += got;
        }
    }
    fclose(fp);
    *n_out = n;
    return rc;
}

int rd_host_gunzip_parallel(const char *path, uint8_t *out, int64_t cap, int64_t *n_out, int threads, int64_t section_bytes, int64_t *stats) {
    if (!path || !out || !n_out) RDH_FAIL("rd_host_gunzip_parallel: null argument");
    FILE *fp = fopen(path, "rb");
    if (!fp) RDH_FAIL("cannot open %s", path);
    setvbuf(fp, nullptr, _IONBF, 0);
    int64_t n = 0;
    int rc = 0;
    {
        rdz::ParallelGzip gz(fp, threads > 0 ? threads : std::max(2, usable_threads() - 2), section_bytes > 0 ? (size_t)section_bytes : (4u << 20));
        for (;;) {
            uint8_t extra;
            const long got = n < cap ? gz.read(out + n, (size_t)(cap - n)) : gz.read(&extra, 1);
            if (got < 0) {
                snprintf(g_err, sizeof(g_err), "%s", gz.err.c_str());
                rc = -1;
                break;
            }
            if (got == 0) break;
            if (n >= cap) {
                snprintf(g_err, sizeof(g_err), "rd_host_gunzip: output exceeds the buffer (%lld bytes)", (long long)cap);
                rc = -1;
                break;
            }
            n += got;
        }
        if (stats) {
            stats[0] = (int64_t)gz.sections_used;
            stats[1] = (int64_t)gz.sections_dropped;
            stats[2] = (int64_t)gz.batches;
            stats[3] = gz.fell_back ? 1 : 0;
        }
    }
    fclose(fp);
    *n_out = n;
    return rc;
}

int rd_writer_threads(const rd_writer *w) { return w ? w->threads : -1; }

int rd_host_set_threads(int threads) {
    g_threads = threads > 0 ? threads : 0;
    return 0;
}

int rd_host_set_gz_threads(int threads) {
    g_gz_threads = threads < 0 ? -1 : threads;
    return 0;
}

int rd_writer_open(const char *path, rd_writer **out) {
    if (!path || !out) RDH_FAIL("rd_writer_open: null argument");
    rd_writer *w = new rd_writer();
    std::string p(path);
    w->gz = ends_with(p, "gz");            // reference detect.py:738: read_file.endswith('gz')
    w->threads = usable_threads();
    w->fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    w->off = 0;
    if (w->fd < 0) {
        delete w;
        RDH_FAIL("cannot open %s for writing", path);
    }
    if (w->gz && libdeflate().ok)
        for (int k = 0; k < w->threads; ++k) w->comp.push_back(libdeflate().alloc(5));   // reference: compresslevel=5
    *out = w;
    return 0;
}

// One run = consecutive selected records = one contiguous byte range of the chunk. Plain output is one pwritev() per
// 1,024 runs straight from the chunk buffer: a single copy into the page cache, by one thread - writes to one file
// serialise on the inode lock anyway (tmpfs on the GPU box: 6.1 GB/s from one thread, 4.2 GB/s from eight pwrite()rs).
bool write_runs(int fd, std::vector<struct iovec> &iov, int64_t off) {
    size_t k = 0;
    while (k < iov.size()) {
        const int cnt = (int)std::min<size_t>(iov.size() - k, 1024);
        ssize_t got = pwritev(fd, iov.data() + k, cnt, (off_t)off);
        if (got <= 0) return false;
        off += got;
        while (got > 0 && k < iov.size()) {   // advance past what was written (a short write can end inside a run)
            if ((size_t)got >= iov[k].iov_len) {
                got -= (ssize_t)iov[k].iov_len;
                ++k;
            } else {
                iov[k].iov_base = (char *)iov[k].iov_base + got;
                iov[k].iov_len -= (size_t)got;
                got = 0;
            }
        }
    }
    return true;
}

int rd_writer_write_selected(rd_writer *w, const uint8_t *buf, const int64_t *rec_start, int64_t n, const int8_t *labels,
                             int32_t want) {
    if (!w || !buf || !rec_start || !labels) RDH_FAIL("rd_writer_write_selected: null argument");
    std::vector<struct iovec> iov;
    size_t total = 0;
    int64_t i = 0;
    while (i < n) {
        if (labels[i] != want) { ++i; continue; }
        int64_t j = i + 1;
        while (j < n && labels[j] == want) ++j;   // one copy/write per run of consecutive selected records
        const uint8_t *p = buf + rec_start[i];
        const size_t len = (size_t)(rec_start[j] - rec_start[i]);
        if (w->gz) w->pending.insert(w->pending.end(), p, p + len);
        else if (len) iov.push_back(iovec{const_cast<uint8_t *>(p), len});
        total += len;
        i = j;
    }
    if (w->gz) {
        const size_t full = (w->pending.size() / GZ_BLOCK) * GZ_BLOCK;
        if (full >= GZ_BLOCK * (size_t)w->threads && gz_flush(w, full) != 0) RDH_FAIL("gzip compression/write failed");
        return 0;
    }
    if (!write_runs(w->fd, iov, w->off)) RDH_FAIL("write failed");
    w->off += (int64_t)total;
    return 0;
}

// Complete gzip members made elsewhere - on the GPU (librd_hip.so rd_gz_compress_selected: the chunk's records of one label, deflated
// where they already lie) - appended as they are. Whatever the host path has buffered for this file comes first (input order).
int rd_writer_write_members(rd_writer *w, const uint8_t *members, int64_t len) {
    if (!w || len < 0 || (!members && len)) RDH_FAIL("rd_writer_write_members: bad argument");
    if (!w->gz) RDH_FAIL("rd_writer_write_members: not a gzip output (the name does not end with 'gz')");
    if (!w->pending.empty() && gz_flush(w, w->pending.size()) != 0) RDH_FAIL("gzip compression/write failed");
    if (len && !pwrite_all(w->fd, members, (size_t)len, w->off)) RDH_FAIL("write failed");
    w->off += len;
    w->bgzf = true;
    return 0;
}

// Text that already IS the selected records in input order (packed on the GPU: librd_hip.so rd_select_pack), appended as it is to a
// plain output; for a gzip output it joins the bytes waiting for the host's compressor.
int rd_writer_write_text(rd_writer *w, const uint8_t *text, int64_t len) {
    if (!w || len < 0 || (!text && len)) RDH_FAIL("rd_writer_write_text: bad argument");
    if (w->gz) {
        w->pending.insert(w->pending.end(), text, text + len);
        const size_t full = (w->pending.size() / GZ_BLOCK) * GZ_BLOCK;
        if (full >= GZ_BLOCK * (size_t)w->threads && gz_flush(w, full) != 0) RDH_FAIL("gzip compression/write failed");
        return 0;
    }
    if (len && !pwrite_all(w->fd, text, (size_t)len, w->off)) RDH_FAIL("write failed");
    w->off += len;
    return 0;
}

// BGZF's end-of-file marker at close (default on): off for a PART of a file that is joined with others afterwards - the joined file
// gets one marker at its end (an empty block in the middle would end the file for readers that stop at the first one)
int rd_writer_set_eof_marker(rd_writer *w, int on) {
    if (!w) RDH_FAIL("rd_writer_set_eof_marker: null writer");
    w->eof_marker = on != 0;
    return 0;
}

int rd_writer_close(rd_writer *w) {
    if (!w) return 0;
    int rc = 0;
    if (w->gz) {
        if (!w->pending.empty()) rc = gz_flush(w, w->pending.size());
        if (w->bgzf && !w->eof_marker) {
        } else if (w->bgzf) {            // device-written members are BGZF blocks: the file ends with BGZF's end-of-file marker (an empty member)
            static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (rc == 0 && pwrite_all(w->fd, eof, sizeof(eof), w->off)) w->off += (int64_t)sizeof(eof);
            else rc = -1;
        } else if (w->off == 0) {   // an empty .gz must still be a valid gzip file (one empty member)
            std::vector<uint8_t> m;
            rc = gz_member(nullptr, 0, m, nullptr) && pwrite_all(w->fd, m.data(), m.size(), 0) ? 0 : -1;
        }
    }
    for (void *c : w->comp) libdeflate().release(c);
    if (close(w->fd) != 0) rc = -1;
    delete w;
    if (rc) RDH_FAIL("close failed");
    return 0;
}

}  // extern "C"
