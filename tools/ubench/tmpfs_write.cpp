// tools/ubench/tmpfs_write.cpp - how fast can ONE output file on tmpfs be written?  g++ -O2 -pthread tmpfs_write.cpp -o tmpfs_write
// (a) pwritev from one thread (what librd_host.so's plain writer does: writes to one file serialise on the inode lock),
// (b) ftruncate + mmap(MAP_SHARED) + N threads memcpy-ing disjoint parts (page faults allocate the pages in parallel).
// Source = runs of ~1.7 KB scattered over a 256 MB buffer, like the selected records of a chunk.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/uio.h>
#include <unistd.h>

#include <chrono>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "/dev/shm/rd_tmpfs_write.bin";
    const size_t chunk = 256u << 20, nchunks = 6, run = 1744, gap = 218;
    std::vector<char> src(chunk);
    for (size_t i = 0; i < chunk; ++i) src[i] = (char)(i * 2654435761u >> 24);
    std::vector<iovec> iov;
    size_t total = 0;
    for (size_t p = 0; p + run <= chunk; p += run + gap) { iov.push_back({src.data() + p, run}); total += run; }
    // (a)
    {
        int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        off_t off = 0;
        for (size_t c = 0; c < nchunks; ++c)
            for (size_t k = 0; k < iov.size(); k += 1024) {
                ssize_t got = pwritev(fd, iov.data() + k, (int)std::min<size_t>(1024, iov.size() - k), off);
                if (got <= 0) { perror("pwritev"); return 1; }
                off += got;
            }
        double dt = now() - t0;
        close(fd);
        printf("pwritev, 1 thread: %.2f GB in %.3f s = %.2f GB/s\n", off / 1e9, dt, off / 1e9 / dt);
        unlink(path);
    }
    // (b)
    for (int nt : {1, 2, 4, 8}) {
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        size_t off = 0;
        for (size_t c = 0; c < nchunks; ++c) {
            const size_t base = off & ~(size_t)4095, len = off + total - base;
            if (ftruncate(fd, (off_t)(off + total)) != 0) { perror("ftruncate"); return 1; }
            char *m = (char *)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)base);
            if (m == MAP_FAILED) { perror("mmap"); return 1; }
            char *dst0 = m + (off - base);
            std::vector<std::thread> th;
            const size_t per = (iov.size() + nt - 1) / nt;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&, t] {
                    const size_t k0 = t * per, k1 = std::min(iov.size(), k0 + per);
                    char *d = dst0 + k0 * run;
                    for (size_t k = k0; k < k1; ++k) { memcpy(d, iov[k].iov_base, iov[k].iov_len); d += iov[k].iov_len; }
                });
            for (auto &x : th) x.join();
            munmap(m, len);
            off += total;
        }
        double dt = now() - t0;
        close(fd);
        printf("mmap + %d thread(s): %.2f GB in %.3f s = %.2f GB/s\n", nt, off / 1e9, dt, off / 1e9 / dt);
        unlink(path);
    }
    return 0;
}
