// rd_sort.hpp - steps per read, length bucketing for rd_classify, the stable sort behind rd_pack_plan, and their host-side launch helpers
// Part of the single translation unit rd_kernels.hip (included from there, in order); see that file for the kernel
// inventory and DESIGN.md §3 for the roofline of each kernel.
#pragma once
#include "rd_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// length bucketing: order[] = read indices sorted by T = min(len,max_len) descending (pack_sequence's sort,
// detect.py:685). Ties are ordered by input index (stable) so that the order is deterministic.
// Kernels: per-block histograms -> exclusive scan over (length desc, block asc) in three small launches -> stable scatter.
// ------------------------------------------------------------------------------------------------
constexpr int SORT_BLOCK = 256;
constexpr int SORT_ITEMS = 2048;   // reads per block


// hist[(T)*(nblk) + blk] = number of reads of truncated length T in block blk
__global__ void rd_len_hist_kernel(const int32_t *__restrict__ len, int64_t n, int max_len, int nblk,
                                   uint32_t *__restrict__ hist) {
    extern __shared__ uint32_t sh[];   // max_len+1
    for (int i = threadIdx.x; i <= max_len; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * SORT_ITEMS;
    for (int k = threadIdx.x; k < SORT_ITEMS; k += blockDim.x) {
        int64_t i = base + k;
        if (i < n) atomicAdd(&sh[rd_T(len, i, max_len)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= max_len; i += blockDim.x) hist[(size_t)i * nblk + blockIdx.x] = sh[i];
}

// Exclusive scan of hist in the order (T descending, blk ascending), in three small launches (rounds 1-2 did it in ONE workgroup:
// 127 us per 2^20 reads - API parity only, but flagged twice): (1) one workgroup per length T scans its row of nblk counts in place
// and leaves the row total; (2) one workgroup scans the max_len+1 row totals, longest first: len_start[T] = first sorted position of
// length T, batch_sizes[t] = #reads with T > t (= len_start[t]: lengths are descending), total_steps; (3) the rows get their start.
__global__ __launch_bounds__(SORT_BLOCK) void rd_len_rowscan_kernel(uint32_t *__restrict__ hist, int nblk, uint32_t *__restrict__ row_total) {
    __shared__ uint32_t part[SORT_BLOCK];
    const int tid = threadIdx.x;
    uint32_t *row = hist + (size_t)blockIdx.x * nblk;
    const int per = (nblk + SORT_BLOCK - 1) / SORT_BLOCK, e0 = tid * per, e1 = min(nblk, e0 + per);
    uint32_t s = 0;
    for (int e = e0; e < e1; ++e) s += row[e];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < SORT_BLOCK; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    for (int e = e0; e < e1; ++e) {
        const uint32_t v = row[e];
        row[e] = run;
        run += v;
    }
    if (tid == SORT_BLOCK - 1) row_total[blockIdx.x] = part[tid];
}

__global__ __launch_bounds__(SORT_BLOCK) void rd_len_rowstart_kernel(const uint32_t *__restrict__ row_total, int max_len,
                                                                     int64_t *__restrict__ len_start, int64_t *__restrict__ batch_sizes,
                                                                     int64_t *__restrict__ total_steps) {
    __shared__ unsigned long long part[SORT_BLOCK];
    __shared__ unsigned long long wsum[SORT_BLOCK / 64];
    const int tid = threadIdx.x, nb = max_len + 1, per = (nb + SORT_BLOCK - 1) / SORT_BLOCK;
    unsigned long long s = 0;
    for (int k = 0; k < per; ++k) {
        const int e = tid * per + k;   // scan position e <-> T = max_len - e
        if (e < nb) s += row_total[max_len - e];
    }
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < SORT_BLOCK; d <<= 1) {
        const unsigned long long v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned long long run = part[tid] - s;
    for (int k = 0; k < per; ++k) {
        const int e = tid * per + k;
        if (e < nb) {
            len_start[max_len - e] = (int64_t)run;
            run += row_total[max_len - e];
        }
    }
    if (batch_sizes || total_steps) {
        __threadfence_block();
        __syncthreads();
        unsigned long long acc = 0;
        for (int t = tid; t < max_len; t += SORT_BLOCK) {
            const int64_t bs = len_start[t];
            if (batch_sizes) batch_sizes[t] = bs;
            acc += (unsigned long long)bs;
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if ((tid & 63) == 0) wsum[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0 && total_steps) {
            unsigned long long t = 0;
            for (int w = 0; w < SORT_BLOCK / 64; ++w) t += wsum[w];
            *total_steps = (int64_t)t;
        }
    }
}

__global__ __launch_bounds__(SORT_BLOCK) void rd_len_rowadd_kernel(uint32_t *__restrict__ hist, int nblk, const int64_t *__restrict__ len_start) {
    uint32_t *row = hist + (size_t)blockIdx.x * nblk;
    const uint32_t base = (uint32_t)len_start[blockIdx.x];   // n < 2^31
    for (int e = threadIdx.x; e < nblk; e += SORT_BLOCK) row[e] += base;
}

// stable scatter: one wave-serial pass per block keeps input order inside a length bucket.
__global__ void rd_len_scatter_kernel(const int32_t *__restrict__ len, int64_t n, int max_len, int nblk,
                                      const uint32_t *__restrict__ hist, int32_t *__restrict__ order,
                                      int64_t *__restrict__ sorted_idx, int64_t *__restrict__ unsorted_idx) {
    extern __shared__ uint32_t cur[];   // max_len+1 running cursors of this block
    for (int i = threadIdx.x; i <= max_len; i += blockDim.x) cur[i] = hist[(size_t)i * nblk + blockIdx.x];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SORT_ITEMS;
    // process 64 reads at a time with wave 0 only (stability needs an order; the work is a few bytes per read)
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        for (int k0 = 0; k0 < SORT_ITEMS; k0 += 64) {
            int64_t i = base + k0 + lane;
            bool valid = i < n;
            int T = valid ? rd_T(len, i, max_len) : -1;
            // rank among earlier lanes with the same T
            unsigned rank = 0, cnt = 0;
            for (int o = 0; o < 64; ++o) {
                int To = __shfl(T, o);
                if (To == T) { cnt++; if (o < lane) rank++; }
            }
            uint32_t pos = 0;
            if (valid) pos = cur[T] + rank;
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == cnt - 1) cur[T] = pos + 1;   // last lane of each group advances the cursor
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                if (order) order[pos] = (int32_t)i;
                if (sorted_idx) sorted_idx[pos] = i;
                if (unsorted_idx) unsorted_idx[i] = pos;
            }
        }
    }
}

// Per-read number of forward steps.
//   packed (model.py:32-37 + detect.py:682): T = min(len, max_len).
//   padded (model_cpu.py:29-37,57-62): the input is zero-padded to max_len rows and the output row is the LAST NON-ZERO row,
//     pos = L-1-argmax(flip(rowsum)); if every row is zero, argmax = 0 and pos = L-1. T = pos + 1.
// It also histograms the step counts (wave-aggregated LDS counters, flushed with one global atomic per non-empty bin and
// workgroup) for the bucketing below.
__device__ __forceinline__ uint32_t rd_bin_add(uint32_t *bins, int T, bool valid) {   // returns the rank inside the bin
    const unsigned long long m = __ballot(valid);
    if (!m) return 0;
    const int lane = threadIdx.x & 63, first = __ffsll((long long)m) - 1;
    const int T0 = __shfl(T, first);
    if (__all(!valid || T == T0)) {   // the common case (fixed-length reads): one atomic per wave
        uint32_t base = 0;
        if (lane == first) base = atomicAdd(&bins[T0], (uint32_t)__popcll(m));
        return __shfl(base, first) + (uint32_t)__popcll(m & ((1ull << lane) - 1));
    }
    return valid ? atomicAdd(&bins[T], 1u) : 0;
}

// With a prefix-state table of pk bases (default kernel): a read whose first pk bases are all A/C/G/T(U) and that
// has at least one more step to run starts from the table row numbered by those bases (first base = most significant base-4 digit):
// pfx[i] = that row and steps[i] = the steps that remain. Every other read gets pfx[i] = 4^pk (the zero row) and all its steps.
__global__ __launch_bounds__(256) void rd_steps_kernel(const uint8_t *__restrict__ arena, const int64_t *__restrict__ off,
                                                       const int32_t *__restrict__ len, int64_t n, int max_len, int sem,
                                                       int32_t *__restrict__ steps, uint32_t *__restrict__ ghist, int pk,
                                                       int32_t *__restrict__ pfx) {
    extern __shared__ uint32_t sh_bins[];   // max_len+1
    for (int i = threadIdx.x; i <= max_len; i += 256) sh_bins[i] = 0;
    __syncthreads();
    for (int64_t base = (int64_t)blockIdx.x * 256; base < n; base += (int64_t)gridDim.x * 256) {
        const int64_t i = base + threadIdx.x;
        const bool valid = i < n;
        int T = 0;
        if (valid) {
            const int lr = rd_T(len, i, max_len);
            T = lr;
            if (sem == RD_SEM_PADDED) {
                const uint8_t *p = arena + off[i];
                int pos = lr - 1;
                while (pos >= 0 && rd_code(p[pos]) == 4) --pos;
                T = pos >= 0 ? pos + 1 : max_len;
            }
            if (pk > 0) {
                int prow = 1 << (2 * pk);
                if (T > pk && lr >= pk) {
                    const uint8_t *p = arena + off[i];
                    u32x4 raw = {0u, 0u, 0u, 0u};                      // the first pk <= 13 bases: ONE unaligned 16-byte load when the
                    if (lr >= 16) __builtin_memcpy(&raw, p, 16);       // read has 16 bytes (never past its end), else byte by byte
                    else for (int t = 0; t < pk; ++t) raw[t >> 2] |= (uint32_t)p[t] << (8 * (t & 3));
                    uint32_t idx = 0;
                    bool ok = true;
                    for (int t = 0; t < pk; ++t) {
                        const int c = rd_code((raw[t >> 2] >> (8 * (t & 3))) & 0xffu);
                        ok = ok && c < 4;
                        idx = idx * 4u + (uint32_t)(c & 3);
                    }
                    if (ok) { prow = (int)idx; T -= pk; }
                }
                pfx[i] = prow;
            }
            steps[i] = T;
        }
        rd_bin_add(sh_bins, T, valid);
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= max_len; i += 256)
        if (sh_bins[i]) atomicAdd(&ghist[i], sh_bins[i]);
}

// Length bucketing for rd_classify: order[] = read indices grouped by step count, longest first, so that the reads of a
// tile run (nearly) the same number of steps. Every read is computed independently of its tile mates, so the order INSIDE
// a bucket does not matter and is left to the atomics (rd_pack_plan, whose output order is visible, uses the stable sort
// above). cursor[T] = first position of bucket T = number of reads with more than T steps.
__global__ __launch_bounds__(256) void rd_bucket_scan_kernel(const uint32_t *__restrict__ ghist, int max_len,
                                                             uint32_t *__restrict__ cursor) {
    __shared__ uint32_t part[256];
    const int tid = threadIdx.x, nb = max_len + 1, per = (nb + 255) / 256;
    uint32_t s = 0;
    for (int k = 0; k < per; ++k) {
        const int e = tid * per + k;   // scan position e <-> T = max_len - e
        if (e < nb) s += ghist[max_len - e];
    }
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    for (int k = 0; k < per; ++k) {
        const int e = tid * per + k;
        if (e < nb) {
            cursor[max_len - e] = run;
            run += ghist[max_len - e];
        }
    }
}

constexpr int BK_ITEMS = 2048;   // reads per workgroup and pass
__global__ __launch_bounds__(256) void rd_bucket_scatter_kernel(const int32_t *__restrict__ steps, int64_t n, int max_len,
                                                                uint32_t *__restrict__ cursor, int32_t *__restrict__ order) {
    extern __shared__ uint32_t sh_bins[];   // max_len+1: counts of this pass, then the buckets' reserved start positions
    for (int64_t b0 = (int64_t)blockIdx.x * BK_ITEMS; b0 < n; b0 += (int64_t)gridDim.x * BK_ITEMS) {
        __syncthreads();
        for (int i = threadIdx.x; i <= max_len; i += 256) sh_bins[i] = 0;
        __syncthreads();
        int Tk[BK_ITEMS / 256];
        uint32_t rk[BK_ITEMS / 256];
#pragma unroll
        for (int k = 0; k < BK_ITEMS / 256; ++k) {
            const int64_t i = b0 + k * 256 + threadIdx.x;
            const bool valid = i < n;
            const int T = valid ? steps[i] : 0;
            rk[k] = rd_bin_add(sh_bins, T, valid);
            Tk[k] = valid ? T : -1;
        }
        __syncthreads();
        for (int i = threadIdx.x; i <= max_len; i += 256) {
            const uint32_t c = sh_bins[i];
            if (c) sh_bins[i] = atomicAdd(&cursor[i], c);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < BK_ITEMS / 256; ++k)
            if (Tk[k] >= 0) order[sh_bins[Tk[k]] + rk[k]] = (int32_t)(b0 + k * 256 + threadIdx.x);
    }
}

// ------------------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------------------
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct SortPlan {
    int nblk;
    size_t hist_bytes, order_bytes, lenstart_bytes, steps_bytes, pfx_bytes, total;
};
inline SortPlan sort_plan(int64_t n, int max_len) {
    SortPlan p;
    p.nblk = (int)((n + SORT_ITEMS - 1) / SORT_ITEMS);
    if (p.nblk < 1) p.nblk = 1;
    p.hist_bytes = align_up((size_t)(max_len + 1) * p.nblk * sizeof(uint32_t), 256);
    p.order_bytes = align_up((size_t)(n > 0 ? n : 1) * sizeof(int32_t), 256);
    p.lenstart_bytes = align_up((size_t)(max_len + 1) * sizeof(int64_t) * 2, 256);   // len_start + cum scratch
    p.steps_bytes = align_up((size_t)(n > 0 ? n : 1) * sizeof(int32_t), 256);
    p.pfx_bytes = p.steps_bytes;                                                      // table row per read (rd_steps_kernel)
    p.total = p.hist_bytes + p.order_bytes + p.lenstart_bytes + p.steps_bytes + p.pfx_bytes;
    return p;
}

constexpr int MAX_LEN_LIMIT = 16000;   // LDS histogram of max_len+1 uint32 must fit in 64 KB

int run_sort(const int32_t *seq_len, int64_t n, int max_len, void *workspace, size_t wbytes, int32_t *&order,
             int64_t *sorted_idx, int64_t *unsorted_idx, int64_t *batch_sizes, int64_t *total_steps, int64_t *&len_start,
             hipStream_t st) {
    SortPlan p = sort_plan(n, max_len);
    if (wbytes < p.total) RD_FAIL(RD_E_WORKSPACE, "workspace too small: %zu < %zu", wbytes, p.total);
    char *w = (char *)workspace;
    uint32_t *hist = (uint32_t *)w;
    order = (int32_t *)(w + p.hist_bytes);
    len_start = (int64_t *)(w + p.hist_bytes + p.order_bytes);
    const size_t sh = (size_t)(max_len + 1) * sizeof(uint32_t);
    hipLaunchKernelGGL(rd_len_hist_kernel, dim3(p.nblk), dim3(SORT_BLOCK), sh, st, seq_len, n, max_len, p.nblk, hist);
    uint32_t *row_total = (uint32_t *)(len_start + (max_len + 1));              // second half of the lenstart area
    hipLaunchKernelGGL(rd_len_rowscan_kernel, dim3(max_len + 1), dim3(SORT_BLOCK), 0, st, hist, p.nblk, row_total);
    hipLaunchKernelGGL(rd_len_rowstart_kernel, dim3(1), dim3(SORT_BLOCK), 0, st, row_total, max_len, len_start, batch_sizes, total_steps);
    hipLaunchKernelGGL(rd_len_rowadd_kernel, dim3(max_len + 1), dim3(SORT_BLOCK), 0, st, hist, p.nblk, len_start);
    hipLaunchKernelGGL(rd_len_scatter_kernel, dim3(p.nblk), dim3(SORT_BLOCK), sh, st, seq_len, n, max_len, p.nblk, hist, order,
                       sorted_idx, unsorted_idx);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

// steps[] + order[] for rd_classify: steps kernel (with histogram) -> bucket starts -> scatter
int run_steps_and_buckets(const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n, int max_len, int sem,
                          void *workspace, size_t wbytes, int32_t *&steps, int32_t *&order, int pk, int32_t *&pfx, hipStream_t st) {
    SortPlan p = sort_plan(n, max_len);
    if (wbytes < p.total) RD_FAIL(RD_E_WORKSPACE, "workspace too small: %zu < %zu", wbytes, p.total);
    char *w = (char *)workspace;
    uint32_t *ghist = (uint32_t *)w;                                          // (max_len+1) u32 fit in hist_bytes
    order = (int32_t *)(w + p.hist_bytes);
    uint32_t *cursor = (uint32_t *)(w + p.hist_bytes + p.order_bytes);        // (max_len+1) u32 fit in lenstart_bytes
    steps = (int32_t *)(w + p.total - p.pfx_bytes - p.steps_bytes);
    pfx = pk > 0 ? (int32_t *)(w + p.total - p.pfx_bytes) : nullptr;
    const size_t sh = (size_t)(max_len + 1) * sizeof(uint32_t);
    RD_HIP(hipMemsetAsync(ghist, 0, sh, st));
    // one workgroup per CU at most: every workgroup ends with one global atomic per non-empty bin, and with fixed-length
    // reads they all hit the same bin
    int64_t nb = (n + 255) / 256;
    const int64_t nb_max = pk > 0 ? 2048 : 256;   // with a prefix table every read costs a (latency-bound) load of its first bases: more
    if (nb > nb_max) nb = nb_max;                 // workgroups in flight; the few atomics per workgroup stay few
    hipLaunchKernelGGL(rd_steps_kernel, dim3((unsigned)nb), dim3(256), sh, st, arena, seq_off, seq_len, n, max_len, sem, steps, ghist, pk, pfx);
    hipLaunchKernelGGL(rd_bucket_scan_kernel, dim3(1), dim3(256), 0, st, ghist, max_len, cursor);
    int64_t nbs = (n + BK_ITEMS - 1) / BK_ITEMS;
    if (nbs > 256) nbs = 256;
    hipLaunchKernelGGL(rd_bucket_scatter_kernel, dim3((unsigned)nbs), dim3(256), sh, st, steps, n, max_len, cursor, order);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

}  // namespace
