// rd_kernels.hip - CDNA4 (gfx950) kernels + C ABI of the RiboDetector BiLSTM inference path.
//
// Reference path being replaced (paths relative to /root/reference/ribodetector):
//   data_loader/seq_encoder.py:11-18,126-127   one-hot nucleotide encoder
//   detect.py:666-726                          collate: truncate to max_len, one-hot, pack_sequence
//   model/model.py:32-37,114-119               forward1: BiLSTM, last-timestep gather, Linear(256->2)
//   detect.py:288,481                          argmax
//   detect.py:616-663                          paired-end label fusion
//
// Structural facts used (verified by tests against the reference's own outputs):
//   * forward1 gathers timestep len-1. There the reverse LSTM has made exactly ONE step from the zero state,
//     so its contribution to the logits is a 5-entry table keyed by the last base (rev_lut).
//   * the input is one-hot/zero, so W_ih x + b_ih + b_hh is a 5-row table (in_lut) - a gather, not a GEMM.
//   => the only dense work is the forward recurrence h[B,128] . W_hh^T[128,512] per timestep.
//
// Kernel inventory (DESIGN.md §3 has the roofline of each)
//   rd_prep_kernel, rd_revtab_kernel   weight pre-packing (once per model); reverse-half table of the padded semantics
//   rd_steps_kernel, rd_bucket_scan/scatter_kernel   steps per read + length bucketing for rd_classify (longest first)
//   rd_len_hist/scan/scatter_kernel    stable counting sort = pack_sequence's sort, for rd_pack_plan
//   rd_lstm_mfma_f16x3_t32_kernel      DEFAULT recurrence: split-precision f16 MFMA 32x32x16, weights resident in AGPRs,
//                                      hand-interleaved gate math; fused encoder + FC + argmax epilogue
//   rd_lstm_mfma_f16x3_kernel, rd_lstm_mfma_f16x3_w8_kernel   earlier / experimental tilings of the same arithmetic
//   rd_lstm_mfma_f32_kernel            exact-fp32 MFMA recurrence (A/B reference for the split-precision kernels)
//   rd_lstm_simple_kernel              plain-FMA cross-check of the same function
//   rd_encode_* / rd_pack_onehot       standalone encoder kernels (reference tensor layouts), HBM-bound
//   rd_pair_fuse_kernel, rd_count_kernel
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ribodetector_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HID = 128;     // hidden size
constexpr int G4 = 512;      // 4 gates x hidden
constexpr int NT = 4;        // 16-read tiles per workgroup (ring)
constexpr int BT = NT * 16;  // reads per workgroup
constexpr int HSTR = 132;    // LDS row stride (floats) of an h tile: 128 + 4 pad -> conflict-free b128 reads
constexpr int TC = 128;      // timesteps per staged code chunk

thread_local char g_err[512] = "";

#define RD_FAIL(code, ...)                              \
    do {                                                \
        snprintf(g_err, sizeof(g_err), __VA_ARGS__);    \
        return (code);                                  \
    } while (0)
#define RD_HIP(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) RD_FAIL(RD_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));  \
    } while (0)

// ------------------------------------------------------------------------------------------------
// encoder: seq_encoder.py:11-18  A C G T U(=T) -> 0 1 2 3 ; anything else (lowercase included) -> 4
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int rd_code(unsigned ch) {
    return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : (ch == 'T' || ch == 'U') ? 3 : 4;
}

__device__ __forceinline__ int rd_T(const int32_t *len, int64_t i, int max_len) {
    int t = len[i];
    t = t < 0 ? 0 : t;
    return t < max_len ? t : max_len;
}

// ------------------------------------------------------------------------------------------------
// activations. v_exp_f32 evaluates 2^x to ~1 ulp; the argument x*log2(e) is formed with an FMA-compensated
// product so the result stays within ~2 ulp of expf over the whole range (SURVEY §7 "transcendental accuracy").
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rd_exp(float x) {
    const float L2E_HI = 1.44269502162933349609375f;    // fl(log2 e)
    const float L2E_LO = 1.925963033500011e-8f;         // log2 e - L2E_HI
    float t = x * L2E_HI;
    float e = __builtin_fmaf(x, L2E_HI, -t);             // rounding error of the product
    e = __builtin_fmaf(x, L2E_LO, e);
    float r = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(r, e * 0.693147182464599609375f, r);   // 2^(t+e) ~= 2^t (1 + e ln2)
}
__device__ __forceinline__ float rd_sigmoid(float x) {
    x = fmaxf(x, -80.0f);                                // keep exp(-x) finite: rcp(inf) would still be 0, but avoid inf*0 downstream
    return __builtin_amdgcn_rcpf(1.0f + rd_exp(-x));
}
__device__ __forceinline__ float rd_tanh(float x) {     // tanh x = 2 sigmoid(2x) - 1
    return __builtin_fmaf(2.0f, rd_sigmoid(2.0f * x), -1.0f);
}

// ------------------------------------------------------------------------------------------------
// model blob (device)
// ------------------------------------------------------------------------------------------------
struct DevModel {
    float *raw;       // uploaded tensors, concatenated
    float *wpack32;   // [4 waves][8 col tiles][32 k-steps][64 lanes]  fp32 MFMA B-operand order
    float *wt_hh;     // [128][512]  W_hh^T  (simple kernel)
    float *in_lut;    // [5][512]    W_ih[:,code] + (b_ih + b_hh); code 4 = bias only
    float *rev_lut;   // [5][2]      W_out[:,128:] . h_rev(one step from zero on `code`)
    float *w_out;     // [2][256]
    float *b_out;     // [2]
    uint32_t *wpack16;  // f16x3 16x16x32 A operand (hi/lo halves), see rd_prep_kernel
    uint32_t *wpack16b; // f16x3 32x32x16 A operand
    uint32_t *wpack16c; // f16x3 32x32x16 A operand of the 8-wave kernel
    float *rev_tab;     // padded (ribodetector_cpu) semantics: [max_len][5][2] reverse-direction logit terms, see rd_revtab_kernel
};

}  // namespace

struct rd_model {
    int device;
    int variant;
    int semantics;      // RD_SEM_PACKED / RD_SEM_PADDED
    int rev_tab_len;    // max_len the padded-semantics table was built for (0 = none)
    DevModel d;
    // profiling of the recurrence kernel (bench.py roofline)
    int prof_enabled;
    int prof_count;
    hipEvent_t prof_ev[2 * 512];
    double prof_ms_accum;
    int64_t prof_launches_accum;
};

namespace {

// raw layout offsets (floats)
constexpr int OFF_WIH = 0, OFF_WHH = OFF_WIH + 512 * 4, OFF_BIH = OFF_WHH + 512 * 128, OFF_BHH = OFF_BIH + 512;
constexpr int OFF_WIHR = OFF_BHH + 512, OFF_WHHR = OFF_WIHR + 512 * 4, OFF_BIHR = OFF_WHHR + 512 * 128;
constexpr int OFF_BHHR = OFF_BIHR + 512, OFF_WOUT = OFF_BHHR + 512, OFF_BOUT = OFF_WOUT + 512, RAW_FLOATS = OFF_BOUT + 2;

// gate column handled by (wave w, column tile c = gate*2 + sub, lane&15)
__device__ __host__ __forceinline__ int gate_col(int w, int c, int l15) { return (c >> 1) * HID + 32 * w + 16 * (c & 1) + l15; }

__global__ void rd_prep_kernel(DevModel d) {
    const float *raw = d.raw;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    // fp32 MFMA B operand: lane l (col = l&15, q = l>>4), k-step s = 4m + j  <->  hidden index 16m + 4q + j
    for (int i = tid; i < 4 * 8 * 32 * 64; i += nth) {
        int lane = i & 63, s = (i >> 6) & 31, c = (i >> 11) & 7, w = i >> 14;
        int m = s >> 2, j = s & 3, q = lane >> 4;
        d.wpack32[i] = raw[OFF_WHH + gate_col(w, c, lane & 15) * HID + 16 * m + 4 * q + j];
    }
    for (int i = tid; i < HID * G4; i += nth) {
        int k = i / G4, col = i % G4;
        d.wt_hh[i] = raw[OFF_WHH + col * HID + k];
    }
    // f16x3 A operand: [wave][W1|W2][tile a][k-step s][lane][8 halves]; lane (i = lane&15, q): row i of tile a is
    // gate i&3 of unit 32w + 8(i>>2) + a; element e is hidden index 32s + 8q + e.  W1 = fp16(16 w), W2 = fp16(2^11 (16 w - W1)).
    for (int i = tid; i < 4 * 8 * 4 * 64 * 8; i += nth) {
        int e = i & 7, lane = (i >> 3) & 63, s = (i >> 9) & 3, a = (i >> 11) & 7, w = i >> 14;
        int row = lane & 15, q = lane >> 4;
        int col = (row & 3) * HID + 32 * w + 8 * (row >> 2) + a;
        float x = 16.0f * raw[OFF_WHH + col * HID + 32 * s + 8 * q + e];
        _Float16 hi = (_Float16)x;
        _Float16 lo = (_Float16)((x - (float)hi) * 2048.0f);
        _Float16 *base = reinterpret_cast<_Float16 *>(d.wpack16);
        base[((((size_t)(w * 2 + 0) * 8 + a) * 4 + s) * 64 + lane) * 8 + e] = hi;
        base[((((size_t)(w * 2 + 1) * 8 + a) * 4 + s) * 64 + lane) * 8 + e] = lo;
    }
    for (int i = tid; i < 5 * G4; i += nth) {
        int code = i / G4, col = i % G4;
        float b = raw[OFF_BIH + col] + raw[OFF_BHH + col];
        d.in_lut[i] = code < 4 ? b + raw[OFF_WIH + col * 4 + code] : b;
    }
    // f16x3 / 32x32x16 A operand: [wave][W1|W2][row-tile a][k-step s][lane][8 halves]; lane (i = lane&31, kh = lane>>5):
    // row i = 8b + 4hf + g is gate g of unit 32w + 16hf + 4a + b; element e is hidden index 16s + 8kh + e.
    for (int i = tid; i < 4 * 4 * 8 * 64 * 8; i += nth) {
        int e = i & 7, lane = (i >> 3) & 63, s = (i >> 9) & 7, a = (i >> 12) & 3, w = i >> 14;
        int row = lane & 31, kh = lane >> 5;
        int g = row & 3, hf = (row >> 2) & 1, b = row >> 3;
        int col = g * HID + 32 * w + 16 * hf + 4 * a + b;
        float x = 16.0f * raw[OFF_WHH + col * HID + 16 * s + 8 * kh + e];
        _Float16 hi = (_Float16)x;
        _Float16 lo = (_Float16)(x - (float)hi);      // unscaled: multiplied with H1s = 2^11 h_hi it carries the common 2^15
        _Float16 *base = reinterpret_cast<_Float16 *>(d.wpack16b);
        base[((((size_t)(w * 2 + 0) * 4 + a) * 8 + s) * 64 + lane) * 8 + e] = hi;
        base[((((size_t)(w * 2 + 1) * 4 + a) * 8 + s) * 64 + lane) * 8 + e] = lo;
    }
    // w8 kernel A operand: [wave(8)][W1|W2][row-tile a(2)][k-step s][lane][8 halves]; row i = 8b + 4hf + g is gate g of unit
    // 16w + 8hf + 4a + b; W2 = unscaled fp16 residual of 16 w.
    for (int i = tid; i < 8 * 2 * 8 * 64 * 8; i += nth) {
        int e = i & 7, lane = (i >> 3) & 63, s = (i >> 9) & 7, a = (i >> 12) & 1, w = i >> 13;
        int row = lane & 31, kh = lane >> 5;
        int g = row & 3, hf = (row >> 2) & 1, b = row >> 3;
        int col = g * HID + 16 * w + 8 * hf + 4 * a + b;
        float x = 16.0f * raw[OFF_WHH + col * HID + 16 * s + 8 * kh + e];
        _Float16 hi = (_Float16)x;
        _Float16 lo = (_Float16)(x - (float)hi);
        _Float16 *base = reinterpret_cast<_Float16 *>(d.wpack16c);
        base[((((size_t)(w * 2 + 0) * 2 + a) * 8 + s) * 64 + lane) * 8 + e] = hi;
        base[((((size_t)(w * 2 + 1) * 2 + a) * 8 + s) * 64 + lane) * 8 + e] = lo;
    }
    for (int i = tid; i < 512; i += nth) d.w_out[i] = raw[OFF_WOUT + i];
    if (tid < 2) d.b_out[tid] = raw[OFF_BOUT + tid];
    // reverse direction: one cell step from (h,c) = 0 on base `code` (W_hh_r . 0 vanishes), then the FC's reverse half.
    if (tid < 10) {
        int code = tid >> 1, k = tid & 1;
        float s = 0.0f;
        for (int u = 0; u < HID; ++u) {
            float g[4];
            for (int gi = 0; gi < 4; ++gi) {
                int col = gi * HID + u;
                float b = raw[OFF_BIHR + col] + raw[OFF_BHHR + col];
                g[gi] = code < 4 ? b + raw[OFF_WIHR + col * 4 + code] : b;
            }
            float ig = 1.0f / (1.0f + expf(-g[0])), gg = tanhf(g[2]), og = 1.0f / (1.0f + expf(-g[3]));
            float c = ig * gg;                    // f * 0 + i * g~
            float h = og * tanhf(c);
            s += raw[OFF_WOUT + k * 256 + HID + u] * h;
        }
        d.rev_lut[code * 2 + k] = s;
    }
}

// Padded (ribodetector_cpu) semantics, reverse half. The output row pos of a read is preceded, in the reverse direction, by
// max_len-1-pos all-zero rows (padding / trailing non-ACGT bases): the reverse state there does not depend on the read.
// tab[k][code][cls] = W_out[cls, 128:] . h_rev  where h_rev = cell(state after k zero-input steps from zero, input `code`).
// One workgroup, max_len sequential cell steps of a 128 x 512 mat-vec: microseconds, built once per max_len.
__global__ __launch_bounds__(512) void rd_revtab_kernel(DevModel d, int max_len) {
    __shared__ float h[HID], c[HID], g[G4], hc[5][HID], hn[HID], cn[HID];
    const float *raw = d.raw;
    const int tid = threadIdx.x;
    if (tid < HID) { h[tid] = 0.0f; c[tid] = 0.0f; }
    __syncthreads();
    for (int k = 0; k < max_len; ++k) {
        float a = raw[OFF_BIHR + tid] + raw[OFF_BHHR + tid];
        for (int u = 0; u < HID; ++u) a = __builtin_fmaf(raw[OFF_WHHR + tid * HID + u], h[u], a);
        g[tid] = a;
        __syncthreads();
        for (int cell = tid; cell < 5 * HID; cell += 512) {
            const int code = cell / HID, u = cell % HID;
            float gi = g[u], gf = g[HID + u], gg = g[2 * HID + u], go = g[3 * HID + u];
            if (code < 4) {
                gi += raw[OFF_WIHR + u * 4 + code];
                gf += raw[OFF_WIHR + (HID + u) * 4 + code];
                gg += raw[OFF_WIHR + (2 * HID + u) * 4 + code];
                go += raw[OFF_WIHR + (3 * HID + u) * 4 + code];
            }
            const float ig = 1.0f / (1.0f + expf(-gi)), fg = 1.0f / (1.0f + expf(-gf)), og = 1.0f / (1.0f + expf(-go));
            const float c2 = fg * c[u] + ig * tanhf(gg);
            const float h2 = og * tanhf(c2);
            hc[code][u] = h2;
            if (code == 4) { hn[u] = h2; cn[u] = c2; }   // the state advances over a zero row
        }
        __syncthreads();
        if (tid < 10) {
            const int code = tid >> 1, cls = tid & 1;
            float s = 0.0f;
            for (int u = 0; u < HID; ++u) s += raw[OFF_WOUT + cls * 256 + HID + u] * hc[code][u];
            d.rev_tab[(k * 5 + code) * 2 + cls] = s;
        }
        if (tid < HID) { h[tid] = hn[tid]; c[tid] = cn[tid]; }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// length bucketing: order[] = read indices sorted by T = min(len,max_len) descending (pack_sequence's sort,
// detect.py:685). Ties are ordered by input index (stable) so that the order is deterministic.
// Three kernels: per-block histograms -> exclusive scan over (length desc, block asc) -> stable scatter.
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

// single block: exclusive scan of hist in the order (T descending, blk ascending); also per-length totals:
// len_start[T] = first sorted position of length T; batch_sizes[t] = #reads with T > t; total_steps.
__global__ void rd_len_scan_kernel(uint32_t *__restrict__ hist, int max_len, int nblk, int64_t *__restrict__ len_start,
                                   int64_t *__restrict__ batch_sizes, int64_t *__restrict__ total_steps) {
    __shared__ unsigned long long carry;
    __shared__ unsigned long long wsum[SORT_BLOCK / 64];
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    const int64_t total = (int64_t)(max_len + 1) * nblk;
    unsigned long long steps = 0;
    for (int64_t base = 0; base < total; base += SORT_BLOCK) {
        int64_t e = base + tid;               // element in scan order
        unsigned v = 0;
        int T = 0;
        if (e < total) {
            T = max_len - (int)(e / nblk);
            v = hist[(size_t)T * nblk + (e % nblk)];
        }
        // inclusive scan within the block
        unsigned long long x = v;
        for (int o = 1; o < 64; o <<= 1) {
            unsigned long long y = __shfl_up(x, o);
            if ((tid & 63) >= o) x += y;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = x;
        __syncthreads();
        unsigned long long pre = carry;
        for (int w = 0; w < (tid >> 6); ++w) pre += wsum[w];
        unsigned long long excl = pre + x - v;
        if (e < total) {
            hist[(size_t)T * nblk + (e % nblk)] = (uint32_t)excl;   // n < 2^31
            if ((e % nblk) == 0 && len_start) len_start[T] = (int64_t)excl;
        }
        __syncthreads();
        if (tid == SORT_BLOCK - 1) carry = pre + x;
        __syncthreads();
    }
    (void)steps;
    if (batch_sizes || total_steps) {
        // batch_sizes[t] = #reads with T >= t+1 = len_start[t] (start of the first length <= t) ... computed from len_start:
        // reads with T > t occupy sorted positions [0, len_start[t]) because lengths are descending.
        __syncthreads();
        unsigned long long acc = 0;
        for (int t = tid; t < max_len; t += SORT_BLOCK) {
            int64_t bs = len_start[t];
            if (batch_sizes) batch_sizes[t] = bs;
            acc += (unsigned long long)bs;
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
        if ((tid & 63) == 0) wsum[tid >> 6] = acc;
        __syncthreads();
        if (tid == 0 && total_steps) {
            unsigned long long s = 0;
            for (int w = 0; w < SORT_BLOCK / 64; ++w) s += wsum[w];
            *total_steps = (int64_t)s;
        }
    }
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

// ------------------------------------------------------------------------------------------------
// shared pieces of the recurrence kernels
// ------------------------------------------------------------------------------------------------
struct ReadBatch {
    const uint8_t *arena;
    const int64_t *off;
    const int32_t *len;
    const int32_t *steps;   // timesteps the forward recurrence runs for each read (rd_steps_kernel)
    const int32_t *order;   // sorted position -> read index (nullptr = identity)
    int64_t n;
    int max_len;
    int sem;                // RD_SEM_PACKED: gather at step len-1 (reference GPU path); RD_SEM_PADDED: ribodetector_cpu
    const float *rev_tab;   // padded semantics only
};

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

__global__ __launch_bounds__(256) void rd_steps_kernel(const uint8_t *__restrict__ arena, const int64_t *__restrict__ off,
                                                       const int32_t *__restrict__ len, int64_t n, int max_len, int sem,
                                                       int32_t *__restrict__ steps, uint32_t *__restrict__ ghist) {
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

// FC + argmax epilogue for one workgroup's reads. hl(row,u) = captured last forward hidden state.
// logits = b_out + W_out[:, :128] . h_fwd + rev_lut[last base]   (model.py:36; reverse half folded, see header)
template <typename HL>
__device__ __forceinline__ void rd_fc_epilogue(int nrows, HL hl, const int *Trow, const int *Lrow, const long long *offrow,
                                               const int *origrow, const float *s_wout, const DevModel &d, const ReadBatch &rb,
                                               float *logits, uint8_t *labels) {
    const int tid = threadIdx.x;
    if (tid < 2 * nrows) {
        const int row = tid >> 1, k = tid & 1;
        float s = d.b_out[k];
        for (int u = 0; u < HID; ++u) s = __builtin_fmaf(s_wout[k * HID + u], hl(row, u), s);
        const int T = Trow[row];
        if (rb.sem == RD_SEM_PADDED && T > 0) {   // (T == 0 only for the filler rows of the last workgroup)
            // reverse half of output row pos = T-1: the reverse LSTM has walked max_len-1-pos zero rows, then x[pos]
            const int pos = T - 1;
            const int code = pos < Lrow[row] ? rd_code(rb.arena[offrow[row] + pos]) : 4;
            s += rb.rev_tab[((rb.max_len - 1 - pos) * 5 + code) * 2 + k];
        } else if (rb.sem != RD_SEM_PADDED && T > 0) {
            s += d.rev_lut[rd_code(rb.arena[offrow[row] + T - 1]) * 2 + k];
        }
        const float other = __shfl_xor(s, 1);
        const int orig = origrow[row];
        if (orig >= 0) {
            logits[(size_t)orig * 2 + k] = s;
            if (labels && k == 0) labels[orig] = other > s ? 1 : 0;   // torch.argmax: first max wins ties -> 0
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rd_lstm_mfma_f32_kernel - persistent-weight fp32-MFMA forward recurrence.
//
// Workgroup = 256 threads = 4 waves (one per SIMD, 512 VGPR/AGPR each), 64 reads = ring of NT=4 tiles x 16 reads.
// Wave w owns hidden units [32w, 32w+32) of all four gates: 8 column tiles of 16 (gate g, sub s), W_hh slice held in
// 256 registers per lane for the whole kernel (64 KiB per wave, 256 KiB per CU = all of W_hh).
// One "phase" = one tile x one timestep = 256 v_mfma_f32_16x16x4_f32 per wave (A = h tile from LDS, B = weights):
//     G[16 reads, 128 cols] += h[16,128] . W^T           (k index permuted identically on both operands)
// The C/D layout (col = lane&15, row = 4*(lane>>4)+reg) puts i,f,g,o of one (read, unit) cell in ONE lane, so the gate
// math needs no cross-lane traffic. Phase p runs the MFMAs of (t, tile) while the VALU does the gate math of phase p-1
// and the LDS prefetches the A fragments of phase p+1; one barrier per phase. h and c live in LDS between phases.
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) LstmSmem {
    float Hs[NT][16][HSTR];        // current h of every tile (A operand source)
    float Hl[NT][16][HSTR];        // h captured at t == T-1 (last_items, model.py:114-119)
    f32x4 cA[NT][256];             // cell state, sub-tile 0 (4 reads per lane)
    f32x4 cB[NT][256];             // cell state, sub-tile 1
    f32x4 lut[5][4][16][2];        // in_lut staged per lane: [code][wave][lane&15][half] -> 8 floats = column tiles c=0..7
    float wout[2][HID];            // forward half of W_out
    float dummy[256];              // sink of predicated-off Hl stores (keeps the phase body branch-free)
    uint8_t codes[2][TC][BT];      // double-buffered code chunks, [t][row]
    int T[BT];
    int Lr[BT];       // readable bytes of the read = min(len, max_len)
    long long off[BT];
    int orig[BT];
    int tmax;
};

__device__ __forceinline__ void rd_stage_codes(LstmSmem &S, const ReadBatch &rb, int chunk) {
    const int t0 = chunk * TC;
    uint8_t(*dst)[BT] = S.codes[chunk & 1];
    for (int idx = threadIdx.x; idx < BT * TC; idx += 256) {
        const int row = idx / TC, tt = idx % TC, t = t0 + tt;
        int code = 4;
        if (t < S.Lr[row]) code = rd_code(rb.arena[S.off[row] + t]);
        dst[tt][row] = (uint8_t)code;
    }
}

// Cheap activations for the recurrence: sigmoid(x) = rcp(1 + 2^(-x log2 e)). The rounding of the product x*log2(e)
// perturbs the exponent by <= |x| 2^-24, i.e. sigmoid by <= s(1-s) |x| ln2 2^-24 < 1.5e-8 |x| e^-|x|... < 1e-7 absolute:
// the same order as one fp32 ulp of the result, so the compensated form (rd_exp) is only kept for the A/B variant.
template <int ACT>
__device__ __forceinline__ float act_sigmoid(float x) {
    if (ACT == 0) return rd_sigmoid(x);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
template <int ACT>
__device__ __forceinline__ float act_tanh(float x) {
    if (ACT == 0) return rd_tanh(x);
    // tanh x = 1 - 2 / (1 + 2^(2x log2 e))
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.88539008177792681f)), 1.0f);
}

// ACT: 0 = compensated exp, 1 = plain v_exp_f32 forms.  SCHED: 0 = compiler's own order, 1 = LDS reads of the gate math
// pinned to the top of the phase + explicit MFMA/VALU interleave (sched_group_barrier).
template <int ACT, int SCHED, int DIAG = 0>   // DIAG (bench diagnosis only, wrong results): 1 = no gate math, 2 = no MFMA
__global__ __launch_bounds__(256, 1) void rd_lstm_mfma_f32_kernel(DevModel d, ReadBatch rb, float *__restrict__ logits,
                                                                  uint8_t *__restrict__ labels) {
    __shared__ LstmSmem S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;

    // ---- per-read metadata, zero state ---------------------------------------------------------
    if (tid < BT) {
        const int64_t g = (int64_t)blockIdx.x * BT + tid;
        int T = 0, lr = 0, orig = -1;
        long long off = 0;
        if (g < rb.n) {
            orig = rb.order ? rb.order[g] : (int)g;
            T = rd_T(rb.steps, orig, rb.max_len);
            lr = rd_T(rb.len, orig, rb.max_len);
            off = rb.off[orig];
        }
        S.T[tid] = T; S.Lr[tid] = lr; S.off[tid] = off; S.orig[tid] = orig;
    }
    if (tid == 0) S.tmax = 0;
    for (int i = tid; i < NT * 16 * HSTR; i += 256) { (&S.Hs[0][0][0])[i] = 0.0f; (&S.Hl[0][0][0])[i] = 0.0f; }
    for (int i = tid; i < NT * 256; i += 256) { (&S.cA[0][0])[i] = f32x4{0, 0, 0, 0}; (&S.cB[0][0])[i] = f32x4{0, 0, 0, 0}; }
    for (int i = tid; i < 5 * G4; i += 256) {      // i = ((code*4 + w)*16 + l15)*8 + c
        const int c = i & 7, l = (i >> 3) & 15, w = (i >> 7) & 3, code = i >> 9;
        (reinterpret_cast<float *>(&S.lut[0][0][0][0]))[i] = d.in_lut[code * G4 + gate_col(w, c, l)];
    }
    S.wout[tid >> 7][tid & 127] = d.w_out[(tid >> 7) * 256 + (tid & 127)];
    __syncthreads();
    if (tid < BT) atomicMax(&S.tmax, S.T[tid]);
    rd_stage_codes(S, rb, 0);

    // ---- resident weights: 8 column tiles x 32 k-steps, one f32 per lane each ------------------
    float Wr[8][32];
    {
        const float *wp = d.wpack32 + (size_t)wave * (8 * 32 * 64) + lane;
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                // Register plan (512 per lane): column tiles 1..7 of the weights are pinned in 224 AGPRs (the MFMAs read
                // them there directly as srcB), the 8 accumulators take the other 32 AGPRs, and tile 0's 32 weights stay
                // in architectural VGPRs next to the h fragments and the gate math.
                const float x = wp[(c * 32 + s) * 64];
                if (c == 0) Wr[c][s] = x;
                else asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(Wr[c][s]) : "v"(x));
            }
    }
    __syncthreads();
    const int tmax = S.tmax;
    const int nphase = tmax * NT;

    f32x4 accP[8];   // gate pre-activations (recurrent part) of the previous phase
#pragma unroll
    for (int c = 0; c < 8; ++c) accP[c] = f32x4{0, 0, 0, 0};
    f32x4 hA[8];     // A fragments of the current phase: h[read l15][16m + 4q .. +3]
#pragma unroll
    for (int m = 0; m < 8; ++m) hA[m] = f32x4{0, 0, 0, 0};

    int tile = 0, t = 0;          // current phase
    int ptile = NT - 1, pt = -1;  // previous phase (dummy before the first: its state update is masked to zero)
    uint32_t cwP = 0x04040404u;   // codes of the previous phase's 4 reads of this lane (loaded one phase ahead)

    // p == nphase is a drain iteration: its MFMAs run on a dummy tile, its gate math finishes the last real phase.
    for (int p = 0; p <= nphase; ++p) {
        // stage the next code chunk one full chunk ahead (visible long before its first use, barriers in between)
        // (tile 1, not 0: the gate math of phase (t, 0) still reads the chunk that this overwrites)
        if (tile == 1 && (t % TC) == 0) {
            const int chunk = t / TC + 1;
            if (chunk * TC < tmax + 1) rd_stage_codes(S, rb, chunk);
        }
        const int ntile = tile + 1 == NT ? 0 : tile + 1;
        const int nt = tile + 1 == NT ? t + 1 : t;

        // ---- LDS reads, all issued at the top of the phase ------------------------------------------
        // A fragments of the next phase (written >= 2 barriers ago)
        f32x4 hN[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) hN[m] = *reinterpret_cast<const f32x4 *>(&S.Hs[ntile][l15][16 * m + 4 * q]);
        if (DIAG >= 3) {
#pragma unroll
            for (int m = 0; m < 8; ++m) hN[m] = hA[m] + accP[m] * 1e-30f;
        }
        // operands of the previous phase's gate math
        const int tcur = t < tmax ? t : 0;
        const uint32_t cwN = *reinterpret_cast<const uint32_t *>(&S.codes[(tcur / TC) & 1][tcur % TC][tile * 16 + 4 * q]);
        const int4 Tr = *reinterpret_cast<const int4 *>(&S.T[ptile * 16 + 4 * q]);
        f32x4 cs[2] = {S.cA[ptile][tid], S.cB[ptile][tid]};
        f32x4 lv[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int code = (int)((cwP >> (8 * r)) & 0xff);
            lv[r][0] = S.lut[code][wave][l15][0];
            lv[r][1] = S.lut[code][wave][l15][1];
        }
        if (SCHED) __builtin_amdgcn_sched_barrier(0);

        // ---- MFMA: 256 x v_mfma_f32_16x16x4_f32, 8 independent accumulators ---------------------
        f32x4 acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = f32x4{0, 0, 0, 0};
        if (DIAG != 2) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = hA[m][j];
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Wr[c][m * 4 + j], acc[c], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] = f32x4{hA[c][0] + Wr[c][0], hA[c][1] + Wr[c][9], hA[c][2] + Wr[c][17], hA[c][3] + Wr[c][31]};
        }

        // ---- gate math of the previous phase (VALU, overlaps the MFMAs above) -------------------
        if (DIAG >= 3) {
        } else if (DIAG == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    S.Hs[ptile][4 * q + r][32 * wave + 16 * s + l15] = accP[s][r] + accP[2 + s][r] + accP[4 + s][r] + accP[6 + s][r] + lv[r][s][0] + cs[s][r];
        } else {
            const int Tq[4] = {Tr.x, Tr.y, Tr.z, Tr.w};
            const float live = pt < 0 ? 0.0f : 1.0f;   // the dummy phase before t = 0 must leave the zero state untouched
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool last = (pt == Tq[r] - 1);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    // column tile c = gate*2 + s ; lv[r][c>>2][c&3]
                    const float gi = accP[0 + s][r] + lv[r][0][0 + s];
                    const float gf = accP[2 + s][r] + lv[r][0][2 + s];
                    const float gg = accP[4 + s][r] + lv[r][1][0 + s];
                    const float go = accP[6 + s][r] + lv[r][1][2 + s];
                    float cn = __builtin_fmaf(act_sigmoid<ACT>(gf), cs[s][r], act_sigmoid<ACT>(gi) * act_tanh<ACT>(gg));
                    cn *= live;
                    const float h = act_sigmoid<ACT>(go) * act_tanh<ACT>(cn) * live;
                    cs[s][r] = cn;
                    const int row = 4 * q + r, u = 32 * wave + 16 * s + l15;
                    S.Hs[ptile][row][u] = h;
                    float *dst = last ? &S.Hl[ptile][row][u] : &S.dummy[tid];
                    *dst = h;
                }
            }
            S.cA[ptile][tid] = cs[0];
            S.cB[ptile][tid] = cs[1];
        }
        if (SCHED) {
            // one MFMA, then up to three VALU/transcendental ops in its shadow (single wave per SIMD issues in order)
#pragma unroll
            for (int i = 0; i < 256; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x402, 3, 0);
            }
        }
        if (DIAG != 3) __syncthreads();
#pragma unroll
        for (int c = 0; c < 8; ++c) accP[c] = acc[c];
#pragma unroll
        for (int m = 0; m < 8; ++m) hA[m] = hN[m];
        cwP = cwN;
        ptile = tile; pt = t; tile = ntile; t = nt;
    }

    if (DIAG >= 3) {   // keep the diagnostic MFMA chain live
        float sink = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) sink += accP[c][0] + accP[c][1] + accP[c][2] + accP[c][3] + hA[c][0];
        S.Hl[0][l15][tid & 127] = sink;
        __syncthreads();
    }
    // ---- epilogue: FC + reverse table + argmax ----------------------------------------------------
    rd_fc_epilogue(
        BT, [&](int row, int u) { return S.Hl[row >> 4][row & 15][u]; }, S.T, S.Lr, S.off, S.orig, &S.wout[0][0], d, rb, logits, labels);
}

// ------------------------------------------------------------------------------------------------
// rd_lstm_simple_kernel - plain fp32 FMA statement of the same function (cross-check / bring-up).
// 512 threads = one per gate column, 8 reads per workgroup, W_hh^T streamed from L2 every step.
// ------------------------------------------------------------------------------------------------
constexpr int SB = 8;
__global__ __launch_bounds__(512) void rd_lstm_simple_kernel(DevModel d, ReadBatch rb, float *__restrict__ logits,
                                                             uint8_t *__restrict__ labels) {
    __shared__ float h[SB][HID], c[SB][HID], hl[SB][HID], g[SB][G4], s_wout[2 * HID];
    __shared__ int T[SB], Lr[SB], orig[SB], tmax_s;
    __shared__ long long off[SB];
    const int tid = threadIdx.x;
    if (tid < SB) {
        const int64_t gi = (int64_t)blockIdx.x * SB + tid;
        int Ti = 0, li = 0, o = -1;
        long long of = 0;
        if (gi < rb.n) { o = rb.order ? rb.order[gi] : (int)gi; Ti = rd_T(rb.steps, o, rb.max_len); li = rd_T(rb.len, o, rb.max_len); of = rb.off[o]; }
        T[tid] = Ti; Lr[tid] = li; orig[tid] = o; off[tid] = of;
    }
    if (tid == 0) tmax_s = 0;
    for (int i = tid; i < SB * HID; i += 512) { (&h[0][0])[i] = 0; (&c[0][0])[i] = 0; (&hl[0][0])[i] = 0; }
    if (tid < 256) s_wout[tid] = d.w_out[(tid >> 7) * 256 + (tid & 127)];
    __syncthreads();
    if (tid < SB) atomicMax(&tmax_s, T[tid]);
    __syncthreads();
    const int tmax = tmax_s;
    for (int t = 0; t < tmax; ++t) {
        float a[SB];
#pragma unroll
        for (int r = 0; r < SB; ++r) {
            int code = 4;
            if (t < Lr[r]) code = rd_code(rb.arena[off[r] + t]);
            a[r] = d.in_lut[code * G4 + tid];
        }
        for (int k = 0; k < HID; ++k) {
            const float w = d.wt_hh[k * G4 + tid];
#pragma unroll
            for (int r = 0; r < SB; ++r) a[r] = __builtin_fmaf(h[r][k], w, a[r]);
        }
#pragma unroll
        for (int r = 0; r < SB; ++r) g[r][tid] = a[r];
        __syncthreads();
        for (int cell = tid; cell < SB * HID; cell += 512) {
            const int r = cell / HID, u = cell % HID;
            const float ig = rd_sigmoid(g[r][u]), fg = rd_sigmoid(g[r][HID + u]), gg = rd_tanh(g[r][2 * HID + u]);
            const float og = rd_sigmoid(g[r][3 * HID + u]);
            const float cn = __builtin_fmaf(fg, c[r][u], ig * gg);
            const float hn = og * rd_tanh(cn);
            c[r][u] = cn; h[r][u] = hn;
            if (t == T[r] - 1) hl[r][u] = hn;
        }
        __syncthreads();
    }
    rd_fc_epilogue(
        SB, [&](int row, int u) { return hl[row][u]; }, T, Lr, off, orig, s_wout, d, rb, logits, labels);
}

// ------------------------------------------------------------------------------------------------
// rd_lstm_mfma_f16x3_kernel - split-precision recurrence on the f16 matrix pipe.
//
// Measured on MI355X (tools/ubench/mfma_fill.hip): v_mfma_f32_16x16x4_f32 issues every 36 cycles and does NOT overlap
// with VALU work of the same wave (one filler VALU op costs +12 cycles: the f32 MFMA runs at the f32 vector rate on the
// same datapath), so the fp32 kernel above pays MFMA time + gate-math time. v_mfma_f32_16x16x32_f16 issues every 17
// cycles with two VALU ops per MFMA hidden for free. This kernel therefore evaluates the fp32 product as three f16
// products accumulated in fp32:
//     h = h_hi + h_lo,  w = w_hi + w_lo   (hi = fp16 rounding, lo = fp16 rounding of the exact residual)
//     h.w ~= h_hi w_hi + h_hi w_lo + h_lo w_hi          (dropped: h_lo w_lo <= 2^-22 |h w|)
// Operands are pre-scaled by powers of two so that no residual lands in the fp16 subnormal range:
//     A (weights, registers):  W1 = 16 w_hi            W2 = 2^11 (16 w - W1)
//     B (hidden state, LDS):   H1s = 2^11 h_hi'  (h_hi' = fp16(2^11 h)/2^11)   H1 = h_hi'   H2 = 2^11 h - H1s
//     acc = LUT*2^15 + W1.H1s + W2.H1 + W1.H2  = 2^15 * (W_ih x + b + W_hh h)      (every term carries 2^15)
// and the 2^-15 is folded into the activation's exp2 argument. Products of fp16 pairs are exact in fp32; the only
// extra error over the fp32 kernel is the 2^-22-relative representation error of each operand (same order as fp32's
// own 2^-24 rounding of the 128-term sum). tests/test_gpu_parity.py holds this path to the same 1e-4 logit bound.
//
// Orientation: A = weights (rows = 16 gate rows of a column tile), B = h^T (cols = 16 reads). A tile's 16 rows are
// {4 units x (i,f,g,o)}: row 4*qr + gate <-> unit 32*wave + 8*qr + a for tile a = 0..7, so the D layout
// (col = lane&15, row = 4*(lane>>4) + reg) hands lane (read l15, q) the four gates of unit 32w + 8q + a in ONE
// accumulator, and over a = 0..7 eight CONTIGUOUS units: h leaves as one 16-byte LDS store per operand array.
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int H16STR = 136;   // f16 per LDS row: 128 + 8 pad = 272 B -> conflict-free ds_read_b128 over 16 rows
constexpr int TC16 = 64;      // timesteps per staged code chunk
constexpr float W_SCALE = 16.0f, H_SCALE = 2048.0f, G_SCALE = 32768.0f;   // 2^4, 2^11, 2^15

struct __attribute__((aligned(16))) Lstm16Smem {
    _Float16 H1s[NT][16][H16STR];
    _Float16 H1[NT][16][H16STR];
    _Float16 H2[NT][16][H16STR];
    float Hl[NT][16][HSTR];        // h captured at t == T-1
    f32x4 cA[NT][256];             // cell state of units a = 0..3
    f32x4 cB[NT][256];             // cell state of units a = 4..7
    f32x4 lut[5][4][4][8];         // [code][wave][q][a] -> exp2 arguments' constant terms of (i,f,g,o), see KI/KG
    float wout[2][HID];
    uint8_t codes[2][TC16][BT];
    int T[BT];
    int Lr[BT];       // readable bytes of the read = min(len, max_len)
    long long off[BT];
    int orig[BT];
    int tmax;
};

__device__ __forceinline__ void rd_stage_codes16(Lstm16Smem &S, const ReadBatch &rb, int chunk) {
    const int t0 = chunk * TC16;
    uint8_t(*dst)[BT] = S.codes[chunk & 1];
    for (int idx = threadIdx.x; idx < BT * TC16; idx += 256) {
        const int row = idx / TC16, tt = idx % TC16, t = t0 + tt;
        int code = 4;
        if (t < S.Lr[row]) code = rd_code(rb.arena[S.off[row] + t]);
        dst[tt][row] = (uint8_t)code;
    }
}

// sigmoid(x) = 1 / (1 + 2^(KS x)),  tanh(x) = 1 - 2 / (1 + 2^(KT x))
constexpr float KS = -1.44269504088896341f, KT = 2.88539008177792681f;

template <int FILL>   // VALU ops scheduled behind each MFMA
__global__ __launch_bounds__(256, 1) void rd_lstm_mfma_f16x3_kernel(DevModel d, ReadBatch rb, float *__restrict__ logits,
                                                                    uint8_t *__restrict__ labels) {
    __shared__ Lstm16Smem S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;

    if (tid < BT) {
        const int64_t g = (int64_t)blockIdx.x * BT + tid;
        int T = 0, lr = 0, orig = -1;
        long long off = 0;
        if (g < rb.n) {
            orig = rb.order ? rb.order[g] : (int)g;
            T = rd_T(rb.steps, orig, rb.max_len);
            lr = rd_T(rb.len, orig, rb.max_len);
            off = rb.off[orig];
        }
        S.T[tid] = T; S.Lr[tid] = lr; S.off[tid] = off; S.orig[tid] = orig;
    }
    if (tid == 0) S.tmax = 0;
    for (int i = tid; i < 3 * NT * 16 * H16STR / 2; i += 256) (reinterpret_cast<uint32_t *>(&S.H1s[0][0][0]))[i] = 0u;
    for (int i = tid; i < NT * 16 * HSTR; i += 256) (&S.Hl[0][0][0])[i] = 0.0f;
    for (int i = tid; i < NT * 256; i += 256) { (&S.cA[0][0])[i] = f32x4{0, 0, 0, 0}; (&S.cB[0][0])[i] = f32x4{0, 0, 0, 0}; }
    // The input/bias term of each gate enters as the constant term of the activation's exp2 argument:
    //   2^(KS (G/2^15 + lut)) = 2^(fma(G, KS/2^15, KS lut))  - no add, no accumulator init.
    for (int i = tid; i < 5 * G4; i += 256) {      // i = (((code*4 + w)*4 + qq)*8 + a)*4 + gate
        const int gate = i & 3, a = (i >> 2) & 7, qq = (i >> 5) & 3, w = (i >> 7) & 3, code = i >> 9;
        (reinterpret_cast<float *>(&S.lut[0][0][0][0]))[i] = (gate == 2 ? KT : KS) * d.in_lut[code * G4 + gate * HID + 32 * w + 8 * qq + a];
    }
    S.wout[tid >> 7][tid & 127] = d.w_out[(tid >> 7) * 256 + (tid & 127)];
    __syncthreads();
    if (tid < BT) atomicMax(&S.tmax, S.T[tid]);
    rd_stage_codes16(S, rb, 0);

    // ---- resident weights: 8 tiles x 4 k-steps x (W1, W2), one f16x8 (4 registers) per lane each = 256 registers.
    // Register plan: tiles 1..7 pinned in 224 AGPRs (read there directly as srcA), the 8 accumulators in the other 32
    // AGPRs, tile 0 (32 registers) in architectural VGPRs with the B fragments and the gate math.
    f16x8 W1[8][4], W2[8][4];
    {
        const uint4 *wp = reinterpret_cast<const uint4 *>(d.wpack16) + (size_t)wave * (2 * 8 * 4 * 64) + lane;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    const uint4 x = wp[((hl * 8 + a) * 4 + s) * 64];
                    uint4 y = x;
                    if (a != 0) {
                        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.x) : "v"(x.x));
                        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.y) : "v"(x.y));
                        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.z) : "v"(x.z));
                        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.w) : "v"(x.w));
                    }
                    if (hl == 0) W1[a][s] = __builtin_bit_cast(f16x8, y);
                    else W2[a][s] = __builtin_bit_cast(f16x8, y);
                }
            }
    }
    __syncthreads();
    const int tmax = S.tmax;
    const int nphase = tmax * NT;

    f32x4 accP[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) accP[a] = f32x4{0, 0, 0, 0};
    int tile = 0, t = 0, ptile = NT - 1, pt = -1;
    int codeP = 4;                         // code of this lane's read in the previous phase (constant terms of its gates)
    const int boff = l15 * H16STR + 8 * q; // f16 offset of this lane's B fragment inside a tile, k-step 0
    // k-step 0 of the B fragments is fetched one phase ahead; k-steps 1..3 stream in behind the MFMAs of the step before
    f16x8 b1s0 = *reinterpret_cast<const f16x8 *>(&S.H1s[0][0][0] + boff);
    f16x8 b10 = *reinterpret_cast<const f16x8 *>(&S.H1[0][0][0] + boff);
    f16x8 b20 = *reinterpret_cast<const f16x8 *>(&S.H2[0][0][0] + boff);

    for (int p = 0; p <= nphase; ++p) {
        if (tile == 1 && (t % TC16) == 0) {
            const int chunk = t / TC16 + 1;
            if (chunk * TC16 < tmax + 1) rd_stage_codes16(S, rb, chunk);
        }
        const int ntile = tile + 1 == NT ? 0 : tile + 1;
        const int nt = tile + 1 == NT ? t + 1 : t;
        const _Float16 *h1s = &S.H1s[tile][0][0] + boff, *h1 = &S.H1[tile][0][0] + boff, *h2 = &S.H2[tile][0][0] + boff;

        // ---- operands of the previous phase's gate math (not latency critical) ------------------------------------
        const int tc = t < tmax ? t : 0;
        const int codeN = S.codes[(tc / TC16) & 1][tc % TC16][tile * 16 + l15];
        const int Tp = S.T[ptile * 16 + l15];
        f32x4 cs[2] = {S.cA[ptile][tid], S.cB[ptile][tid]};
        f32x4 kc[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) kc[a] = S.lut[codeP][wave][q][a];

        // ---- 96 x v_mfma_f32_16x16x32_f16: same accumulator every 8th instruction --------------------------------
        f32x4 acc[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) acc[a] = f32x4{0, 0, 0, 0};
        f16x8 bs = b1s0, bh = b10, bl = b20;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f16x8 ns = bs, nh = bh, nl = bl;
            if (s < 3) {
                ns = *reinterpret_cast<const f16x8 *>(h1s + 32 * (s + 1));
                nh = *reinterpret_cast<const f16x8 *>(h1 + 32 * (s + 1));
                nl = *reinterpret_cast<const f16x8 *>(h2 + 32 * (s + 1));
            } else {                                  // next phase's k-step 0 (its tile was written >= 2 barriers ago)
                ns = *reinterpret_cast<const f16x8 *>(&S.H1s[ntile][0][0] + boff);
                nh = *reinterpret_cast<const f16x8 *>(&S.H1[ntile][0][0] + boff);
                nl = *reinterpret_cast<const f16x8 *>(&S.H2[ntile][0][0] + boff);
            }
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W1[a][s], bs, acc[a], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W2[a][s], bh, acc[a], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W1[a][s], bl, acc[a], 0, 0, 0);
            bs = ns; bh = nh; bl = nl;
        }
        b1s0 = bs; b10 = bh; b20 = bl;

        // ---- gate math of the previous phase: lane (read l15) x units 32w + 8q + a -------------------------------
        const float live = pt < 0 ? 0.0f : 1.0f;
        float hv[8];
        f16x8 o1s, o1, o2;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const f32x4 G = accP[a];
            const float cold = cs[a >> 2][a & 3];
            const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(G[0], KS / G_SCALE, kc[a][0])));
            const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(G[1], KS / G_SCALE, kc[a][1])));
            const float gr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(G[2], KT / G_SCALE, kc[a][2])));
            const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(G[3], KS / G_SCALE, kc[a][3])));
            const float gg = __builtin_fmaf(-2.0f, gr, 1.0f);
            float cn = __builtin_fmaf(fg, cold, ig * gg);
            cn *= live;
            const float tc2 = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cn * KT)), 1.0f);
            const float h = og * tc2 * live;
            cs[a >> 2][a & 3] = cn;
            hv[a] = h;
            const float hs = h * H_SCALE;
            const _Float16 p16 = (_Float16)hs;                  // 2^11 h_hi'
            o1s[a] = p16;
            o1[a] = p16 * (_Float16)(1.0f / H_SCALE);           // exact power-of-two scaling
            o2[a] = (_Float16)(hs - (float)p16);                // exact residual, rounded once
        }
        {
            const int wo = l15 * H16STR + 32 * wave + 8 * q;
            *reinterpret_cast<f16x8 *>(&S.H1s[ptile][0][0] + wo) = o1s;
            *reinterpret_cast<f16x8 *>(&S.H1[ptile][0][0] + wo) = o1;
            *reinterpret_cast<f16x8 *>(&S.H2[ptile][0][0] + wo) = o2;
            S.cA[ptile][tid] = cs[0];
            S.cB[ptile][tid] = cs[1];
        }
        // the f16 MFMA hides two VALU/transcendental ops per instruction (tools/ubench/mfma_fill.hip): pin that interleave
#pragma unroll
        for (int i = 0; i < 96; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x402, FILL, 0);
        }
        if (pt == Tp - 1) {
            float *hl = &S.Hl[ptile][l15][32 * wave + 8 * q];
            *reinterpret_cast<f32x4 *>(hl) = f32x4{hv[0], hv[1], hv[2], hv[3]};
            *reinterpret_cast<f32x4 *>(hl + 4) = f32x4{hv[4], hv[5], hv[6], hv[7]};
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 8; ++a) accP[a] = acc[a];
        codeP = codeN;
        ptile = tile; pt = t; tile = ntile; t = nt;
    }

    rd_fc_epilogue(
        BT, [&](int row, int u) { return S.Hl[row >> 4][row & 15][u]; }, S.T, S.Lr, S.off, S.orig, &S.wout[0][0], d, rb, logits, labels);
}

// ------------------------------------------------------------------------------------------------
// rd_lstm_mfma_f16x3_t32_kernel - the same split-precision recurrence on v_mfma_f32_32x32x16_f16 with 32-read tiles.
//
// Why: one wave per SIMD issues one instruction per ~4.6 cycles, and the 16x16x32 kernel above spends 2,215 of its
// 3,900 cycles per phase issuing ~450 instructions (PMC, profiles/r01_summary.txt). The 32x32x16 MFMA does twice the
// work per instruction (37 cycles) and hides six VALU ops instead of two (tools/ubench/mfma_fill.hip), so the whole
// gate math fits in MFMA shadows.
//   * workgroup = 4 waves, 64 reads = 2 tiles x 32 reads; the two tiles alternate (phase A: tile 0, phase B: tile 1),
//     so tile indices, LDS addresses and the accumulator set of each half are compile-time constants;
//   * accumulators ping-pong between two VGPR sets (X: tile 0, Y: tile 1): the gate math reads the other set in place
//     - no copies, no v_accvgpr_read; ALL 256 AGPRs hold weights (read directly as MFMA srcA);
//   * A = weights: row-tile a (0..3) of 32 rows = 8 units x (i,f,g,o); row 8b + 4hf + g  <->  gate g of unit
//     32w + 16hf + 4a + b.  With the 32x32 C/D layout (col = lane&31, row = (reg&3) + 8(reg>>2) + 4(lane>>5)) lane
//     (read j, half) holds in acc[a][4b + g] the four gates of unit 32w + 16half + 4a + b: 16 contiguous units/lane;
//   * the dummy gate pass before t = 0 uses an all-zero table row (code 5): sigmoid -> 1/2, tanh -> 0 => c = h = 0;
//   * only two B arrays: W2 is kept as the UNSCALED fp16 residual of 16 w, so W2 . H1s carries the same 2^15 as W1 . H1s and
//     W1 . H2 - the separate unscaled copy of h_hi (H1) of the 16x16 kernel is gone (8 LDS reads, 4 stores, 8 VALU per phase).
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct __attribute__((aligned(16))) Lstm16bSmem {
    // hot arrays first: everything the phase loop touches per cell sits below 64 KiB, so its LDS addresses are one base
    // register + a 16-bit immediate offset (no per-access address arithmetic)
    f32x4 lut[4][2][4][4][6];      // [wave][half][a][b][code] -> exp2-argument constants of (i,f,g,o); code 5 = zeros
    _Float16 H1s[2][32][H16STR];   // 2^11 h_hi   (B operand of the W1 and of the unscaled-W2 products)
    _Float16 H2[2][32][H16STR];    // 2^11 h - H1s
    f32x4 cS[2][4][256];           // cell state [tile][row-tile a][tid] -> units b = 0..3
    float Hl[64][HSTR];            // h captured at t == T-1
    f32x4 dummy[256];              // sink of predicated-off Hl stores
    float wout[2][HID];
    uint8_t codes[2][TC16][64];
    int T[64];
    int Lr[64];       // readable bytes of the read = min(len, max_len)
    long long off[64];
    int orig[64];
    int tmax;
};

__device__ __forceinline__ void rd_stage_codes16b(Lstm16bSmem &S, const ReadBatch &rb, int chunk) {
    const int t0 = chunk * TC16;
    uint8_t(*dst)[64] = S.codes[chunk & 1];
    for (int idx = threadIdx.x; idx < 64 * TC16; idx += 256) {
        const int row = idx / TC16, tt = idx % TC16, t = t0 + tt;
        int code = 4;
        if (t < S.Lr[row]) code = rd_code(rb.arena[S.off[row] + t]);
        dst[tt][row] = (uint8_t)code;
    }
}

// One phase: MFMAs of (tile TL, current step) into accC; gate math of (tile TL^1, step tEW) from accP.
//
// The compiler's scheduler neither interleaves the two streams on its own nor honours a 96-group sched_group_barrier
// pipeline in a region this large, so the interleave is written out: the gate math is cut into 212 "units" of 1-5
// instructions (13 stages per cell, two cells in flight and never in the same stage, at most two transcendentals per
// unit, table rows fetched one cell ahead) and the units are dealt out behind the 96 MFMAs, ~2.2 units (about 5 VALU
// ops) per MFMA - what a 32x32x16 MFMA mostly hides (tools/ubench/mfma_fill.hip: 38.7 cycles bare, 48 with 2 exp + 3 fma).
// A sched_barrier after every slot pins the order.
constexpr int EW_NU = 212;
constexpr unsigned char EW_CELL[EW_NU] = {0,0,0,0,0,0,0,1,0,1,0,1,0,1,0,1,0,1,0,1,1,2,1,2,1,2,1,2,1,2,1,2,2,3,2,3,2,3,2,3,2,3,2,3,2,3,3,4,3,4,3,4,3,4,3,4,3,4,3,4,5,4,5,4,5,4,5,4,5,4,5,4,5,5,6,5,6,5,6,5,6,5,6,5,6,6,7,6,7,6,7,6,7,6,7,6,7,6,7,7,8,7,8,7,8,7,8,7,8,7,8,7,8,9,8,9,8,9,8,9,8,9,8,9,8,9,9,10,9,10,9,10,9,10,9,10,9,10,10,11,10,11,10,11,10,11,10,11,10,11,10,11,11,12,11,12,11,12,11,12,11,12,11,12,11,12,13,12,13,12,13,12,13,12,13,12,13,12,13,13,14,13,14,13,14,13,14,13,14,13,14,14,15,14,15,14,15,14,15,14,15,14,15,14,15,15,15,15,15,15,15,15};
constexpr unsigned char EW_STAGE[EW_NU] = {0,1,2,3,4,5,6,0,7,1,8,2,9,3,10,4,11,5,12,6,7,0,8,1,9,2,10,3,11,4,12,5,6,0,7,1,8,2,9,3,10,4,11,5,12,6,7,0,8,1,9,2,10,3,11,4,12,5,13,6,0,7,1,8,2,9,3,10,4,11,5,12,6,7,0,8,1,9,2,10,3,11,4,12,5,6,0,7,1,8,2,9,3,10,4,11,5,12,6,7,0,8,1,9,2,10,3,11,4,12,5,13,6,0,7,1,8,2,9,3,10,4,11,5,12,6,7,0,8,1,9,2,10,3,11,4,12,5,6,0,7,1,8,2,9,3,10,4,11,5,12,6,7,0,8,1,9,2,10,3,11,4,12,5,13,6,0,7,1,8,2,9,3,10,4,11,5,12,6,7,0,8,1,9,2,10,3,11,4,12,5,6,0,7,1,8,2,9,3,10,4,11,5,12,6,7,8,9,10,11,12,13};
// (schedule above: 16 cells x 13 stages, cell c starts at step floor(6.5 c) so that two cells are in flight and never in
//  the same stage; stage 13 = stores of a finished row-tile; generated offline, units are dealt out in this order)
struct EwRegs {
    f32x4 kc[2];        // table rows of the cells in flight, by cell parity
    f32x2 v[2][2];      // gate pipeline values by cell parity: {i,f} and {g,o} as register pairs (packed fp32 math)
    float y[2], og[2], hs[2];
    f32x4 cs[2], hv[2]; // per row-tile, by row-tile parity
    f16x4 o1s[2], o2[2];
};

struct PhaseCtx {       // per-lane constants of a phase
    int codeEW, wave, half, j, tid;
    bool last;
};

template <int TP, int U>
__device__ __forceinline__ void rd_ew_unit(Lstm16bSmem &S, EwRegs &R, const f32x16 (&accP)[4], const PhaseCtx &c) {
    constexpr int cell = EW_CELL[U], stage = EW_STAGE[U];
    constexpr int a = cell >> 2, b = cell & 3, k = cell & 1, ap = a & 1;
    if constexpr (stage == 0) {   // table row of the NEXT cell (this cell's row was fetched one cell ago); cell state per row-tile
        constexpr int nc = cell + 1;
        if constexpr (nc < 16) R.kc[k ^ 1] = S.lut[c.wave][c.half][nc >> 2][nc & 3][c.codeEW];
        if constexpr (b == 0) R.cs[ap] = S.cS[TP][a][c.tid];
    } else if constexpr (stage == 1) {
        // exp2 arguments. Scalar FMAs on purpose: packed fp32 ops (v_pk_fma_f32 / v_pk_add_f32) cost ~10 cycles each beside
        // f16 MFMAs against ~1 for a scalar op (tools/ubench/mfma_fill.hip), so the build also passes -fno-slp-vectorize.
        R.v[k][0][0] = __builtin_fmaf(accP[a][4 * b + 0], KS / G_SCALE, R.kc[k][0]);
        R.v[k][0][1] = __builtin_fmaf(accP[a][4 * b + 1], KS / G_SCALE, R.kc[k][1]);
        R.v[k][1][0] = __builtin_fmaf(accP[a][4 * b + 2], KT / G_SCALE, R.kc[k][2]);
        R.v[k][1][1] = __builtin_fmaf(accP[a][4 * b + 3], KS / G_SCALE, R.kc[k][3]);
    } else if constexpr (stage == 2) {
        R.v[k][0][0] = __builtin_amdgcn_exp2f(R.v[k][0][0]); R.v[k][0][1] = __builtin_amdgcn_exp2f(R.v[k][0][1]);
    } else if constexpr (stage == 3) {
        R.v[k][1][0] = __builtin_amdgcn_exp2f(R.v[k][1][0]); R.v[k][1][1] = __builtin_amdgcn_exp2f(R.v[k][1][1]);
    } else if constexpr (stage == 4) {
        R.v[k][0][0] += 1.0f; R.v[k][0][1] += 1.0f;
        R.v[k][1][0] += 1.0f; R.v[k][1][1] += 1.0f;
    } else if constexpr (stage == 5) {
        R.v[k][0][0] = __builtin_amdgcn_rcpf(R.v[k][0][0]); R.v[k][0][1] = __builtin_amdgcn_rcpf(R.v[k][0][1]);
    } else if constexpr (stage == 6) {
        R.v[k][1][0] = __builtin_amdgcn_rcpf(R.v[k][1][0]); R.v[k][1][1] = __builtin_amdgcn_rcpf(R.v[k][1][1]);
    } else if constexpr (stage == 7) {
        // the cell state is kept pre-multiplied by KT (c' = KT c): c' = f c'_old + i (KT tanh g), and tanh(c) = 1 - 2/(1 + 2^c')
        const float gg = __builtin_fmaf(-2.0f * KT, R.v[k][1][0], KT);
        const float cn = __builtin_fmaf(R.v[k][0][1], R.cs[ap][b], R.v[k][0][0] * gg);
        R.cs[ap][b] = cn;
        R.y[k] = cn;
        R.og[k] = R.v[k][1][1];
    } else if constexpr (stage == 8) {
        R.y[k] = __builtin_amdgcn_exp2f(R.y[k]);
    } else if constexpr (stage == 9) {
        R.y[k] = 1.0f + R.y[k];
    } else if constexpr (stage == 10) {
        R.y[k] = __builtin_amdgcn_rcpf(R.y[k]);
    } else if constexpr (stage == 11) {
        R.hs[k] = R.og[k] * __builtin_fmaf(-2.0f * H_SCALE, R.y[k], H_SCALE);   // 2^11 h = 2^11 o tanh(c)
        R.hv[ap][b] = R.hs[k];                                       // captured state is kept at scale 2^11 (epilogue divides)
    } else if constexpr (stage == 12) {
        // hi/lo split, two cells at a time (cells 2i and 2i+1 of a row-tile; the even cell's 2^11 h waits in R.hs[0]):
        //   P  = {fp16(hs0), fp16(hs1)}                 one v_cvt_pk_f16_f32
        //   r  = hs - fp32(P.half)  (exact, in fp32)    one v_fma_mix_f32 each (fp32 result: the fp16-output form
        //                                               v_fma_mixlo_f16 measurably loses accuracy)
        //   O2 = {fp16(r0), fp16(r1)}                   one v_cvt_pk_f16_f32
        if constexpr (k == 1) {
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            const f16x2 P = {(_Float16)R.hs[0], (_Float16)R.hs[1]};
            unsigned pbits = __builtin_bit_cast(unsigned, P);
            float r0, r1;
            asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(R.hs[0]), "v"(pbits));
            asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(R.hs[1]), "v"(pbits));
            const f16x2 O = {(_Float16)r0, (_Float16)r1};
            R.o1s[ap][b - 1] = P[0]; R.o1s[ap][b] = P[1];
            R.o2[ap][b - 1] = O[0]; R.o2[ap][b] = O[1];
        }
    } else {   // 13: the row-tile's 4 cells are complete
        const int wo = c.j * H16STR + 32 * c.wave + 16 * c.half + 4 * a;
        *reinterpret_cast<f16x4 *>(&S.H1s[TP][0][0] + wo) = R.o1s[ap];
        *reinterpret_cast<f16x4 *>(&S.H2[TP][0][0] + wo) = R.o2[ap];
        S.cS[TP][a][c.tid] = R.cs[ap];
        f32x4 *dst = c.last ? reinterpret_cast<f32x4 *>(&S.Hl[TP * 32 + c.j][32 * c.wave + 16 * c.half + 4 * a]) : &S.dummy[c.tid];
        *dst = R.hv[ap];
    }
}

template <int TP, int U0, int U1>
__device__ __forceinline__ void rd_ew_units(Lstm16bSmem &S, EwRegs &R, const f32x16 (&accP)[4], const PhaseCtx &c) {
    if constexpr (U0 < U1) {
        rd_ew_unit<TP, U0>(S, R, accP, c);
        rd_ew_units<TP, U0 + 1, U1>(S, R, accP, c);
    }
}

// slot M = MFMA number M (k-step s = M/12, product (M%12)/4, row-tile M%4) followed by its share of gate-math units
template <int TL, int FILL, int M>
__device__ __forceinline__ void rd_slots(Lstm16bSmem &S, const f16x8 (&W1)[4][8], const f16x8 (&W2)[4][8], f32x16 (&accC)[4],
                                         const f32x16 (&accP)[4], f16x8 (&Bf)[2][2], EwRegs &R, const PhaseCtx &c,
                                         const _Float16 *h1s, const _Float16 *h2) {
    if constexpr (M < 96) {
        constexpr int s = M / 12, pr = (M % 12) / 4, a = M % 4;
        if constexpr (M % 12 == 0 && s < 7) {       // B fragments of the next k-step stream in behind this one's MFMAs
            Bf[(s + 1) & 1][0] = *reinterpret_cast<const f16x8 *>(h1s + 16 * (s + 1));
            Bf[(s + 1) & 1][1] = *reinterpret_cast<const f16x8 *>(h2 + 16 * (s + 1));
        }
        // products: W1.H1s, W2.H1s (W2 = unscaled residual of 16 w, so this pair also carries 2^15), W1.H2
        const f16x8 A = pr == 1 ? W2[a][s] : W1[a][s];
        const f16x8 B = Bf[s & 1][pr == 2 ? 1 : 0];
        if constexpr (M < 4) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.0f;
            accC[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, z, 0, 0, 0);
        } else {
            accC[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, accC[a], 0, 0, 0);
        }
        if constexpr (FILL > 0) {
            rd_ew_units<TL ^ 1, (M * EW_NU) / 96, ((M + 1) * EW_NU) / 96>(S, R, accP, c);
            __builtin_amdgcn_sched_barrier(0);
        }
        rd_slots<TL, FILL, M + 1>(S, W1, W2, accC, accP, Bf, R, c, h1s, h2);
    }
}

// One phase: MFMAs of (tile TL, current step) into accC; gate math of (tile TL^1, step tEW) from accP.
//
// The compiler's scheduler neither interleaves the two streams on its own nor honours a 96-group sched_group_barrier
// pipeline in a region this large, so the interleave is written out: the gate math is cut into 212 "units" of 1-5
// instructions (13 stages per cell, two cells in flight and never in the same stage, at most two transcendentals per
// unit, table rows fetched one cell ahead) and the units are dealt out behind the 96 MFMAs, ~2.2 units (about 5 VALU
// ops) per MFMA - what a 32x32x16 MFMA mostly hides (tools/ubench/mfma_fill.hip: 38.7 cycles bare, 48 with 2 exp + 3 fma).
// A sched_barrier after every slot pins the order.
template <int TL, int FILL>
__device__ __forceinline__ void rd_phase_t32(Lstm16bSmem &S, const f16x8 (&W1)[4][8], const f16x8 (&W2)[4][8], f32x16 (&accC)[4],
                                             f32x16 (&accP)[4], int tEW, int codeEW, int wave, int half, int j, int tid) {
    constexpr int TP = TL ^ 1;
    const int boff = j * H16STR + 8 * half;     // this lane's B fragment: row j, k = 16s + 8half + e
    const _Float16 *h1s = &S.H1s[TL][0][0] + boff, *h2 = &S.H2[TL][0][0] + boff;
    f16x8 Bf[2][2];
    Bf[0][0] = *reinterpret_cast<const f16x8 *>(h1s);
    Bf[0][1] = *reinterpret_cast<const f16x8 *>(h2);
    PhaseCtx c;
    c.codeEW = codeEW; c.wave = wave; c.half = half; c.j = j; c.tid = tid;
    c.last = (tEW == S.T[TP * 32 + j] - 1);
    EwRegs R;
    R.kc[0] = S.lut[wave][half][0][0][codeEW];
    if (FILL > 0) __builtin_amdgcn_sched_barrier(0);
    rd_slots<TL, (FILL > 0 ? FILL : 0), 0>(S, W1, W2, accC, accP, Bf, R, c, h1s, h2);
    if constexpr (FILL == 0) rd_ew_units<TP, 0, EW_NU>(S, R, accP, c);
    if constexpr (FILL < 0) {   // bench diagnosis only (wrong results): no gate math, keep the accumulators live
        if (accC[0][0] + accC[1][5] + accC[2][9] + accC[3][15] == 123.456f) S.Hl[TP * 32 + j][tid & 127] = accC[0][1];
    }
    if constexpr (FILL != 7) __syncthreads();   // FILL 7: bench diagnosis only (racy, wrong results): what the barrier costs
}

template <int FILL>
__global__ __launch_bounds__(256, 1) void rd_lstm_mfma_f16x3_t32_kernel(DevModel d, ReadBatch rb, float *__restrict__ logits,
                                                                        uint8_t *__restrict__ labels) {
    __shared__ Lstm16bSmem S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, j = lane & 31;

    if (tid < 64) {
        const int64_t g = (int64_t)blockIdx.x * 64 + tid;
        int T = 0, lr = 0, orig = -1;
        long long off = 0;
        if (g < rb.n) {
            orig = rb.order ? rb.order[g] : (int)g;
            T = rd_T(rb.steps, orig, rb.max_len);
            lr = rd_T(rb.len, orig, rb.max_len);
            off = rb.off[orig];
        }
        S.T[tid] = T; S.Lr[tid] = lr; S.off[tid] = off; S.orig[tid] = orig;
    }
    if (tid == 0) S.tmax = 0;
    for (int i = tid; i < 2 * 32 * H16STR / 2; i += 256) { (reinterpret_cast<uint32_t *>(&S.H1s[0][0][0]))[i] = 0u; (reinterpret_cast<uint32_t *>(&S.H2[0][0][0]))[i] = 0u; }
    for (int i = tid; i < 64 * HSTR; i += 256) (&S.Hl[0][0])[i] = 0.0f;
    for (int i = tid; i < 2 * 4 * 256; i += 256) (&S.cS[0][0][0])[i] = f32x4{0, 0, 0, 0};
    for (int i = tid; i < 4 * 2 * 4 * 4 * 6 * 4; i += 256) {   // i = ((((w*2 + hf)*4 + a)*4 + b)*6 + code)*4 + gate
        const int gate = i & 3, rest = i >> 2, code = rest % 6, cell = rest / 6;
        const int b = cell & 3, a = (cell >> 2) & 3, hf = (cell >> 4) & 1, w = cell >> 5;
        float v = 0.0f;
        if (code < 5) v = (gate == 2 ? KT : KS) * d.in_lut[code * G4 + gate * HID + 32 * w + 16 * hf + 4 * a + b];
        (reinterpret_cast<float *>(&S.lut[0][0][0][0][0]))[i] = v;
    }
    S.wout[tid >> 7][tid & 127] = d.w_out[(tid >> 7) * 256 + (tid & 127)];
    __syncthreads();
    if (tid < 64) atomicMax(&S.tmax, S.T[tid]);
    rd_stage_codes16b(S, rb, 0);
    if (FILL < 0) {   // diagnosis: realistic (pseudo-random) B operands that are never updated
        for (int i = tid; i < 2 * 32 * H16STR / 2; i += 256) {
            uint32_t x = (uint32_t)i * 2654435761u + blockIdx.x * 40503u;
            x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
            (reinterpret_cast<uint32_t *>(&S.H1s[0][0][0]))[i] = (x & 0x83ff83ffu) | 0x34003400u;   // |v| in [0.25, 0.5), random sign+mantissa
            (reinterpret_cast<uint32_t *>(&S.H2[0][0][0]))[i] = ((x * 31u) & 0x83ff83ffu) | 0x34003400u;
        }
    }

    // ---- resident weights: 4 row-tiles x 8 k-steps x (W1, W2) x 4 registers = 256 registers, all pinned in AGPRs ----
    f16x8 W1[4][8], W2[4][8];
    {
        const uint4 *wp = reinterpret_cast<const uint4 *>(d.wpack16b) + (size_t)wave * (2 * 4 * 8 * 64) + lane;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    const uint4 x = wp[((hl * 4 + a) * 8 + s) * 64];
                    uint4 y;
                    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.x) : "v"(x.x));
                    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.y) : "v"(x.y));
                    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.z) : "v"(x.z));
                    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.w) : "v"(x.w));
                    if (hl == 0) W1[a][s] = __builtin_bit_cast(f16x8, y);
                    else W2[a][s] = __builtin_bit_cast(f16x8, y);
                }
            }
    }
    __syncthreads();
    const int tmax = S.tmax;

    f32x16 X[4], Y[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { X[a][r] = 0.0f; Y[a][r] = 0.0f; }
    int codeY = 5;   // code of (tile 1, step t-1): zero row before the first step

    for (int t = 0; t <= tmax; ++t) {
        const int tc = t < tmax ? t : 0;
        const uint8_t *crow = &S.codes[(tc / TC16) & 1][tc % TC16][0];
        const int codeX = crow[j];            // (tile 0, step t): consumed by phase B's gate math
        const int codeYn = crow[32 + j];      // (tile 1, step t): consumed by the next iteration's phase A
        // phase A: MFMAs of (tile 0, t) -> X ; gate math of (tile 1, t-1) <- Y
        rd_phase_t32<0, FILL>(S, W1, W2, X, Y, t - 1, codeY, wave, half, j, tid);
        if (t < tmax) {
            // next code chunk: its buffer was last read by the gate math of phase A above (step t-1)
            if ((t % TC16) == 0) {
                const int chunk = t / TC16 + 1;
                if (chunk * TC16 < tmax + 1) rd_stage_codes16b(S, rb, chunk);
            }
            // phase B: MFMAs of (tile 1, t) -> Y ; gate math of (tile 0, t) <- X
            rd_phase_t32<1, FILL>(S, W1, W2, Y, X, t, codeX, wave, half, j, tid);
        }
        codeY = codeYn;
    }

    rd_fc_epilogue(
        64, [&](int row, int u) { return S.Hl[row][u] * (1.0f / H_SCALE); }, S.T, S.Lr, S.off, S.orig, &S.wout[0][0], d, rb, logits,
        labels);
}

// ------------------------------------------------------------------------------------------------
// rd_lstm_mfma_f16x3_w8_kernel - the split-precision recurrence with TWO waves per SIMD.
//
// The t32 kernel above runs one wave per SIMD: its gate math costs +31 % over the matrix-pipe floor because a single wave
// cannot issue VALU work while it waits for the MFMA pipe (profiles/README.md). Here a workgroup has 8 waves; wave w owns
// 16 hidden units (its W_hh slice = 128 AGPRs, accumulators and gate math in <= 128 VGPRs) and the two waves that share a
// SIMD run one stage out of phase: while waves 0-3 (group A) issue the MFMAs of a (tile, step), waves 4-7 (group B) do
// the gate math of their previous MFMAs, and vice versa - the hardware interleaves the two instruction streams.
//   stage s:  group A: s even -> MFMA(q), s odd -> GATES(q),  q = s/2        (q = 2*step + tile)
//             group B: s odd  -> MFMA(q), s even -> GATES(q), q = (s-1)/2
// h is double buffered per tile (by step parity): GATES(tile,t) writes h(t+1) while the other group may still be reading
// h(t) in its MFMA(tile,t). One workgroup barrier per stage.
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) Lstm16cSmem {
    _Float16 H1s[2][2][32][H16STR];   // [tile][step parity][read][unit]  2^11 h_hi
    _Float16 H2[2][2][32][H16STR];    // 2^11 h - H1s
    float Hl[64][HSTR];               // h captured at t == T-1
    f32x4 cS[2][2][512];              // cell state [tile][row-tile a][tid]
    f32x4 lut[8][2][2][4][6];         // [wave][half][a][b][code]
    float wout[2][HID];
    uint8_t codes[2][TC16][64];
    int T[64];
    int Lr[64];
    long long off[64];
    int orig[64];
    int tmax;
};

__device__ __forceinline__ void rd_stage_codes16c(Lstm16cSmem &S, const ReadBatch &rb, int chunk) {
    const int t0 = chunk * TC16;
    uint8_t(*dst)[64] = S.codes[chunk & 1];
    for (int idx = threadIdx.x; idx < 64 * TC16; idx += 512) {
        const int row = idx / TC16, tt = idx % TC16, t = t0 + tt;
        int code = 4;
        if (t < S.Lr[row]) code = rd_code(rb.arena[S.off[row] + t]);
        dst[tt][row] = (uint8_t)code;
    }
}

__global__ __launch_bounds__(512, 2) void rd_lstm_mfma_f16x3_w8_kernel(DevModel d, ReadBatch rb, float *__restrict__ logits,
                                                                       uint8_t *__restrict__ labels) {
    __shared__ Lstm16cSmem S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;               // 0: waves 0-3, 1: waves 4-7 (the second wave of each SIMD)
    const int half = lane >> 5, j = lane & 31;

    if (tid < 64) {
        const int64_t g = (int64_t)blockIdx.x * 64 + tid;
        int T = 0, lr = 0, orig = -1;
        long long off = 0;
        if (g < rb.n) {
            orig = rb.order ? rb.order[g] : (int)g;
            T = rd_T(rb.steps, orig, rb.max_len);
            lr = rd_T(rb.len, orig, rb.max_len);
            off = rb.off[orig];
        }
        S.T[tid] = T; S.Lr[tid] = lr; S.off[tid] = off; S.orig[tid] = orig;
    }
    if (tid == 0) S.tmax = 0;
    for (int i = tid; i < 2 * 2 * 2 * 32 * H16STR / 2; i += 512) (reinterpret_cast<uint32_t *>(&S.H1s[0][0][0][0]))[i] = 0u;
    for (int i = tid; i < 64 * HSTR; i += 512) (&S.Hl[0][0])[i] = 0.0f;
    for (int i = tid; i < 2 * 2 * 512; i += 512) (&S.cS[0][0][0])[i] = f32x4{0, 0, 0, 0};
    for (int i = tid; i < 8 * 2 * 2 * 4 * 6 * 4; i += 512) {   // i = ((((w*2 + hf)*2 + a)*4 + b)*6 + code)*4 + gate
        const int gate = i & 3, rest = i >> 2, code = rest % 6, cell = rest / 6;
        const int b = cell & 3, a = (cell >> 2) & 1, hf = (cell >> 3) & 1, w = cell >> 4;
        float v = 0.0f;
        if (code < 5) v = (gate == 2 ? KT : KS) * d.in_lut[code * G4 + gate * HID + 16 * w + 8 * hf + 4 * a + b];
        (reinterpret_cast<float *>(&S.lut[0][0][0][0][0]))[i] = v;
    }
    if (tid < 256) S.wout[tid >> 7][tid & 127] = d.w_out[(tid >> 7) * 256 + (tid & 127)];
    __syncthreads();
    if (tid < 64) atomicMax(&S.tmax, S.T[tid]);
    rd_stage_codes16c(S, rb, 0);

    // ---- resident weights: 2 row-tiles x 8 k-steps x (W1, W2) x 4 registers = 128 registers, pinned in AGPRs ----
    f16x8 W1[2][8], W2[2][8];
    {
        const uint4 *wp = reinterpret_cast<const uint4 *>(d.wpack16c) + (size_t)wave * (2 * 2 * 8 * 64) + lane;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    const uint4 x = wp[((hl * 2 + a) * 8 + s) * 64];
                    uint4 y;
                    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.x) : "v"(x.x));
                    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.y) : "v"(x.y));
                    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.z) : "v"(x.z));
                    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.w) : "v"(x.w));
                    if (hl == 0) W1[a][s] = __builtin_bit_cast(f16x8, y);
                    else W2[a][s] = __builtin_bit_cast(f16x8, y);
                }
            }
    }
    __syncthreads();
    const int tmax = S.tmax;
    const int nq = 2 * tmax;                    // phases q = 2*step + tile
    const int nstage = 2 * nq + 1;              // group B trails group A by one stage

    f32x16 acc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    const int boff = j * H16STR + 8 * half;     // B fragment: row j, k = 16s + 8half + e
    const int woff = j * H16STR + 16 * wave + 8 * half;

    for (int sg = 0; sg < nstage; ++sg) {
        // next code chunk: all reads of chunk k-1 are done by stage 4 k TC16 (see header); first use of chunk k+1 is far later
        if (sg >= 2 && ((sg - 2) % (4 * TC16)) == 0) {
            const int chunk = (sg - 2) / (4 * TC16) + 1;
            if (chunk * TC16 < tmax + 1) rd_stage_codes16c(S, rb, chunk);
        }
        const int sl = sg - group;              // this wave's local stage
        const int q = sl >> 1;
        if (sl >= 0 && q < nq) {
            const int tile = q & 1, t = q >> 1;
            if ((sl & 1) == 0) {
                // ---- MFMA(q): 48 x v_mfma_f32_32x32x16_f16, two accumulators alternating --------------------------
                const _Float16 *h1s = &S.H1s[tile][t & 1][0][0] + boff, *h2 = &S.H2[tile][t & 1][0][0] + boff;
                f16x8 bs = *reinterpret_cast<const f16x8 *>(h1s), bl = *reinterpret_cast<const f16x8 *>(h2);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    f16x8 ns = bs, nl = bl;
                    if (s < 7) {
                        ns = *reinterpret_cast<const f16x8 *>(h1s + 16 * (s + 1));
                        nl = *reinterpret_cast<const f16x8 *>(h2 + 16 * (s + 1));
                    }
                    if (s == 0) {
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            f32x16 z;
#pragma unroll
                            for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                            acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W1[a][s], bs, z, 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W1[a][s], bs, acc[a], 0, 0, 0);
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W2[a][s], bs, acc[a], 0, 0, 0);
#pragma unroll
                    for (int a = 0; a < 2; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(W1[a][s], bl, acc[a], 0, 0, 0);
                    bs = ns; bl = nl;
                }
            } else {
                // ---- GATES(q): lane (read j) x units 16w + 8half + 4a + b ; writes h(t+1) into the other buffer -----
                const int code = S.codes[(t / TC16) & 1][t % TC16][tile * 32 + j];
                const bool last = (t == S.T[tile * 32 + j] - 1);
                _Float16 *o1 = &S.H1s[tile][(t + 1) & 1][0][0] + woff, *o2 = &S.H2[tile][(t + 1) & 1][0][0] + woff;
                f16x8 v1, v2;
                float hsv[8];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    f32x4 cs = S.cS[tile][a][tid];
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const f32x4 kc = S.lut[wave][half][a][b][code];
                        const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc[a][4 * b + 0], KS / G_SCALE, kc[0])));
                        const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc[a][4 * b + 1], KS / G_SCALE, kc[1])));
                        const float gr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc[a][4 * b + 2], KT / G_SCALE, kc[2])));
                        const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc[a][4 * b + 3], KS / G_SCALE, kc[3])));
                        const float cn = __builtin_fmaf(fg, cs[b], ig * __builtin_fmaf(-2.0f, gr, 1.0f));
                        cs[b] = cn;
                        const float yc = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cn * KT));
                        float hs = og * __builtin_fmaf(-2.0f * H_SCALE, yc, H_SCALE);     // 2^11 h
                        asm volatile("" : "+v"(hs));             // no v_fma_mix fusion (see rd_ew_unit)
                        const _Float16 p16 = (_Float16)hs;
                        float res = hs - (float)p16;
                        asm volatile("" : "+v"(res));
                        v1[4 * a + b] = p16;
                        v2[4 * a + b] = (_Float16)res;
                        hsv[4 * a + b] = hs;
                    }
                    S.cS[tile][a][tid] = cs;
                }
                *reinterpret_cast<f16x8 *>(o1) = v1;
                *reinterpret_cast<f16x8 *>(o2) = v2;
                if (last) {
                    float *hl = &S.Hl[tile * 32 + j][16 * wave + 8 * half];
                    *reinterpret_cast<f32x4 *>(hl) = f32x4{hsv[0], hsv[1], hsv[2], hsv[3]} * (1.0f / H_SCALE);
                    *reinterpret_cast<f32x4 *>(hl + 4) = f32x4{hsv[4], hsv[5], hsv[6], hsv[7]} * (1.0f / H_SCALE);
                }
            }
        }
        __syncthreads();
    }

    rd_fc_epilogue(
        64, [&](int row, int u) { return S.Hl[row][u]; }, S.T, S.Lr, S.off, S.orig, &S.wout[0][0], d, rb, logits, labels);
}

// ------------------------------------------------------------------------------------------------
// standalone encoders (reference tensor layouts). HBM-bound streaming kernels.
// ------------------------------------------------------------------------------------------------
// All three give one workgroup a block of ENC_R reads whose output range is contiguous, stage the reads' offsets and
// lengths in LDS once, and let consecutive lanes write consecutive 4-/16-byte pieces of that range, so every wave store
// covers whole cache lines whatever the read length is.
constexpr int ENC_R = 64;

__device__ __forceinline__ f32x4 rd_onehot(int code) {
    return f32x4{code == 0 ? 1.f : 0.f, code == 1 ? 1.f : 0.f, code == 2 ? 1.f : 0.f, code == 3 ? 1.f : 0.f};
}

// codes[n][stride] u8 (4 = pad / not ACGTU): 4 output bytes per lane and iteration, one aligned dword store when VEC
template <bool VEC>
__global__ __launch_bounds__(256) void rd_encode_codes_kernel(const uint8_t *__restrict__ arena, const int64_t *__restrict__ off,
                                                              const int32_t *__restrict__ len, int64_t n, int max_len, int stride,
                                                              uint8_t *__restrict__ codes) {
    __shared__ int64_t s_off[ENC_R];
    __shared__ int s_T[ENC_R];
    for (int64_t r0 = (int64_t)blockIdx.x * ENC_R; r0 < n; r0 += (int64_t)gridDim.x * ENC_R) {
        const int R = (int)(n - r0 < ENC_R ? n - r0 : ENC_R);
        __syncthreads();
        if ((int)threadIdx.x < R) {
            s_off[threadIdx.x] = off[r0 + threadIdx.x];
            s_T[threadIdx.x] = rd_T(len, r0 + threadIdx.x, max_len);
        }
        __syncthreads();
        const unsigned total = (unsigned)R * (unsigned)stride;
        uint8_t *dst = codes + (size_t)r0 * stride;
        for (unsigned e = threadIdx.x * 4; e < total; e += 1024) {
            unsigned i = e / (unsigned)stride, j = e - i * (unsigned)stride;
            uint32_t w = 0;
            if (j + 4 <= (unsigned)s_T[i]) {   // four bases of one read: one (unaligned) dword load
                uint32_t raw;
                __builtin_memcpy(&raw, arena + s_off[i] + j, 4);
#pragma unroll
                for (int b = 0; b < 4; ++b) w |= (unsigned)rd_code((raw >> (8 * b)) & 0xff) << (8 * b);
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    unsigned c = 4;
                    if (e + b < total && j < (unsigned)s_T[i]) c = (unsigned)rd_code(arena[s_off[i] + j]);
                    w |= c << (8 * b);
                    if (++j == (unsigned)stride) {
                        j = 0;
                        ++i;
                    }
                }
            }
            if (VEC && e + 4 <= total) {
                *(uint32_t *)(dst + e) = w;
            } else {
                for (int b = 0; b < 4 && e + b < total; ++b) dst[e + b] = (uint8_t)(w >> (8 * b));
            }
        }
    }
}

// onehot[n][max_len][4] fp32 (encode_variable_len_read): one 16-byte store per lane, consecutive lanes consecutive rows
__global__ __launch_bounds__(256) void rd_encode_onehot_padded_kernel(const uint8_t *__restrict__ arena, const int64_t *__restrict__ off,
                                                                      const int32_t *__restrict__ len, int64_t n, int max_len,
                                                                      f32x4 *__restrict__ out) {
    __shared__ int64_t s_off[ENC_R];
    __shared__ int s_T[ENC_R];
    const unsigned L = (unsigned)max_len, dq = 256u / L, dr = 256u % L;
    for (int64_t r0 = (int64_t)blockIdx.x * ENC_R; r0 < n; r0 += (int64_t)gridDim.x * ENC_R) {
        const int R = (int)(n - r0 < ENC_R ? n - r0 : ENC_R);
        __syncthreads();
        if ((int)threadIdx.x < R) {
            s_off[threadIdx.x] = off[r0 + threadIdx.x];
            s_T[threadIdx.x] = rd_T(len, r0 + threadIdx.x, max_len);
        }
        __syncthreads();
        const unsigned total = (unsigned)R * L;
        f32x4 *dst = out + (size_t)r0 * L;
        unsigned i = threadIdx.x / L, j = threadIdx.x - i * L;
        for (unsigned e = threadIdx.x; e < total; e += 256) {
            int code = 4;
            if (j < (unsigned)s_T[i]) code = rd_code(arena[s_off[i] + j]);
            __builtin_nontemporal_store(rd_onehot(code), dst + e);
            i += dq;
            j += dr;
            if (j >= L) {
                j -= L;
                ++i;
            }
        }
    }
}

// PackedSequence.data [sum T][4]: row(t, j) = cum[t] + j, cum[t] = sum_{t'<t} batch_sizes[t'], j = position of the read in
// the length-sorted order. A workgroup takes ENC_R consecutive sorted reads and walks the timesteps in chunks of PK_TC:
// each read's bases are loaded once, as contiguous bytes, into an LDS tile; the tile is then written out transposed, one
// timestep per wave instruction = 64 consecutive 16-byte rows. cum[] is carried from chunk to chunk.
constexpr int PK_TC = 128;
__global__ __launch_bounds__(256) void rd_pack_onehot_kernel(const uint8_t *__restrict__ arena, const int64_t *__restrict__ off,
                                                             const int32_t *__restrict__ len, int64_t n, int max_len,
                                                             const int64_t *__restrict__ sorted_idx,
                                                             const int64_t *__restrict__ batch_sizes, f32x4 *__restrict__ data) {
    __shared__ int64_t s_off[ENC_R];
    __shared__ int s_T[ENC_R];
    __shared__ int64_t s_bs[PK_TC], s_cum[PK_TC], s_scan[2][PK_TC];
    __shared__ uint8_t s_code[ENC_R][PK_TC + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * ENC_R;
    const int R = (int)(n - j0 < ENC_R ? n - j0 : ENC_R);
    if (tid < ENC_R) {
        int T = 0;
        int64_t o = 0;
        if (tid < R) {
            const int64_t i = sorted_idx[j0 + tid];
            o = off[i];
            T = rd_T(len, i, max_len);
        }
        s_off[tid] = o;
        s_T[tid] = T;
    }
    __syncthreads();
    const int Tmax = s_T[0];   // sorted by length, descending: the first read of the block is its longest
    int64_t carry = 0;         // cum[t0]
    for (int t0 = 0; t0 < Tmax; t0 += PK_TC) {
        const int TC = Tmax - t0 < PK_TC ? Tmax - t0 : PK_TC;
        // batch_sizes of this chunk and their exclusive prefix sums (Hillis-Steele over PK_TC entries)
        if (tid < PK_TC) {
            const int64_t b = tid < TC ? batch_sizes[t0 + tid] : 0;
            s_bs[tid] = b;
            s_scan[0][tid] = b;
        }
        __syncthreads();
        int cur = 0;
        for (int d = 1; d < PK_TC; d <<= 1) {
            if (tid < PK_TC) s_scan[cur ^ 1][tid] = s_scan[cur][tid] + (tid >= d ? s_scan[cur][tid - d] : 0);
            cur ^= 1;
            __syncthreads();
        }
        if (tid < PK_TC) s_cum[tid] = carry + s_scan[cur][tid] - s_bs[tid];
        const int64_t chunk_total = s_scan[cur][PK_TC - 1];
        // the reads' bases of this chunk -> LDS tile (16 reads per wave, 64 consecutive bytes per wave load)
        for (int r = wave * 16; r < wave * 16 + 16; ++r) {
            const int T = s_T[r];
            const uint8_t *src = arena + s_off[r] + t0;
            for (int tt = lane; tt < TC; tt += 64) s_code[r][tt] = (uint8_t)(t0 + tt < T ? rd_code(src[tt]) : 4);
        }
        __syncthreads();
        for (int tt = wave; tt < TC; tt += 4) {
            const int64_t bs = s_bs[tt];
            if (j0 + lane < bs) __builtin_nontemporal_store(rd_onehot(s_code[lane][tt]), data + s_cum[tt] + j0 + lane);
        }
        carry += chunk_total;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// label logic
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rd_block_count3(unsigned c0, unsigned c1, unsigned c2, uint64_t *counts) {
    for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_down(c0, o); c1 += __shfl_down(c1, o); c2 += __shfl_down(c2, o); }
    __shared__ unsigned sh[3];
    if (threadIdx.x < 3) sh[threadIdx.x] = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sh[0], c0); atomicAdd(&sh[1], c1); atomicAdd(&sh[2], c2); }
    __syncthreads();
    if (threadIdx.x < 3 && sh[threadIdx.x]) atomicAdd((unsigned long long *)&counts[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
}

__device__ __forceinline__ int rd_fuse(float2 a, float2 b, int mode) {   // detect.py:616-663
    const int la = a.y > a.x, lb = b.y > b.x;
    if (mode == RD_ENSURE_RRNA) return la & lb;
    if (mode == RD_ENSURE_NORRNA) return la | lb;
    if (mode == RD_ENSURE_BOTH) return (la == lb) ? la : -1;
    return __fadd_rn(a.y, b.y) > __fadd_rn(a.x, b.x) ? 1 : 0;   // argmax(r1_outs + r2_outs), :657
}

// VEC: two pairs per lane and iteration (16-byte loads; needs 16-byte aligned logits and 2-byte aligned labels)
template <bool VEC>
__global__ __launch_bounds__(256) void rd_pair_fuse_kernel(const float2 *__restrict__ l1, const float2 *__restrict__ l2, int64_t n, int mode,
                                                           int8_t *__restrict__ out, uint64_t *__restrict__ counts) {
    unsigned c0 = 0, c1 = 0, c2 = 0;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    if (VEC) {
        const int64_t n2 = n >> 1;
        for (int64_t i = gtid; i < n2; i += gsz) {
            const f32x4 a = ((const f32x4 *)l1)[i], b = ((const f32x4 *)l2)[i];
            const int f0 = rd_fuse(float2{a[0], a[1]}, float2{b[0], b[1]}, mode), f1 = rd_fuse(float2{a[2], a[3]}, float2{b[2], b[3]}, mode);
            ((uint16_t *)out)[i] = (uint16_t)((f0 & 0xff) | ((f1 & 0xff) << 8));
            c0 += (f0 == 0) + (f1 == 0); c1 += (f0 == 1) + (f1 == 1); c2 += (f0 < 0) + (f1 < 0);
        }
        if ((n & 1) && gtid == 0) {
            const int f = rd_fuse(l1[n - 1], l2[n - 1], mode);
            out[n - 1] = (int8_t)f;
            c0 += f == 0; c1 += f == 1; c2 += f < 0;
        }
    } else {
        for (int64_t i = gtid; i < n; i += gsz) {
            const int f = rd_fuse(l1[i], l2[i], mode);
            out[i] = (int8_t)f;
            c0 += f == 0; c1 += f == 1; c2 += f < 0;
        }
    }
    if (counts) rd_block_count3(c0, c1, c2, counts);
}

// VEC: 16 labels per lane and iteration (needs a 16-byte aligned pointer)
template <bool VEC>
__global__ __launch_bounds__(256) void rd_count_kernel(const uint8_t *__restrict__ labels, int64_t n, uint64_t *__restrict__ counts) {
    unsigned c0 = 0, c1 = 0;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    if (VEC) {
        const int64_t n16 = n >> 4;
        for (int64_t i = gtid; i < n16; i += gsz) {
            const u32x4 v = ((const u32x4 *)labels)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const unsigned f = (v[k] >> (8 * b)) & 0xff;
                    c0 += f == 0; c1 += f == 1;
                }
        }
        done = n16 << 4;
    }
    for (int64_t i = done + gtid; i < n; i += gsz) {
        const int f = labels[i];
        c0 += f == 0; c1 += f == 1;
    }
    rd_block_count3(c0, c1, 0, counts);
}

// ------------------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------------------
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct SortPlan {
    int nblk;
    size_t hist_bytes, order_bytes, lenstart_bytes, steps_bytes, total;
};
inline SortPlan sort_plan(int64_t n, int max_len) {
    SortPlan p;
    p.nblk = (int)((n + SORT_ITEMS - 1) / SORT_ITEMS);
    if (p.nblk < 1) p.nblk = 1;
    p.hist_bytes = align_up((size_t)(max_len + 1) * p.nblk * sizeof(uint32_t), 256);
    p.order_bytes = align_up((size_t)(n > 0 ? n : 1) * sizeof(int32_t), 256);
    p.lenstart_bytes = align_up((size_t)(max_len + 1) * sizeof(int64_t) * 2, 256);   // len_start + cum scratch
    p.steps_bytes = align_up((size_t)(n > 0 ? n : 1) * sizeof(int32_t), 256);
    p.total = p.hist_bytes + p.order_bytes + p.lenstart_bytes + p.steps_bytes;
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
    hipLaunchKernelGGL(rd_len_scan_kernel, dim3(1), dim3(SORT_BLOCK), 0, st, hist, max_len, p.nblk, len_start, batch_sizes,
                       total_steps);
    hipLaunchKernelGGL(rd_len_scatter_kernel, dim3(p.nblk), dim3(SORT_BLOCK), sh, st, seq_len, n, max_len, p.nblk, hist, order,
                       sorted_idx, unsorted_idx);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

// steps[] + order[] for rd_classify: steps kernel (with histogram) -> bucket starts -> scatter
int run_steps_and_buckets(const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n, int max_len, int sem,
                          void *workspace, size_t wbytes, int32_t *&steps, int32_t *&order, hipStream_t st) {
    SortPlan p = sort_plan(n, max_len);
    if (wbytes < p.total) RD_FAIL(RD_E_WORKSPACE, "workspace too small: %zu < %zu", wbytes, p.total);
    char *w = (char *)workspace;
    uint32_t *ghist = (uint32_t *)w;                                          // (max_len+1) u32 fit in hist_bytes
    order = (int32_t *)(w + p.hist_bytes);
    uint32_t *cursor = (uint32_t *)(w + p.hist_bytes + p.order_bytes);        // (max_len+1) u32 fit in lenstart_bytes
    steps = (int32_t *)(w + p.total - p.steps_bytes);
    const size_t sh = (size_t)(max_len + 1) * sizeof(uint32_t);
    RD_HIP(hipMemsetAsync(ghist, 0, sh, st));
    // one workgroup per CU at most: every workgroup ends with one global atomic per non-empty bin, and with fixed-length
    // reads they all hit the same bin
    int64_t nb = (n + 255) / 256;
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(rd_steps_kernel, dim3((unsigned)nb), dim3(256), sh, st, arena, seq_off, seq_len, n, max_len, sem, steps, ghist);
    hipLaunchKernelGGL(rd_bucket_scan_kernel, dim3(1), dim3(256), 0, st, ghist, max_len, cursor);
    int64_t nbs = (n + BK_ITEMS - 1) / BK_ITEMS;
    if (nbs > 256) nbs = 256;
    hipLaunchKernelGGL(rd_bucket_scatter_kernel, dim3((unsigned)nbs), dim3(256), sh, st, steps, n, max_len, cursor, order);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

const char *rd_last_error(void) { return g_err; }
const char *rd_version(void) { return "ribodetector_amd 0.1.0 (gfx950)"; }

int rd_model_create(const rd_weights *w, int device, rd_model **out) {
    if (!w || !out) RD_FAIL(RD_E_INVALID, "rd_model_create: null argument");
    if (w->input_size != 4 || w->hidden_size != HID || w->num_classes != 2)
        RD_FAIL(RD_E_UNSUPPORTED, "rd_model_create: kernels cover input_size=4, hidden_size=128, num_classes=2 (got %d,%d,%d)",
                w->input_size, w->hidden_size, w->num_classes);
    const float *src[10] = {w->w_ih, w->w_hh, w->b_ih, w->b_hh, w->w_ih_r, w->w_hh_r, w->b_ih_r, w->b_hh_r, w->w_out, w->b_out};
    const int offs[11] = {OFF_WIH, OFF_WHH, OFF_BIH, OFF_BHH, OFF_WIHR, OFF_WHHR, OFF_BIHR, OFF_BHHR, OFF_WOUT, OFF_BOUT, RAW_FLOATS};
    for (int i = 0; i < 10; ++i)
        if (!src[i]) RD_FAIL(RD_E_INVALID, "rd_model_create: weight pointer %d is null", i);
    RD_HIP(hipSetDevice(device));
    rd_model *m = new rd_model();
    memset(m, 0, sizeof(*m));
    m->device = device;
    m->variant = RD_VARIANT_MFMA_F16X3_T32;
    float *host = new float[RAW_FLOATS];
    for (int i = 0; i < 10; ++i) memcpy(host + offs[i], src[i], sizeof(float) * (size_t)(offs[i + 1] - offs[i]));
    hipError_t e = hipSuccess;
    auto A = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    A((void **)&m->d.raw, sizeof(float) * RAW_FLOATS);
    A((void **)&m->d.wpack32, sizeof(float) * 4 * 8 * 32 * 64);
    A((void **)&m->d.wt_hh, sizeof(float) * HID * G4);
    A((void **)&m->d.wpack16, sizeof(uint16_t) * 4 * 2 * 8 * 4 * 64 * 8);
    A((void **)&m->d.wpack16b, sizeof(uint16_t) * 4 * 2 * 4 * 8 * 64 * 8);
    A((void **)&m->d.wpack16c, sizeof(uint16_t) * 8 * 2 * 2 * 8 * 64 * 8);
    A((void **)&m->d.in_lut, sizeof(float) * 5 * G4);
    A((void **)&m->d.rev_lut, sizeof(float) * 10);
    A((void **)&m->d.rev_tab, sizeof(float) * (size_t)MAX_LEN_LIMIT * 10);
    A((void **)&m->d.w_out, sizeof(float) * 512);
    A((void **)&m->d.b_out, sizeof(float) * 2);
    if (e == hipSuccess) e = hipMemcpy(m->d.raw, host, sizeof(float) * RAW_FLOATS, hipMemcpyHostToDevice);
    delete[] host;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rd_prep_kernel, dim3(64), dim3(256), 0, 0, m->d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "rd_model_create: %s", hipGetErrorString(e));
        rd_model_destroy(m);
        return RD_E_HIP;
    }
    *out = m;
    return RD_OK;
}

void rd_model_destroy(rd_model *m) {
    if (!m) return;
    hipSetDevice(m->device);
    hipFree(m->d.raw); hipFree(m->d.wpack32); hipFree(m->d.wt_hh); hipFree(m->d.in_lut);
    hipFree(m->d.rev_lut); hipFree(m->d.rev_tab); hipFree(m->d.w_out); hipFree(m->d.b_out);
    if (m->d.wpack16) hipFree(m->d.wpack16);
    if (m->d.wpack16b) hipFree(m->d.wpack16b);
    if (m->d.wpack16c) hipFree(m->d.wpack16c);
    for (int i = 0; i < 2 * 512; ++i)
        if (m->prof_ev[i]) hipEventDestroy(m->prof_ev[i]);
    delete m;
}

int rd_set_variant(rd_model *m, int variant) {
    if (!m) RD_FAIL(RD_E_INVALID, "rd_set_variant: null model");
    if (variant == RD_VARIANT_AUTO) variant = RD_VARIANT_MFMA_F16X3_T32;
    if (variant != RD_VARIANT_MFMA_F32 && variant != RD_VARIANT_SIMPLE && variant != RD_VARIANT_MFMA_F16X3 && variant != RD_VARIANT_MFMA_F16X3_T32 && variant != 5 && !(variant >= 10 && variant <= 42))
        RD_FAIL(RD_E_UNSUPPORTED, "rd_set_variant: variant %d not available in this build", variant);
    m->variant = variant;
    return RD_OK;
}

int rd_set_semantics(rd_model *m, int semantics) {
    if (!m) RD_FAIL(RD_E_INVALID, "rd_set_semantics: null model");
    if (semantics != RD_SEM_PACKED && semantics != RD_SEM_PADDED) RD_FAIL(RD_E_INVALID, "rd_set_semantics: unknown semantics %d", semantics);
    m->semantics = semantics;
    return RD_OK;
}

size_t rd_classify_workspace_bytes(int64_t n, int32_t max_len) {
    if (n < 0 || max_len < 1) return 0;
    return sort_plan(n, max_len).total;
}

int rd_profile_enable(rd_model *m, int enable) {
    if (!m) RD_FAIL(RD_E_INVALID, "rd_profile_enable: null model");
    m->prof_enabled = enable;
    m->prof_count = 0;
    m->prof_ms_accum = 0;
    m->prof_launches_accum = 0;
    return RD_OK;
}

static int rd_profile_drain(rd_model *m) {
    for (int i = 0; i < m->prof_count; ++i) {
        float ms = 0;
        RD_HIP(hipEventSynchronize(m->prof_ev[2 * i + 1]));
        RD_HIP(hipEventElapsedTime(&ms, m->prof_ev[2 * i], m->prof_ev[2 * i + 1]));
        m->prof_ms_accum += ms;
        m->prof_launches_accum += 1;
    }
    m->prof_count = 0;
    return RD_OK;
}

int rd_profile_read(rd_model *m, int64_t *launches, double *total_ms) {
    if (!m) RD_FAIL(RD_E_INVALID, "rd_profile_read: null model");
    int rc = rd_profile_drain(m);
    if (rc) return rc;
    if (launches) *launches = m->prof_launches_accum;
    if (total_ms) *total_ms = m->prof_ms_accum;
    return RD_OK;
}

int rd_classify(const rd_model *cm, const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n,
                int32_t max_len, float *logits, uint8_t *labels, void *workspace, size_t workspace_bytes, void *stream) {
    rd_model *m = const_cast<rd_model *>(cm);
    if (!m || !logits) RD_FAIL(RD_E_INVALID, "rd_classify: null model or logits");
    if (n < 0 || n > 0x7fffffffLL) RD_FAIL(RD_E_INVALID, "rd_classify: n=%lld out of range", (long long)n);
    if (max_len < 1 || max_len > MAX_LEN_LIMIT) RD_FAIL(RD_E_INVALID, "rd_classify: max_len=%d out of range [1,%d]", max_len, MAX_LEN_LIMIT);
    if (n == 0) return RD_OK;
    if (!arena || !seq_off || !seq_len || !workspace) RD_FAIL(RD_E_INVALID, "rd_classify: null input pointer");
    hipStream_t st = (hipStream_t)stream;
    const SortPlan sp = sort_plan(n, max_len);
    if (workspace_bytes < sp.total) RD_FAIL(RD_E_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes, sp.total);
    if (m->semantics == RD_SEM_PADDED && m->rev_tab_len != max_len) {
        hipLaunchKernelGGL(rd_revtab_kernel, dim3(1), dim3(512), 0, st, m->d, max_len);
        m->rev_tab_len = max_len;
    }
    int32_t *steps = nullptr, *order = nullptr;
    int rc = run_steps_and_buckets(arena, seq_off, seq_len, n, max_len, m->semantics, workspace, workspace_bytes, steps, order, st);
    if (rc) return rc;
    ReadBatch rb{arena, seq_off, seq_len, steps, order, n, max_len, m->semantics, m->d.rev_tab};
    hipEvent_t *ev = nullptr;
    if (m->prof_enabled) {
        if (m->prof_count == 512) { rc = rd_profile_drain(m); if (rc) return rc; }
        ev = &m->prof_ev[2 * m->prof_count];
        for (int i = 0; i < 2; ++i)
            if (!ev[i]) RD_HIP(hipEventCreate(&ev[i]));
        RD_HIP(hipEventRecord(ev[0], st));
    }
    if (m->variant == RD_VARIANT_SIMPLE) {
        const int64_t nwg = (n + SB - 1) / SB;
        hipLaunchKernelGGL(rd_lstm_simple_kernel, dim3((unsigned)nwg), dim3(512), 0, st, m->d, rb, logits, labels);
    } else {
        const int64_t nwg = (n + BT - 1) / BT;
        switch (m->variant) {
        case 10: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<0, 0>), dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case 11: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<1, 0>), dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case 12: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<0, 1>), dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case 13: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<1, 1>), dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case 20: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<1, 0, 1>), dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case 21: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<1, 0, 2>), dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case 22: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<1, 0, 3>), dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case 23: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<1, 0, 4>), dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case RD_VARIANT_MFMA_F16X3: hipLaunchKernelGGL(rd_lstm_mfma_f16x3_kernel<2>, dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case RD_VARIANT_MFMA_F16X3_T32: hipLaunchKernelGGL(rd_lstm_mfma_f16x3_t32_kernel<6>, dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case 5: hipLaunchKernelGGL(rd_lstm_mfma_f16x3_w8_kernel, dim3((unsigned)nwg), dim3(512), 0, st, m->d, rb, logits, labels); break;
        case 42: hipLaunchKernelGGL(rd_lstm_mfma_f16x3_t32_kernel<7>, dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case 41: hipLaunchKernelGGL(rd_lstm_mfma_f16x3_t32_kernel<-1>, dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case 40: hipLaunchKernelGGL(rd_lstm_mfma_f16x3_t32_kernel<0>, dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case 30: hipLaunchKernelGGL(rd_lstm_mfma_f16x3_kernel<0>, dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case 31: hipLaunchKernelGGL(rd_lstm_mfma_f16x3_kernel<3>, dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case 32: hipLaunchKernelGGL(rd_lstm_mfma_f16x3_kernel<4>, dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        case RD_VARIANT_MFMA_F32:
        default: hipLaunchKernelGGL((rd_lstm_mfma_f32_kernel<1, 0>), dim3((unsigned)nwg), dim3(256), 0, st, m->d, rb, logits, labels); break;
        }
    }
    RD_HIP(hipGetLastError());
    if (ev) { RD_HIP(hipEventRecord(ev[1], st)); m->prof_count++; }
    return RD_OK;
}

int rd_pair_fuse(const float *logits1, const float *logits2, int64_t n, int32_t ensure_mode, int8_t *pair_labels,
                 uint64_t *counts, void *stream) {
    if (n < 0 || ensure_mode < 0 || ensure_mode > 3) RD_FAIL(RD_E_INVALID, "rd_pair_fuse: bad n or ensure_mode");
    if (n == 0) return RD_OK;
    if (!logits1 || !logits2 || !pair_labels) RD_FAIL(RD_E_INVALID, "rd_pair_fuse: null pointer");
    // few, fat workgroups: every workgroup ends with three global atomics on the same counters
    int64_t nb = (n + 2047) / 2048;
    if (nb > 1024) nb = 1024;
    const bool vec = (((uintptr_t)logits1 | (uintptr_t)logits2) & 15) == 0 && ((uintptr_t)pair_labels & 1) == 0;
    if (vec)
        hipLaunchKernelGGL(rd_pair_fuse_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float2 *)logits1,
                           (const float2 *)logits2, n, ensure_mode, pair_labels, counts);
    else
        hipLaunchKernelGGL(rd_pair_fuse_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float2 *)logits1,
                           (const float2 *)logits2, n, ensure_mode, pair_labels, counts);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_count_labels(const uint8_t *labels, int64_t n, uint64_t *counts, void *stream) {
    if (n < 0) RD_FAIL(RD_E_INVALID, "rd_count_labels: bad n");
    if (n == 0) return RD_OK;
    if (!labels || !counts) RD_FAIL(RD_E_INVALID, "rd_count_labels: null pointer");
    int64_t nb = (n + 16383) / 16384;
    if (nb > 1024) nb = 1024;
    if (((uintptr_t)labels & 15) == 0)
        hipLaunchKernelGGL(rd_count_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, labels, n, counts);
    else
        hipLaunchKernelGGL(rd_count_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, labels, n, counts);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_encode_codes(const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n, int32_t max_len,
                    int32_t stride, uint8_t *codes, void *stream) {
    if (n < 0 || max_len < 1 || stride < max_len) RD_FAIL(RD_E_INVALID, "rd_encode_codes: bad n/max_len/stride");
    if (n == 0) return RD_OK;
    if (!arena || !seq_off || !seq_len || !codes) RD_FAIL(RD_E_INVALID, "rd_encode_codes: null pointer");
    if ((int64_t)ENC_R * stride > 0x7fffffffLL) RD_FAIL(RD_E_INVALID, "rd_encode_codes: stride too large");
    int64_t nb = (n + ENC_R - 1) / ENC_R;
    if (nb > 256 * 64) nb = 256 * 64;
    if (((uintptr_t)codes & 3) == 0)   // a block's output starts at r0 * stride with r0 a multiple of 64
        hipLaunchKernelGGL(rd_encode_codes_kernel<true>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, arena, seq_off, seq_len,
                           n, max_len, stride, codes);
    else
        hipLaunchKernelGGL(rd_encode_codes_kernel<false>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, arena, seq_off, seq_len,
                           n, max_len, stride, codes);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_encode_onehot_padded(const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n,
                            int32_t max_len, float *onehot, void *stream) {
    if (n < 0 || max_len < 1) RD_FAIL(RD_E_INVALID, "rd_encode_onehot_padded: bad n/max_len");
    if (n == 0) return RD_OK;
    if (!arena || !seq_off || !seq_len || !onehot) RD_FAIL(RD_E_INVALID, "rd_encode_onehot_padded: null pointer");
    if (max_len > MAX_LEN_LIMIT) RD_FAIL(RD_E_INVALID, "rd_encode_onehot_padded: max_len=%d out of range [1,%d]", max_len, MAX_LEN_LIMIT);
    int64_t nb = (n + ENC_R - 1) / ENC_R;
    if (nb > 256 * 64) nb = 256 * 64;
    hipLaunchKernelGGL(rd_encode_onehot_padded_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, arena, seq_off,
                       seq_len, n, max_len, (f32x4 *)onehot);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

int rd_pack_plan(const int32_t *seq_len, int64_t n, int32_t max_len, int64_t *sorted_idx, int64_t *unsorted_idx,
                 int64_t *batch_sizes, int64_t *total_steps, void *workspace, size_t workspace_bytes, void *stream) {
    if (n < 1 || max_len < 1 || max_len > MAX_LEN_LIMIT) RD_FAIL(RD_E_INVALID, "rd_pack_plan: bad n/max_len");
    if (!seq_len || !sorted_idx || !unsorted_idx || !batch_sizes || !total_steps || !workspace)
        RD_FAIL(RD_E_INVALID, "rd_pack_plan: null pointer");
    int32_t *order = nullptr;
    int64_t *len_start = nullptr;
    return run_sort(seq_len, n, max_len, workspace, workspace_bytes, order, sorted_idx, unsorted_idx, batch_sizes, total_steps,
                    len_start, (hipStream_t)stream);
}

int rd_pack_onehot(const uint8_t *arena, const int64_t *seq_off, const int32_t *seq_len, int64_t n, int32_t max_len,
                   const int64_t *sorted_idx, const int64_t *batch_sizes, float *data, void *stream) {
    if (n < 1 || max_len < 1) RD_FAIL(RD_E_INVALID, "rd_pack_onehot: bad n/max_len");
    if (!arena || !seq_off || !seq_len || !sorted_idx || !batch_sizes || !data) RD_FAIL(RD_E_INVALID, "rd_pack_onehot: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = (n + ENC_R - 1) / ENC_R;
    if (nb > 0x7fffffffLL) RD_FAIL(RD_E_INVALID, "rd_pack_onehot: n too large");
    hipLaunchKernelGGL(rd_pack_onehot_kernel, dim3((unsigned)nb), dim3(256), 0, st, arena, seq_off, seq_len, n, max_len, sorted_idx,
                       batch_sizes, (f32x4 *)data);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

}  // extern "C"
