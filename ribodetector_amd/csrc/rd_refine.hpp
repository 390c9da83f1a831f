// rd_refine.hpp - float64 re-evaluation of the reads whose decision sits inside the fp32 noise band (rd_refine_kernel)
// Part of the single translation unit rd_kernels.hip (included from there, in order); see that file for the kernel
// inventory and DESIGN.md §3 for the roofline of each kernel.
//
// Why: the label is argmax(logits) (detect.py:288,481). Any fp32 evaluation of the recurrence - the reference's torch/cuDNN
// arithmetic, the CPU oracle, either MFMA kernel here - carries ~3e-6 rms / ~1e-4 worst-case rounding noise on the logits
// (DESIGN.md §4), so for the ~6 reads per million whose margin |logit1 - logit0| is below that, the label is decided by rounding
// noise and differs between implementations. Those reads (margin below the model's refine threshold, default 2.5e-4: ~15 per
// million) are evaluated again in float64 - model.py:32-37 with every product, sum and activation in double - and their
// logits and labels replaced. The result is the label of the exact function wherever its margin exceeds ~1e-7, whatever the
// batch size or kernel variant.
#pragma once
#include "rd_prep.hpp"
#include "rd_recurrence.hpp"

namespace {

constexpr int REFINE_SLICE = 512;    // reads scanned per workgroup; a workgroup re-evaluates the candidates of its own slice

__device__ __forceinline__ double rd_sigmoid64(double x) { return 1.0 / (1.0 + exp(-x)); }
// float -> double at the point of use: the 128 weights of a column stay in 128 registers as floats (a hoisted conversion would
// need 256 and spill)
__device__ __forceinline__ double rd_f64(float x) {
    double d;
    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(x));
    return d;
}

struct RefineSmem {
    double h[HID], g[G4], part[G4], hr[HID], red[2][2];
};

// Thread (col, half) owns half of the 128 recurrent weights of gate column col (torch order i,f,g,o) in registers.
struct RefineWeights {
    float w[HID / 2];
    double bias, wi[5];
};
__device__ __forceinline__ void rd_refine_load(const DevModel &d, RefineWeights &W) {
    constexpr int KH = HID / 2;
    const int tid = threadIdx.x, col = tid & (G4 - 1), half = tid >> 9;
    const float *raw = d.raw;
#pragma unroll
    for (int u = 0; u < KH; ++u) W.w[u] = d.wt_hh[(KH * half + u) * G4 + col];
    W.bias = (double)raw[OFF_BIH + col] + (double)raw[OFF_BHH + col];
#pragma unroll
    for (int k = 0; k < 4; ++k) W.wi[k] = (double)raw[OFF_WIH + col * 4 + k];
    W.wi[4] = 0.0;
}

// model.py:32-37 for ONE read in float64, by the 1,024 threads of a workgroup; writes the read's logits (and label)
__device__ __forceinline__ void rd_refine_eval(const DevModel &d, RefineSmem &S, const RefineWeights &W, const uint8_t *p, int lr, int sem,
                                               int max_len, const float *rev_tab, float *out_logits, uint8_t *out_label) {
    constexpr int KH = HID / 2;
    const int tid = threadIdx.x, col = tid & (G4 - 1), half = tid >> 9;
    const float *raw = d.raw;
    int T = lr;
    if (sem == RD_SEM_PADDED) {   // steps = last non-zero row + 1 (rd_steps_kernel)
        int pos = lr - 1;
        while (pos >= 0 && rd_code(p[pos]) == 4) --pos;
        T = pos >= 0 ? pos + 1 : max_len;
    }
    double c = 0.0;   // cell state of unit tid (threads < 128)
    __syncthreads();
    if (tid < HID) S.h[tid] = 0.0;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int code = t < lr ? rd_code(p[t]) : 4;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // four chains: the float64 FMA latency, not its rate, bounds a step
        const double *hh = S.h + KH * half;
#pragma unroll
        for (int u = 0; u < KH; u += 4) {
            a0 = fma(rd_f64(W.w[u]), hh[u], a0);
            a1 = fma(rd_f64(W.w[u + 1]), hh[u + 1], a1);
            a2 = fma(rd_f64(W.w[u + 2]), hh[u + 2], a2);
            a3 = fma(rd_f64(W.w[u + 3]), hh[u + 3], a3);
        }
        const double a = (a0 + a1) + (a2 + a3);
        if (half) S.part[col] = a;
        __syncthreads();
        if (!half) {   // every column activates its own gate
            const double x = (W.bias + W.wi[code]) + (a + S.part[col]);
            S.g[col] = (col >> 7) == 2 ? tanh(x) : rd_sigmoid64(x);
        }
        __syncthreads();
        if (tid < HID) {
            c = S.g[HID + tid] * c + S.g[tid] * S.g[2 * HID + tid];
            S.h[tid] = S.g[3 * HID + tid] * tanh(c);
        }
        __syncthreads();
    }
    // reverse half of the output row
    const int last = T > 0 ? ((T - 1) < lr ? rd_code(p[T - 1]) : 4) : 4;
    if (tid < HID) {
        double v = 0.0;
        if (sem != RD_SEM_PADDED && T > 0) {   // one reverse step from the zero state on the last base (model.py:33, forward1)
            double gr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cc = q * HID + tid;
                gr[q] = (double)raw[OFF_BIHR + cc] + (double)raw[OFF_BHHR + cc] + (last < 4 ? (double)raw[OFF_WIHR + cc * 4 + last] : 0.0);
            }
            const double cr = rd_sigmoid64(gr[0]) * tanh(gr[2]);
            v = rd_sigmoid64(gr[3]) * tanh(cr);
        }
        S.hr[tid] = v;
    }
    __syncthreads();
    // FC (model.py:36): 2 x 256 dot product, 128 lanes each
    if (tid < 2 * HID) {
        const int cls = tid >> 7, u = tid & 127;
        double s = (double)raw[OFF_WOUT + cls * 256 + u] * S.h[u] + (double)raw[OFF_WOUT + cls * 256 + HID + u] * S.hr[u];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
        if ((tid & 63) == 0) S.red[cls][(tid >> 6) & 1] = s;
    }
    __syncthreads();
    if (tid < 2) {
        double s = (double)raw[OFF_BOUT + tid] + S.red[tid][0] + S.red[tid][1];
        if (sem == RD_SEM_PADDED && T > 0) s += (double)rev_tab[((max_len - 1 - (T - 1)) * 5 + last) * 2 + tid];
        const float f = (float)s;
        const float other = __shfl_xor(f, 1);
        out_logits[tid] = f;
        if (out_label && tid == 0) *out_label = other > f ? 1 : 0;
    }
}

// One kernel, no global candidate list: workgroup b scans the logits of reads [512 b, 512 (b+1)) (4 KB), collects the reads whose
// margin is inside the band in LDS and re-evaluates them itself, one read at a time (at the default band 1.5 % of the slices
// hold a candidate, next to none holds two: the pass takes the ~100 us latency of one read). 1,024 threads; the weights are loaded
// only when the slice has a candidate. candidate = own margin inside the band; with mate logits (paired end, --ensure none: the
// pair label is argmax of the SUMMED logits, detect.py:657) also a read whose pair margin is inside twice the band.
// Every read is evaluated on its own, so the result does not depend on the order the atomics collect the candidates in.
// Deferred mode (rd_set_refine_async): the recurrence kernel's epilogue records the candidates in the model's queue
// (rd_fc_epilogue) and rd_refine_flush_kernel evaluates the candidates of a group of calls together; this kernel then runs only as
// the second tier, gated on an overflow of that queue.
__global__ __launch_bounds__(1024) void rd_refine_kernel(DevModel d, ReadBatch rb, const float2 *__restrict__ mate, float thresh,
                                                         float *__restrict__ logits, uint8_t *__restrict__ labels, RefineQueue q,
                                                         const uint32_t *gate) {
    // gate (deferred mode, second tier): this launch only has work if the queue of its group overflowed - then it re-evaluates every
    // in-band read of its call itself (reads the flush has already refined come out the same)
    if (gate && *gate <= q.cap) return;
    if (gate) q.e = nullptr;
    __shared__ RefineSmem S;
    __shared__ int cand[REFINE_SLICE];
    __shared__ int ncand;
    const int tid = threadIdx.x;
    bool loaded = false;
    RefineWeights W;
    // (the second tier is launched with a handful of workgroups that walk the slices: when its gate is closed - always, unless a queue
    // overflowed - it must cost nothing, and 2,048 workgroups of 1,024 threads that each need a whole CU do not cost nothing)
    for (int64_t slice = blockIdx.x; slice * REFINE_SLICE < rb.n; slice += gridDim.x) {
        const int64_t s0 = slice * REFINE_SLICE;
        const int64_t s1 = s0 + REFINE_SLICE < rb.n ? s0 + REFINE_SLICE : rb.n;
        __syncthreads();
        if (tid == 0) ncand = 0;
        __syncthreads();
        for (int64_t i = s0 + tid; i < s1; i += 1024) {
            const float2 a = reinterpret_cast<const float2 *>(logits)[i];
            bool hit = fabsf(a.y - a.x) < thresh;
            if (mate) {
                const float2 m = mate[i];
                hit = hit || fabsf((a.y + m.y) - (a.x + m.x)) < 2.0f * thresh;
            }
            if (hit && q.e) {   // record it; evaluated by the flush
                const uint32_t slot = atomicAdd(q.count, 1u);
                if (slot < q.cap) {
                    q.e[slot] = RefineEntry{rb.arena + rb.off[i], logits + 2 * i, labels ? labels + i : nullptr, rd_T(rb.len, i, rb.max_len),
                                            rb.max_len, rb.sem, 0};
                    hit = false;
                }
            }
            if (hit) cand[atomicAdd(&ncand, 1)] = (int)(i - s0);
        }
        __syncthreads();
        const int total = ncand;
        if (total == 0) continue;
        if (!loaded) { rd_refine_load(d, W); loaded = true; }
        for (int k = 0; k < total; ++k) {
            const int64_t i = s0 + cand[k];
            rd_refine_eval(d, S, W, rb.arena + rb.off[i], rd_T(rb.len, i, rb.max_len), rb.sem, rb.max_len, rb.rev_tab, logits + 2 * i,
                           labels ? labels + i : nullptr);
        }
    }
}

// The candidates of the calls of one group (rd_set_refine_async), each by one workgroup: they all finish in the time one takes.
constexpr int REFINE_FLUSH_WGS = 32;
__global__ __launch_bounds__(1024) void rd_refine_flush_kernel(DevModel d, RefineQueue q) {
    __shared__ RefineSmem S;
    const uint32_t total = min(*q.count, q.cap);
    if (blockIdx.x >= total) return;
    RefineWeights W;
    rd_refine_load(d, W);
    for (uint32_t k = blockIdx.x; k < total; k += gridDim.x) {
        const RefineEntry e = q.e[k];
        rd_refine_eval(d, S, W, e.bases, e.lr, e.sem, e.max_len, d.rev_tab, e.logits, e.label);
    }
}

}  // namespace
