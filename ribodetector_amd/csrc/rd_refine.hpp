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

// One kernel, no global candidate list: workgroup b scans the logits of reads [512 b, 512 (b+1)) (4 KB), collects the reads whose
// margin is inside the band in LDS and re-evaluates them itself, one read at a time (at the default band 1.5 % of the slices
// hold a candidate, next to none holds two: the pass takes the ~100 us latency of one read). 1,024 threads: thread (col, half)
// owns half of the 128 recurrent weights of gate column col (torch order i,f,g,o) in registers, loaded only when the slice
// has a candidate. candidate = own margin inside the band; with mate logits (paired end, --ensure none: the pair label is
// argmax of the SUMMED logits, detect.py:657) also a read whose pair margin is inside twice the band.
// Every read is evaluated on its own, so the result does not depend on the order the atomics collect the candidates in.
__global__ __launch_bounds__(1024) void rd_refine_kernel(DevModel d, ReadBatch rb, const float2 *__restrict__ mate, float thresh,
                                                         float *__restrict__ logits, uint8_t *__restrict__ labels) {
    __shared__ double h[HID], g[G4], part[G4], hr[HID], red[2][2];
    __shared__ int cand[REFINE_SLICE];
    __shared__ int ncand;
    const int tid = threadIdx.x, col = tid & (G4 - 1), half = tid >> 9;
    const float *raw = d.raw;
    const int64_t s0 = (int64_t)blockIdx.x * REFINE_SLICE;
    const int64_t s1 = s0 + REFINE_SLICE < rb.n ? s0 + REFINE_SLICE : rb.n;
    if (tid == 0) ncand = 0;
    __syncthreads();
    for (int64_t i = s0 + tid; i < s1; i += 1024) {
        const float2 a = reinterpret_cast<const float2 *>(logits)[i];
        bool hit = fabsf(a.y - a.x) < thresh;
        if (mate) {
            const float2 m = mate[i];
            hit = hit || fabsf((a.y + m.y) - (a.x + m.x)) < 2.0f * thresh;
        }
        if (hit) cand[atomicAdd(&ncand, 1)] = (int)(i - s0);
    }
    __syncthreads();
    const int total = ncand;
    if (total == 0) return;

    constexpr int KH = HID / 2;
    float w[KH];
#pragma unroll
    for (int u = 0; u < KH; ++u) w[u] = d.wt_hh[(KH * half + u) * G4 + col];
    const double bias = (double)raw[OFF_BIH + col] + (double)raw[OFF_BHH + col];
    double wi[5];
#pragma unroll
    for (int k = 0; k < 4; ++k) wi[k] = (double)raw[OFF_WIH + col * 4 + k];
    wi[4] = 0.0;
    for (int k = 0; k < total; ++k) {
        const int64_t i = s0 + cand[k];
        const int lr = rd_T(rb.len, i, rb.max_len);
        const uint8_t *p = rb.arena + rb.off[i];
        int T = lr;
        if (rb.sem == RD_SEM_PADDED) {   // steps = last non-zero row + 1 (rd_steps_kernel)
            int pos = lr - 1;
            while (pos >= 0 && rd_code(p[pos]) == 4) --pos;
            T = pos >= 0 ? pos + 1 : rb.max_len;
        }
        double c = 0.0;   // cell state of unit tid (threads < 128)
        __syncthreads();
        if (tid < HID) h[tid] = 0.0;
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const int code = t < lr ? rd_code(p[t]) : 4;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // four chains: the float64 FMA latency, not its rate, bounds a step
            const double *hh = h + KH * half;
#pragma unroll
            for (int u = 0; u < KH; u += 4) {
                a0 = fma(rd_f64(w[u]), hh[u], a0);
                a1 = fma(rd_f64(w[u + 1]), hh[u + 1], a1);
                a2 = fma(rd_f64(w[u + 2]), hh[u + 2], a2);
                a3 = fma(rd_f64(w[u + 3]), hh[u + 3], a3);
            }
            const double a = (a0 + a1) + (a2 + a3);
            if (half) part[col] = a;
            __syncthreads();
            if (!half) {   // every column activates its own gate
                const double x = (bias + wi[code]) + (a + part[col]);
                g[col] = (col >> 7) == 2 ? tanh(x) : rd_sigmoid64(x);
            }
            __syncthreads();
            if (tid < HID) {
                c = g[HID + tid] * c + g[tid] * g[2 * HID + tid];
                h[tid] = g[3 * HID + tid] * tanh(c);
            }
            __syncthreads();
        }
        // reverse half of the output row
        const int last = T > 0 ? ((T - 1) < lr ? rd_code(p[T - 1]) : 4) : 4;
        if (tid < HID) {
            double v = 0.0;
            if (rb.sem != RD_SEM_PADDED && T > 0) {   // one reverse step from the zero state on the last base (model.py:33, forward1)
                double gr[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cc = q * HID + tid;
                    gr[q] = (double)raw[OFF_BIHR + cc] + (double)raw[OFF_BHHR + cc] + (last < 4 ? (double)raw[OFF_WIHR + cc * 4 + last] : 0.0);
                }
                const double cr = rd_sigmoid64(gr[0]) * tanh(gr[2]);
                v = rd_sigmoid64(gr[3]) * tanh(cr);
            }
            hr[tid] = v;
        }
        __syncthreads();
        // FC (model.py:36): 2 x 256 dot product, 128 lanes each
        if (tid < 2 * HID) {
            const int cls = tid >> 7, u = tid & 127;
            double s = (double)raw[OFF_WOUT + cls * 256 + u] * h[u] + (double)raw[OFF_WOUT + cls * 256 + HID + u] * hr[u];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
            if ((tid & 63) == 0) red[cls][(tid >> 6) & 1] = s;
        }
        __syncthreads();
        if (tid < 2) {
            double s = (double)raw[OFF_BOUT + tid] + red[tid][0] + red[tid][1];
            if (rb.sem == RD_SEM_PADDED && T > 0) s += (double)rb.rev_tab[((rb.max_len - 1 - (T - 1)) * 5 + last) * 2 + tid];
            const float f = (float)s;
            const float other = __shfl_xor(f, 1);
            logits[(size_t)i * 2 + tid] = f;
            if (labels && tid == 0) labels[i] = other > f ? 1 : 0;
        }
    }
}

}  // namespace
