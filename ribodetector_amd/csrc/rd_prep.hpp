// rd_prep.hpp - weight pre-packing (rd_prep_kernel) and the reverse-half table of the padded semantics (rd_revtab_kernel)
// Part of the single translation unit rd_kernels.hip (included from there, in order); see that file for the kernel
// inventory and DESIGN.md §3 for the roofline of each kernel.
#pragma once
#include "rd_common.hpp"

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
    // the input table of the default kernel, in its LDS layout: i = ((((w*2 + hf)*4 + a)*4 + b)*6 + code)*4 + gate; the rows are the
    // exp2 arguments' additive constants (gate g: 2 log2 e, the others: -log2 e); code 5 = the all-zero row of the dummy first pass
    for (int i = tid; i < 4 * 2 * 4 * 4 * 6 * 4; i += nth) {
        const int gate = i & 3, rest = i >> 2, code = rest % 6, cell = rest / 6;
        const int b = cell & 3, a = (cell >> 2) & 3, hf = (cell >> 4) & 1, w = cell >> 5;
        float v = 0.0f;
        if (code < 5) {
            const int col = gate * HID + 32 * w + 16 * hf + 4 * a + b;
            const float bb = raw[OFF_BIH + col] + raw[OFF_BHH + col];
            v = (gate == 2 ? 2.88539008177792681f : -1.44269504088896341f) * (code < 4 ? bb + raw[OFF_WIH + col * 4 + code] : bb);
        }
        d.lut_t32[i] = v;
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

}  // namespace
