// rd_lstm_w8.hpp - experiment: the split-precision recurrence with two waves per SIMD (rd_lstm_mfma_f16x3_w8_kernel); slower, kept as evidence
// Part of the single translation unit rd_kernels.hip (included from there, in order); see that file for the kernel
// inventory and DESIGN.md §3 for the roofline of each kernel.
#pragma once
#include "rd_recurrence.hpp"

namespace {

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

}  // namespace
