// rd_lstm_f16x3.hpp - split-precision recurrence on 16x16x32 f16 MFMAs (rd_lstm_mfma_f16x3_kernel): the first f16x3 tiling, kept for A/B runs
// Part of the single translation unit rd_kernels.hip (included from there, in order); see that file for the kernel
// inventory and DESIGN.md §3 for the roofline of each kernel.
#pragma once
#include "rd_recurrence.hpp"

namespace {

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

}  // namespace
