// rd_lstm_f32.hpp - exact-fp32 MFMA recurrence (rd_lstm_mfma_f32_kernel) and the plain-FMA cross-check (rd_lstm_simple_kernel)
// Part of the single translation unit rd_kernels.hip (included from there, in order); see that file for the kernel
// inventory and DESIGN.md §3 for the roofline of each kernel.
#pragma once
#include "rd_recurrence.hpp"

namespace {

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
    int k0[BT];       // bases of the read covered by its prefix-table row (0 or pk)
    int prow[BT];     // the table row the read starts from
    int tmax;
};

// Prefix-state table of THIS kernel (round 4; the default kernel's is described in rd_lstm_t32.hpp / DESIGN.md §3.9): a row is the
// state as this kernel holds it - h fp32[128] | c fp32[128], 1 KiB, same addressing as the default kernel's rows - so that a start
// from the table is the same bits as stepping over the bases. The rows of one model's table belong to the kernel that built it
// (rd_set_prefix_table builds with the model's current variant; rd_classify uses the table only with that variant).
__device__ __forceinline__ void rd_f32_load_h(LstmSmem &S, const uint8_t *ptab, int read) {   // thread (read, piece): 32 floats of h
    const int piece = threadIdx.x & 3;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(ptab + (size_t)S.prow[read] * PFX_ROW) + 8 * piece;
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[k];
    f32x4 *dst = reinterpret_cast<f32x4 *>(&S.Hs[read >> 4][read & 15][32 * piece]);
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] = v[k];
}
__device__ __forceinline__ void rd_f32_load_c(LstmSmem &S, const uint8_t *ptab, int tile) {   // the lane's own 8 cells of the tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, l15 = lane & 15;
    f32x4 a, b;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float *c = reinterpret_cast<const float *>(ptab + (size_t)S.prow[tile * 16 + 4 * q + r] * PFX_ROW + 512) + 32 * wave + l15;
        a[r] = c[0];
        b[r] = c[16];
    }
    S.cA[tile][tid] = a;
    S.cB[tile][tid] = b;
}

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

// ACT: 0 = compensated exp, 1 = plain v_exp_f32 forms (round 1), 2 = shared reciprocals (the product).  SCHED: 0 = compiler's own order, 1 = LDS reads of the gate math
// pinned to the top of the phase + explicit MFMA/VALU interleave (sched_group_barrier).
// BUILD = true: the same code builds the table one level per launch (as rd_lstm_mfma_f16x3_t32_kernel<true> does): "read" g of level
// rb.pk starts from row g >> 2 of the level below (rb.ptab), steps over base g & 3 and writes its state into row g of `logits`.
template <int ACT, int SCHED, int DIAG = 0, bool BUILD = false>   // DIAG (bench diagnosis only, wrong results): 1 = no gate math, 2 = no MFMA
__global__ __launch_bounds__(256, 1) void rd_lstm_mfma_f32_kernel(DevModel d, ReadBatch rb, float *__restrict__ logits,
                                                                  uint8_t *__restrict__ labels) {
    __shared__ LstmSmem S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, l15 = lane & 15;

    // ---- per-read metadata, zero state ---------------------------------------------------------
    const int nz = BUILD ? 0 : (rb.pk > 0 ? 1 << (2 * rb.pk) : 0);   // the zero row (without a table: row 0 of the model's one-row table)
    if (tid < BT) {
        const int64_t g = (int64_t)blockIdx.x * BT + tid;
        int T = 0, lr = 0, orig = -1, k0 = 0, prow = nz;
        long long off = 0;
        if (BUILD) {
            if (g < rb.n) { orig = (int)g; T = 1; lr = 1; if (rb.pk > 1) prow = (int)(g >> 2); }
        } else if (g < rb.n) {
            orig = rb.order ? rb.order[g] : (int)g;
            T = rd_T(rb.steps, orig, rb.max_len);
            lr = rd_T(rb.len, orig, rb.max_len);
            off = rb.off[orig];
            if (rb.pfx) {
                prow = rb.pfx[orig];
                if (prow != nz) { k0 = rb.pk; lr -= k0; off += k0; }   // the row covers the first pk bases (rd_steps_kernel took them off T)
            }
        }
        S.T[tid] = T; S.Lr[tid] = lr; S.off[tid] = off; S.orig[tid] = orig; S.k0[tid] = k0; S.prow[tid] = prow;
    }
    if (tid == 0) S.tmax = 0;
    for (int i = tid; i < NT * 16 * HSTR; i += 256) (&S.Hl[0][0][0])[i] = 0.0f;
    for (int i = tid; i < 5 * G4; i += 256) {      // i = ((code*4 + w)*16 + l15)*8 + c
        const int c = i & 7, l = (i >> 3) & 15, w = (i >> 7) & 3, code = i >> 9;
        (reinterpret_cast<float *>(&S.lut[0][0][0][0]))[i] = d.in_lut[code * G4 + gate_col(w, c, l)];
    }
    S.wout[tid >> 7][tid & 127] = d.w_out[(tid >> 7) * 256 + (tid & 127)];
    __syncthreads();
    if (tid < BT) atomicMax(&S.tmax, S.T[tid]);
    // start state: the read's table row (the zero row without a table). Thread (read, piece) moves 128 B of h; a lane takes the cells
    // it owns. Tile NT-1 is loaded AGAIN after the first phase: the dummy gate pass before t = 0 (ptile = NT-1, masked by `live`)
    // writes a zero state there, which is harmless for a zero start and would wipe a loaded one.
    rd_f32_load_h(S, rb.ptab, tid >> 2);
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) rd_f32_load_c(S, rb.ptab, tl);
    if (BUILD) {   // the one base of prefix g = its least significant base-4 digit
        for (int idx = tid; idx < BT * TC; idx += 256) {
            const int row = idx / TC, tt = idx % TC;
            S.codes[0][tt][row] = (uint8_t)((tt == 0 && S.T[row] > 0) ? (int)(((int64_t)blockIdx.x * BT + row) & 3) : 4);
        }
    } else {
        rd_stage_codes(S, rb, 0);
    }

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
    for (int m = 0; m < 8; ++m) hA[m] = *reinterpret_cast<const f32x4 *>(&S.Hs[0][l15][16 * m + 4 * q]);

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
                    float cn, h;
                    if constexpr (ACT == 2) {
                        // shared reciprocals (as in the default kernel, rd_lstm_t32.hpp): sigmoid(i) tanh(g) = (e_g - 1) / ((1 + e_i)(1 + e_g)),
                        // sigmoid(o) tanh(c) likewise - 8 instead of 10 transcendentals per cell and less rounding noise; the exp2
                        // arguments of g and c are capped so that e stays finite (a saturated neighbour then gives finite * 0)
                        const float ei = __builtin_amdgcn_exp2f(gi * -1.44269504088896341f), ef = __builtin_amdgcn_exp2f(gf * -1.44269504088896341f);
                        const float eg = __builtin_amdgcn_exp2f(fminf(gg * 2.88539008177792681f, 64.0f));
                        const float eo = __builtin_amdgcn_exp2f(go * -1.44269504088896341f);
                        const float u = (eg - 1.0f) * __builtin_amdgcn_rcpf((1.0f + ei) * (1.0f + eg));
                        cn = __builtin_fmaf(__builtin_amdgcn_rcpf(1.0f + ef), cs[s][r], u) * live;
                        const float ec = __builtin_amdgcn_exp2f(fminf(cn * 2.88539008177792681f, 64.0f));
                        h = (ec - 1.0f) * __builtin_amdgcn_rcpf((1.0f + eo) * (1.0f + ec)) * live;
                    } else {
                        cn = __builtin_fmaf(act_sigmoid<ACT>(gf), cs[s][r], act_sigmoid<ACT>(gi) * act_tanh<ACT>(gg));
                        cn *= live;
                        h = act_sigmoid<ACT>(go) * act_tanh<ACT>(cn) * live;
                    }
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
        if (p == 0 && (BUILD || rb.pfx)) {   // tile NT-1's start state, wiped by the dummy gate pass above (first read at p = NT-2)
            if (wave == 3) rd_f32_load_h(S, rb.ptab, tid >> 2);
            rd_f32_load_c(S, rb.ptab, NT - 1);
        }
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
    if constexpr (BUILD) {   // row g of the level-rb.pk table <- the state after the step (layout: see rd_f32_load_h / _c)
        uint8_t *tab = reinterpret_cast<uint8_t *>(logits);
        const int64_t g0 = (int64_t)blockIdx.x * BT;
        {
            const int read = tid >> 2, piece = tid & 3;
            if (g0 + read < rb.n) {
                f32x4 *dst = reinterpret_cast<f32x4 *>(tab + (size_t)(g0 + read) * PFX_ROW) + 8 * piece;
                const f32x4 *src = reinterpret_cast<const f32x4 *>(&S.Hs[read >> 4][read & 15][32 * piece]);
#pragma unroll
                for (int k = 0; k < 8; ++k) dst[k] = src[k];
            }
        }
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
            const f32x4 a = S.cA[tl][tid], b = S.cB[tl][tid];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t g = g0 + tl * 16 + 4 * q + r;
                if (g < rb.n) {
                    float *c = reinterpret_cast<float *>(tab + (size_t)g * PFX_ROW + 512) + 32 * wave + l15;
                    c[0] = a[r];
                    c[16] = b[r];
                }
            }
        }
    } else {
        // ---- epilogue: FC + reverse table + argmax ----------------------------------------------------
        rd_fc_epilogue(
            BT, [&](int row, int u) { return S.Hl[row >> 4][row & 15][u]; }, S.T, S.Lr, S.off, S.orig, &S.wout[0][0], d, rb, logits, labels,
            S.k0);
    }
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

}  // namespace
