// rd_lstm_t32_diag.hpp - DIAGNOSTIC BUILDS ONLY (-DRD_DIAG, librd_hip_diag.so): the default kernel as a template over the accuracy /
// timing experiments of rounds 1-2 (ACC bits, FILL modes). Several instantiations compute WRONG results by design. The product
// library never contains any of this: rd_kernels.hip includes this file only under RD_DIAG, and the product kernel
// (rd_lstm_t32.hpp) is the straight-line form of ACC = 16|32|64|128, FILL > 0 with no switches left in it.
// Everything lives in namespace t32diag; only the code-staging helpers and the gate-math schedule are shared with the product file.
// Results of the experiments: DESIGN.md §4 (accuracy table) and §8 (explored and rejected).
#pragma once
#include "rd_lstm_t32.hpp"

namespace {
namespace t32diag {
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
    uint8_t codes[2][64][CSTR];    // [buffer][row][t % TC16], rows padded to 17 words: the lanes' byte reads hit 32 banks
    int T[64];
    int Lr[64];       // readable bytes of the read = min(len, max_len)
    long long off[64];
    int orig[64];
    int tmax;
    int wmax[4];      // per wave: the largest step count of its 16 reads
};

// One phase: MFMAs of (tile TL, current step) into accC; gate math of (tile TL^1, step tEW) from accP.
//
// The compiler's scheduler neither interleaves the two streams on its own nor honours a 96-group sched_group_barrier
// pipeline in a region this large, so the interleave is written out: the gate math is cut into 212 "units" of 1-5
// instructions (13 stages per cell, two cells in flight and never in the same stage, at most two transcendentals per
// unit, table rows fetched one cell ahead) and the units are dealt out behind the 96 MFMAs, ~2.2 units (about 5 VALU
// ops) per MFMA - what a 32x32x16 MFMA mostly hides (tools/ubench/mfma_fill.hip: 38.7 cycles bare, 48 with 2 exp + 3 fma).
// A sched_barrier after every slot pins the order.
struct EwRegs {
    // slots of the cells in flight: cell % 2 in the phases (two cells in flight), cell % 4 in the gate-math-only tail (a row-tile's
    // four cells side by side)
    f32x4 kc[4];        // table rows
    f32x2 v[4][2];      // gate pipeline values: {i,f} and {g,o} as register pairs
    float y[4], og[4], hs[4];
    f32x4 cs[2], hv[2]; // per row-tile, by row-tile parity
    f16x4 o1s[2], o2[2];
    f32x4 call[4];      // ACC & 32: the tile's cell state, resident across phases
    float nm[4];        // ACC & 128: numerators K (e_g - 1) resp. (e_c - 1) of the shared-reciprocal form
};

// ACC: build options of the kernel. The product instantiates T32_PRODUCT (bits 16 + 32 + 64 + 128); the other bits are accuracy experiments
// that exist only in diagnostic builds (-DRD_DIAG, tools/acc_experiment.py; results in DESIGN.md §4):
//   1 = fourth product W2.H2 (the dropped lo x lo term)       2 = the small products first, W1.H1s last (H1s fragments read twice)
//   4 = exp2 arguments formed from the fp32 pre-activation with a compensated product (table holds the raw in_lut rows)
//   8 = one Newton step on every v_rcp_f32
//   16 = gate math in 24 instead of 26 VALU ops per cell (scales folded into the reciprocals' arguments, see stage 4)
//   32 = cell state kept in registers across phases instead of the LDS round trip
//   64 = captured-h stores only in phases where a read of the wave finishes (wave-uniform branch; -0.26 %)
//   128 = shared reciprocals: sigmoid(i) tanh(g) = (e_g - 1) / ((1 + e_i)(1 + e_g)) and sigmoid(o) tanh(c) likewise: 8 instead of 10
//         transcendentals per cell for 4 more plain VALU ops; the only clamp needed is on the exp2 argument of g and c (<= 64)
//   256 = timing diagnosis (wrong results): table rows not fetched
//   1024 = timing diagnosis (wrong results): every 32x32x16 MFMA issued as two 16x16x32 (rd_slots)
//   512 = (with 128) ONE reciprocal for the whole cell update: c' = (c' (1+e_i)(1+e_g) + K (e_g - 1)(1+e_f)) / ((1+e_f)(1+e_i)(1+e_g)),
//         exp2 arguments of i, f, g clamped to <= 40 so that the triple product stays below 2^128: 7 transcendentals per cell
constexpr int T32_PRODUCT = 16 | 32 | 64 | 128;
__device__ __forceinline__ float rd_exp2c(float x, float khi, float klo) {   // 2^(x (khi + klo)), product error compensated
    const float t = x * khi;
    float e = __builtin_fmaf(x, khi, -t);
    e = __builtin_fmaf(x, klo, e);
    const float r = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(r, e * 0.693147182464599609375f, r);
}
__device__ __forceinline__ float rd_rcp_nr(float d) {
    const float y = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, y, 1.0f), y, y);
}
constexpr float KS_HI = -1.44269502162933349609375f, KS_LO = -1.925963033500011e-8f;   // -log2 e = KS_HI + KS_LO
constexpr float KT_HI = 2.8853900432586669921875f, KT_LO = 3.851926067000022e-8f;      // 2 log2 e

// One stage of one cell. NS = register slots of the cells in flight (slot = cell % NS); OWNROW: stage 0 fetches the cell's own
// table row (the tail) instead of the next cell's (the phases, whose first row is fetched before the first unit).
template <int TP, int cell, int stage, int ACC, int NS, bool OWNROW>
__device__ __forceinline__ void rd_ew_cs(Lstm16bSmem &S, EwRegs &R, const f32x16 (&accP)[4], const PhaseCtx &c) {
    constexpr int a = cell >> 2, b = cell & 3, k = cell % NS, ap = a & 1;
    if constexpr (stage == 0 && OWNROW) {
        R.kc[k] = S.lut[c.wave][c.half][a][b][c.codeEW];
        if constexpr (b == 0 && !(ACC & 32)) R.cs[ap] = S.cS[TP][a][c.tid];
    } else if constexpr (stage == 0) {   // table row of the NEXT cell (this cell's row was fetched one cell ago); cell state per row-tile
        constexpr int nc = cell + 1;
        if constexpr (nc < 16) {
            if constexpr (ACC & 256) R.kc[k ^ 1] = R.kc[k];   // timing diagnosis only (WRONG results): what the 16 table-row reads cost
            else R.kc[k ^ 1] = S.lut[c.wave][c.half][nc >> 2][nc & 3][c.codeEW];
        }
        if constexpr (b == 0 && !(ACC & 32)) R.cs[ap] = S.cS[TP][a][c.tid];
    } else if constexpr (stage == 1) {
        // exp2 arguments. Scalar FMAs on purpose: packed fp32 ops (v_pk_fma_f32 / v_pk_add_f32) cost ~10 cycles each beside
        // f16 MFMAs against ~1 for a scalar op (tools/ubench/mfma_fill.hip), so the build also passes -fno-slp-vectorize.
        if constexpr (ACC & 4) {   // fp32 pre-activations (exact scaling, one rounding like the reference's bias add)
            R.v[k][0][0] = __builtin_fmaf(accP[a][4 * b + 0], 1.0f / G_SCALE, R.kc[k][0]);
            R.v[k][0][1] = __builtin_fmaf(accP[a][4 * b + 1], 1.0f / G_SCALE, R.kc[k][1]);
            R.v[k][1][0] = __builtin_fmaf(accP[a][4 * b + 2], 1.0f / G_SCALE, R.kc[k][2]);
            R.v[k][1][1] = __builtin_fmaf(accP[a][4 * b + 3], 1.0f / G_SCALE, R.kc[k][3]);
        } else {
        R.v[k][0][0] = __builtin_fmaf(accP[a][4 * b + 0], KS / G_SCALE, R.kc[k][0]);
        R.v[k][0][1] = __builtin_fmaf(accP[a][4 * b + 1], KS / G_SCALE, R.kc[k][1]);
        R.v[k][1][0] = __builtin_fmaf(accP[a][4 * b + 2], KT / G_SCALE, R.kc[k][2]);
        R.v[k][1][1] = __builtin_fmaf(accP[a][4 * b + 3], KS / G_SCALE, R.kc[k][3]);
        }
    } else if constexpr (stage == 2) {
        if constexpr (ACC & 4) {
            R.v[k][0][0] = rd_exp2c(R.v[k][0][0], KS_HI, KS_LO); R.v[k][0][1] = rd_exp2c(R.v[k][0][1], KS_HI, KS_LO);
        } else {
        if constexpr (ACC & 512) { R.v[k][0][0] = __builtin_amdgcn_exp2f(fminf(R.v[k][0][0], 40.0f)); R.v[k][0][1] = __builtin_amdgcn_exp2f(fminf(R.v[k][0][1], 40.0f)); }
        else { R.v[k][0][0] = __builtin_amdgcn_exp2f(R.v[k][0][0]); R.v[k][0][1] = __builtin_amdgcn_exp2f(R.v[k][0][1]); }
        }
    } else if constexpr (stage == 3) {
        if constexpr (ACC & 4) {
            R.v[k][1][0] = rd_exp2c(R.v[k][1][0], KT_HI, KT_LO); R.v[k][1][1] = rd_exp2c(R.v[k][1][1], KS_HI, KS_LO);
        } else {
        if constexpr (ACC & 128) R.v[k][1][0] = __builtin_amdgcn_exp2f(fminf(R.v[k][1][0], (ACC & 512) ? 40.0f : 64.0f));   // e_g stays finite: (e_g - 1) * rcp(inf) must not be inf * 0
        else R.v[k][1][0] = __builtin_amdgcn_exp2f(R.v[k][1][0]);
        R.v[k][1][1] = __builtin_amdgcn_exp2f(R.v[k][1][1]);
        }
    } else if constexpr (stage == 4) {
        if constexpr (ACC & 128) {
            R.nm[k] = __builtin_fmaf(R.v[k][1][0], KT, -KT);                                        // KT (e_g - 1)
            R.v[k][0][0] += 1.0f; R.v[k][0][1] += 1.0f; R.v[k][1][0] += 1.0f;                      // 1 + e_i, 1 + e_f, 1 + e_g
            R.v[k][1][1] = __builtin_fmaf(R.v[k][1][1], 1.0f / H_SCALE, 1.0f / H_SCALE);            // (1 + e_o) 2^-11
        } else if constexpr (ACC & 16) {
            // the constants the gates are multiplied with later are folded into the reciprocals' arguments (an FMA instead of
            // an add, nothing else): 1/((1+e)/KT) = KT sigmoid(i),  1/(-(1+e)/2) = -2/(1+e) = tanh(g) - 1,  1/((1+e) 2^-11) = 2^11 sigmoid(o)
            R.v[k][0][0] = __builtin_fmaf(R.v[k][0][0], 1.0f / KT, 1.0f / KT); R.v[k][0][1] += 1.0f;
            R.v[k][1][0] = __builtin_fmaf(R.v[k][1][0], -0.5f, -0.5f); R.v[k][1][1] = __builtin_fmaf(R.v[k][1][1], 1.0f / H_SCALE, 1.0f / H_SCALE);
        } else {
        R.v[k][0][0] += 1.0f; R.v[k][0][1] += 1.0f;
        R.v[k][1][0] += 1.0f; R.v[k][1][1] += 1.0f;
        }
    } else if constexpr (stage == 5) {
        if constexpr (ACC & 512) { R.v[k][0][0] *= R.v[k][1][0]; R.v[k][1][0] = R.v[k][0][1] * R.v[k][0][0]; }   // AB = (1+e_i)(1+e_g); F AB
        else if constexpr (ACC & 128) { R.v[k][0][1] = (ACC & 8) ? rd_rcp_nr(R.v[k][0][1]) : __builtin_amdgcn_rcpf(R.v[k][0][1]); R.v[k][0][0] *= R.v[k][1][0]; }   // f; (1+e_i)(1+e_g)
        else
        if constexpr (ACC & 8) { R.v[k][0][0] = rd_rcp_nr(R.v[k][0][0]); R.v[k][0][1] = rd_rcp_nr(R.v[k][0][1]); }
        else { R.v[k][0][0] = __builtin_amdgcn_rcpf(R.v[k][0][0]); R.v[k][0][1] = __builtin_amdgcn_rcpf(R.v[k][0][1]); }
    } else if constexpr (stage == 6) {
        if constexpr (ACC & 512) R.v[k][1][0] = __builtin_amdgcn_rcpf(R.v[k][1][0]);   // 1 / (F AB)
        else if constexpr (ACC & 128) R.v[k][0][0] = (ACC & 8) ? rd_rcp_nr(R.v[k][0][0]) : __builtin_amdgcn_rcpf(R.v[k][0][0]);
        else
        if constexpr (ACC & 8) { R.v[k][1][0] = rd_rcp_nr(R.v[k][1][0]); R.v[k][1][1] = rd_rcp_nr(R.v[k][1][1]); }
        else { R.v[k][1][0] = __builtin_amdgcn_rcpf(R.v[k][1][0]); R.v[k][1][1] = __builtin_amdgcn_rcpf(R.v[k][1][1]); }
    } else if constexpr (stage == 7) {
        // the cell state is kept pre-multiplied by KT (c' = KT c): c' = f c'_old + i (KT tanh g), and tanh(c) = 1 - 2/(1 + 2^c')
        float cst;
        if constexpr (ACC & 32) cst = R.call[a][b];
        else cst = R.cs[ap][b];
        float cn;
        if constexpr (ACC & 512) {   // (c' AB + KT (e_g - 1) F) / (F AB)
            cn = __builtin_fmaf(R.nm[k], R.v[k][0][1], cst * R.v[k][0][0]) * R.v[k][1][0];
        } else if constexpr (ACC & 128) {   // KT sigmoid(i) tanh(g) = KT (e_g - 1) / ((1 + e_i)(1 + e_g))
            cn = __builtin_fmaf(R.v[k][0][1], cst, R.nm[k] * R.v[k][0][0]);
        } else if constexpr (ACC & 16) {   // KT i tanh(g) = i' (1 + y') with i' = KT sigmoid(i), y' = tanh(g) - 1: one FMA
            cn = __builtin_fmaf(R.v[k][0][1], cst, __builtin_fmaf(R.v[k][0][0], R.v[k][1][0], R.v[k][0][0]));
        } else {
        const float gg = __builtin_fmaf(-2.0f * KT, R.v[k][1][0], KT);
        cn = __builtin_fmaf(R.v[k][0][1], cst, R.v[k][0][0] * gg);
        }
        if constexpr (ACC & 32) R.call[a][b] = cn;
        else R.cs[ap][b] = cn;
        R.y[k] = cn;
        R.og[k] = R.v[k][1][1];
    } else if constexpr (stage == 8) {
        if constexpr (ACC & 128) R.y[k] = __builtin_amdgcn_exp2f(fminf(R.y[k], 64.0f));
        else R.y[k] = __builtin_amdgcn_exp2f(R.y[k]);
    } else if constexpr (stage == 9) {
        if constexpr (ACC & 128) {   // 2^11 sigmoid(o) tanh(c) = (e_c - 1) / ((1 + e_o) 2^-11 (1 + e_c))
            R.nm[k] = R.y[k] - 1.0f;
            R.y[k] = R.og[k] * (R.y[k] + 1.0f);
        } else
        if constexpr (ACC & 16) R.y[k] = __builtin_fmaf(R.y[k], -0.5f, -0.5f);   // reciprocal = tanh(c) - 1
        else R.y[k] = 1.0f + R.y[k];
    } else if constexpr (stage == 10) {
        if constexpr (ACC & 8) R.y[k] = rd_rcp_nr(R.y[k]);
        else R.y[k] = __builtin_amdgcn_rcpf(R.y[k]);
    } else if constexpr (stage == 11) {
        if constexpr (ACC & 128) R.hs[k] = R.nm[k] * R.y[k];
        else
        if constexpr (ACC & 16) R.hs[k] = __builtin_fmaf(R.og[k], R.y[k], R.og[k]);   // o' (1 + (tanh(c) - 1)), o' = 2^11 sigmoid(o)
        else
        R.hs[k] = R.og[k] * __builtin_fmaf(-2.0f * H_SCALE, R.y[k], H_SCALE);   // 2^11 h = 2^11 o tanh(c)
        R.hv[ap][b] = R.hs[k];                                       // captured state is kept at scale 2^11 (epilogue divides)
    } else if constexpr (stage == 12) {
        // hi/lo split, two cells at a time (cells 2i and 2i+1 of a row-tile; the even cell's 2^11 h waits in R.hs[0]):
        //   P  = {fp16(hs0), fp16(hs1)}                 one v_cvt_pk_f16_f32
        //   r  = hs - fp32(P.half)  (exact, in fp32)    one v_fma_mix_f32 each (fp32 result: the fp16-output form
        //                                               v_fma_mixlo_f16 measurably loses accuracy)
        //   O2 = {fp16(r0), fp16(r1)}                   one v_cvt_pk_f16_f32
        if constexpr (cell & 1) {
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            const f16x2 P = {(_Float16)R.hs[k - 1], (_Float16)R.hs[k]};
            unsigned pbits = __builtin_bit_cast(unsigned, P);
            float r0, r1;
            asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(R.hs[k - 1]), "v"(pbits));
            asm volatile("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(R.hs[k]), "v"(pbits));
            const f16x2 O = {(_Float16)r0, (_Float16)r1};
            R.o1s[ap][b - 1] = P[0]; R.o1s[ap][b] = P[1];
            R.o2[ap][b - 1] = O[0]; R.o2[ap][b] = O[1];
        }
    } else {   // 13: the row-tile's 4 cells are complete
        const int wo = c.j * H16STR + 32 * c.wave + 16 * c.half + 4 * a;
        *reinterpret_cast<f16x4 *>(&S.H1s[TP][0][0] + wo) = R.o1s[ap];
        *reinterpret_cast<f16x4 *>(&S.H2[TP][0][0] + wo) = R.o2[ap];
        if constexpr (!(ACC & 32)) S.cS[TP][a][c.tid] = R.cs[ap];
        if constexpr (ACC & 64) {   // experiment: skip the captured-h store unless a read of this wave finishes in this phase
            if (c.any_last) {
                f32x4 *dst = c.last ? reinterpret_cast<f32x4 *>(&S.Hl[TP * 32 + c.j][32 * c.wave + 16 * c.half + 4 * a]) : &S.dummy[c.tid];
                *dst = R.hv[ap];
            }
        } else {
        f32x4 *dst = c.last ? reinterpret_cast<f32x4 *>(&S.Hl[TP * 32 + c.j][32 * c.wave + 16 * c.half + 4 * a]) : &S.dummy[c.tid];
        *dst = R.hv[ap];
        }
    }
}

template <int TP, int U, int ACC = 0>
__device__ __forceinline__ void rd_ew_unit(Lstm16bSmem &S, EwRegs &R, const f32x16 (&accP)[4], const PhaseCtx &c) {
    rd_ew_cs<TP, EW_CELL[U], EW_STAGE[U], ACC, 2, false>(S, R, accP, c);
}

template <int TP, int U0, int U1, int ACC = 0>
__device__ __forceinline__ void rd_ew_units(Lstm16bSmem &S, EwRegs &R, const f32x16 (&accP)[4], const PhaseCtx &c) {
    if constexpr (U0 < U1) {
        rd_ew_unit<TP, U0, ACC>(S, R, accP, c);
        rd_ew_units<TP, U0 + 1, U1, ACC>(S, R, accP, c);
    }
}

// slot M = MFMA number M followed by its share of gate-math units. Product build (ACC = 0): k-step s = M/12, product (M%12)/4,
// row-tile M%4; products: 0 = W1.H1s, 1 = W2.H1s (W2 = unscaled residual of 16 w, so this pair also carries 2^15), 2 = W1.H2,
// 3 (ACC & 1 only) = W2.H2.
template <int ACC>
struct SlotMap {
    static constexpr int NPR = (ACC & 1) ? 4 : 3;     // products per k-step
    static constexpr int NM = 32 * NPR;               // MFMAs per phase
    static constexpr int NS = NPR - 1;                // small products (pass 1 of the small-first order)
    static constexpr bool SF = (ACC & 2) != 0;
    static constexpr int pass(int M) { return SF ? (M < 32 * NS ? 1 : 2) : 0; }
    static constexpr int s(int M) { return !SF ? M / (4 * NPR) : (M < 32 * NS ? M / (4 * NS) : (M - 32 * NS) / 4); }
    static constexpr int pr(int M) { return !SF ? (M % (4 * NPR)) / 4 : (M < 32 * NS ? 1 + (M % (4 * NS)) / 4 : 0); }
    static constexpr bool first_of_step(int M) { return !SF ? M % (4 * NPR) == 0 : (M < 32 * NS ? M % (4 * NS) == 0 : (M - 32 * NS) % 4 == 0); }
};

template <int TL, int FILL, int M, int ACC = 0>
__device__ __forceinline__ void rd_slots(Lstm16bSmem &S, const f16x8 (&W1)[4][8], const f16x8 (&W2)[4][8], f32x16 (&accC)[4],
                                         const f32x16 (&accP)[4], f16x8 (&Bf)[2][2], EwRegs &R, const PhaseCtx &c,
                                         const _Float16 *h1s, const _Float16 *h2) {
    typedef SlotMap<ACC> SM;
    if constexpr (M < SM::NM) {
        constexpr int s = SM::s(M), pr = SM::pr(M), a = M % 4, pass = SM::pass(M);
        if constexpr (SM::first_of_step(M)) {       // B fragments of the next k-step stream in behind this one's MFMAs
            if constexpr (pass == 0 && s < 7) {
                Bf[(s + 1) & 1][0] = *reinterpret_cast<const f16x8 *>(h1s + 16 * (s + 1));
                Bf[(s + 1) & 1][1] = *reinterpret_cast<const f16x8 *>(h2 + 16 * (s + 1));
            } else if constexpr (pass == 1 && s < 7) {
                Bf[(s + 1) & 1][0] = *reinterpret_cast<const f16x8 *>(h1s + 16 * (s + 1));
                Bf[(s + 1) & 1][1] = *reinterpret_cast<const f16x8 *>(h2 + 16 * (s + 1));
            } else if constexpr (pass == 1 && s == 7) {
                Bf[0][0] = *reinterpret_cast<const f16x8 *>(h1s);                       // pass 2 reads the H1s fragments again
            } else if constexpr (pass == 2 && s < 7) {
                Bf[(s + 1) & 1][0] = *reinterpret_cast<const f16x8 *>(h1s + 16 * (s + 1));
            }
        }
        const f16x8 A = (pr == 1 || pr == 3) ? W2[a][s] : W1[a][s];
        const f16x8 B = Bf[s & 1][pr >= 2 ? 1 : 0];
        if constexpr (ACC & 1024) {
            // timing diagnosis (WRONG results, same magnitudes): the slot's 16,384 MACs as two v_mfma_f32_16x16x32_f16 on two 4-register
            // slices of the accumulator, half of the slot's gate-math units behind each - what a 16x16x32 form of this kernel would
            // cost in time and energy, before writing it (8 of the 192 MFMAs per phase are merged away by the compiler: identical operands)
            constexpr int sa = 2 * ((M / 4) & 1);
            constexpr int U0 = (M * EW_NU) / SM::NM, U1 = ((M + 1) * EW_NU) / SM::NM, UM = (U0 + U1 + 1) / 2;
            f32x4 c0 = {accC[a][4 * sa], accC[a][4 * sa + 1], accC[a][4 * sa + 2], accC[a][4 * sa + 3]};
            f32x4 c1 = {accC[a][4 * sa + 4], accC[a][4 * sa + 5], accC[a][4 * sa + 6], accC[a][4 * sa + 7]};
            if constexpr (M < 8) { c0 = f32x4{0, 0, 0, 0}; c1 = f32x4{0, 0, 0, 0}; }
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c0, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) accC[a][4 * sa + r] = c0[r];
            if constexpr (FILL > 0) {
                rd_ew_units<TL ^ 1, U0, UM, ACC>(S, R, accP, c);
                __builtin_amdgcn_sched_barrier(0);
            }
            const f16x8 A2 = (pr == 1 || pr == 3) ? W1[a][s] : W2[a][s];   // (not the same product again: the compiler would merge the two)
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2, B, c1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) accC[a][4 * sa + 4 + r] = c1[r];
            if constexpr (FILL > 0) {
                rd_ew_units<TL ^ 1, UM, U1, ACC>(S, R, accP, c);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        if constexpr (M < 4) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.0f;
            accC[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, z, 0, 0, 0);
        } else {
            accC[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, accC[a], 0, 0, 0);
        }
        if constexpr (FILL > 0) {
            rd_ew_units<TL ^ 1, (M * EW_NU) / SM::NM, ((M + 1) * EW_NU) / SM::NM, ACC>(S, R, accP, c);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        rd_slots<TL, FILL, M + 1, ACC>(S, W1, W2, accC, accP, Bf, R, c, h1s, h2);
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
template <int TL, int FILL, int ACC = 0>
__device__ __forceinline__ void rd_phase_t32(Lstm16bSmem &S, const f16x8 (&W1)[4][8], const f16x8 (&W2)[4][8], f32x16 (&accC)[4],
                                             f32x16 (&accP)[4], EwRegs &R, int tEW, int codeEW, int wave, int half, int j, int tid) {
    constexpr int TP = TL ^ 1;
    const int boff = j * H16STR + 8 * half;     // this lane's B fragment: row j, k = 16s + 8half + e
    const _Float16 *h1s = &S.H1s[TL][0][0] + boff, *h2 = &S.H2[TL][0][0] + boff;
    f16x8 Bf[2][2];
    Bf[0][0] = *reinterpret_cast<const f16x8 *>(h1s);
    Bf[0][1] = *reinterpret_cast<const f16x8 *>(h2);
    PhaseCtx c;
    c.codeEW = codeEW; c.wave = wave; c.half = half; c.j = j; c.tid = tid;
    c.last = (tEW == S.T[TP * 32 + j] - 1);
    c.any_last = __builtin_amdgcn_ballot_w64(c.last) != 0;
    R.kc[0] = S.lut[wave][half][0][0][codeEW];
    if (FILL > 0) __builtin_amdgcn_sched_barrier(0);
    rd_slots<TL, (FILL > 0 ? FILL : 0), 0, ACC>(S, W1, W2, accC, accP, Bf, R, c, h1s, h2);
    if constexpr (FILL == 0) rd_ew_units<TP, 0, EW_NU, ACC>(S, R, accP, c);
    if constexpr (FILL < 0) {   // bench diagnosis only (wrong results): no gate math, keep the accumulators live
        if (accC[0][0] + accC[1][5] + accC[2][9] + accC[3][15] == 123.456f) S.Hl[TP * 32 + j][tid & 127] = accC[0][1];
    }
    if constexpr (FILL != 7) __syncthreads();   // FILL 7: bench diagnosis only (racy, wrong results): what the barrier costs
}

template <int TP, int ACC, int A, int STAGE, int B>
__device__ __forceinline__ void rd_tail_cells(Lstm16bSmem &S, EwRegs &R, const f32x16 (&accP)[4], const PhaseCtx &c) {
    if constexpr (B < 4) {
        if constexpr (STAGE < 13 || B == 3) rd_ew_cs<TP, 4 * A + B, STAGE, ACC, 4, true>(S, R, accP, c);   // stage 13 = the row-tile's stores
        rd_tail_cells<TP, ACC, A, STAGE, B + 1>(S, R, accP, c);
    }
}
template <int TP, int ACC, int A, int STAGE>
__device__ __forceinline__ void rd_tail_stages(Lstm16bSmem &S, EwRegs &R, const f32x16 (&accP)[4], const PhaseCtx &c) {
    if constexpr (STAGE < 14) {
        rd_tail_cells<TP, ACC, A, STAGE, 0>(S, R, accP, c);
        rd_tail_stages<TP, ACC, A, STAGE + 1>(S, R, accP, c);
    }
}
template <int TP, int ACC, int A>
__device__ __forceinline__ void rd_tail_rowtiles(Lstm16bSmem &S, EwRegs &R, const f32x16 (&accP)[4], const PhaseCtx &c) {
    if constexpr (A < 4) {
        rd_tail_stages<TP, ACC, A, 0>(S, R, accP, c);
        rd_tail_rowtiles<TP, ACC, A + 1>(S, R, accP, c);
    }
}

// The gate math of the last step of tile TP, after the loop: no tile is left whose MFMAs it could hide behind.
template <int TP, int ACC>
__device__ __forceinline__ void rd_phase_ewonly(Lstm16bSmem &S, f32x16 (&accP)[4], EwRegs &R, int tEW, int codeEW, int wave, int half,
                                                int j, int tid) {
    PhaseCtx c;
    c.codeEW = codeEW; c.wave = wave; c.half = half; c.j = j; c.tid = tid;
    c.last = (tEW == S.T[TP * 32 + j] - 1);
    c.any_last = __builtin_amdgcn_ballot_w64(c.last) != 0;
    // No MFMAs to hide behind, so no pipeline of two cells either: the four cells of a row-tile go through every stage side by side
    // (the same operations on the same values as in a phase - the results are bit-identical - but 4 independent chains per lane
    // instead of 2: -0.3 % on one CU, -0.13 % on the full chip).
    rd_tail_rowtiles<TP, ACC, 0>(S, R, accP, c);
    __syncthreads();
}

template <int FILL, int ACC = 0>
__global__ __launch_bounds__(256, 1) void rd_lstm_mfma_f16x3_t32_kernel(DevModel d, ReadBatch rb, float *__restrict__ logits,
                                                                        uint8_t *__restrict__ labels) {
    __shared__ Lstm16bSmem S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, j = lane & 31;

    // Everything a workgroup fetches before its first phase is three dependent round trips (order entry -> steps, length, offset ->
    // the first 64 bases) plus 64 KiB of weights per wave and the input table. Every thread fetches the metadata of read tid >> 2
    // itself (four threads share a read and its addresses), so no barrier separates the trips, and the weight loads - whose
    // v_accvgpr_write statements are scheduling barriers - are issued between them and cover their latency.
    const int row = tid >> 2;
    const int64_t g = (int64_t)blockIdx.x * 64 + row;
    const bool valid = g < rb.n;
    int orig = -1;
    if (valid) orig = rb.order ? rb.order[g] : (int)g;                                            // round trip 1
    f32x4 lut_v[3];
    if constexpr (!(ACC & 4)) {   // the table in this kernel's layout, rows pre-multiplied by -log2 e resp. 2 log2 e (rd_prep_kernel): a straight copy
        const f32x4 *src = reinterpret_cast<const f32x4 *>(d.lut_t32);
#pragma unroll
        for (int k = 0; k < 3; ++k) lut_v[k] = src[tid + 256 * k];
    }
    const float wout_v = d.w_out[(tid >> 7) * 256 + (tid & 127)];
    // resident weights: 4 row-tiles x 8 k-steps x (W1, W2) x 4 registers = 256 registers, all pinned in AGPRs
    f16x8 W1[4][8], W2[4][8];
    const uint4 *wp = reinterpret_cast<const uint4 *>(d.wpack16b) + (size_t)wave * (2 * 4 * 8 * 64) + lane;
    auto load_row_tile = [&](auto a_c) {   // 16 loads in flight, then their 64 v_accvgpr_write
        constexpr int a = decltype(a_c)::value;
        uint4 x[2][8];
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
            for (int s = 0; s < 8; ++s) x[hl][s] = wp[((hl * 4 + a) * 8 + s) * 64];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                uint4 y;
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.x) : "v"(x[hl][s].x));
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.y) : "v"(x[hl][s].y));
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.z) : "v"(x[hl][s].z));
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.w) : "v"(x[hl][s].w));
                if (hl == 0) W1[a][s] = __builtin_bit_cast(f16x8, y);
                else W2[a][s] = __builtin_bit_cast(f16x8, y);
            }
        }
    };
    load_row_tile(std::integral_constant<int, 0>());
    int T = 0, lr = 0;
    long long off = 0;
    if (valid) {                                                                                   // round trip 2
        T = rd_T(rb.steps, orig, rb.max_len);
        lr = rd_T(rb.len, orig, rb.max_len);
        off = rb.off[orig];
    }
    load_row_tile(std::integral_constant<int, 1>());
    const uint8_t *src0 = rb.arena + off;
    const u32x4 raw0 = rd_codes_load(lr, src0, 0);                                                 // round trip 3
    u32x4 raw1 = {0u, 0u, 0u, 0u};   // both code buffers are free now: reads of up to 128 steps never stage inside the phase loop
    if (lr > TC16) raw1 = rd_codes_load(lr, src0, 1);
    load_row_tile(std::integral_constant<int, 2>());
    load_row_tile(std::integral_constant<int, 3>());
    // LDS: state, tables, this tile's metadata and first code chunk
    if ((tid & 3) == 0) { S.T[row] = T; S.Lr[row] = lr; S.off[row] = off; S.orig[row] = orig; }
    {
        int m = T;   // wave maximum of T (16 reads per wave)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
        if (lane == 0) S.wmax[wave] = m;
    }
    for (int i = tid; i < 2 * 32 * H16STR / 2; i += 256) { (reinterpret_cast<uint32_t *>(&S.H1s[0][0][0]))[i] = 0u; (reinterpret_cast<uint32_t *>(&S.H2[0][0][0]))[i] = 0u; }
    for (int i = tid; i < 64 * HSTR; i += 256) (&S.Hl[0][0])[i] = 0.0f;
    for (int i = tid; i < 2 * 4 * 256; i += 256) (&S.cS[0][0][0])[i] = f32x4{0, 0, 0, 0};
    if constexpr (ACC & 4) {
        for (int i = tid; i < 4 * 2 * 4 * 4 * 6 * 4; i += 256) {   // i = ((((w*2 + hf)*4 + a)*4 + b)*6 + code)*4 + gate
            const int gate = i & 3, rest = i >> 2, code = rest % 6, cell = rest / 6;
            const int b = cell & 3, a = (cell >> 2) & 3, hf = (cell >> 4) & 1, w = cell >> 5;
            (reinterpret_cast<float *>(&S.lut[0][0][0][0][0]))[i] = code < 5 ? d.in_lut[code * G4 + gate * HID + 32 * w + 16 * hf + 4 * a + b] : 0.0f;
        }
    } else {
        f32x4 *dst = &S.lut[0][0][0][0][0];
#pragma unroll
        for (int k = 0; k < 3; ++k) dst[tid + 256 * k] = lut_v[k];
    }
    S.wout[tid >> 7][tid & 127] = wout_v;
    rd_codes_store(S, lr, 0, 0, raw0);
    rd_codes_store(S, lr, 1, 1, raw1);
    if (FILL < 0) {   // diagnosis: realistic (pseudo-random) B operands that are never updated
        for (int i = tid; i < 2 * 32 * H16STR / 2; i += 256) {
            uint32_t x = (uint32_t)i * 2654435761u + blockIdx.x * 40503u;
            x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
            (reinterpret_cast<uint32_t *>(&S.H1s[0][0][0]))[i] = (x & 0x83ff83ffu) | 0x34003400u;   // |v| in [0.25, 0.5), random sign+mantissa
            (reinterpret_cast<uint32_t *>(&S.H2[0][0][0]))[i] = ((x * 31u) & 0x83ff83ffu) | 0x34003400u;
        }
    }
    __syncthreads();
    const int tmax = max(max(S.wmax[0], S.wmax[1]), max(S.wmax[2], S.wmax[3]));

    f32x16 X[4], Y[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { X[a][r] = 0.0f; Y[a][r] = 0.0f; }
    int codeY = 5;   // code of (tile 1, step t-1): zero row before the first step
    EwRegs R0, R1;   // gate-math registers of tile 0 / tile 1 (transient per phase, except call[] under ACC & 32)
#pragma unroll
    for (int a = 0; a < 4; ++a) { R0.call[a] = f32x4{0, 0, 0, 0}; R1.call[a] = f32x4{0, 0, 0, 0}; }

    for (int t = 0; t < tmax; ++t) {
        const uint8_t *ccol = &S.codes[(t / TC16) & 1][0][t % TC16];
        const int codeX = ccol[j * CSTR];            // (tile 0, step t): consumed by phase B's gate math
        const int codeYn = ccol[(32 + j) * CSTR];    // (tile 1, step t): consumed by the next iteration's phase A
        // phase A: MFMAs of (tile 0, t) -> X ; gate math of (tile 1, t-1) <- Y. (At t = 0 the phase changes nothing - h = 0, zero
        // accumulators, the all-zero table row - but skipping it behind a branch made the loop 1.2 % slower: measured, left in.)
        rd_phase_t32<0, FILL, ACC>(S, W1, W2, X, Y, R1, t - 1, codeY, wave, half, j, tid);
        // next code chunk (chunks 0 and 1 were staged before the loop): its buffer was last read by the gate math of phase A above (step t-1)
        if ((t % TC16) == 0 && t > 0) {
            const int chunk = t / TC16 + 1;
            if (chunk * TC16 < tmax + 1) rd_codes_store(S, lr, chunk, chunk & 1, rd_codes_load(lr, src0, chunk));
        }
        // phase B: MFMAs of (tile 1, t) -> Y ; gate math of (tile 0, t) <- X
        rd_phase_t32<1, FILL, ACC>(S, W1, W2, Y, X, R0, t, codeX, wave, half, j, tid);
        codeY = codeYn;
    }
    // the gate math of (tile 1, tmax-1): gate math only (the loop used to run one more phase A whose 96 MFMAs computed nothing)
    if constexpr (FILL >= 0) rd_phase_ewonly<1, ACC>(S, Y, R1, tmax - 1, codeY, wave, half, j, tid);

    rd_fc_epilogue(
        64, [&](int row, int u) { return S.Hl[row][u] * (1.0f / H_SCALE); }, S.T, S.Lr, S.off, S.orig, &S.wout[0][0], d, rb, logits,
        labels);
}

}  // namespace t32diag
}  // namespace
