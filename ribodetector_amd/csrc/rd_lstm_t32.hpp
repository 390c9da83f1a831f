// rd_lstm_t32.hpp - THE DEFAULT KERNEL: split-precision recurrence on 32x32x16 f16 MFMAs with hand-interleaved gate math (rd_lstm_mfma_f16x3_t32_kernel)
// Part of the single translation unit rd_kernels.hip (included from there, in order); see that file for the kernel
// inventory and DESIGN.md §3 for the roofline of each kernel.
#pragma once
#include "rd_recurrence.hpp"

namespace {

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
//     W1 . H2 - the separate unscaled copy of h_hi (H1) of the 16x16 kernel is gone (8 LDS reads, 4 stores, 8 VALU per phase);
//   * round 2: the cell state of both tiles stays in registers; the gate products
//     share a reciprocal - sigmoid(i) tanh(g) = (e_g - 1) / ((1 + e_i)(1 + e_g)), sigmoid(o) tanh(c) likewise: 5 v_exp_f32 +
//     3 v_rcp_f32 per cell instead of 5 + 5, and 30 % less rounding noise (2.24e-6 rms against float64, below the reference's
//     own torch arithmetic); the captured-h stores are skipped unless a read of the wave finishes; the last step's gate math
//     runs as a gate-math-only tail instead of a whole extra phase.
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CSTR = TC16 + 4;
struct __attribute__((aligned(16))) Lstm16bSmem {
    // hot arrays first: everything the phase loop touches per cell sits below 64 KiB, so its LDS addresses are one base
    // register + a 16-bit immediate offset (no per-access address arithmetic)
    f32x4 lut[4][2][4][4][6];      // [wave][half][a][b][code] -> exp2-argument constants of (i,f,g,o); code 5 = zeros
    _Float16 H1s[2][32][H16STR];   // 2^11 h_hi   (B operand of the W1 and of the unscaled-W2 products)
    _Float16 H2[2][32][H16STR];    // 2^11 h - H1s
    float Hl[64][HSTR];            // h captured at t == T-1
    f32x4 dummy[256];              // sink of predicated-off Hl stores
    float wout[2][HID];
    uint8_t codes[2][64][CSTR];    // [buffer][row][t % TC16], rows padded to 17 words: the lanes' byte reads hit 32 banks
    int T[64];
    int Lr[64];       // readable bytes of the read = min(len, max_len)
    long long off[64];
    int orig[64];
    int k0[64];       // bases of the read covered by its prefix-table row (0 or pk)
    // tile 1's start state waits here until phase A of t = 0 has run (see the kernel): h rows of threads 128..255, cell state per lane
    u32x4 st_h[128][8];
    f32x4 st_c[4][256];
    int tmax;
    int wmax[4];      // per wave: the largest step count of its 16 reads
};

// Code chunks: thread (row = tid >> 2, piece = tid & 3) moves the 16 bases t0 + 16 piece .. + 15 of read `row` with ONE unaligned 16-byte
// load (the first version walked bytes, one dependent global load per loop trip: 16 serialized round trips per chunk and workgroup,
// ~10 us of the ~18 us a workgroup spent outside its phase loop). The piece that holds the read's end loads the 16 bytes that END at
// the read's end (never past it: the arena's size is not known here) and shifts them down; bytes past the end are 0 = code 4.
// (Issuing the load some steps before its use - and a persistent-workgroup form that fetches the next tile's metadata and bases
// inside the phase loop - was built and measured in round 2: equal to this form on a full chip)
__device__ __forceinline__ u32x4 rd_codes_load(int lr, const uint8_t *src, int chunk) {   // lr, src: readable bytes and first byte of read tid >> 2
    const int j0 = chunk * TC16 + 16 * (threadIdx.x & 3);
    const int m = lr - j0;                                     // valid bytes from j0 on
    u32x4 raw = {0u, 0u, 0u, 0u};
    if (m >= 16) {
        __builtin_memcpy(&raw, src + j0, 16);
    } else if (m > 0 && lr >= 16) {
        __builtin_memcpy(&raw, src + lr - 16, 16);
    } else if (m > 0) {                                        // read shorter than 16 bases: bytes, already in place
        for (int k = 0; k < m; ++k) raw[k >> 2] |= (uint32_t)src[j0 + k] << (8 * (k & 3));
    }
    return raw;
}
template <typename SMEM>
__device__ __forceinline__ void rd_codes_store(SMEM &S, int lr, int chunk, int buf, u32x4 raw) {
    const int row = threadIdx.x >> 2, piece = threadIdx.x & 3, j0 = chunk * TC16 + 16 * piece;
    const int m = lr - j0;
    if (m > 0 && m < 16 && lr >= 16) {                         // window that ends at the read's end: shift down by 16 - m bytes
        const unsigned sh = 16u - (unsigned)m, b = sh & 3u;
        if (sh & 8u) raw = u32x4{raw[2], raw[3], 0u, 0u};
        if (sh & 4u) raw = u32x4{raw[1], raw[2], raw[3], 0u};
        raw = u32x4{__builtin_amdgcn_alignbyte(raw[1], raw[0], b), __builtin_amdgcn_alignbyte(raw[2], raw[1], b),
                    __builtin_amdgcn_alignbyte(raw[3], raw[2], b), __builtin_amdgcn_alignbyte(0u, raw[3], b)};
    }
    u32x4 w;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t o = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // rd_code without branches (its ternary chain compiles to sixteen divergent branch ladders here)
            const uint32_t ch = (raw[q] >> (8 * k)) & 0xffu;
            const uint32_t c = 4u - 4u * (ch == 'A') - 3u * (ch == 'C') - 2u * (ch == 'G') - (uint32_t)(ch == 'T') - (uint32_t)(ch == 'U');
            o |= c << (8 * k);
        }
        w[q] = o;
    }
    uint32_t *dst = reinterpret_cast<uint32_t *>(&S.codes[buf][row][16 * piece]);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = w[q];
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
// (schedule above, generated by tools/gen_ew_schedule.py: 16 cells x 13 stages, cell c starts at step floor(6.5 c) so that two cells are in flight and never in
//  the same stage; stage 13 = stores of a finished row-tile; generated offline, units are dealt out in this order)
struct EwRegs {
    // slots of the cells in flight: cell % 2 in the phases (two cells in flight), cell % 4 in the gate-math-only tail (a row-tile's
    // four cells side by side)
    f32x4 kc[4];        // table rows
    f32x2 v[4][2];      // gate pipeline values: {i,f} and {g,o} as register pairs
    float y[4], og[4], hs[4];
    f32x4 hv[2];        // per row-tile, by row-tile parity
    f16x4 o1s[2], o2[2];
    f32x4 call[4];      // the tile's cell state (pre-multiplied by KT), resident in registers across phases
    float nm[4];        // numerators K (e_g - 1) resp. (e_c - 1) of the shared-reciprocal form
};

struct PhaseCtx {       // per-lane constants of a phase
    int codeEW, wave, half, j, tid;
    bool last;
    bool any_last;      // wave-uniform: some read of this wave's tile finishes in this phase
};

// One stage of one cell (the gate math of DESIGN.md §3.1: shared reciprocals, 5 v_exp_f32 + 3 v_rcp_f32 + 20 plain VALU ops per
// cell). NS = register slots of the cells in flight (slot = cell % NS); OWNROW: stage 0 fetches the cell's own table row (the
// tail) instead of the next cell's (the phases, whose first row is fetched before the first unit).
// (The experiment switches this body carried in rounds 1-2 - and their separate copy of round 3 - are gone: git history)
template <int TP, int cell, int stage, int NS, bool OWNROW>
__device__ __forceinline__ void rd_ew_cs(Lstm16bSmem &S, EwRegs &R, const f32x16 (&accP)[4], const PhaseCtx &c) {
    constexpr int a = cell >> 2, b = cell & 3, k = cell % NS, ap = a & 1;
    if constexpr (stage == 0 && OWNROW) {
        R.kc[k] = S.lut[c.wave][c.half][a][b][c.codeEW];
    } else if constexpr (stage == 0) {   // table row of the NEXT cell (this cell's row was fetched one cell ago)
        constexpr int nc = cell + 1;
        if constexpr (nc < 16) R.kc[k ^ 1] = S.lut[c.wave][c.half][nc >> 2][nc & 3][c.codeEW];
    } else if constexpr (stage == 1) {
        // exp2 arguments. Scalar FMAs on purpose: packed fp32 ops (v_pk_fma_f32 / v_pk_add_f32) cost ~10 cycles each beside
        // f16 MFMAs against ~1 for a scalar op (tools/ubench/mfma_fill.hip), so the build also passes -fno-slp-vectorize.
        R.v[k][0][0] = __builtin_fmaf(accP[a][4 * b + 0], KS / G_SCALE, R.kc[k][0]);
        R.v[k][0][1] = __builtin_fmaf(accP[a][4 * b + 1], KS / G_SCALE, R.kc[k][1]);
        R.v[k][1][0] = __builtin_fmaf(accP[a][4 * b + 2], KT / G_SCALE, R.kc[k][2]);
        R.v[k][1][1] = __builtin_fmaf(accP[a][4 * b + 3], KS / G_SCALE, R.kc[k][3]);
    } else if constexpr (stage == 2) {
        R.v[k][0][0] = __builtin_amdgcn_exp2f(R.v[k][0][0]); R.v[k][0][1] = __builtin_amdgcn_exp2f(R.v[k][0][1]);      // e_i, e_f
    } else if constexpr (stage == 3) {
        R.v[k][1][0] = __builtin_amdgcn_exp2f(fminf(R.v[k][1][0], 64.0f));   // e_g stays finite: (e_g - 1) * rcp(inf) must not be inf * 0
        R.v[k][1][1] = __builtin_amdgcn_exp2f(R.v[k][1][1]);                  // e_o
    } else if constexpr (stage == 4) {
        R.nm[k] = __builtin_fmaf(R.v[k][1][0], KT, -KT);                                        // KT (e_g - 1)
        R.v[k][0][0] += 1.0f; R.v[k][0][1] += 1.0f; R.v[k][1][0] += 1.0f;                      // 1 + e_i, 1 + e_f, 1 + e_g
        R.v[k][1][1] = __builtin_fmaf(R.v[k][1][1], 1.0f / H_SCALE, 1.0f / H_SCALE);            // (1 + e_o) 2^-11
    } else if constexpr (stage == 5) {
        R.v[k][0][1] = __builtin_amdgcn_rcpf(R.v[k][0][1]); R.v[k][0][0] *= R.v[k][1][0];       // sigmoid(f); (1 + e_i)(1 + e_g)
    } else if constexpr (stage == 6) {
        R.v[k][0][0] = __builtin_amdgcn_rcpf(R.v[k][0][0]);
    } else if constexpr (stage == 7) {
        // the cell state is kept pre-multiplied by KT (c' = KT c): c' = f c'_old + KT sigmoid(i) tanh(g), with
        // KT sigmoid(i) tanh(g) = KT (e_g - 1) / ((1 + e_i)(1 + e_g)); tanh(c) = 1 - 2/(1 + 2^c')
        const float cst = R.call[a][b];
        const float cn = __builtin_fmaf(R.v[k][0][1], cst, R.nm[k] * R.v[k][0][0]);
        R.call[a][b] = cn;
        R.y[k] = cn;
        R.og[k] = R.v[k][1][1];
    } else if constexpr (stage == 8) {
        R.y[k] = __builtin_amdgcn_exp2f(fminf(R.y[k], 64.0f));                                  // e_c
    } else if constexpr (stage == 9) {   // 2^11 sigmoid(o) tanh(c) = (e_c - 1) / ((1 + e_o) 2^-11 (1 + e_c))
        R.nm[k] = R.y[k] - 1.0f;
        R.y[k] = R.og[k] * (R.y[k] + 1.0f);
    } else if constexpr (stage == 10) {
        R.y[k] = __builtin_amdgcn_rcpf(R.y[k]);
    } else if constexpr (stage == 11) {
        R.hs[k] = R.nm[k] * R.y[k];                                  // 2^11 h
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
        if (c.any_last) {   // the captured-h store only in phases where a read of this wave finishes (wave-uniform branch)
            f32x4 *dst = c.last ? reinterpret_cast<f32x4 *>(&S.Hl[TP * 32 + c.j][32 * c.wave + 16 * c.half + 4 * a]) : &S.dummy[c.tid];
            *dst = R.hv[ap];
        }
    }
}

template <int TP, int U0, int U1>
__device__ __forceinline__ void rd_ew_units(Lstm16bSmem &S, EwRegs &R, const f32x16 (&accP)[4], const PhaseCtx &c) {
    if constexpr (U0 < U1) {
        rd_ew_cs<TP, EW_CELL[U0], EW_STAGE[U0], 2, false>(S, R, accP, c);
        rd_ew_units<TP, U0 + 1, U1>(S, R, accP, c);
    }
}

// slot M = MFMA number M followed by its share of gate-math units: k-step s = M/12, product (M%12)/4, row-tile M%4; products:
// 0 = W1.H1s, 1 = W2.H1s (W2 = unscaled residual of 16 w, so this pair also carries 2^15), 2 = W1.H2.
constexpr int T32_NM = 96;   // MFMAs per phase
template <int TL, int M>
__device__ __forceinline__ void rd_slots(Lstm16bSmem &S, const f16x8 (&W1)[4][8], const f16x8 (&W2)[4][8], f32x16 (&accC)[4],
                                         const f32x16 (&accP)[4], f16x8 (&Bf)[2][2], EwRegs &R, const PhaseCtx &c,
                                         const _Float16 *h1s, const _Float16 *h2) {
    if constexpr (M < T32_NM) {
        constexpr int s = M / 12, pr = (M % 12) / 4, a = M % 4;
        if constexpr (M % 12 == 0 && s < 7) {       // B fragments of the next k-step stream in behind this one's MFMAs
            Bf[(s + 1) & 1][0] = *reinterpret_cast<const f16x8 *>(h1s + 16 * (s + 1));
            Bf[(s + 1) & 1][1] = *reinterpret_cast<const f16x8 *>(h2 + 16 * (s + 1));
        }
        const f16x8 A = pr == 1 ? W2[a][s] : W1[a][s];
        const f16x8 B = Bf[s & 1][pr >= 2 ? 1 : 0];
        if constexpr (M < 4) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.0f;
            accC[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, z, 0, 0, 0);
        } else {
            accC[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, accC[a], 0, 0, 0);
        }
        rd_ew_units<TL ^ 1, (M * EW_NU) / T32_NM, ((M + 1) * EW_NU) / T32_NM>(S, R, accP, c);
        __builtin_amdgcn_sched_barrier(0);
        rd_slots<TL, M + 1>(S, W1, W2, accC, accP, Bf, R, c, h1s, h2);
    }
}

// One phase (see the comment above the schedule): MFMAs of (tile TL, current step) into accC; gate math of (tile TL^1, step tEW)
// from accP; one workgroup barrier.
template <int TL>
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
    __builtin_amdgcn_sched_barrier(0);
    rd_slots<TL, 0>(S, W1, W2, accC, accP, Bf, R, c, h1s, h2);
    __syncthreads();
}

template <int TP, int A, int STAGE, int B>
__device__ __forceinline__ void rd_tail_cells(Lstm16bSmem &S, EwRegs &R, const f32x16 (&accP)[4], const PhaseCtx &c) {
    if constexpr (B < 4) {
        if constexpr (STAGE < 13 || B == 3) rd_ew_cs<TP, 4 * A + B, STAGE, 4, true>(S, R, accP, c);   // stage 13 = the row-tile's stores
        rd_tail_cells<TP, A, STAGE, B + 1>(S, R, accP, c);
    }
}
template <int TP, int A, int STAGE>
__device__ __forceinline__ void rd_tail_stages(Lstm16bSmem &S, EwRegs &R, const f32x16 (&accP)[4], const PhaseCtx &c) {
    if constexpr (STAGE < 14) {
        rd_tail_cells<TP, A, STAGE, 0>(S, R, accP, c);
        rd_tail_stages<TP, A, STAGE + 1>(S, R, accP, c);
    }
}
template <int TP, int A>
__device__ __forceinline__ void rd_tail_rowtiles(Lstm16bSmem &S, EwRegs &R, const f32x16 (&accP)[4], const PhaseCtx &c) {
    if constexpr (A < 4) {
        rd_tail_stages<TP, A, 0>(S, R, accP, c);
        rd_tail_rowtiles<TP, A + 1>(S, R, accP, c);
    }
}

// The gate math of the last step of tile TP, after the loop: no tile is left whose MFMAs it could hide behind.
template <int TP>
__device__ __forceinline__ void rd_phase_ewonly(Lstm16bSmem &S, f32x16 (&accP)[4], EwRegs &R, int tEW, int codeEW, int wave, int half,
                                                int j, int tid) {
    PhaseCtx c;
    c.codeEW = codeEW; c.wave = wave; c.half = half; c.j = j; c.tid = tid;
    c.last = (tEW == S.T[TP * 32 + j] - 1);
    c.any_last = __builtin_amdgcn_ballot_w64(c.last) != 0;
    // No MFMAs to hide behind, so no pipeline of two cells either: the four cells of a row-tile go through every stage side by side
    // (the same operations on the same values as in a phase - the results are bit-identical - but 4 independent chains per lane
    // instead of 2: -0.3 % on one CU, -0.13 % on the full chip).
    rd_tail_rowtiles<TP, 0>(S, R, accP, c);
    __syncthreads();
}

// BUILD = false: classification (the product path). Every read starts from a row of the prefix-state table: the recurrence state after
// its first pk bases (rd_steps_kernel chose the row and took pk off the step count), or the all-zero row.
// BUILD = true: the same code builds that table, one level per launch (rd_set_prefix_table): "read" g of level j = rb.pk is the
// j-base prefix whose base-4 number is g; it starts from row g >> 2 of the level j-1 table (rb.ptab; level 1: a zero row), steps
// over its last base g & 3, and instead of the FC epilogue the workgroup writes the state - the two fp16 arrays of h as they stand
// in LDS and the cell state as it stands in the registers - into row g of the level-j table (`logits` then points to it).
// Building with the classifying code is what makes a table start bit-identical to stepping over the bases; building level by
// level costs 4/3 4^k steps instead of k 4^k.
template <bool BUILD>
__global__ __launch_bounds__(256, 1) void rd_lstm_mfma_f16x3_t32_kernel(DevModel d, ReadBatch rb, float *__restrict__ logits,
                                                                        uint8_t *__restrict__ labels) {
    __shared__ Lstm16bSmem S;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, j = lane & 31;

    // Everything a workgroup fetches before its first phase is three dependent round trips (order entry -> steps, length, offset,
    // table row -> the first 64 bases and the 1 KiB of start state) plus 64 KiB of weights per wave and the input table. Every thread
    // fetches the metadata of read tid >> 2 itself (four threads share a read and its addresses) and of the two reads whose cell
    // state its lane holds (j and 32 + j), so no barrier separates the trips, and the weight loads - whose v_accvgpr_write
    // statements are scheduling barriers - are issued between them and cover their latency.
    const int row = tid >> 2, piece = tid & 3;
    const int64_t g = (int64_t)blockIdx.x * 64 + row;
    const bool valid = g < rb.n;
    const int nz = BUILD ? 0 : (rb.pk > 0 ? 1 << (2 * rb.pk) : 0);   // the zero row (BUILD: only level 1 starts there, from a one-row table)
    const int64_t gc0 = (int64_t)blockIdx.x * 64 + j, gc1 = gc0 + 32;
    const bool use_pfx = !BUILD && rb.pfx != nullptr;
    int orig = -1, oc0 = -1, oc1 = -1;
    if (valid) orig = (!BUILD && rb.order) ? rb.order[g] : (int)g;                                 // round trip 1
    if (use_pfx) {
        if (gc0 < rb.n) oc0 = rb.order ? rb.order[gc0] : (int)gc0;
        if (gc1 < rb.n) oc1 = rb.order ? rb.order[gc1] : (int)gc1;
    }
    f32x4 lut_v[3];
    {   // the table in this kernel's layout, rows pre-multiplied by -log2 e resp. 2 log2 e (rd_prep_kernel): a straight copy
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
    int T = 0, lr = 0, k0 = 0, prow = nz, pc0 = nz, pc1 = nz;
    long long off = 0;
    if (BUILD) {
        if (valid) { T = 1; lr = 1; if (rb.pk > 1) prow = (int)(g >> 2); }
        if (rb.pk > 1) { if (gc0 < rb.n) pc0 = (int)(gc0 >> 2); if (gc1 < rb.n) pc1 = (int)(gc1 >> 2); }
    } else if (valid) {                                                                            // round trip 2
        T = rd_T(rb.steps, orig, rb.max_len);
        lr = rd_T(rb.len, orig, rb.max_len);
        off = rb.off[orig];
        if (use_pfx) prow = rb.pfx[orig];
    }
    if (use_pfx) {
        if (oc0 >= 0) pc0 = rb.pfx[oc0];
        if (oc1 >= 0) pc1 = rb.pfx[oc1];
    }
    if (!BUILD && prow != nz) { k0 = rb.pk; lr -= k0; off += k0; }      // the row covers the first pk bases: the kernel steps over the rest
    load_row_tile(std::integral_constant<int, 1>());
    const uint8_t *src0 = rb.arena + off;
    u32x4 raw0, raw1 = {0u, 0u, 0u, 0u};   // both code buffers are free now: reads of up to 128 steps never stage inside the phase loop
    if (BUILD) {   // the last base of prefix g = its least significant digit (rd_steps_kernel: first base = most significant digit)
        raw0 = u32x4{0u, 0u, 0u, 0u};
        if (valid && piece == 0) raw0[0] = (uint32_t)"ACGT"[g & 3];
    } else {
        raw0 = rd_codes_load(lr, src0, 0);                                                         // round trip 3
        if (lr > TC16) raw1 = rd_codes_load(lr, src0, 1);
    }
    // start state: thread (row, piece) moves 64 B of each fp16 array of h of its read into LDS; lane (j, half) of wave w takes the
    // cell state of units [32w + 16half, +16) of reads j (tile 0) and 32 + j (tile 1) into its registers
    u32x4 hrow[2][4];
    f32x4 c0[4], c1[4];
    {
        const u32x4 *tr = reinterpret_cast<const u32x4 *>(rb.ptab + (size_t)prow * PFX_ROW);
        const f32x4 *t0 = reinterpret_cast<const f32x4 *>(rb.ptab + (size_t)pc0 * PFX_ROW + 512) + 8 * wave + 4 * half;
        const f32x4 *t1 = reinterpret_cast<const f32x4 *>(rb.ptab + (size_t)pc1 * PFX_ROW + 512) + 8 * wave + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q) { hrow[0][q] = tr[4 * piece + q]; hrow[1][q] = tr[16 + 4 * piece + q]; c0[q] = t0[q]; c1[q] = t1[q]; }
    }
    load_row_tile(std::integral_constant<int, 2>());
    load_row_tile(std::integral_constant<int, 3>());
    // LDS: state, tables, this tile's metadata and first code chunk
    if (piece == 0) { S.T[row] = T; S.Lr[row] = lr; S.off[row] = off; S.orig[row] = orig; S.k0[row] = k0; }
    {
        int m = T;   // wave maximum of T (16 reads per wave)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
        if (lane == 0) S.wmax[wave] = m;
    }
    // tile 0 (rows 0-31 = waves 0, 1) starts from its table rows now. Tile 1 must wait: phase A of t = 0 runs tile 1's gate math on
    // zero accumulators with the all-zero table row, which leaves a ZERO state unchanged (and writes h = 0 into tile 1's LDS rows)
    // but would halve a loaded cell state - its rows and registers are filled between the two phases of t = 0, from an LDS staging
    // area (held in registers until then they would be live across the whole loop: spills).
    u32x4 *const h1row = reinterpret_cast<u32x4 *>(&S.H1s[row >> 5][row & 31][32 * piece]);
    u32x4 *const h2row = reinterpret_cast<u32x4 *>(&S.H2[row >> 5][row & 31][32 * piece]);
    if (wave < 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { h1row[q] = hrow[0][q]; h2row[q] = hrow[1][q]; }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { S.st_h[tid - 128][q] = hrow[0][q]; S.st_h[tid - 128][4 + q] = hrow[1][q]; }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) S.st_c[a][tid] = c1[a];
    for (int i = tid; i < 64 * HSTR; i += 256) (&S.Hl[0][0])[i] = 0.0f;
    {
        f32x4 *dst = &S.lut[0][0][0][0][0];
#pragma unroll
        for (int k = 0; k < 3; ++k) dst[tid + 256 * k] = lut_v[k];
    }
    S.wout[tid >> 7][tid & 127] = wout_v;
    rd_codes_store(S, lr, 0, 0, raw0);
    rd_codes_store(S, lr, 1, 1, raw1);
    __syncthreads();
    const int tmax = max(max(S.wmax[0], S.wmax[1]), max(S.wmax[2], S.wmax[3]));

    f32x16 X[4], Y[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { X[a][r] = 0.0f; Y[a][r] = 0.0f; }
    int codeY = 5;   // code of (tile 1, step t-1): zero row before the first step
    EwRegs R0, R1;   // gate-math registers of tile 0 / tile 1 (transient per phase, except the cell state call[])
#pragma unroll
    for (int a = 0; a < 4; ++a) { R0.call[a] = c0[a]; R1.call[a] = f32x4{0, 0, 0, 0}; }

    for (int t = 0; t < tmax; ++t) {
        const uint8_t *ccol = &S.codes[(t / TC16) & 1][0][t % TC16];
        const int codeX = ccol[j * CSTR];            // (tile 0, step t): consumed by phase B's gate math
        const int codeYn = ccol[(32 + j) * CSTR];    // (tile 1, step t): consumed by the next iteration's phase A
        // phase A: MFMAs of (tile 0, t) -> X ; gate math of (tile 1, t-1) <- Y. (At t = 0 the gate math changes nothing - zero state,
        // zero accumulators, the all-zero table row - but skipping it behind a branch made the loop 1.2 % slower: measured, left in.)
        rd_phase_t32<0>(S, W1, W2, X, Y, R1, t - 1, codeY, wave, half, j, tid);
        if (t == 0) {   // tile 1's start state (see above); phase B's MFMAs read the rows
            if (wave >= 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { h1row[q] = S.st_h[tid - 128][q]; h2row[q] = S.st_h[tid - 128][4 + q]; }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) R1.call[a] = S.st_c[a][tid];
            __syncthreads();
        }
        // next code chunk (chunks 0 and 1 were staged before the loop): its buffer was last read by the gate math of phase A above (step t-1)
        if ((t % TC16) == 0 && t > 0) {
            const int chunk = t / TC16 + 1;
            if (chunk * TC16 < tmax + 1) rd_codes_store(S, lr, chunk, chunk & 1, rd_codes_load(lr, src0, chunk));
        }
        // phase B: MFMAs of (tile 1, t) -> Y ; gate math of (tile 0, t) <- X
        rd_phase_t32<1>(S, W1, W2, Y, X, R0, t, codeX, wave, half, j, tid);
        codeY = codeYn;
    }
    // the gate math of (tile 1, tmax-1): gate math only (the loop used to run one more phase A whose 96 MFMAs computed nothing)
    rd_phase_ewonly<1>(S, Y, R1, tmax - 1, codeY, wave, half, j, tid);

    if constexpr (BUILD) {
        uint8_t *tab = reinterpret_cast<uint8_t *>(logits);
        if (valid) {
            u32x4 *tr = reinterpret_cast<u32x4 *>(tab + (size_t)g * PFX_ROW);
            const u32x4 *h1 = reinterpret_cast<const u32x4 *>(&S.H1s[row >> 5][row & 31][32 * piece]);
            const u32x4 *h2 = reinterpret_cast<const u32x4 *>(&S.H2[row >> 5][row & 31][32 * piece]);
#pragma unroll
            for (int q = 0; q < 4; ++q) { tr[4 * piece + q] = h1[q]; tr[16 + 4 * piece + q] = h2[q]; }
        }
        f32x4 *t0 = reinterpret_cast<f32x4 *>(tab + (size_t)gc0 * PFX_ROW + 512) + 8 * wave + 4 * half;
        f32x4 *t1 = reinterpret_cast<f32x4 *>(tab + (size_t)gc1 * PFX_ROW + 512) + 8 * wave + 4 * half;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (gc0 < rb.n) t0[a] = R0.call[a];
            if (gc1 < rb.n) t1[a] = R1.call[a];
        }
    } else {
        rd_fc_epilogue(
            64, [&](int row, int u) { return S.Hl[row][u] * (1.0f / H_SCALE); }, S.T, S.Lr, S.off, S.orig, &S.wout[0][0], d, rb, logits,
            labels, S.k0);
    }
}

}  // namespace
