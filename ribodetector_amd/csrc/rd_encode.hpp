// rd_encode.hpp - standalone encoder kernels (reference tensor layouts) and the label kernels (pair fusion, counters)
// Part of the single translation unit rd_kernels.hip (included from there, in order); see that file for the kernel
// inventory and DESIGN.md §3 for the roofline of each kernel.
#pragma once
#include "rd_common.hpp"

namespace {

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

// codes[n][stride] u8 (4 = pad / not ACGTU). HBM-bound only if the byte mapping is cheap and no lane walks bytes: the first
// version (compare chains per byte, a division per 4 bytes, a byte loop wherever 4 bytes crossed a row) ran at 0.28 of the HBM
// peak. Now a lane takes one 16-byte PIECE OF ONE ROW (rows are cut into ceil(stride/16) pieces, so a piece never crosses a row):
//   piece inside the read    one unaligned 16-byte load at its bytes
//   piece holding the read's end (m < 16 valid bytes)   the 16 bytes that END at the read's end (still inside the read, never past
//                            it), mapped, then shifted down by 16-m bytes with the padding code shifted in
//   piece past the end       padding
// and the mapping is a 256-entry table in LDS: 256 bytes = one dword per bank, so two lanes that hit the same bank hit the same
// dword (broadcast) - ds_read_u8 is conflict-free whatever the bases are. Reads shorter than 16 bases take a byte loop.
__device__ __forceinline__ uint32_t rd_map4(const uint8_t *lut, uint32_t raw) {
    return (uint32_t)lut[raw & 0xff] | ((uint32_t)lut[(raw >> 8) & 0xff] << 8) | ((uint32_t)lut[(raw >> 16) & 0xff] << 16) |
           ((uint32_t)lut[raw >> 24] << 24);
}

template <bool VEC>
__global__ __launch_bounds__(256) void rd_encode_codes_kernel(const uint8_t *__restrict__ arena, const int64_t *__restrict__ off,
                                                              const int32_t *__restrict__ len, int64_t n, int max_len, int stride,
                                                              uint8_t *__restrict__ codes) {
    __shared__ int64_t s_off[ENC_R];
    __shared__ int s_T[ENC_R];
    __shared__ __attribute__((aligned(256))) uint8_t s_lut[256];
    s_lut[threadIdx.x] = (uint8_t)rd_code(threadIdx.x);
    const unsigned PP = ((unsigned)stride + 15u) / 16u;   // pieces per row
    for (int64_t r0 = (int64_t)blockIdx.x * ENC_R; r0 < n; r0 += (int64_t)gridDim.x * ENC_R) {
        const int R = (int)(n - r0 < ENC_R ? n - r0 : ENC_R);
        __syncthreads();
        if ((int)threadIdx.x < R) {
            s_off[threadIdx.x] = off[r0 + threadIdx.x];
            s_T[threadIdx.x] = rd_T(len, r0 + threadIdx.x, max_len);
        }
        __syncthreads();
        const unsigned pieces = (unsigned)R * PP;
        uint8_t *dst = codes + (size_t)r0 * stride;
        for (unsigned g = threadIdx.x; g < pieces; g += 256) {
            const unsigned i = g / PP, j0 = (g - i * PP) * 16u;
            const int T = s_T[i], m = T - (int)j0;             // valid input bytes from j0 on
            const uint8_t *src = arena + s_off[i];
            u32x4 w = {0x04040404u, 0x04040404u, 0x04040404u, 0x04040404u};
            if (m >= 16) {
                u32x4 raw;
                __builtin_memcpy(&raw, src + j0, 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = rd_map4(s_lut, raw[q]);
            } else if (m > 0 && T >= 16) {                      // the 16 bytes that end at the read's end, shifted down by 16-m bytes
                u32x4 raw;
                __builtin_memcpy(&raw, src + T - 16, 16);
                uint32_t c[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) { c[q] = rd_map4(s_lut, raw[q]); c[4 + q] = 0x04040404u; }
                const unsigned sh = 16u - (unsigned)m, a = sh >> 2, bsh = (sh & 3u) * 8u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t lo = 0x04040404u, hi = 0x04040404u;   // c[q + a], c[q + a + 1] without dynamic register indexing
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        lo = ((unsigned)k == q + a) ? c[k] : lo;
                        hi = ((unsigned)k == q + a + 1) ? c[k] : hi;
                    }
                    w[q] = bsh ? ((lo >> bsh) | (hi << (32u - bsh))) : lo;
                }
            } else if (m > 0) {                                  // read shorter than 16 bases
                for (int k = 0; k < m; ++k) {
                    const uint32_t cd = s_lut[src[j0 + k]];
                    w[k >> 2] = (w[k >> 2] & ~(0xffu << (8 * (k & 3)))) | (cd << (8 * (k & 3)));
                }
            }
            uint8_t *o = dst + (size_t)i * stride + j0;
            const unsigned ob = (unsigned)stride - j0 < 16u ? (unsigned)stride - j0 : 16u;   // output bytes of this piece
            if (VEC && ob == 16) {
                __builtin_nontemporal_store(w, (u32x4 *)o);       // 4-byte aligned at least (stride % 4 == 0 when VEC)
            } else if (VEC) {
                for (unsigned q = 0; q < ob / 4; ++q) *(uint32_t *)(o + 4 * q) = w[q];
                for (unsigned k = ob & ~3u; k < ob; ++k) o[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
            } else {
                for (unsigned k = 0; k < ob; ++k) o[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
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

}  // namespace
