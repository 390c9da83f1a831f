// Microbenchmark: what do byte-aligned (unaligned) LDS reads cost on gfx950? The deflate kernel's candidate check reads 16 bytes at eight
// arbitrary byte positions per lane and strip; as aligned dwords + v_alignbyte that is 5 ds_read_b32 per candidate. gfx950 runs with
// unaligned DS access enabled (the compiler itself emits ds_read_b64/b96/b128 for align-1 pointers), so one ds_read_b128 could do it.
// One workgroup of 16 waves on one CU (the deflate kernel's shape), 8 independent reads per lane and iteration, cycles per iteration
// of the whole workgroup (clock64 of wave 0, barrier on both sides) -> LDS cycles per wave-instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -w lds_unaligned.hip -o lds_unaligned
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// MODE 0: 5 x ds_read_b32, aligned dwords around a random byte position   (today's candidate read)
//      1: 1 x ds_read_b128 at the random BYTE position
//      2: 1 x ds_read_b128 at the position rounded down to 16 bytes         (aligned reference)
//      3: 2 x ds_read_b64 at the byte position
//      4: 1 x ds_read_b128 at a random 4-byte aligned position
//      5: 5 x ds_read_b32 at consecutive positions (lane l: byte base + l)   (today's own-position read)
//      6: 1 x ds_read_b128 at byte base + l
//      7: 1 x ds_read_b64 at the random byte position (8 bytes: the first check only)
//      8: 3 x ds_read_b32 aligned dwords (8 bytes at any alignment)
//      9: 3 x ds_read_b64 at the position rounded down to 8 bytes (24 bytes: 16 at any alignment)
template <int MODE>
__global__ __launch_bounds__(1024, 1) void kern(uint32_t *out, long long *clk, int iters) {
    __shared__ uint32_t text[16384 + 64];   // 64 KiB like a member
    for (int i = threadIdx.x; i < 16384 + 64; i += 1024) text[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    uint32_t rng = threadIdx.x * 747796405u + 2891336453u;
    uint32_t acc = 0;
    __syncthreads();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        u32x4 r[8];
        uint32_t e[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            rng = rng * 1664525u + 1013904223u;
            uint32_t c = (rng >> 8) & 0xffffu;              // a byte position in the member
            if (MODE == 2) c &= ~15u;
            if (MODE == 4) c &= ~3u;
            if (MODE == 5 || MODE == 6) c = (uint32_t)((it * 8 + w) * 1777 + (threadIdx.x >> 6) * 4099) + lane;   // wave-uniform base
            if (MODE == 9) c &= ~7u;
            c &= 0xffffu;
            r[w] = u32x4{0, 0, 0, 0};
            e[w] = 0;
            if (MODE == 0 || MODE == 5) {
                const uint32_t a = c & ~3u;
                asm volatile("ds_read_b32 %0, %1" : "=v"(r[w].x) : "v"(a));
                asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(r[w].y) : "v"(a));
                asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(r[w].z) : "v"(a));
                asm volatile("ds_read_b32 %0, %1 offset:12" : "=v"(r[w].w) : "v"(a));
                asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(e[w]) : "v"(a));
            } else if (MODE == 8) {
                const uint32_t a = c & ~3u;
                asm volatile("ds_read_b32 %0, %1" : "=v"(r[w].x) : "v"(a));
                asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(r[w].y) : "v"(a));
                asm volatile("ds_read_b32 %0, %1 offset:8" : "=v"(r[w].z) : "v"(a));
            } else if (MODE == 9) {
                u32x2 a0, a1, a2;
                asm volatile("ds_read_b64 %0, %1" : "=v"(a0) : "v"(c));
                asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(a1) : "v"(c));
                asm volatile("ds_read_b64 %0, %1 offset:16" : "=v"(a2) : "v"(c));
                r[w] = u32x4{a0.x ^ a2.x, a0.y ^ a2.y, a1.x, a1.y};
            } else if (MODE == 3) {
                u32x2 lo, hi;
                asm volatile("ds_read_b64 %0, %1" : "=v"(lo) : "v"(c));
                asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(hi) : "v"(c));
                r[w] = u32x4{lo.x, lo.y, hi.x, hi.y};
            } else if (MODE == 7) {
                u32x2 lo;
                asm volatile("ds_read_b64 %0, %1" : "=v"(lo) : "v"(c));
                r[w].x = lo.x; r[w].y = lo.y;
            } else {
                asm volatile("ds_read_b128 %0, %1" : "=v"(r[w]) : "v"(c));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int w = 0; w < 8; ++w) acc ^= r[w].x + r[w].y + r[w].z + r[w].w + e[w];
    }
    __syncthreads();
    const long long c1 = clock64();
    out[threadIdx.x + blockIdx.x * 1024] = acc;
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

// correctness of the unaligned forms: bytes == a byte-wise copy
__global__ void check(uint32_t *bad) {
    __shared__ uint8_t t[4096 + 32];
    for (int i = threadIdx.x; i < 4096 + 32; i += 64) t[i] = (uint8_t)(i * 31 + (i >> 5));
    __syncthreads();
    for (int c = threadIdx.x; c < 4096; c += 64) {
        u32x4 v;
        u32x2 q;
        asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)0 + c));
        asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"((uint32_t)c));
        for (int k = 0; k < 16; ++k)
            if (((v[k >> 2] >> (8 * (k & 3))) & 0xff) != t[c + k]) atomicAdd(bad, 1u);
        for (int k = 0; k < 8; ++k)
            if (((q[k >> 2] >> (8 * (k & 3))) & 0xff) != t[c + k]) atomicAdd(bad + 1, 1u);
    }
}

template <int MODE>
static void run(const char *what, int per_iter_instr) {
    const int iters = 2000;
    uint32_t *out; long long *clk;
    hipMalloc(&out, 1024 * 4); hipMalloc(&clk, 8);
    kern<MODE><<<1, 1024>>>(out, clk, 10);
    kern<MODE><<<1, 1024>>>(out, clk, iters);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    // clock64 = s_memtime at 100 MHz: convert with the wall clock instead -> report both
    const double per_iter = (double)c / iters;
    printf("%-70s %8.1f clk64 ticks / iteration (16 waves x 8 candidates), %6.2f per wave-candidate, %5.2f per LDS instruction\n", what, per_iter,
           per_iter / 128, per_iter / 128 / per_iter_instr);
    hipFree(out); hipFree(clk);
}

int main() {
    uint32_t *bad; hipMalloc(&bad, 8); hipMemset(bad, 0, 8);
    // the check kernel's LDS array sits at LDS address 0 (its only __shared__ object)
    check<<<1, 64>>>(bad);
    uint32_t hb[2]; hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    printf("unaligned ds_read_b128 wrong bytes: %u, ds_read_b64: %u\n", hb[0], hb[1]);
    run<0>("0: 5 x ds_read_b32 aligned around a random byte position", 5);
    run<1>("1: ds_read_b128 at a random BYTE position", 1);
    run<4>("4: ds_read_b128 at a random 4-byte aligned position", 1);
    run<2>("2: ds_read_b128 at a random 16-byte aligned position", 1);
    run<3>("3: 2 x ds_read_b64 at a random byte position", 2);
    run<7>("7: 1 x ds_read_b64 at a random byte position (8 bytes)", 1);
    run<8>("8: 3 x ds_read_b32 aligned (8 bytes at any alignment)", 3);
    run<9>("9: 3 x ds_read_b64, 8-byte aligned window around a random byte position", 3);
    run<5>("5: 5 x ds_read_b32, lane l at byte base + l", 5);
    run<6>("6: ds_read_b128, lane l at byte base + l", 1);
    return 0;
}
