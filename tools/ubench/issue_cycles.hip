// Microbenchmark: ISSUE cost in shader cycles (clock64, one workgroup = 4 waves on one CU, the chip far from its power cap) of an MFMA
// slot with the recurrence's gate-math mix behind it, for both f16 MFMA shapes. Per 16,384 MACs (one 32x32x16 or two 16x16x32) the
// product kernel issues 1.33 transcendentals, 3.33 plain VALU ops and 0.4 LDS instructions.
// Build: hipcc --offload-arch=gfx950 -O3 -w issue_cycles.hip -o issue_cycles
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define TR(i) asm volatile("v_exp_f32 %0, %1" : "=v"(D[(i) & 3]) : "v"(P[(i) & 7]))
#define FM(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(D[(i) & 3]) : "v"(P[(i) & 7]), "v"(P[((i) + 3) & 7]), "v"(P[((i) + 5) & 7]))
#define LD(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l4) : "v"(addr), "n"(((i) & 3) * 1024))

// SHAPE 0: 32x32x16, 1: 16x16x32. MIX: 0 bare; 1 product mix; 2 only the transcendentals; 3 only the plain VALU ops; 4 mix without LDS
template <int SHAPE, int MIX>
__global__ __launch_bounds__(256, 1) void kern(float *out, long long *clk, int iters) {
    __shared__ float pad[16384];
    f16x8 A[8], B[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { A[i][j] = (_Float16)(0.01f * (threadIdx.x % 17 + i + j)); B[i][j] = (_Float16)(0.02f * (threadIdx.x % 13 + i)); }
    float P[8], D[4] = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) P[i] = 0.1f * i + 0.001f * threadIdx.x;
    f32x4 l4 = {0, 0, 0, 0};
    const unsigned addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    for (int i = threadIdx.x; i < 16384; i += 256) pad[i] = i;
    __syncthreads();
    float s = 0;
    long long c0, c1;
    if constexpr (SHAPE == 0) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        c0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 30; ++m) {   // 30 slots: 40 transcendentals, 100 plain, 12 LDS
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(A[m & 7]), "v"(B[(m >> 2) & 7]));
                if constexpr (MIX == 1 || MIX == 2 || MIX == 4) { TR(m); if (m % 3 == 0) TR(m + 1); }
                if constexpr (MIX == 1 || MIX == 3 || MIX == 4) { FM(m); FM(m + 1); FM(m + 2); if (m % 3 == 0) FM(m + 4); }
                if constexpr (MIX == 1) { if (m % 5 == 0 || m % 5 == 2) LD(m); }
            }
            if constexpr (MIX == 1) asm volatile("s_waitcnt lgkmcnt(0)");
        }
        c1 = clock64();
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    } else {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
        c0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 60; ++m) {   // 60 slots = the same MACs: 40 transcendentals, 100 plain, 12 LDS
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(A[m & 7]), "v"(B[(m >> 3) & 7]));
                if constexpr (MIX == 1 || MIX == 2 || MIX == 4) { if (m % 3 != 2) TR(m); }
                if constexpr (MIX == 1 || MIX == 3 || MIX == 4) { FM(m); if (m % 3 != 1) FM(m + 1); }
                if constexpr (MIX == 1) { if (m % 5 == 0) LD(m); }
            }
            if constexpr (MIX == 1) asm volatile("s_waitcnt lgkmcnt(0)");
        }
        c1 = clock64();
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    }
    for (int i = 0; i < 4; ++i) s += D[i];
    s += l4[0] + l4[3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) clk[0] = c1 - c0;
}

template <int SHAPE, int MIX>
static void run(const char *name, float *out, long long *clk) {
    const int iters = 2000;
    long long h = 0;
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL((kern<SHAPE, MIX>), dim3(1), dim3(256), 0, 0, out, clk, iters);
        hipDeviceSynchronize();
    }
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%-52s %7.2f cycles per 16,384 MACs\n", name, (double)h / (iters * 30.0));
}

int main() {
    float *out;
    long long *clk;
    hipMalloc(&out, 4096);
    hipMalloc(&clk, 16);
    run<0, 0>("32x32x16 bare", out, clk);
    run<0, 2>("32x32x16 + 1.33 v_exp_f32", out, clk);
    run<0, 3>("32x32x16 + 3.33 v_fma_f32", out, clk);
    run<0, 4>("32x32x16 + 1.33 exp + 3.33 fma", out, clk);
    run<0, 1>("32x32x16 + 1.33 exp + 3.33 fma + 0.4 ds_read_b128", out, clk);
    run<1, 0>("2 x 16x16x32 bare", out, clk);
    run<1, 2>("2 x 16x16x32 + 1.33 v_exp_f32", out, clk);
    run<1, 3>("2 x 16x16x32 + 3.33 v_fma_f32", out, clk);
    run<1, 4>("2 x 16x16x32 + 1.33 exp + 3.33 fma", out, clk);
    run<1, 1>("2 x 16x16x32 + 1.33 exp + 3.33 fma + 0.4 ds_read_b128", out, clk);
    return 0;
}
