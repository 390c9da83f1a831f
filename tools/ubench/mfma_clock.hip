// Microbenchmark: what clock does the chip run at as a function of the number of busy CUs? G workgroups (one per CU, 4 waves) issue
// a bare stream of v_mfma_f32_32x32x16_f16 (8 passes = 32 cycles each, four accumulators round robin, register operands); the wall
// time per MFMA / 32 is the cycle time. Also prints the in-kernel s_memtime and s_memrealtime deltas.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_clock.hip -o mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void kern(float *out, long long *clk, int iters) {
    __shared__ float pad[30000];
    f16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (_Float16)(0.001f * (threadIdx.x + i)); B[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(A), "v"(B));
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    pad[threadIdx.x] = s;
    out[blockIdx.x * 256 + threadIdx.x] = pad[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main() {
    float *out;
    long long *clk, h[2];
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;   // 64,000 MFMAs per wave = ~0.85 ms at 2.4 GHz
    const int gs[] = {1, 8, 32, 64, 128, 192, 256, 512};
    for (int g : gs) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern<1>, dim3(g), dim3(256), 0, 0, out, clk, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        const int reps = 20;
        for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(kern<1>, dim3(g), dim3(256), 0, 0, out, clk, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const int rounds = (g + 255) / 256;
        const double ns_per_mfma = ms * 1e6 / reps / rounds / (iters * 32.0);
        printf("workgroups %4d: %.3f ns per MFMA -> %.3f GHz if 32 cycles each; clock64 delta %lld (%.2f per MFMA), wall_clock64 delta %lld\n", g,
               ns_per_mfma, 32.0 / ns_per_mfma, h[0], (double)h[0] / (iters * 32.0), h[1]);
    }
    return 0;
}
