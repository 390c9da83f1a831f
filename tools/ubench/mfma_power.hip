// Microbenchmark: SUSTAINED f16 MFMA throughput of the whole chip at the package power cap, by instruction shape and by
// where the A operand lives (AGPR as in the recurrence kernel, or VGPR). One wave per SIMD, 256 workgroups, pseudo-random
// fp16 operands (operand bits toggle like real data), ~8 s per case so that the power management settles.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power ; run next to `rocm-smi --showpower --showclocks`.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f16x8 rnd8(unsigned &s) {
    f16x8 r;
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u;
        r[i] = (_Float16)(((int)(s >> 9) & 0x3fff) * (1.0f / 8192.0f) - 1.0f);   // uniform in [-1, 1)
    }
    return r;
}

// SHAPE 0: 16x16x32 (4 acc regs), SHAPE 1: 32x32x16 (16 acc regs). NB independent B fragments, NA A fragments.
// FILL 1: the gate-math mix of the recurrence kernel dealt out behind the MFMAs - per 16,384 MACs ~0.83 transcendentals,
// ~1.35 VALU ops and 1/12 ds_read_b128 (i.e. per 16x16x32 MFMA; twice that per 32x32x16 MFMA).
#define TR(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define FM(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(fb), "v"(fa))
#define LD(x) asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(addr))
template <int SHAPE, int FILL>
__global__ __launch_bounds__(256, 1) void kern(float *out, int iters) {
    __shared__ float pad[30000];   // one workgroup per CU
    float v[8];
    f32x4 l4 = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.1f;
    const float fa = 1e-3f, fb = 0.999f;
    const unsigned addr = (threadIdx.x & 63) * 16;
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x;
    f16x8 A[8], B[4];
    for (int i = 0; i < 8; ++i) A[i] = rnd8(seed);
    for (int i = 0; i < 4; ++i) B[i] = rnd8(seed);
    float s = 0;
    if (SHAPE == 0) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                {
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(A[(c + m) & 7]), "v"(B[m & 3]));
                    if (FILL) {   // 5 trans + 8 fma per 6 MFMAs, one LDS read per 12
                        constexpr int ph = 0;
                        const int q = (m * 8 + c) % 6;
                        if (q != 5) TR(v[(c + 1) & 7]);
                        FM(v[(c + 3) & 7]);
                        if (q == 1 || q == 4) FM(v[(c + 5) & 7]);
                        if ((m * 8 + c) % 12 == 0) LD(l4);
                        (void)ph;
                    }
                }
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(A[(c + m) & 7]), "v"(B[m & 3]));
                    if (FILL) {   // 5 trans + 8 fma per 3 MFMAs, one LDS read per 6
                        const int q = (m * 4 + c) % 3;
                        TR(v[(c + 1) & 7]);
                        if (q != 2) TR(v[(c + 2) & 7]);
                        FM(v[(c + 3) & 7]); FM(v[(c + 4) & 7]);
                        if (q != 0) FM(v[(c + 5) & 7]);
                        if ((m * 4 + c) % 6 == 0) LD(l4);
                    }
                }
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    }
    for (int i = 0; i < 8; ++i) s += v[i];
    s += l4[0];
    pad[threadIdx.x] = s;
    out[blockIdx.x * 256 + threadIdx.x] = pad[threadIdx.x];
}

template <int SHAPE, int FILL>
static void run(const char *name, float *out, double seconds) {
    const int iters = 20000;
    const double macs_per_launch = 256.0 * 4 /*waves*/ * iters * 16.0 * (SHAPE == 0 ? 8 : 4) * (SHAPE == 0 ? 16.0 * 16 * 32 : 32.0 * 32 * 16);
    hipLaunchKernelGGL((kern<SHAPE, FILL>), dim3(256), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    double dt = 0;
    while (dt < seconds) {
        hipLaunchKernelGGL((kern<SHAPE, FILL>), dim3(256), dim3(256), 0, 0, out, iters);
        hipDeviceSynchronize();
        ++n;
        dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // report the second half only (settled clocks)
    auto t1 = std::chrono::steady_clock::now();
    int n2 = 0;
    double dt2 = 0;
    while (dt2 < seconds) {
        hipLaunchKernelGGL((kern<SHAPE, FILL>), dim3(256), dim3(256), 0, 0, out, iters);
        hipDeviceSynchronize();
        ++n2;
        dt2 = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    }
    printf("%-28s %7.1f TFLOP/s sustained (%.1f s)\n", name, 2.0 * macs_per_launch * n2 / dt2 / 1e12, dt2);
    fflush(stdout);
}

int main(int argc, char **argv) {
    double seconds = argc > 1 ? atof(argv[1]) : 5.0;
    float *out;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    run<0, 0>("16x16x32_f16 bare", out, seconds);
    run<1, 0>("32x32x16_f16 bare", out, seconds);
    run<0, 1>("16x16x32_f16 + gate-math mix", out, seconds);
    run<1, 1>("32x32x16_f16 + gate-math mix", out, seconds);
    return 0;
}
