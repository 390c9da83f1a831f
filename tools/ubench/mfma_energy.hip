// Microbenchmark: which properties of an MFMA stream change its ENERGY? All 256 CUs, one wave per SIMD, 8 s per case: at the package
// power cap every case draws the same ~1,300 W, so the sustained rate (= the clock the power management settles at, the pipe is
// 100 % busy in every case) measures energy per MFMA. Cases: accumulator rotation (4 accumulators round robin) vs a dependent chain
// on one accumulator (SrcC forwarded inside the matrix core), A operand in AGPRs vs VGPRs, constant vs pseudo-random operands,
// 32x32x16 vs 16x16x32.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_energy.hip -o mfma_energy ; ./mfma_energy [seconds per case]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f16x8 rnd8(unsigned &s, float scale) {
    f16x8 r;
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u;
        r[i] = (_Float16)((((int)(s >> 9) & 0x3fff) * (1.0f / 8192.0f) - 1.0f) * scale);
    }
    return r;
}

// bit 5: the recurrence's gate math dealt out behind the MFMAs - per 16,384 MACs (one 32x32x16 or two 16x16x32) 1.33 transcendentals,
// 3 plain VALU ops and ~0.4 ds_read_b128 (product kernel of round 2: 128 + 288 + 40 per 96 MFMAs of 32x32x16)
#define TR(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define FM(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(fb), "v"(fa))
#define LD(x) asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"(addr))
// bit 6 / 7: the A operand changes only every 2nd / 4th MFMA; bit 8: B changes every MFMA
// MODE bit 0: chain (one accumulator) instead of rotating over 4; bit 1: A operands in AGPRs; bit 2: 16x16x32; bit 3: 8 distinct
// B fragments cycling (instead of 1); bit 4: small-magnitude B (like the residual arrays: |x| < 2^-11)
template <int MODE>
__global__ __launch_bounds__(256, 1) void kern(float *out, int iters) {
    __shared__ float pad[30000];
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 977u + 12345u;
    f16x8 A[8], B[8];
    for (int i = 0; i < 8; ++i) A[i] = rnd8(seed, 1.0f);
    for (int i = 0; i < 8; ++i) B[i] = rnd8(seed, (MODE & 16) ? (1.0f / 2048.0f) : 1.0f);
    if constexpr (MODE & 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint4 x = __builtin_bit_cast(uint4, A[i]), y;
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.x) : "v"(x.x));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.y) : "v"(x.y));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.z) : "v"(x.z));
            asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.w) : "v"(x.w));
            A[i] = __builtin_bit_cast(f16x8, y);
        }
    }
    float s = 0;
    float v[8];
    f32x4 l4 = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.1f;
    const float fa = 1e-3f, fb = 0.999f;
    const unsigned addr = (threadIdx.x & 63) * 16;
    if constexpr (MODE & 4) {
        f32x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 64; ++m) {
                constexpr int dummy = 0; (void)dummy;
                const int c = (MODE & 1) ? ((m >> 4) & 3) : (m & 3);
                const int b = (MODE & 8) ? ((m >> 2) & 7) : 0;
                if constexpr (MODE & 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "a"(A[m & 7]), "v"(B[b]));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(A[m & 7]), "v"(B[b]));
                if constexpr (MODE & 32) {
                    const int q = m % 6;
                    if (q == 0 || q == 1 || q == 3 || q == 4) TR(v[(m + 1) & 7]);
                    FM(v[(m + 3) & 7]);
                    if (q == 2 || q == 5 || q == 0) FM(v[(m + 5) & 7]);
                    if (m % 5 == 0) LD(l4);
                }
            }
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 32; ++m) {
                const int c = (MODE & 1) ? ((m >> 3) & 3) : (m & 3);
                const int b = (MODE & 8) ? ((m >> 2) & 7) : 0;
                const int ai = (MODE & 64) ? ((m >> 1) & 7) : (MODE & 128) ? ((m >> 2) & 7) : (m & 7);
                if constexpr (MODE & 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "a"(A[ai]), "v"(B[(MODE & 256) ? (m & 7) : b]));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(A[ai]), "v"(B[(MODE & 256) ? (m & 7) : b]));
                if constexpr (MODE & 32) {
                    const int q = m % 3;
                    TR(v[(m + 1) & 7]);
                    if (q != 2) TR(v[(m + 2) & 7]);
                    FM(v[(m + 3) & 7]); FM(v[(m + 4) & 7]); FM(v[(m + 6) & 7]);
                    if (m % 5 == 0 || m % 5 == 2) LD(l4);
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

template <int MODE>
static void run(const char *name, float *out, double seconds) {
    const int iters = 10000;
    const double macs = 256.0 * 4 * iters * 32.0 * 16384.0;    // both shapes: 32 x 16,384 MACs per iteration and wave
    hipLaunchKernelGGL(kern<MODE>, dim3(256), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    double rate = 0;
    for (int half = 0; half < 2; ++half) {   // the second half is reported (settled clocks)
        auto t0 = std::chrono::steady_clock::now();
        int n = 0;
        double dt = 0;
        while (dt < seconds) {
            hipLaunchKernelGGL(kern<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
            hipDeviceSynchronize();
            ++n;
            dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        rate = 2.0 * macs * n / dt / 1e12;
    }
    printf("%-64s %7.1f TFLOP/s sustained = %.3f GHz at a full pipe\n", name, rate, rate / 2500.0 * 2.4);
    fflush(stdout);
}

int main(int argc, char **argv) {
    double seconds = argc > 1 ? atof(argv[1]) : 4.0;
    float *out;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    run<0>("32x32x16 rotate 4 acc, A vgpr, 1 B", out, seconds);
    run<1>("32x32x16 chain 8 per acc, A vgpr, 1 B", out, seconds);
    run<2>("32x32x16 rotate, A agpr, 1 B", out, seconds);
    run<3>("32x32x16 chain, A agpr, 1 B", out, seconds);
    run<8>("32x32x16 rotate, A vgpr, 8 B cycling every 4", out, seconds);
    run<2 + 8>("32x32x16 rotate, A agpr changing every MFMA, B every 4", out, seconds);
    run<2 + 8 + 64>("32x32x16 rotate, A agpr changing every 2nd MFMA, B every 4", out, seconds);
    run<2 + 8 + 128>("32x32x16 rotate, A agpr changing every 4th MFMA, B every 4", out, seconds);
    run<2 + 256>("32x32x16 rotate, A agpr and B changing every MFMA", out, seconds);
    run<2 + 256 + 128>("32x32x16 rotate, A agpr every 4th, B every MFMA", out, seconds);
    run<9>("32x32x16 chain, A vgpr, 8 B cycling every 4", out, seconds);
    run<8 + 16>("32x32x16 rotate, A vgpr, 8 small B", out, seconds);
    run<4>("16x16x32 rotate 4 acc, A vgpr, 1 B", out, seconds);
    run<5>("16x16x32 chain 16 per acc, A vgpr, 1 B", out, seconds);
    run<4 + 8>("16x16x32 rotate, A vgpr, 8 B", out, seconds);
    run<5 + 8>("16x16x32 chain, A vgpr, 8 B", out, seconds);
    run<5 + 8 + 2>("16x16x32 chain, A agpr, 8 B", out, seconds);
    run<32 + 8 + 2>("32x32x16 rotate, A agpr, 8 B + gate-math mix", out, seconds);
    run<32 + 4 + 8 + 2>("16x16x32 rotate, A agpr, 8 B + gate-math mix", out, seconds);
    run<32 + 5 + 8 + 2>("16x16x32 chain, A agpr, 8 B + gate-math mix", out, seconds);
    return 0;
}
