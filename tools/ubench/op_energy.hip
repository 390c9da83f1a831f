// Microbenchmark: the ENERGY price list of the instructions of the recurrence kernel. All 256 CUs, one wave per SIMD, a stream of
// v_mfma_f32_32x32x16_f16 (A in AGPRs, 8 B fragments, 4 accumulators round robin) with K filler instructions of one kind after
// every MFMA. At the package power cap every case draws the same ~1,300 W, so time per MFMA slot is proportional to the energy of the
// slot: price(filler) = (t_with - t_bare) / K, in units of the bare MFMA's energy. Fillers read pseudo-random per-lane operands
// from a pool of registers (operand bits toggle between consecutive instructions) and write to sink registers.
// Build: hipcc --offload-arch=gfx950 -O3 -w op_energy.hip -o op_energy ; ./op_energy [seconds per case]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float rndf(unsigned &s, float scale) {
    s = s * 1664525u + 1013904223u;
    return (((int)(s >> 9) & 0x3fff) * (1.0f / 8192.0f) - 1.0f) * scale;
}

// KIND: 0 none, 1 v_exp_f32, 2 v_rcp_f32, 3 v_fma_f32, 4 v_pk_fma_f32, 5 ds_read_b128, 6 v_cvt_pk_f16_f32 (as v_cvt_pkrtz), 7 v_add_f32,
// 8 ds_write_b64, 9 v_mul_f32, 10 v_min_f32, 11 v_mov_b32
template <int KIND, int K, bool MFMA>
__global__ __launch_bounds__(256, 1) void kern(float *out, int iters) {
    __shared__ float pad[16384];
    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 977u + 12345u;
    f16x8 A[8], B[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) { A[i][j] = (_Float16)rndf(seed, 1.0f); B[i][j] = (_Float16)rndf(seed, 1.0f); }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint4 x = __builtin_bit_cast(uint4, A[i]), y;
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.x) : "v"(x.x));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.y) : "v"(x.y));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.z) : "v"(x.z));
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(y.w) : "v"(x.w));
        A[i] = __builtin_bit_cast(f16x8, y);
    }
    float P[8], D[4] = {0, 0, 0, 0};
    f32x2 P2[4], D2[2] = {{0, 0}, {0, 0}};
    for (int i = 0; i < 8; ++i) P[i] = rndf(seed, 4.0f);
    for (int i = 0; i < 4; ++i) P2[i] = f32x2{rndf(seed, 4.0f), rndf(seed, 4.0f)};
    f32x4 l4 = {0, 0, 0, 0};
    const unsigned addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    for (int i = threadIdx.x; i < 16384; i += 256) pad[i] = rndf(seed, 1.0f);
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            if constexpr (MFMA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "a"(A[m & 7]), "v"(B[(m >> 2) & 7]));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int i = (m * K + k);
                if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %1" : "=v"(D[i & 3]) : "v"(P[i & 7]));
                if constexpr (KIND == 2) asm volatile("v_rcp_f32 %0, %1" : "=v"(D[i & 3]) : "v"(P[i & 7]));
                if constexpr (KIND == 3) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(D[i & 3]) : "v"(P[i & 7]), "v"(P[(i + 3) & 7]), "v"(P[(i + 5) & 7]));
                if constexpr (KIND == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(D2[i & 1]) : "v"(P2[i & 3]), "v"(P2[(i + 1) & 3]), "v"(P2[(i + 2) & 3]));
                if constexpr (KIND == 5) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l4) : "v"(addr), "n"((i & 3) * 1024));
                if constexpr (KIND == 6) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(D[i & 3]) : "v"(P[i & 7]), "v"(P[(i + 3) & 7]));
                if constexpr (KIND == 7) asm volatile("v_add_f32 %0, %1, %2" : "=v"(D[i & 3]) : "v"(P[i & 7]), "v"(P[(i + 3) & 7]));
                if constexpr (KIND == 8) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(P2[i & 3]), "n"((i & 3) * 1024));
                if constexpr (KIND == 9) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(D[i & 3]) : "v"(P[i & 7]), "v"(P[(i + 3) & 7]));
                if constexpr (KIND == 10) asm volatile("v_min_f32 %0, %1, %2" : "=v"(D[i & 3]) : "v"(P[i & 7]), "v"(P[(i + 3) & 7]));
                if constexpr (KIND == 11) asm volatile("v_mov_b32 %0, %1" : "=v"(D[i & 3]) : "v"(P[i & 7]));
            }
        }
        if constexpr (KIND == 5 || KIND == 8) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15] + D[i];
    s += l4[0] + l4[3] + D2[0][0] + D2[1][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double g_base = 0;
template <int KIND, int K, bool MFMA>
static void run(const char *name, float *out, double seconds) {
    const int iters = 5000;
    hipLaunchKernelGGL((kern<KIND, K, MFMA>), dim3(256), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    double ns = 0;
    for (int half = 0; half < 2; ++half) {
        auto t0 = std::chrono::steady_clock::now();
        int n = 0;
        double dt = 0;
        while (dt < seconds) {
            hipLaunchKernelGGL((kern<KIND, K, MFMA>), dim3(256), dim3(256), 0, 0, out, iters);
            hipDeviceSynchronize();
            ++n;
            dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        ns = dt * 1e9 / ((double)n * iters * 32);
    }
    if (KIND == 0) g_base = ns;
    printf("%-44s %7.3f ns per slot", name, ns);
    if (K > 0 && MFMA) printf("   price per filler = %.4f of a bare MFMA slot", (ns - g_base) / K / g_base);
    printf("\n");
    fflush(stdout);
}

int main(int argc, char **argv) {
    double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    float *out;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    run<0, 0, true>("bare 32x32x16 MFMA", out, seconds);
    run<1, 1, true>("+ 1 v_exp_f32", out, seconds);
    run<1, 2, true>("+ 2 v_exp_f32", out, seconds);
    run<2, 1, true>("+ 1 v_rcp_f32", out, seconds);
    run<3, 2, true>("+ 2 v_fma_f32", out, seconds);
    run<3, 4, true>("+ 4 v_fma_f32", out, seconds);
    run<7, 4, true>("+ 4 v_add_f32", out, seconds);
    run<9, 4, true>("+ 4 v_mul_f32", out, seconds);
    run<10, 4, true>("+ 4 v_min_f32", out, seconds);
    run<11, 4, true>("+ 4 v_mov_b32", out, seconds);
    run<4, 2, true>("+ 2 v_pk_fma_f32", out, seconds);
    run<6, 2, true>("+ 2 v_cvt_pkrtz_f16_f32", out, seconds);
    run<5, 1, true>("+ 1 ds_read_b128", out, seconds);
    run<8, 1, true>("+ 1 ds_write_b64", out, seconds);
    return 0;
}
