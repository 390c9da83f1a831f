// Microbenchmark: how many independent VALU instructions hide behind one MFMA when ONE wave runs per SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 mfma_fill.hip -o mfma_fill ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int KIND>   // K filler VALU ops per MFMA; KIND 0: f32 16x16x4 MFMA, 1: f16 16x16x32 MFMA, 2: f16 32x32x16
__global__ __launch_bounds__(256, 1) void kern(float *out, int iters) {
    __shared__ float pad[30000];   // > 80 KB: one workgroup per CU
    f32x4 acc[8];
    float v[8];
    for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0, 0, 0, 0}; v[i] = threadIdx.x * 0.001f + i; }
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(a + i); bh[i] = (_Float16)(b - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(ah), "v"(bh));
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(c + k) & 7]) : "v"(b), "v"(a));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    pad[threadIdx.x] = s;
    out[blockIdx.x * 256 + threadIdx.x] = pad[threadIdx.x];
}

// 32x32x16 f16 MFMA (16 accumulator regs), 4 independent accumulators; FILL 0: v_fma_f32, 1: v_exp_f32, 2: ds_read_b128
template <int K, int FILL>
__global__ __launch_bounds__(256, 1) void kern32(float *out, int iters) {
    __shared__ float pad[30000];
    f32x16 acc[4];
    float v[8];
    f32x4 l[8];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pk[4], pka = {1.0f + threadIdx.x, 0.5f}, pkb = {1.0001f, 0.9999f};
    for (int i = 0; i < 4; ++i) pk[i] = f32x2{(float)i, (float)threadIdx.x};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.001f + i; l[i] = f32x4{0,0,0,0}; }
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(a + i); bh[i] = (_Float16)(b - i); }
    for (int i = threadIdx.x; i < 30000; i += 256) pad[i] = i;
    __syncthreads();
    unsigned addr = (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 32; ++m) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(ah), "v"(bh));
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(c + k) & 7]) : "v"(b), "v"(a));
                    else if (FILL == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(c + k) & 7]));
                    else if (FILL == 3) {   // mixed: first two fillers transcendental, the rest fma
                        if (k < 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(c + k) & 7]));
                        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(c + k) & 7]) : "v"(b), "v"(a));
                    } else if (FILL == 4) {   // mixed: one exp, one rcp, rest fma
                        if (k == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(c + k) & 7]));
                        else if (k == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[(c + k) & 7]));
                        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(c + k) & 7]) : "v"(b), "v"(a));
                    } else if (FILL == 6) {   // packed fp32 fma (two lanes-worth of work per instruction)
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[(c + k) & 3]) : "v"(pkb), "v"(pka));
                    } else if (FILL == 7) {   // packed fp32 add
                        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[(c + k) & 3]) : "v"(pkb));
                    } else if (FILL == 5) {   // one trans + rest fma
                        if (k == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(c + k) & 7]));
                        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(c + k) & 7]) : "v"(b), "v"(a));
                    }
                    else asm volatile("ds_read_b128 %0, %1" : "=v"(l[(c + k) & 7]) : "v"(addr));
                }
            }
        }
        if (FILL == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    for (int i = 0; i < 8; ++i) s += v[i] + l[i][0];
    for (int i = 0; i < 4; ++i) s += pk[i][0] + pk[i][1];
    pad[threadIdx.x] = s;
    out[blockIdx.x * 256 + threadIdx.x] = pad[threadIdx.x];
}

template <int K, int FILL>
void run32(float *d, const char *name) {
    const int iters = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern32<K, FILL><<<256, 256>>>(d, 10);
    hipEventRecord(e0);
    kern32<K, FILL><<<256, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 128;
    printf("%s K=%d: %.3f ms, %.1f ns/MFMA = %.1f cycles @2.39GHz\n", name, K, ms, ms * 1e6 / n, ms * 1e6 / n * 2.39);
}

template <int K, int KIND>
void run(float *d, const char *name) {
    const int iters = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<K, KIND><<<256, 256>>>(d, 10);
    hipEventRecord(e0);
    kern<K, KIND><<<256, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double n = (double)iters * 256;
    printf("%s K=%d: %.3f ms, %.1f ns/MFMA = %.1f cycles @2.39GHz\n", name, K, ms, ms * 1e6 / n, ms * 1e6 / n * 2.39);
}

int main() {
    float *d;
    hipMalloc(&d, 256 * 256 * 4);
    run<0, 0>(d, "f32 16x16x4"); run<1, 0>(d, "f32 16x16x4"); run<2, 0>(d, "f32 16x16x4"); run<3, 0>(d, "f32 16x16x4");
    run<4, 0>(d, "f32 16x16x4"); run<6, 0>(d, "f32 16x16x4"); run<8, 0>(d, "f32 16x16x4");
    run<0, 1>(d, "f16 16x16x32"); run<1, 1>(d, "f16 16x16x32"); run<2, 1>(d, "f16 16x16x32"); run<3, 1>(d, "f16 16x16x32");
    run<4, 1>(d, "f16 16x16x32");
    run32<0, 0>(d, "f16 32x32x16 +fma"); run32<2, 0>(d, "f16 32x32x16 +fma"); run32<4, 0>(d, "f16 32x32x16 +fma"); run32<5, 0>(d, "f16 32x32x16 +fma");
    run32<6, 0>(d, "f16 32x32x16 +fma"); run32<8, 0>(d, "f16 32x32x16 +fma");
    run32<1, 1>(d, "f16 32x32x16 +exp"); run32<2, 1>(d, "f16 32x32x16 +exp"); run32<4, 1>(d, "f16 32x32x16 +exp");
    run32<4, 3>(d, "f16 32x32x16 +2exp+fma"); run32<5, 3>(d, "f16 32x32x16 +2exp+fma"); run32<6, 3>(d, "f16 32x32x16 +2exp+fma"); run32<7, 3>(d, "f16 32x32x16 +2exp+fma");
    run32<5, 4>(d, "f16 32x32x16 +exp+rcp+fma"); run32<6, 4>(d, "f16 32x32x16 +exp+rcp+fma");
    run32<4, 5>(d, "f16 32x32x16 +1exp+fma"); run32<5, 5>(d, "f16 32x32x16 +1exp+fma"); run32<6, 5>(d, "f16 32x32x16 +1exp+fma");
    run32<2, 6>(d, "f16 32x32x16 +pk_fma"); run32<4, 6>(d, "f16 32x32x16 +pk_fma"); run32<6, 6>(d, "f16 32x32x16 +pk_fma");
    run32<2, 7>(d, "f16 32x32x16 +pk_add"); run32<4, 7>(d, "f16 32x32x16 +pk_add");
    run32<1, 2>(d, "f16 32x32x16 +ds_read_b128"); run32<2, 2>(d, "f16 32x32x16 +ds_read_b128");
    return 0;
}
