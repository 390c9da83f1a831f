// rd_common.hpp - types, constants and small device helpers shared by every kernel of librd_hip.so (encoder table, length
// clamp, activations, the device-side model, the opaque rd_model of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ribodetector_amd.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int HID = 128;     // hidden size
constexpr int G4 = 512;      // 4 gates x hidden
constexpr int NT = 4;        // 16-read tiles per workgroup (ring)
constexpr int BT = NT * 16;  // reads per workgroup
constexpr int HSTR = 132;    // LDS row stride (floats) of an h tile: 128 + 4 pad -> conflict-free b128 reads
constexpr int TC = 128;      // timesteps per staged code chunk

thread_local char g_err[512] = "";

#define RD_FAIL(code, ...)                              \
    do {                                                \
        snprintf(g_err, sizeof(g_err), __VA_ARGS__);    \
        return (code);                                  \
    } while (0)
#define RD_HIP(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) RD_FAIL(RD_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));  \
    } while (0)

// ------------------------------------------------------------------------------------------------
// encoder: seq_encoder.py:11-18  A C G T U(=T) -> 0 1 2 3 ; anything else (lowercase included) -> 4
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int rd_code(unsigned ch) {
    return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : (ch == 'T' || ch == 'U') ? 3 : 4;
}

__device__ __forceinline__ int rd_T(const int32_t *len, int64_t i, int max_len) {
    int t = len[i];
    t = t < 0 ? 0 : t;
    return t < max_len ? t : max_len;
}

// ------------------------------------------------------------------------------------------------
// activations. v_exp_f32 evaluates 2^x to ~1 ulp; the argument x*log2(e) is formed with an FMA-compensated
// product so the result stays within ~2 ulp of expf over the whole range (SURVEY §7 "transcendental accuracy").
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rd_exp(float x) {
    const float L2E_HI = 1.44269502162933349609375f;    // fl(log2 e)
    const float L2E_LO = 1.925963033500011e-8f;         // log2 e - L2E_HI
    float t = x * L2E_HI;
    float e = __builtin_fmaf(x, L2E_HI, -t);             // rounding error of the product
    e = __builtin_fmaf(x, L2E_LO, e);
    float r = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(r, e * 0.693147182464599609375f, r);   // 2^(t+e) ~= 2^t (1 + e ln2)
}
__device__ __forceinline__ float rd_sigmoid(float x) {
    x = fmaxf(x, -80.0f);                                // keep exp(-x) finite: rcp(inf) would still be 0, but avoid inf*0 downstream
    return __builtin_amdgcn_rcpf(1.0f + rd_exp(-x));
}
__device__ __forceinline__ float rd_tanh(float x) {     // tanh x = 2 sigmoid(2x) - 1
    return __builtin_fmaf(2.0f, rd_sigmoid(2.0f * x), -1.0f);
}

// ------------------------------------------------------------------------------------------------
// model blob (device)
// ------------------------------------------------------------------------------------------------
struct DevModel {
    float *raw;       // uploaded tensors, concatenated
    float *wpack32;   // [4 waves][8 col tiles][32 k-steps][64 lanes]  fp32 MFMA B-operand order
    float *wt_hh;     // [128][512]  W_hh^T  (simple kernel)
    float *in_lut;    // [5][512]    W_ih[:,code] + (b_ih + b_hh); code 4 = bias only
    float *rev_lut;   // [5][2]      W_out[:,128:] . h_rev(one step from zero on `code`)
    float *w_out;     // [2][256]
    float *b_out;     // [2]
    uint32_t *wpack16b; // f16x3 32x32x16 A operand
    float *lut_t32;     // in_lut in the LDS layout of the default kernel: [wave][half][a][b][code 0..5][gate], exp2-argument scale folded in
    float *rev_tab;     // padded (ribodetector_cpu) semantics: [max_len][5][2] reverse-direction logit terms, see rd_revtab_kernel
    uint8_t *zero_row;  // one all-zero prefix-table row (1 KiB): where every read starts while no prefix-state table is attached
};

}  // namespace

struct rd_model {
    int device;
    int variant;
    int semantics;      // RD_SEM_PACKED / RD_SEM_PADDED
    int rev_tab_len;    // max_len the padded-semantics table was built for (0 = none)
    float refine_thresh;  // margin below which a read is re-evaluated in float64 (rd_refine.hpp); 0 = off
    int prefix_k;         // bases covered by a row of the attached prefix-state table (0 = none attached)
    const uint8_t *ptab;  // the table (caller-owned memory, rd_set_prefix_table), (4^prefix_k + 1) rows of 1 KiB
    int ptab_variant;     // the kernel whose state the rows hold (the variant that was current when the table was built)
    DevModel d;
    // deferred float64 pass (rd_set_refine_async, rd_kernels.hip): the candidates of refine_async consecutive calls are recorded
    // in a device queue and evaluated together on a stream the model owns; two queues alternate
    int refine_async;              // calls per group (0 = the pass runs inline in rd_classify)
    hipStream_t side;
    hipEvent_t ev_fork, ev_join[2];
    void *q_e[2];                  // RefineEntry[RD_REFINE_QCAP]
    uint32_t *q_count[2];
    int q_cur, q_calls, q_flushing[2];
    int q_wait[2];                 // a flush of queue x has been issued: whoever records into x next waits for ev_join[x] first
    struct { const void *p[5]; int64_t n; int max_len, sem; float thresh; } q_pend[2][16];   // buffers of the calls whose candidates wait in queue x
    int q_npend[2];
    // profiling of the recurrence kernel (bench.py roofline)
    int prof_enabled;
    int prof_count;
    hipEvent_t prof_ev[2 * 512];
    double prof_ms_accum;
    int64_t prof_launches_accum;
};

