// rd_recurrence.hpp - what the recurrence kernels share: the read batch descriptor, the fused FC + argmax epilogue, the split-precision scales
// Part of the single translation unit rd_kernels.hip (included from there, in order); see that file for the kernel
// inventory and DESIGN.md §3 for the roofline of each kernel.
#pragma once
#include "rd_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------
// shared pieces of the recurrence kernels
// ------------------------------------------------------------------------------------------------
// a candidate waiting in the model's queue (rd_set_refine_async): where its bases lie, where its results go
struct RefineEntry {
    const uint8_t *bases;
    float *logits;       // the read's logit row
    uint8_t *label;      // or nullptr
    int32_t lr;          // readable bytes = min(len, max_len)
    int32_t max_len;
    int32_t sem;
    int32_t pad_;
};
struct RefineQueue {
    RefineEntry *e;
    uint32_t *count;
    uint32_t cap;
};

struct ReadBatch {
    const uint8_t *arena;
    const int64_t *off;
    const int32_t *len;
    const int32_t *steps;   // timesteps the forward recurrence runs for each read (rd_steps_kernel)
    const int32_t *order;   // sorted position -> read index (nullptr = identity)
    int64_t n;
    int max_len;
    int sem;                // RD_SEM_PACKED: gather at step len-1 (reference GPU path); RD_SEM_PADDED: ribodetector_cpu
    const float *rev_tab;   // padded semantics only
    // prefix-state table of the default kernel (rd_lstm_t32.hpp): row p = the recurrence state after the pk bases
    // whose base-4 number is p; row 4^pk = zeros. pfx[i] = the row read i starts from (4^pk: from the zero state, all its steps);
    // pfx == nullptr: every read starts from row 0 of ptab (a zero row), steps[] are the full step counts.
    const uint8_t *ptab;
    const int32_t *pfx;
    int pk;
    // deferred float64 pass (rd_set_refine_async): the epilogue records the reads whose margin is below rthresh in rq
    RefineQueue rq;
    float rthresh;
};
constexpr int PFX_ROW = 1024;   // bytes per table row: 2^11 h_hi fp16[128] | residual fp16[128] | KT c fp32[128]

// FC + argmax epilogue for one workgroup's reads. hl(row,u) = captured last forward hidden state. Trow / Lrow / offrow count from
// the first base the kernel stepped over; k0row (default kernel only) = the bases before it that a prefix-table row covered.
// logits = b_out + W_out[:, :128] . h_fwd + rev_lut[last base]   (model.py:36; reverse half folded, see header)
template <typename HL>
__device__ __forceinline__ void rd_fc_epilogue(int nrows, HL hl, const int *Trow, const int *Lrow, const long long *offrow,
                                               const int *origrow, const float *s_wout, const DevModel &d, const ReadBatch &rb,
                                               float *logits, uint8_t *labels, const int *k0row = nullptr) {
    const int tid = threadIdx.x;
    if (tid < 2 * nrows) {
        const int row = tid >> 1, k = tid & 1;
        float s = d.b_out[k];
        for (int u = 0; u < HID; ++u) s = __builtin_fmaf(s_wout[k * HID + u], hl(row, u), s);
        const int T = Trow[row];
        if (rb.sem == RD_SEM_PADDED && T > 0) {   // (T == 0 only for the filler rows of the last workgroup)
            // reverse half of output row pos = T-1: the reverse LSTM has walked max_len-1-pos zero rows, then x[pos]
            const int pos = T - 1;
            const int code = pos < Lrow[row] ? rd_code(rb.arena[offrow[row] + pos]) : 4;
            s += rb.rev_tab[((rb.max_len - 1 - pos - (k0row ? k0row[row] : 0)) * 5 + code) * 2 + k];
        } else if (rb.sem != RD_SEM_PADDED && T > 0) {
            s += d.rev_lut[rd_code(rb.arena[offrow[row] + T - 1]) * 2 + k];
        }
        const float other = __shfl_xor(s, 1);
        const int orig = origrow[row];
        if (orig >= 0) {
            logits[(size_t)orig * 2 + k] = s;
            if (labels && k == 0) labels[orig] = other > s ? 1 : 0;   // torch.argmax: first max wins ties -> 0
            if (rb.rq.e && k == 0 && fabsf(other - s) < rb.rthresh) {   // inside the fp32 noise band: float64 later (rd_refine.hpp)
                const uint32_t slot = atomicAdd(rb.rq.count, 1u);      // (a count beyond cap tells the flush that entries are missing)
                const int k0 = k0row ? k0row[row] : 0;
                if (slot < rb.rq.cap)
                    rb.rq.e[slot] = RefineEntry{rb.arena + offrow[row] - k0, logits + (size_t)orig * 2, labels ? labels + orig : nullptr,
                                                Lrow[row] + k0, rb.max_len, rb.sem, 0};
            }
        }
    }
}

// split-precision (f16x3) kernels: operand layout and scales, see rd_lstm_f16x3.hpp
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int H16STR = 136;   // f16 per LDS row: 128 + 8 pad = 272 B -> conflict-free ds_read_b128 over 16 rows
constexpr int TC16 = 64;      // timesteps per staged code chunk
constexpr float W_SCALE = 16.0f, H_SCALE = 2048.0f, G_SCALE = 32768.0f;   // 2^4, 2^11, 2^15
// sigmoid(x) = 1 / (1 + 2^(KS x)),  tanh(x) = 1 - 2 / (1 + 2^(KT x))
constexpr float KS = -1.44269504088896341f, KT = 2.88539008177792681f;

}  // namespace
