// classify_fastq.cpp - the whole path through the two C ABIs, without Python or torch:
//   librd_host.so  (include/ribodetector_amd_host.h)  FASTQ/FASTA(.gz) -> byte arena + offsets + lengths
//   librd_hip.so   (include/ribodetector_amd.h)       arena -> logits + labels on the GPU
// Usage: classify_fastq <weights.safetensors> <reads.fq[.gz]> [max_len=100] [labels_out.txt]
// Prints the counters the reference logs (detect.py:331-333) and, optionally, one label per line.
// Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O2 -I include examples/classify_fastq.cpp -L ribodetector_amd/csrc -lrd_hip -lrd_host \
//         -Wl,-rpath,'$ORIGIN/../ribodetector_amd/csrc' -o examples/classify_fastq
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "ribodetector_amd.h"
#include "ribodetector_amd_host.h"

#define HIP_OK(x)                                                                            \
    do {                                                                                     \
        hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)
#define RD_CHECK(x)                                                                  \
    do {                                                                             \
        if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, rd_last_error()); return 1; } \
    } while (0)

// safetensors = u64 header length, JSON header, raw little-endian tensor data; the ten tensors are F32
static bool load_tensor(const std::vector<uint8_t> &file, const std::string &header, const char *name, size_t count,
                        std::vector<float> &out) {
    const std::string key = std::string("\"") + name + "\"";
    size_t p = header.find(key);
    if (p == std::string::npos) return false;
    p = header.find("\"data_offsets\"", p);
    if (p == std::string::npos) return false;
    p = header.find('[', p);
    unsigned long long a = 0, b = 0;
    if (sscanf(header.c_str() + p, "[%llu,%llu]", &a, &b) != 2 && sscanf(header.c_str() + p, "[%llu, %llu]", &a, &b) != 2) return false;
    if (b - a != count * sizeof(float)) return false;
    const size_t base = 8 + header.size();
    if (base + b > file.size()) return false;
    out.resize(count);
    memcpy(out.data(), file.data() + base + a, count * sizeof(float));
    return true;
}

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <weights.safetensors> <reads.fq[.gz]> [max_len=100] [labels_out.txt]\n", argv[0]);
        return 2;
    }
    const int max_len = argc > 3 ? atoi(argv[3]) : 100;

    // ---- weights ----
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    std::vector<uint8_t> file((size_t)ftell(f));
    fseek(f, 0, SEEK_SET);
    if (fread(file.data(), 1, file.size(), f) != file.size()) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    uint64_t hlen = 0;
    memcpy(&hlen, file.data(), 8);
    const std::string header((const char *)file.data() + 8, (size_t)hlen);
    const char *names[10] = {"rnn.weight_ih_l0", "rnn.weight_hh_l0", "rnn.bias_ih_l0", "rnn.bias_hh_l0", "rnn.weight_ih_l0_reverse",
                             "rnn.weight_hh_l0_reverse", "rnn.bias_ih_l0_reverse", "rnn.bias_hh_l0_reverse", "out.weight", "out.bias"};
    const size_t counts[10] = {512 * 4, 512 * 128, 512, 512, 512 * 4, 512 * 128, 512, 512, 2 * 256, 2};
    std::vector<float> t[10];
    for (int i = 0; i < 10; ++i)
        if (!load_tensor(file, header, names[i], counts[i], t[i])) { fprintf(stderr, "tensor %s not found / wrong size\n", names[i]); return 1; }
    rd_weights w = {t[0].data(), t[1].data(), t[2].data(), t[3].data(), t[4].data(), t[5].data(), t[6].data(), t[7].data(), t[8].data(), t[9].data(), 4, 128, 2};
    rd_model *model = nullptr;
    RD_CHECK(rd_model_create(&w, 0, &model));
    // optional: the prefix-state table (include/ribodetector_amd.h) - the recurrence state after every sequence of k bases in
    // caller-owned memory, built in milliseconds; reads then start k steps in with the same logits. Here k = 10: 1 GiB.
    void *d_table = nullptr, *d_scratch = nullptr;
    const int prefix_k = getenv("RD_PREFIX_K") ? atoi(getenv("RD_PREFIX_K")) : 10;
    if (prefix_k) {
        HIP_OK(hipMalloc(&d_table, rd_prefix_table_bytes(prefix_k)));
        HIP_OK(hipMalloc(&d_scratch, rd_prefix_scratch_bytes(prefix_k)));
        RD_CHECK(rd_set_prefix_table(model, prefix_k, d_table, rd_prefix_table_bytes(prefix_k), d_scratch, rd_prefix_scratch_bytes(prefix_k), nullptr));
        HIP_OK(hipFree(d_scratch));   // only the build needs it
    }

    // ---- read the file in chunks, classify each on the GPU ----
    rd_reader *reader = nullptr;
    if (rd_reader_open(argv[2], -1, &reader) != 0) { fprintf(stderr, "%s\n", rd_host_last_error()); return 1; }
    const int64_t CHUNK = 1 << 20, CAP = CHUNK * 400;
    std::vector<uint8_t> buf((size_t)CAP);
    std::vector<int64_t> rec_start(CHUNK + 1), seq_off(CHUNK);
    std::vector<int32_t> seq_len(CHUNK);
    uint8_t *d_arena = nullptr, *d_labels = nullptr;
    int64_t *d_off = nullptr;
    int32_t *d_len = nullptr;
    float *d_logits = nullptr;
    void *d_ws = nullptr;
    uint64_t *d_counts = nullptr;
    const size_t ws_bytes = rd_classify_workspace_bytes(CHUNK, max_len);
    HIP_OK(hipMalloc((void **)&d_arena, (size_t)CAP));
    HIP_OK(hipMalloc((void **)&d_off, CHUNK * sizeof(int64_t)));
    HIP_OK(hipMalloc((void **)&d_len, CHUNK * sizeof(int32_t)));
    HIP_OK(hipMalloc((void **)&d_logits, CHUNK * 2 * sizeof(float)));
    HIP_OK(hipMalloc((void **)&d_labels, CHUNK));
    HIP_OK(hipMalloc(&d_ws, ws_bytes));
    HIP_OK(hipMalloc((void **)&d_counts, 3 * sizeof(uint64_t)));
    HIP_OK(hipMemset(d_counts, 0, 3 * sizeof(uint64_t)));
    FILE *lab_out = argc > 4 ? fopen(argv[4], "w") : nullptr;
    std::vector<uint8_t> labels(CHUNK);
    int64_t total = 0;
    for (;;) {
        int64_t n = 0, nbytes = 0;
        const int rc = rd_reader_next(reader, CHUNK, buf.data(), CAP, rec_start.data(), seq_off.data(), seq_len.data(), &n, &nbytes);
        if (rc < 0) { fprintf(stderr, "%s\n", rd_host_last_error()); return 1; }
        if (n == 0 && rc == 0) {   // the next record alone needs `nbytes` bytes and the buffer is smaller: this example gives up
            fprintf(stderr, "a record of %lld bytes does not fit into the %lld-byte chunk buffer\n", (long long)nbytes, (long long)CAP);
            return 1;              // (ribodetector_amd/data_loader/fastx_parser.py grows its buffer and calls again)
        }
        if (n > 0) {
            HIP_OK(hipMemcpy(d_arena, buf.data(), (size_t)nbytes, hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(d_off, seq_off.data(), n * sizeof(int64_t), hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(d_len, seq_len.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
            RD_CHECK(rd_classify(model, d_arena, d_off, d_len, n, max_len, d_logits, d_labels, d_ws, ws_bytes, nullptr));
            RD_CHECK(rd_count_labels(d_labels, n, d_counts, nullptr));
            if (lab_out) {
                HIP_OK(hipMemcpy(labels.data(), d_labels, (size_t)n, hipMemcpyDeviceToHost));
                for (int64_t i = 0; i < n; ++i) fputc('0' + labels[i], lab_out), fputc('\n', lab_out);
            }
            total += n;
        }
        if (rc == 1) break;
    }
    uint64_t c[3];
    HIP_OK(hipMemcpy(c, d_counts, sizeof(c), hipMemcpyDeviceToHost));
    if (lab_out) fclose(lab_out);
    printf("Processed %lld sequences in total\nDetected %llu non-rRNA sequences\nDetected %llu rRNA sequences\n", (long long)total,
           (unsigned long long)c[0], (unsigned long long)c[1]);
    rd_reader_close(reader);
    rd_model_destroy(model);
    if (d_table) hipFree(d_table);   // after the model that used it
    hipFree(d_arena); hipFree(d_off); hipFree(d_len); hipFree(d_logits); hipFree(d_labels); hipFree(d_ws); hipFree(d_counts);
    return 0;
}
