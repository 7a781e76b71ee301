// q3_gguf.h — read-only GGUF (v2/v3) and NPY readers for real model files.
//
// Replaces, on the host side of libq3tts:
//   * the crate's own minimal GGUF reader for qwen3_assets.gguf (/root/reference/src/assets_manager.rs:28-265) and its NPY
//     fallback (:267-381) — same tensor / file names, same "text_embd may be absent" rule;
//   * llama.cpp's model loader for qwen3_tts_talker.gguf / qwen3_tts_predictor.gguf (behind LlamaModel::load,
//     /root/reference/src/models/llama/mod.rs:360-398), for the tensor types the released quant dirs use that this engine
//     can take: F32, F16, BF16, Q8_0 and the K-quants Q4_K / Q5_K / Q6_K of the gguf_q5_k_m directory (src/tts/engine.rs:91-95);
//     every type is de-quantised on the host and stored as bf16 (the decoder's weight format), other types are refused loudly.
// Files are mmap'ed; tensors are converted on the host and uploaded by the engine (q3_engine.hip).
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

enum { Q3_GGML_F32 = 0, Q3_GGML_F16 = 1, Q3_GGML_Q8_0 = 8, Q3_GGML_Q4_K = 12, Q3_GGML_Q5_K = 13, Q3_GGML_Q6_K = 14, Q3_GGML_BF16 = 30 };

struct Q3GgufTensor {
    std::string name;
    uint32_t type = 0;
    std::vector<uint64_t> dims;  // ne[0] is the contiguous (input / K) dimension, as in ggml
    uint64_t offset = 0;         // relative to the data section
    size_t nelem = 0, nbytes = 0;
    const uint8_t* data = nullptr;
};

class Q3Gguf {
public:
    Q3Gguf() = default;
    ~Q3Gguf();
    Q3Gguf(const Q3Gguf&) = delete;
    Q3Gguf& operator=(const Q3Gguf&) = delete;
    // 0 on success; err receives the reason otherwise (never throws, never aborts)
    int open(const std::string& path, std::string& err);
    const Q3GgufTensor* find(const std::string& name) const;
    const std::vector<Q3GgufTensor>& tensors() const { return tensors_; }
    uint32_t version() const { return version_; }
    uint32_t alignment() const { return alignment_; }
    // scalar metadata that was an integer / float (arrays and strings are skipped); false when absent
    bool meta_u64(const std::string& key, uint64_t* v) const;

private:
    void* map_ = nullptr;
    size_t size_ = 0;
    uint32_t version_ = 0, alignment_ = 32;
    std::vector<Q3GgufTensor> tensors_;
    std::map<std::string, size_t> index_;
    std::map<std::string, uint64_t> meta_;
};

// element conversions (ggml semantics: F16 -> f32 exact, Q8_0: f32(d) * q, BF16 -> f32 exact, K-quants: dequantize_row_q{4,5,6}_K)
int q3_gguf_to_f32(const Q3GgufTensor& t, float* dst, std::string& err);
// to bf16 with round-to-nearest-even from the f32 value above (BF16 sources are copied bit for bit)
int q3_gguf_to_bf16(const Q3GgufTensor& t, uint16_t* dst, std::string& err);

// NPY v1/v2, little-endian f32, C order (the reference assumes exactly that: src/assets_manager.rs:302-377)
int q3_npy_load_f32(const std::string& path, std::vector<float>& out, std::vector<size_t>& shape, std::string& err);
