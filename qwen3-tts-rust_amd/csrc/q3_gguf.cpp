// q3_gguf.cpp — GGUF v2/v3 + NPY readers (host only). See q3_gguf.h for what they replace in the reference.
#include "q3_gguf.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>

namespace {

struct Cursor {
    const uint8_t* p; size_t n, pos;
    bool take(void* dst, size_t k) {
        if (k > n - pos) return false;
        memcpy(dst, p + pos, k); pos += k;
        return true;
    }
    bool skip(size_t k) { if (k > n - pos) return false; pos += k; return true; }
    template <class T> bool get(T* v) { return take(v, sizeof(T)); }
    bool str(std::string* s) {
        uint64_t len = 0;
        if (!get(&len) || len > n - pos) return false;
        s->assign((const char*)p + pos, (size_t)len); pos += (size_t)len;
        return true;
    }
};

// GGUF metadata value types
enum { GV_U8 = 0, GV_I8, GV_U16, GV_I16, GV_U32, GV_I32, GV_F32, GV_BOOL, GV_STR, GV_ARR, GV_U64, GV_I64, GV_F64 };
size_t scalar_size(uint32_t t) {
    switch (t) {
        case GV_U8: case GV_I8: case GV_BOOL: return 1;
        case GV_U16: case GV_I16: return 2;
        case GV_U32: case GV_I32: case GV_F32: return 4;
        case GV_U64: case GV_I64: case GV_F64: return 8;
        default: return 0;
    }
}

inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) u = sign;
        else {  // subnormal: normalise
            int e = -1; uint32_t m = man;
            do { ++e; m <<= 1; } while (!(m & 0x400u));
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (exp == 31) u = sign | 0x7f800000u | (man << 13);
    else u = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f; memcpy(&f, &u, 4);
    return f;
}
inline uint16_t f32_to_bf16_rne(float f) {  // same rounding as q3_bf16 (q3_common.h): RNE, NaN stays NaN
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
size_t type_bytes(uint32_t type, size_t nelem, uint64_t ne0, bool* ok) {
    *ok = true;
    switch (type) {
        case Q3_GGML_F32: return nelem * 4;
        case Q3_GGML_F16: case Q3_GGML_BF16: return nelem * 2;
        case Q3_GGML_Q8_0: if (ne0 % 32) { *ok = false; return 0; } return nelem / 32 * 34;
        case Q3_GGML_Q4_K: if (ne0 % 256) { *ok = false; return 0; } return nelem / 256 * 144;
        case Q3_GGML_Q5_K: if (ne0 % 256) { *ok = false; return 0; } return nelem / 256 * 176;
        case Q3_GGML_Q6_K: if (ne0 % 256) { *ok = false; return 0; } return nelem / 256 * 210;
        default: *ok = false; return 0;
    }
}

// ---- ggml's K-quants (super-blocks of 256; the gguf_q5_k_m directory of the reference, src/tts/engine.rs:91-95, holds Q5_K and Q6_K
// tensors). The layouts are llama.cpp's (ggml-quants.c dequantize_row_q4_K / q5_K / q6_K), restated from their published definition:
// llama.cpp is not in /root/reference, so this row is PARITY UNPINNED; tests/_gguf.py holds the independent numpy restatement. ----
// 6-bit (scale, min) pair j of a Q4_K / Q5_K super-block (get_scale_min_k4)
inline void scale_min_k4(int j, const uint8_t* q, uint8_t* sc, uint8_t* m) {
    if (j < 4) { *sc = q[j] & 63; *m = q[j + 4] & 63; }
    else { *sc = (uint8_t)((q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4)); *m = (uint8_t)((q[j + 4] >> 4) | ((q[j] >> 6) << 4)); }
}
// one super-block -> 256 floats
void deq_q4_k(const uint8_t* b, float* y) {   // { f16 d, dmin; u8 scales[12]; u8 qs[128] }
    uint16_t h; memcpy(&h, b, 2); const float d = f16_to_f32(h); memcpy(&h, b + 2, 2); const float dmin = f16_to_f32(h);
    const uint8_t* scales = b + 4; const uint8_t* q = b + 16;
    int is = 0;
    for (int j = 0; j < 256; j += 64) {
        uint8_t sc, m;
        scale_min_k4(is + 0, scales, &sc, &m); const float d1 = d * (float)sc, m1 = dmin * (float)m;
        scale_min_k4(is + 1, scales, &sc, &m); const float d2 = d * (float)sc, m2 = dmin * (float)m;
        for (int l = 0; l < 32; ++l) *y++ = d1 * (float)(q[l] & 0xF) - m1;
        for (int l = 0; l < 32; ++l) *y++ = d2 * (float)(q[l] >> 4) - m2;
        q += 32; is += 2;
    }
}
void deq_q5_k(const uint8_t* b, float* y) {   // { f16 d, dmin; u8 scales[12]; u8 qh[32]; u8 qs[128] }
    uint16_t h; memcpy(&h, b, 2); const float d = f16_to_f32(h); memcpy(&h, b + 2, 2); const float dmin = f16_to_f32(h);
    const uint8_t* scales = b + 4; const uint8_t* qh = b + 16; const uint8_t* ql = b + 48;
    int is = 0; uint8_t u1 = 1, u2 = 2;
    for (int j = 0; j < 256; j += 64) {
        uint8_t sc, m;
        scale_min_k4(is + 0, scales, &sc, &m); const float d1 = d * (float)sc, m1 = dmin * (float)m;
        scale_min_k4(is + 1, scales, &sc, &m); const float d2 = d * (float)sc, m2 = dmin * (float)m;
        for (int l = 0; l < 32; ++l) *y++ = d1 * (float)((ql[l] & 0xF) + ((qh[l] & u1) ? 16 : 0)) - m1;
        for (int l = 0; l < 32; ++l) *y++ = d2 * (float)((ql[l] >> 4) + ((qh[l] & u2) ? 16 : 0)) - m2;
        ql += 32; is += 2; u1 = (uint8_t)(u1 << 2); u2 = (uint8_t)(u2 << 2);
    }
}
void deq_q6_k(const uint8_t* b, float* y) {   // { u8 ql[128]; u8 qh[64]; i8 scales[16]; f16 d }
    const uint8_t* ql = b; const uint8_t* qh = b + 128; const int8_t* sc = (const int8_t*)(b + 192);
    uint16_t h; memcpy(&h, b + 208, 2); const float d = f16_to_f32(h);
    for (int n = 0; n < 256; n += 128) {
        for (int l = 0; l < 32; ++l) {
            const int is = l / 16;
            const int q1 = (int)(int8_t)((ql[l + 0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
            const int q2 = (int)(int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
            const int q3 = (int)(int8_t)((ql[l + 0] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
            const int q4 = (int)(int8_t)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
            y[l + 0] = d * (float)sc[is + 0] * (float)q1;
            y[l + 32] = d * (float)sc[is + 2] * (float)q2;
            y[l + 64] = d * (float)sc[is + 4] * (float)q3;
            y[l + 96] = d * (float)sc[is + 6] * (float)q4;
        }
        y += 128; ql += 64; qh += 32; sc += 8;
    }
}
// K-quant super-block kb of tensor t -> 256 floats; false for other types
bool deq_k_block(const Q3GgufTensor& t, size_t kb, float* y) {
    switch (t.type) {
        case Q3_GGML_Q4_K: deq_q4_k(t.data + kb * 144, y); return true;
        case Q3_GGML_Q5_K: deq_q5_k(t.data + kb * 176, y); return true;
        case Q3_GGML_Q6_K: deq_q6_k(t.data + kb * 210, y); return true;
        default: return false;
    }
}

}  // namespace

Q3Gguf::~Q3Gguf() { if (map_) munmap(map_, size_); }

int Q3Gguf::open(const std::string& path, std::string& err) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) { err = "cannot open " + path; return -1; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 24) { ::close(fd); err = path + ": too small for a GGUF header"; return -1; }
    size_ = (size_t)st.st_size;
    map_ = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (map_ == MAP_FAILED) { map_ = nullptr; err = "mmap failed for " + path; return -1; }
    Cursor c{(const uint8_t*)map_, size_, 0};
    char magic[4];
    uint64_t n_tensors = 0, n_kv = 0;
    if (!c.take(magic, 4) || memcmp(magic, "GGUF", 4) != 0) { err = path + ": not a GGUF file"; return -1; }
    if (!c.get(&version_) || version_ < 2 || version_ > 3) {  // v1 has 32-bit counts; the reference refuses it too (:42-45)
        err = path + ": unsupported GGUF version " + std::to_string(version_); return -1;
    }
    if (!c.get(&n_tensors) || !c.get(&n_kv) || n_tensors > (1u << 24) || n_kv > (1u << 24)) { err = path + ": corrupt GGUF header"; return -1; }
    for (uint64_t i = 0; i < n_kv; ++i) {
        std::string key; uint32_t vt = 0;
        if (!c.str(&key) || !c.get(&vt)) { err = path + ": truncated metadata"; return -1; }
        if (vt == GV_STR) { std::string s; if (!c.str(&s)) { err = path + ": truncated metadata string"; return -1; } }
        else if (vt == GV_ARR) {  // (the reference's reader gives up on arrays, :93-97; real llama.cpp files have them)
            uint32_t et = 0; uint64_t cnt = 0;
            if (!c.get(&et) || !c.get(&cnt)) { err = path + ": truncated metadata array"; return -1; }
            if (et == GV_STR) { for (uint64_t j = 0; j < cnt; ++j) { std::string s; if (!c.str(&s)) { err = path + ": truncated string array"; return -1; } } }
            else {
                const size_t es = scalar_size(et);
                if (!es || cnt > size_ || !c.skip((size_t)cnt * es)) { err = path + ": bad metadata array '" + key + "'"; return -1; }
            }
        } else {
            const size_t es = scalar_size(vt);
            uint8_t buf[8] = {0};
            if (!es || !c.take(buf, es)) { err = path + ": unknown metadata value type " + std::to_string(vt) + " for '" + key + "'"; return -1; }
            if (vt != GV_F32 && vt != GV_F64) {
                uint64_t v = 0; memcpy(&v, buf, es);
                if (vt == GV_I8) v = (uint64_t)(int64_t)(int8_t)buf[0];
                else if (vt == GV_I16) { int16_t s; memcpy(&s, buf, 2); v = (uint64_t)(int64_t)s; }
                else if (vt == GV_I32) { int32_t s; memcpy(&s, buf, 4); v = (uint64_t)(int64_t)s; }
                meta_[key] = v;
            }
        }
    }
    uint64_t al = 0;
    if (meta_u64("general.alignment", &al) && al >= 1 && al <= (1u << 20) && (al & (al - 1)) == 0) alignment_ = (uint32_t)al;
    tensors_.resize((size_t)n_tensors);
    for (auto& t : tensors_) {
        uint32_t nd = 0;
        if (!c.str(&t.name) || !c.get(&nd) || nd < 1 || nd > 4) { err = path + ": bad tensor info"; return -1; }
        t.dims.resize(nd); t.nelem = 1;
        for (uint32_t d = 0; d < nd; ++d) {
            if (!c.get(&t.dims[d]) || t.dims[d] == 0 || t.dims[d] > ((uint64_t)1 << 40)) { err = path + ": bad tensor shape for '" + t.name + "'"; return -1; }
            t.nelem *= (size_t)t.dims[d];
        }
        if (!c.get(&t.type) || !c.get(&t.offset)) { err = path + ": truncated tensor info"; return -1; }
    }
    const size_t data_start = (c.pos + alignment_ - 1) / alignment_ * alignment_;
    for (size_t i = 0; i < tensors_.size(); ++i) {
        Q3GgufTensor& t = tensors_[i];
        bool ok = false;
        t.nbytes = type_bytes(t.type, t.nelem, t.dims[0], &ok);
        if (ok) {  // unsupported types stay listed (data == nullptr) and fail only if somebody asks for them
            if (t.offset % alignment_ || data_start > size_ || t.offset > size_ - data_start || t.nbytes > size_ - data_start - t.offset) {
                err = path + ": tensor '" + t.name + "' lies outside the file"; return -1;
            }
            t.data = (const uint8_t*)map_ + data_start + t.offset;
        }
        index_[t.name] = i;
    }
    return 0;
}

const Q3GgufTensor* Q3Gguf::find(const std::string& name) const {
    auto it = index_.find(name);
    return it == index_.end() ? nullptr : &tensors_[it->second];
}
bool Q3Gguf::meta_u64(const std::string& key, uint64_t* v) const {
    auto it = meta_.find(key);
    if (it == meta_.end()) return false;
    *v = it->second;
    return true;
}

int q3_gguf_to_f32(const Q3GgufTensor& t, float* dst, std::string& err) {
    if (!t.data) { err = "tensor '" + t.name + "': unsupported ggml type " + std::to_string(t.type) + " (supported: F32, F16, BF16, Q8_0, Q4_K, Q5_K, Q6_K)"; return -1; }
    const size_t n = t.nelem;
    if (t.type == Q3_GGML_Q4_K || t.type == Q3_GGML_Q5_K || t.type == Q3_GGML_Q6_K) { for (size_t kb = 0; kb < n / 256; ++kb) deq_k_block(t, kb, dst + kb * 256); return 0; }
    if (t.type == Q3_GGML_F32) memcpy(dst, t.data, n * 4);
    else if (t.type == Q3_GGML_F16) { for (size_t i = 0; i < n; ++i) { uint16_t h; memcpy(&h, t.data + 2 * i, 2); dst[i] = f16_to_f32(h); } }
    else if (t.type == Q3_GGML_BF16) { for (size_t i = 0; i < n; ++i) { uint16_t h; memcpy(&h, t.data + 2 * i, 2); const uint32_t u = (uint32_t)h << 16; memcpy(&dst[i], &u, 4); } }
    else {  // Q8_0: blocks of { f16 d; int8 q[32] }, y = q * f32(d)
        for (size_t b = 0; b < n / 32; ++b) {
            const uint8_t* blk = t.data + b * 34;
            uint16_t h; memcpy(&h, blk, 2);
            const float d = f16_to_f32(h);
            for (int j = 0; j < 32; ++j) dst[b * 32 + j] = (float)(int8_t)blk[2 + j] * d;
        }
    }
    return 0;
}

int q3_gguf_to_bf16(const Q3GgufTensor& t, uint16_t* dst, std::string& err) {
    if (!t.data) { err = "tensor '" + t.name + "': unsupported ggml type " + std::to_string(t.type) + " (supported: F32, F16, BF16, Q8_0, Q4_K, Q5_K, Q6_K)"; return -1; }
    if (t.type == Q3_GGML_BF16) { memcpy(dst, t.data, t.nelem * 2); return 0; }
    if (t.type == Q3_GGML_Q4_K || t.type == Q3_GGML_Q5_K || t.type == Q3_GGML_Q6_K) {
        float y[256];
        for (size_t kb = 0; kb < t.nelem / 256; ++kb) { deq_k_block(t, kb, y); for (int j = 0; j < 256; ++j) dst[kb * 256 + j] = f32_to_bf16_rne(y[j]); }
        return 0;
    }
    const size_t chunk = 1 << 16;
    std::vector<float> tmp(chunk);
    if (t.type == Q3_GGML_F32) {
        for (size_t i = 0; i < t.nelem; ++i) { float f; memcpy(&f, t.data + 4 * i, 4); dst[i] = f32_to_bf16_rne(f); }
    } else if (t.type == Q3_GGML_F16) {
        for (size_t i = 0; i < t.nelem; ++i) { uint16_t h; memcpy(&h, t.data + 2 * i, 2); dst[i] = f32_to_bf16_rne(f16_to_f32(h)); }
    } else {
        for (size_t b = 0; b < t.nelem / 32; ++b) {
            const uint8_t* blk = t.data + b * 34;
            uint16_t h; memcpy(&h, blk, 2);
            const float d = f16_to_f32(h);
            for (int j = 0; j < 32; ++j) dst[b * 32 + j] = f32_to_bf16_rne((float)(int8_t)blk[2 + j] * d);
        }
    }
    return 0;
}

int q3_npy_load_f32(const std::string& path, std::vector<float>& out, std::vector<size_t>& shape, std::string& err) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { err = "cannot open " + path; return -1; }
    unsigned char magic[10];
    if (fread(magic, 1, 10, f) != 10 || memcmp(magic, "\x93NUMPY", 6) != 0) { fclose(f); err = path + ": not a numpy file"; return -1; }
    size_t hlen = 0, hoff = 0;
    if (magic[6] == 1) { hlen = (size_t)magic[8] | ((size_t)magic[9] << 8); hoff = 10; }
    else if (magic[6] == 2) {
        unsigned char more[2];
        if (fread(more, 1, 2, f) != 2) { fclose(f); err = path + ": truncated header"; return -1; }
        hlen = (size_t)magic[8] | ((size_t)magic[9] << 8) | ((size_t)more[0] << 16) | ((size_t)more[1] << 24); hoff = 12;
    } else { fclose(f); err = path + ": unsupported numpy version"; return -1; }
    if (hlen > (1u << 20)) { fclose(f); err = path + ": header too long"; return -1; }
    std::string header(hlen, '\0');
    if (fread(&header[0], 1, hlen, f) != hlen) { fclose(f); err = path + ": truncated header"; return -1; }
    (void)hoff;
    if (header.find("'<f4'") == std::string::npos && header.find("\"<f4\"") == std::string::npos) { fclose(f); err = path + ": dtype is not little-endian f32"; return -1; }
    if (header.find("'fortran_order': True") != std::string::npos) { fclose(f); err = path + ": fortran_order arrays are not supported"; return -1; }
    shape.clear();
    const size_t sp = header.find("shape");
    const size_t lp = sp == std::string::npos ? std::string::npos : header.find('(', sp), rp = lp == std::string::npos ? std::string::npos : header.find(')', lp);
    if (rp == std::string::npos) { fclose(f); err = path + ": no shape in header"; return -1; }
    size_t v = 0; bool have = false;
    for (size_t i = lp + 1; i <= rp; ++i) {
        const char ch = header[i];
        if (ch >= '0' && ch <= '9') { v = v * 10 + (size_t)(ch - '0'); have = true; }
        else { if (have) shape.push_back(v); v = 0; have = false; }
    }
    size_t n = 1;
    for (size_t d : shape) n *= d;
    out.resize(n);
    if (n && fread(out.data(), 4, n, f) != n) { fclose(f); err = path + ": truncated data"; return -1; }
    fclose(f);
    return 0;
}

// ---- C ABI test hook (host only) -------------------------------------------------------------------------------
#include "../../include/q3tts.h"
struct q3tts_engine;
int q3_set_err(q3tts_engine* e, int code, const std::string& msg);  // q3_engine.hip (engine == nullptr: the global error slot)

extern "C" int q3tts_k_gguf_read(const char* path, const char* tensor, float* out, int64_t cap, int64_t* nelem, int64_t* dims4, int32_t* ggml_type) {
    if (!path || !tensor || !nelem) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "null argument");
    std::string err;
    const std::string p = path;
    if (p.size() > 4 && p.compare(p.size() - 4, 4, ".npy") == 0) {
        std::vector<float> v; std::vector<size_t> shape;
        if (q3_npy_load_f32(p, v, shape, err)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, err);
        *nelem = (int64_t)v.size();
        if (dims4) for (int i = 0; i < 4; ++i) dims4[i] = i < (int)shape.size() ? (int64_t)shape[i] : 0;
        if (ggml_type) *ggml_type = Q3_GGML_F32;
        if (out) { if (cap < *nelem) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "output buffer too small"); memcpy(out, v.data(), v.size() * 4); }
        return Q3TTS_OK;
    }
    Q3Gguf g;
    if (g.open(p, err)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, err);
    const Q3GgufTensor* t = g.find(tensor);
    if (!t) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, std::string("tensor '") + tensor + "' is missing");
    *nelem = (int64_t)t->nelem;
    if (dims4) for (int i = 0; i < 4; ++i) dims4[i] = i < (int)t->dims.size() ? (int64_t)t->dims[i] : 0;
    if (ggml_type) *ggml_type = (int32_t)t->type;
    if (out) {
        if (cap < *nelem) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "output buffer too small");
        if (q3_gguf_to_f32(*t, out, err)) return q3_set_err(nullptr, Q3TTS_ERR_UNSUPPORTED, err);
    }
    return Q3TTS_OK;
}
