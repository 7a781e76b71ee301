// q3_tokenizer.cpp — host-side byte-level BPE reader for tokenizer.json.
//
// Replaces, for hosts that are not the Rust crate, /root/reference/src/utils/tokenizer.rs:1-37: a thin wrapper over the
// `tokenizers` crate ("0.22", Cargo.toml:20) — Tokenizer::from_file(model_dir/tokenizer/tokenizer.json) and
// encode(text, add_special_tokens = false).get_ids() / decode(ids, skip_special_tokens = false). The arithmetic lives in
// that third-party crate; this file restates the pipeline of the Qwen2-family tokenizer.json it is used with:
//   added tokens (leftmost-longest, matched on the raw text) -> NFC normaliser -> Split(the Qwen2 regex, Isolated) ->
//   ByteLevel(add_prefix_space = false, use_regex = false) -> BPE(merges by rank) -> no post-processing.
// Anything else in the file (another normaliser / pre-tokeniser / model type, added tokens with lstrip / rstrip /
// single_word / normalized) is refused at load time with a message, not approximated. NFC is implemented in full (UAX #15:
// generated decomposition / combining-class / composition tables in q3_unicode_tables.h, algorithmic Hangul), behind a quick
// check that lets already-normalised text through untouched.
// Parity is pinned: the Python `tokenizers` package in this image (0.22.2) is the same crate, and tests/test_tokenizer_cpu.py
// compares ids on tokenizers trained in the test (no tokenizer.json of the real model exists offline).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/q3tts.h"
#include "q3_unicode_tables.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// minimal JSON (objects, arrays, strings with \u escapes and surrogate pairs, numbers, true / false / null)
// ---------------------------------------------------------------------------------------------------------------
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false; double num = 0; std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* key) const {
        if (kind != Obj) return nullptr;
        for (auto& kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
};
void put_utf8(std::string& s, uint32_t cp) {
    if (cp < 0x80) s += (char)cp;
    else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
    else { s += (char)(0xF0 | (cp >> 18)); s += (char)(0x80 | ((cp >> 12) & 0x3F)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
}
struct JParser {
    const char* p; const char* end; std::string err; int depth = 0;
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool hex4(uint32_t& v) {
        if (end - p < 4) return fail("truncated \\u escape");
        v = 0;
        for (int i = 0; i < 4; ++i) {
            const char c = *p++; v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0'; else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10; else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else return fail("bad \\u escape");
        }
        return true;
    }
    bool string(std::string& out) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p; out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("truncated escape");
                const char c = *p++;
                switch (c) {
                case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
                case 'b': out += '\b'; break; case 'f': out += '\f'; break; case 'n': out += '\n'; break;
                case 'r': out += '\r'; break; case 't': out += '\t'; break;
                case 'u': {
                    uint32_t v; if (!hex4(v)) return false;
                    if (v >= 0xD800 && v < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                        p += 2; uint32_t lo; if (!hex4(lo)) return false;
                        if (lo >= 0xDC00 && lo < 0xE000) v = 0x10000 + ((v - 0xD800) << 10) + (lo - 0xDC00);
                        else { put_utf8(out, 0xFFFD); v = lo; }
                    }
                    put_utf8(out, v); break;
                }
                default: return fail("bad escape");
                }
            } else out += *p++;
        }
        if (p >= end) return fail("unterminated string");
        ++p; return true;
    }
    bool value(JVal& v) {
        if (++depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        bool ok = true;
        if (*p == '{') {
            v.kind = JVal::Obj; ++p; ws();
            if (p < end && *p == '}') ++p;
            else for (;;) {
                ws(); std::string k; if (!string(k)) { ok = false; break; }
                ws(); if (p >= end || *p != ':') { ok = fail("expected ':'"); break; } ++p;
                v.obj.emplace_back(std::move(k), JVal());
                if (!value(v.obj.back().second)) { ok = false; break; }
                ws(); if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; break; }
                ok = fail("expected ',' or '}'"); break;
            }
        } else if (*p == '[') {
            v.kind = JVal::Arr; ++p; ws();
            if (p < end && *p == ']') ++p;
            else for (;;) {
                v.arr.emplace_back();
                if (!value(v.arr.back())) { ok = false; break; }
                ws(); if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; break; }
                ok = fail("expected ',' or ']'"); break;
            }
        } else if (*p == '"') { v.kind = JVal::Str; ok = string(v.str); }
        else if (end - p >= 4 && !strncmp(p, "true", 4)) { v.kind = JVal::Bool; v.b = true; p += 4; }
        else if (end - p >= 5 && !strncmp(p, "false", 5)) { v.kind = JVal::Bool; v.b = false; p += 5; }
        else if (end - p >= 4 && !strncmp(p, "null", 4)) { v.kind = JVal::Null; p += 4; }
        else {
            const char* s = p;
            while (p < end && (strchr("+-0123456789.eE", *p) != nullptr)) ++p;
            if (p == s) ok = fail("unexpected character");
            else { v.kind = JVal::Num; v.num = atof(std::string(s, p).c_str()); }
        }
        --depth;
        return ok;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Unicode helpers
// ---------------------------------------------------------------------------------------------------------------
bool in_ranges(const Q3URange* r, int n, uint32_t cp) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (cp < r[mid].lo) hi = mid - 1; else if (cp > r[mid].hi) lo = mid + 1; else return true;
    }
    return false;
}
inline bool is_L(uint32_t c) { return in_ranges(Q3U_LETTER, Q3U_LETTER_N, c); }
inline bool is_N(uint32_t c) { return in_ranges(Q3U_NUMBER, Q3U_NUMBER_N, c); }
inline bool is_S(uint32_t c) {  // \s of the regex engine = Unicode White_Space
    return (c >= 0x9 && c <= 0xD) || c == 0x20 || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) || c == 0x2028 ||
           c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}
inline bool is_nl(uint32_t c) { return c == '\r' || c == '\n'; }

// UTF-8 -> code points with byte offsets (invalid bytes are refused by the caller)
bool decode_utf8(const char* s, size_t n, std::vector<uint32_t>& cps, std::vector<uint32_t>& offs) {
    size_t i = 0;
    while (i < n) {
        const unsigned char c = (unsigned char)s[i];
        uint32_t cp; int len;
        if (c < 0x80) { cp = c; len = 1; }
        else if ((c >> 5) == 6) { cp = c & 0x1F; len = 2; }
        else if ((c >> 4) == 14) { cp = c & 0x0F; len = 3; }
        else if ((c >> 3) == 30) { cp = c & 0x07; len = 4; }
        else return false;
        if (i + len > n) return false;
        for (int k = 1; k < len; ++k) { const unsigned char d = (unsigned char)s[i + k]; if ((d >> 6) != 2) return false; cp = (cp << 6) | (d & 0x3F); }
        if ((len == 2 && cp < 0x80) || (len == 3 && cp < 0x800) || (len == 4 && cp < 0x10000) || cp > 0x10FFFF || (cp >= 0xD800 && cp < 0xE000)) return false;
        cps.push_back(cp); offs.push_back((uint32_t)i);
        i += len;
    }
    offs.push_back((uint32_t)n);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// NFC (UAX #15): canonical decomposition (generated table + algorithmic Hangul), canonical ordering, canonical composition
// ---------------------------------------------------------------------------------------------------------------
uint8_t ccc_of(uint32_t cp) {
    int lo = 0, hi = Q3U_CCC_N - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (cp < Q3U_CCC[mid].cp) hi = mid - 1; else if (cp > Q3U_CCC[mid].cp) lo = mid + 1; else return Q3U_CCC[mid].cls;
    }
    return 0;
}
void decompose(uint32_t cp, std::vector<uint32_t>& out) {
    if (cp >= 0xAC00 && cp <= 0xD7A3) {  // Hangul syllable -> L V (T)
        const uint32_t si = cp - 0xAC00;
        out.push_back(0x1100 + si / 588); out.push_back(0x1161 + (si % 588) / 28);
        if (si % 28) out.push_back(0x11A7 + si % 28);
        return;
    }
    int lo = 0, hi = Q3U_DECOMP_N - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (cp < Q3U_DECOMP[mid].cp) hi = mid - 1;
        else if (cp > Q3U_DECOMP[mid].cp) lo = mid + 1;
        else { for (int i = 0; i < Q3U_DECOMP[mid].len; ++i) out.push_back(Q3U_DECOMP_POOL[Q3U_DECOMP[mid].off + i]); return; }
    }
    out.push_back(cp);
}
uint32_t compose_pair(uint32_t a, uint32_t b) {  // 0: no primary composite
    if (a >= 0x1100 && a <= 0x1112 && b >= 0x1161 && b <= 0x1175) return 0xAC00 + ((a - 0x1100) * 21 + (b - 0x1161)) * 28;
    if (a >= 0xAC00 && a <= 0xD7A3 && (a - 0xAC00) % 28 == 0 && b >= 0x11A8 && b <= 0x11C2) return a + (b - 0x11A7);
    int lo = 0, hi = Q3U_COMP_N - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const Q3UComp& c = Q3U_COMP[mid];
        if (a < c.a || (a == c.a && b < c.b)) hi = mid - 1; else if (a > c.a || (a == c.a && b > c.b)) lo = mid + 1; else return c.c;
    }
    return 0;
}
void nfc(std::vector<uint32_t>& cps) {
    bool clean = true;
    for (uint32_t c : cps) if (in_ranges(Q3U_NFC_UNSAFE, Q3U_NFC_UNSAFE_N, c)) { clean = false; break; }
    if (clean) return;  // quick check: nothing that could decompose, reorder or compose
    std::vector<uint32_t> d;
    d.reserve(cps.size() + 8);
    for (uint32_t c : cps) decompose(c, d);
    for (size_t i = 1; i < d.size(); ++i) {  // canonical ordering: stable insertion sort of the non-starters by combining class
        const uint8_t ci = ccc_of(d[i]);
        if (!ci) continue;
        size_t j = i;
        while (j > 0) { const uint8_t cj = ccc_of(d[j - 1]); if (cj == 0 || cj <= ci) break; std::swap(d[j], d[j - 1]); --j; }
    }
    if (d.empty()) { cps.clear(); return; }
    size_t starter_pos = 0, comp_pos = 1;
    uint32_t starter = d[0];
    int last_class = ccc_of(starter);
    if (last_class != 0) last_class = 256;  // a leading non-starter never composes
    for (size_t i = 1; i < d.size(); ++i) {
        const uint32_t ch = d[i];
        const int cls = ccc_of(ch);
        const uint32_t composite = compose_pair(starter, ch);
        if (composite && (last_class < cls || last_class == 0)) { d[starter_pos] = composite; starter = composite; }
        else {
            if (cls == 0) { starter_pos = comp_pos; starter = ch; }
            last_class = cls;
            d[comp_pos++] = ch;
        }
    }
    d.resize(comp_pos);
    cps.swap(d);
}

const char* QWEN2_SPLIT =
    "(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\\r\\n\\p{L}\\p{N}]?\\p{L}+|\\p{N}| ?[^\\s\\p{L}\\p{N}]+[\\r\\n]*|\\s*[\\r\\n]+|\\s+(?!\\S)|\\s+";

inline uint32_t fold(uint32_t c) { return (c >= 'A' && c <= 'Z') ? c + 32 : (c == 0x17F ? 's' : c); }  // (?i:) over the seven suffixes
// length (in code points) of the match of QWEN2_SPLIT at position i: ordered alternation, greedy, with the backtracking the
// pattern needs spelled out per alternative
size_t split_match(const std::vector<uint32_t>& c, size_t i) {
    const size_t n = c.size();
    auto at = [&](size_t k) -> uint32_t { return k < n ? c[k] : 0xFFFFFFFFu; };
    // 1: contractions
    if (c[i] == '\'' && i + 1 < n) {
        const uint32_t a = fold(c[i + 1]), b = i + 2 < n ? fold(c[i + 2]) : 0;
        if (a == 's' || a == 't') return 2;
        if (a == 'r' && b == 'e') return 3;
        if (a == 'v' && b == 'e') return 3;
        if (a == 'm') return 2;
        if (a == 'l' && b == 'l') return 3;
        if (a == 'd') return 2;
    }
    // 2: [^\r\n\p{L}\p{N}]?\p{L}+
    {
        size_t j = i;
        if (!is_nl(c[i]) && !is_L(c[i]) && !is_N(c[i]) && i + 1 < n && is_L(c[i + 1])) j = i + 1;
        if (is_L(at(j))) { while (j < n && is_L(c[j])) ++j; return j - i; }
    }
    // 3: \p{N}
    if (is_N(c[i])) return 1;
    // 4:  ?[^\s\p{L}\p{N}]+[\r\n]*
    {
        auto other = [&](uint32_t x) { return x != 0xFFFFFFFFu && !is_S(x) && !is_L(x) && !is_N(x); };
        size_t j = i;
        if (c[i] == ' ' && other(at(i + 1))) j = i + 1;
        if (other(at(j))) {
            while (j < n && other(c[j])) ++j;
            while (j < n && is_nl(c[j])) ++j;
            return j - i;
        }
    }
    if (is_S(c[i])) {
        size_t e = i;
        while (e < n && is_S(c[e])) ++e;
        // 5: \s*[\r\n]+ — up to the last newline of the whitespace run
        for (size_t p = e; p > i; --p) if (is_nl(c[p - 1])) return p - i;
        // 6: \s+(?!\S) — the run if it ends the text, else all but its last character
        if (e == n) return e - i;
        if (e - i >= 2) return e - i - 1;
        // 7: \s+
        return e - i;
    }
    return 1;  // unreachable: every code point is a letter, a number, white space or "other"
}

}  // namespace

struct q3tts_tokenizer {
    std::unordered_map<std::string, uint32_t> vocab;
    std::vector<std::string> id_to_token;
    std::unordered_map<uint64_t, std::pair<uint32_t, uint32_t>> merges;  // (left id, right id) -> (rank, merged id)
    struct Added { std::string content; uint32_t id; };
    std::vector<Added> added;                     // longest first
    std::string byte_to_tok[256];                 // GPT-2 byte -> printable code point (as UTF-8)
    uint32_t byte_id[256];                        // vocab id of each single-byte token
    std::unordered_map<uint32_t, uint8_t> cp_to_byte;
    mutable std::unordered_map<std::string, std::vector<uint32_t>> cache;
    bool nfc = false;                             // "normalizer": {"type": "NFC"}
};

namespace {

int set_err(char* err, int cap, int code, const std::string& m) {
    if (err && cap > 0) snprintf(err, (size_t)cap, "%s", m.c_str());
    return code;
}
void bpe_word(const q3tts_tokenizer& t, const std::string& piece, std::vector<uint32_t>& out) {
    auto it = t.cache.find(piece);
    if (it != t.cache.end()) { out.insert(out.end(), it->second.begin(), it->second.end()); return; }
    std::vector<uint32_t> sym;
    sym.reserve(piece.size());
    for (unsigned char b : piece) sym.push_back(t.byte_id[b]);
    for (;;) {  // lowest rank first, leftmost among equals
        uint32_t best_rank = 0xFFFFFFFFu, best_id = 0; size_t best_pos = 0;
        for (size_t k = 0; k + 1 < sym.size(); ++k) {
            auto m = t.merges.find(((uint64_t)sym[k] << 32) | sym[k + 1]);
            if (m != t.merges.end() && m->second.first < best_rank) { best_rank = m->second.first; best_id = m->second.second; best_pos = k; }
        }
        if (best_rank == 0xFFFFFFFFu) break;
        sym[best_pos] = best_id;
        sym.erase(sym.begin() + (long)best_pos + 1);
    }
    if (t.cache.size() < 100000) t.cache.emplace(piece, sym);
    out.insert(out.end(), sym.begin(), sym.end());
}

}  // namespace

extern "C" int q3tts_tokenizer_load(const char* path, q3tts_tokenizer** out, char* err, int32_t err_cap) {
    if (!path || !out) return set_err(err, err_cap, Q3TTS_ERR_INVALID, "null argument");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return set_err(err, err_cap, Q3TTS_ERR_IO, std::string("Failed to load tokenizer: cannot open ") + path);
    std::string text;
    char buf[1 << 16]; size_t got;
    while ((got = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, got);
    fclose(f);
    JParser P{text.data(), text.data() + text.size(), "", 0};
    JVal root;
    if (!P.value(root) || root.kind != JVal::Obj) return set_err(err, err_cap, Q3TTS_ERR_IO, "Failed to load tokenizer: JSON: " + (P.err.empty() ? std::string("not an object") : P.err));
    auto unsupported = [&](const std::string& m) { return set_err(err, err_cap, Q3TTS_ERR_UNSUPPORTED, "tokenizer.json: " + m); };
    const JVal* model = root.get("model");
    if (!model || !model->get("type") || model->get("type")->str != "BPE") return unsupported("model.type must be BPE");
    for (const char* k : {"dropout", "unk_token", "continuing_subword_prefix", "end_of_word_suffix"})
        if (model->get(k) && model->get(k)->kind != JVal::Null && !(model->get(k)->kind == JVal::Str && model->get(k)->str.empty())) return unsupported(std::string("model.") + k + " is set");
    for (const char* k : {"byte_fallback", "ignore_merges", "fuse_unk"})
        if (model->get(k) && model->get(k)->kind == JVal::Bool && model->get(k)->b) return unsupported(std::string("model.") + k + " = true");
    const JVal* norm = root.get("normalizer");
    if (norm && norm->kind != JVal::Null && !(norm->get("type") && norm->get("type")->str == "NFC")) return unsupported("normalizer must be NFC or null");
    const JVal* pre = root.get("pre_tokenizer");
    bool pre_ok = false;
    if (pre && pre->get("type") && pre->get("type")->str == "Sequence" && pre->get("pretokenizers") && pre->get("pretokenizers")->arr.size() == 2) {
        const JVal& a = pre->get("pretokenizers")->arr[0]; const JVal& b = pre->get("pretokenizers")->arr[1];
        const JVal* pat = a.get("pattern");
        pre_ok = a.get("type") && a.get("type")->str == "Split" && pat && pat->get("Regex") && pat->get("Regex")->str == QWEN2_SPLIT &&
                 a.get("behavior") && a.get("behavior")->str == "Isolated" && !(a.get("invert") && a.get("invert")->b) &&
                 b.get("type") && b.get("type")->str == "ByteLevel" && !(b.get("add_prefix_space") && b.get("add_prefix_space")->b) &&
                 !(b.get("use_regex") && b.get("use_regex")->b);
    }
    if (!pre_ok) return unsupported("pre_tokenizer must be Sequence[Split(the Qwen2 pattern, Isolated), ByteLevel(add_prefix_space=false, use_regex=false)]");
    const JVal* post = root.get("post_processor");
    if (post && post->kind != JVal::Null && !(post->get("type") && post->get("type")->str == "ByteLevel")) return unsupported("post_processor must be ByteLevel or null");

    q3tts_tokenizer* t = new q3tts_tokenizer();
    t->nfc = norm && norm->kind != JVal::Null;
    auto bail = [&](int code, const std::string& m) { delete t; return set_err(err, err_cap, code, m); };
    const JVal* vocab = model->get("vocab");
    if (!vocab || vocab->kind != JVal::Obj) return bail(Q3TTS_ERR_IO, "tokenizer.json: model.vocab missing");
    for (auto& kv : vocab->obj) {
        if (kv.second.kind != JVal::Num || kv.second.num < 0) return bail(Q3TTS_ERR_IO, "tokenizer.json: bad vocab id for '" + kv.first + "'");
        const uint32_t id = (uint32_t)kv.second.num;
        t->vocab[kv.first] = id;
        if (t->id_to_token.size() <= id) t->id_to_token.resize((size_t)id + 1);
        t->id_to_token[id] = kv.first;
    }
    // GPT-2 byte <-> printable code point table (bytes_to_unicode)
    {
        int extra = 0;
        for (int b = 0; b < 256; ++b) {
            const bool printable = (b >= 0x21 && b <= 0x7E) || (b >= 0xA1 && b <= 0xAC) || (b >= 0xAE && b <= 0xFF);
            const uint32_t cp = printable ? (uint32_t)b : 256u + (uint32_t)extra++;
            put_utf8(t->byte_to_tok[b], cp);
            t->cp_to_byte[cp] = (uint8_t)b;
            auto it = t->vocab.find(t->byte_to_tok[b]);
            if (it == t->vocab.end()) return bail(Q3TTS_ERR_UNSUPPORTED, "tokenizer.json: the vocabulary lacks a single-byte token (byte-level alphabet incomplete)");
            t->byte_id[b] = it->second;
        }
    }
    const JVal* merges = model->get("merges");
    if (!merges || merges->kind != JVal::Arr) return bail(Q3TTS_ERR_IO, "tokenizer.json: model.merges missing");
    uint32_t rank = 0;
    for (auto& m : merges->arr) {
        std::string a, b;
        if (m.kind == JVal::Arr && m.arr.size() == 2 && m.arr[0].kind == JVal::Str && m.arr[1].kind == JVal::Str) { a = m.arr[0].str; b = m.arr[1].str; }
        else if (m.kind == JVal::Str) {
            const size_t sp = m.str.find(' ');
            if (sp == std::string::npos) return bail(Q3TTS_ERR_IO, "tokenizer.json: bad merge '" + m.str + "'");
            a = m.str.substr(0, sp); b = m.str.substr(sp + 1);
        } else return bail(Q3TTS_ERR_IO, "tokenizer.json: bad merges entry");
        auto ia = t->vocab.find(a), ib = t->vocab.find(b), iab = t->vocab.find(a + b);
        if (ia == t->vocab.end() || ib == t->vocab.end() || iab == t->vocab.end()) return bail(Q3TTS_ERR_IO, "tokenizer.json: merge '" + a + " " + b + "' refers to tokens outside the vocabulary");
        t->merges.emplace(((uint64_t)ia->second << 32) | ib->second, std::make_pair(rank, iab->second));
        ++rank;
    }
    if (const JVal* added = root.get("added_tokens")) {
        for (auto& a : added->arr) {
            const JVal* c = a.get("content"); const JVal* id = a.get("id");
            if (!c || c->kind != JVal::Str || !id || id->kind != JVal::Num || c->str.empty()) return bail(Q3TTS_ERR_IO, "tokenizer.json: bad added_tokens entry");
            for (const char* k : {"single_word", "lstrip", "rstrip", "normalized"})
                if (a.get(k) && a.get(k)->kind == JVal::Bool && a.get(k)->b) return bail(Q3TTS_ERR_UNSUPPORTED, "tokenizer.json: added token '" + c->str + "' sets " + k);
            t->added.push_back({c->str, (uint32_t)id->num});
            if (t->id_to_token.size() <= (size_t)id->num) t->id_to_token.resize((size_t)id->num + 1);
            t->id_to_token[(size_t)id->num] = c->str;
        }
        std::stable_sort(t->added.begin(), t->added.end(), [](const q3tts_tokenizer::Added& x, const q3tts_tokenizer::Added& y) { return x.content.size() > y.content.size(); });
    }
    *out = t;
    return Q3TTS_OK;
}

extern "C" void q3tts_tokenizer_free(q3tts_tokenizer* t) { delete t; }

extern "C" int32_t q3tts_tokenizer_vocab_size(const q3tts_tokenizer* t) { return t ? (int32_t)t->id_to_token.size() : 0; }

extern "C" int q3tts_tokenizer_encode(const q3tts_tokenizer* t, const char* utf8, int64_t n_bytes, uint32_t* ids, int32_t cap, int32_t* n_ids,
                                      char* err, int32_t err_cap) {
    if (!t || !n_ids || n_bytes < 0 || (n_bytes > 0 && !utf8)) return set_err(err, err_cap, Q3TTS_ERR_INVALID, "null argument");
    std::vector<uint32_t> out;
    const std::string text(utf8 ? utf8 : "", (size_t)n_bytes);
    // plain segments between added tokens
    auto plain = [&](size_t b, size_t e) -> int {
        if (b >= e) return Q3TTS_OK;
        std::vector<uint32_t> cps, offs;
        if (!decode_utf8(text.data() + b, e - b, cps, offs)) return set_err(err, err_cap, Q3TTS_ERR_INVALID, "input is not valid UTF-8");
        if (t->nfc) nfc(cps);
        size_t i = 0;
        std::string piece;
        while (i < cps.size()) {
            const size_t len = split_match(cps, i);
            piece.clear();
            for (size_t k = i; k < i + len; ++k) put_utf8(piece, cps[k]);
            bpe_word(*t, piece, out);
            i += len;
        }
        return Q3TTS_OK;
    };
    size_t seg = 0, pos = 0;
    while (pos < text.size()) {
        const q3tts_tokenizer::Added* hit = nullptr;
        for (auto& a : t->added)  // sorted longest first: the first hit at this position is the longest
            if (a.content.size() <= text.size() - pos && !memcmp(text.data() + pos, a.content.data(), a.content.size())) { hit = &a; break; }
        if (!hit) { ++pos; continue; }
        int rc = plain(seg, pos);
        if (rc) return rc;
        out.push_back(hit->id);
        pos += hit->content.size(); seg = pos;
    }
    int rc = plain(seg, text.size());
    if (rc) return rc;
    *n_ids = (int32_t)out.size();
    if ((int64_t)out.size() > cap) return set_err(err, err_cap, Q3TTS_ERR_INVALID, "id buffer too small");
    if (!out.empty() && !ids) return set_err(err, err_cap, Q3TTS_ERR_INVALID, "null argument");
    if (!out.empty()) memcpy(ids, out.data(), out.size() * 4);
    return Q3TTS_OK;
}

extern "C" int q3tts_tokenizer_decode(const q3tts_tokenizer* t, const uint32_t* ids, int32_t n, char* out, int64_t cap, int64_t* n_bytes,
                                      char* err, int32_t err_cap) {
    if (!t || !n_bytes || n < 0 || (n > 0 && !ids)) return set_err(err, err_cap, Q3TTS_ERR_INVALID, "null argument");
    std::string bytes;
    for (int32_t i = 0; i < n; ++i) {
        if (ids[i] >= t->id_to_token.size()) continue;  // the crate skips ids it does not know
        const std::string& tok = t->id_to_token[ids[i]];
        bool is_added = false;
        for (auto& a : t->added) if (a.id == ids[i]) { is_added = true; break; }
        if (is_added) { bytes += tok; continue; }
        std::vector<uint32_t> cps, offs;
        if (!decode_utf8(tok.data(), tok.size(), cps, offs)) continue;
        for (uint32_t c : cps) {
            auto it = t->cp_to_byte.find(c);
            if (it != t->cp_to_byte.end()) bytes += (char)it->second; else put_utf8(bytes, c);
        }
    }
    *n_bytes = (int64_t)bytes.size();
    if ((int64_t)bytes.size() > cap) return set_err(err, err_cap, Q3TTS_ERR_INVALID, "output buffer too small");
    if (!bytes.empty() && !out) return set_err(err, err_cap, Q3TTS_ERR_INVALID, "null argument");
    if (!bytes.empty()) memcpy(out, bytes.data(), bytes.size());
    return Q3TTS_OK;
}
