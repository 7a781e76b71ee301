// Where do the microseconds of a decoder block go? Replays a chain of the engine's own kernels (csrc/q3_bgemm.hip, q3_kernels.hip built
// with -DQ3_STAMPS) inside one hipGraph and prints, per launch, constant-rate (100 MHz) timestamps taken INSIDE the kernels by every
// workgroup: when the first / last workgroup started, the phases of the median workgroup, when the last one finished, and the gap between
// the previous kernel's last store and this kernel's first instruction (= the real launch-to-launch cost, which a tracer cannot see).
//   Predictor chain: 3 passes x 5 blocks x [QKV, attention (<= 17 keys), O, gate/up, down] + head, 64 rows, weights hot (157 MB re-read)
//   Talker chain   : 4 blocks x [QKV, attention (T = 150), O, gate/up, down], 64 rows, weights cold (rotating through 2 GiB)
// Build: bash tools/build_stamps.sh chain (here or on the GPU box)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "q3_kernels.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Launch { std::string name; int wgs; };
static const int SLOTS = 1024;  // stamp slots (workgroups) per launch

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 64;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t WBYTES = (size_t)2 << 30;
    char* w; CK(hipMalloc(&w, WBYTES)); CK(hipMemset(w, 0x3c, WBYTES));
    uint16_t *xb, *att, *h; float *x, *qkv, *ssp, *nw, *cs, *sn; uint16_t *kc, *vc; unsigned long long *keys, *dbg;
    CK(hipMalloc(&xb, 256 * 8192 * 2)); CK(hipMemset(xb, 0x3c, 256 * 8192 * 2));
    CK(hipMalloc(&att, 256 * 8192 * 2)); CK(hipMemset(att, 0x3c, 256 * 8192 * 2));
    CK(hipMalloc(&h, 256 * 8192 * 2)); CK(hipMemset(h, 0x3c, 256 * 8192 * 2));
    CK(hipMalloc(&x, 256 * 8192 * 4)); CK(hipMemset(x, 0, 256 * 8192 * 4));
    CK(hipMalloc(&qkv, 256 * 8192 * 4)); CK(hipMemset(qkv, 0x3c, 256 * 8192 * 4));
    CK(hipMalloc(&ssp, 256 * 512 * 4)); CK(hipMemset(ssp, 0x3c, 256 * 512 * 4));
    CK(hipMalloc(&nw, 16384 * 4)); CK(hipMemset(nw, 0x3c, 16384 * 4));
    const int n_ctx_t = 4096;
    CK(hipMalloc(&cs, (size_t)n_ctx_t * 64 * 4)); CK(hipMemset(cs, 0x3c, (size_t)n_ctx_t * 64 * 4));
    CK(hipMalloc(&sn, (size_t)n_ctx_t * 64 * 4)); CK(hipMemset(sn, 0x3c, (size_t)n_ctx_t * 64 * 4));
    const size_t kvn = (size_t)64 * 8 * n_ctx_t * 128;  // one layer, 64 slots
    CK(hipMalloc(&kc, kvn * 2)); CK(hipMemset(kc, 0x3c, kvn * 2)); CK(hipMalloc(&vc, kvn * 2)); CK(hipMemset(vc, 0x3c, kvn * 2));
    CK(hipMalloc(&keys, 256 * 1024 * 8));
    const int MAXL = 128;
    CK(hipMalloc(&dbg, (size_t)MAXL * SLOTS * 8 * 8));
    q3_bgemm_prepare();
    std::vector<Launch> L;
    size_t woff = 0;
    auto wnext = [&](size_t bytes, bool cold) { if (woff + bytes > WBYTES) woff = 0; char* p = w + woff; woff += bytes; return (const uint4*)p; };
    uint16_t* wsc; CK(hipMalloc(&wsc, (size_t)16384 * 256 * 2)); CK(hipMemset(wsc, 0x2c, (size_t)16384 * 256 * 2));  // f16 block scales (any finite value)
    bool q8 = false;
    auto gemm = [&](const char* name, const uint16_t* a, int K, int N, int epi, bool scaled, bool cold, const uint4* wt) -> int {
        Q3BGemm q{}; q.a = a; q.B = rows; q.w = wt; q.K = K; q.N = N; q.w_once = cold ? 1 : 0; q.wscale = q8 ? wsc : nullptr;
        if (scaled) { q.ssp = ssp; q.ld_ssp = K / 16; q.ntiles = K / 16; q.d_norm = K; q.eps = 1e-6f; }
        q.epi = epi; q.y = epi == Q3_EPI_STORE ? qkv : x; q.ldy = N; q.yb = epi == Q3_EPI_SWIGLU ? h : xb; q.keys = keys; q.key_stride = N / 16;
        if (epi == Q3_EPI_RESID) { q.nw_next = nw; q.ssp_out = ssp; q.ld_ssp_out = N / 16; }
        q.dbg = dbg + (size_t)L.size() * SLOTS * 8;
        if (q3_launch_bgemm(q, s)) { printf("launch refused\n"); return 1; }
        L.push_back({name, 0});
        return 0;
    };
    auto attend = [&](const char* name, int nqkv, int n_ctx, int pos_const, bool small) {
        Q3QkPrep qp{}; qp.qkv = qkv; qp.ld = nqkv; qp.rows = rows; qp.Hq = 16; qp.Hkv = 8; qp.hd = 128; qp.qnw = nw; qp.knw = nw; qp.eps = 1e-6f; qp.cs = cs; qp.sn = sn;
        qp.kc = kc; qp.vc = vc; qp.n_ctx = n_ctx; qp.slot_mod = rows; qp.pos_const = pos_const;
        Q3Attend at{}; at.qkv = qkv; at.ld = nqkv; at.rows = rows; at.out = (float*)att; at.ldo = 2048; at.out_bf16 = 1; at.Hq = 16; at.Hkv = 8; at.hd = 128;
        at.kc = kc; at.vc = vc; at.n_ctx = n_ctx; at.slot_mod = rows; at.pos_const = pos_const; at.fused = 1; at.prep = qp;
        at.dbg = dbg + (size_t)L.size() * SLOTS * 8;
        q3_launch_attend(at, s);
        L.push_back({name, 0});
    };
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    // Predictor: 5 distinct blocks (31.5 MB each) + head, revisited by every pass: hot in the Infinity Cache
    std::vector<const uint4*> pw;
    for (int l = 0; l < 5; ++l) { pw.push_back(wnext((size_t)4096 * 1024 * 2, false)); pw.push_back(wnext((size_t)1024 * 2048 * 2, false)); pw.push_back(wnext((size_t)6144 * 1024 * 2, false)); pw.push_back(wnext((size_t)1024 * 3072 * 2, false)); }
    const uint4* phead = wnext((size_t)2048 * 1024 * 2, false);
    for (int pass = 0; pass < 3; ++pass) {
        for (int l = 0; l < 5; ++l) {
            if (gemm("P qkv", xb, 1024, 4096, Q3_EPI_STORE, true, false, pw[l * 4])) return 1;
            attend("P attend_small", 4096, 64, 2 + pass * 6, true);
            if (gemm("P o", att, 2048, 1024, Q3_EPI_RESID, false, false, pw[l * 4 + 1])) return 1;
            if (gemm("P gate/up", xb, 1024, 6144, Q3_EPI_SWIGLU, true, false, pw[l * 4 + 2])) return 1;
            if (gemm("P down", h, 3072, 1024, Q3_EPI_RESID, false, false, pw[l * 4 + 3])) return 1;
        }
        if (gemm("P head", xb, 1024, 2048, Q3_EPI_ARGMAX, true, false, phead)) return 1;
    }
    for (int l = 0; l < 4; ++l) {
        if (gemm("T qkv", xb, 2048, 4096, Q3_EPI_STORE, true, true, wnext((size_t)4096 * 2048 * 2, true))) return 1;
        attend("T attend", 4096, n_ctx_t, 150, false);
        if (gemm("T o", att, 2048, 2048, Q3_EPI_RESID, false, true, wnext((size_t)2048 * 2048 * 2, true))) return 1;
        if (gemm("T gate/up", xb, 2048, 12288, Q3_EPI_SWIGLU, true, true, wnext((size_t)12288 * 2048 * 2, true))) return 1;
        if (gemm("T down", h, 6144, 2048, Q3_EPI_RESID, false, true, wnext((size_t)2048 * 6144 * 2, true))) return 1;
    }
    q8 = true;  // the Talker block on ggml Q8_0 blocks (1.0625 bytes per weight; the weight buffers are read as int8 tiles)
    for (int l = 0; l < 4; ++l) {
        if (gemm("T8 qkv", xb, 2048, 4096, Q3_EPI_STORE, true, true, wnext((size_t)4096 * 2048, true))) return 1;
        attend("T8 attend", 4096, n_ctx_t, 150, false);
        if (gemm("T8 o", att, 2048, 2048, Q3_EPI_RESID, false, true, wnext((size_t)2048 * 2048, true))) return 1;
        if (gemm("T8 gate/up", xb, 2048, 12288, Q3_EPI_SWIGLU, true, true, wnext((size_t)12288 * 2048, true))) return 1;
        if (gemm("T8 down", h, 6144, 2048, Q3_EPI_RESID, false, true, wnext((size_t)2048 * 6144, true))) return 1;
    }
    q8 = false;
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipMemsetAsync(dbg, 0, (size_t)MAXL * SLOTS * 8 * 8, s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st((size_t)L.size() * SLOTS * 8);
    CK(hipMemcpy(st.data(), dbg, st.size() * 8, hipMemcpyDeviceToHost));
    printf("chain of %zu launches at %d rows: %.1f us by events (%.2f us per launch)\n", L.size(), rows, ms * 1e3f, ms * 1e3f / L.size());
    printf("%-16s %5s | %7s %7s | %-44s | %7s %7s | %7s\n", "kernel", "wgs", "gap", "spread", "median workgroup: phase stamps after its start (us)", "last", "period", "");
    unsigned long long prev_end = 0;
    struct Agg { double period = 0, gap = 0, spread = 0, body = 0; int n = 0; double ph[8] = {0}; };
    std::vector<std::pair<std::string, Agg>> aggs;
    for (size_t k = 0; k < L.size(); ++k) {
        const unsigned long long* b = st.data() + k * SLOTS * 8;
        std::vector<unsigned long long> starts; unsigned long long end = 0; int wgs = 0;
        std::vector<std::vector<double>> ph(8);
        for (int wgi = 0; wgi < SLOTS; ++wgi) {
            const unsigned long long* q = b + wgi * 8;
            if (!q[0]) continue;
            ++wgs; starts.push_back(q[0]);
            for (int i = 1; i < 8; ++i) if (q[i]) { end = std::max(end, q[i]); ph[i].push_back((double)(q[i] - q[0]) * 0.01); }
        }
        if (!wgs) { printf("%-16s (no stamps)\n", L[k].name.c_str()); continue; }
        std::sort(starts.begin(), starts.end());
        const double gap = prev_end ? ((double)starts[0] - (double)prev_end) * 0.01 : 0.0, spread = (double)(starts.back() - starts[0]) * 0.01;
        const double last = (double)(end - starts[0]) * 0.01, period = prev_end ? (double)(end - prev_end) * 0.01 : 0.0;
        char buf[256]; int o = 0; double med[8] = {0};
        for (int i = 1; i < 8; ++i) if (!ph[i].empty()) { std::sort(ph[i].begin(), ph[i].end()); med[i] = ph[i][ph[i].size() / 2]; o += snprintf(buf + o, sizeof(buf) - o, "%d:%5.2f ", i, med[i]); }
        buf[o] = 0;
        if (k < 12 || k >= L.size() - 10) printf("%-16s %5d | %7.2f %7.2f | %-44s | %7.2f %7.2f\n", L[k].name.c_str(), wgs, gap, spread, buf, last, period);
        if (prev_end) {
            auto it = std::find_if(aggs.begin(), aggs.end(), [&](const std::pair<std::string, Agg>& a) { return a.first == L[k].name; });
            if (it == aggs.end()) { aggs.push_back({L[k].name, Agg{}}); it = aggs.end() - 1; }
            Agg& a = it->second; a.period += period; a.gap += gap; a.spread += spread; a.body += last; ++a.n;
            for (int i = 1; i < 8; ++i) a.ph[i] += med[i];
        }
        prev_end = end;
    }
    printf("\nmeans per kernel kind (us): period = previous kernel's last stamp -> this kernel's last stamp; gap = previous last stamp -> first workgroup's first instruction;\n"
           "spread = first -> last workgroup start; phases (median workgroup, after its own start): k_bgemm 1 loads issued, 2 first MFMA step done, 3 K loop done, 4 slices in LDS + barrier, 5 epilogue stores issued, 6 stores landed;\n"
           "attention 1 prep done (before barrier), 2 barrier passed, 3 softmax done, 4 stores issued, 5 stores landed\n");
    for (auto& pr : aggs) {
        const Agg& a = pr.second;
        printf("%-16s n=%2d period %6.2f | gap %5.2f spread %5.2f body %6.2f |", pr.first.c_str(), a.n, a.period / a.n, a.gap / a.n, a.spread / a.n, a.body / a.n);
        for (int i = 1; i < 8; ++i) if (a.ph[i] > 0) printf(" %d:%5.2f", i, a.ph[i] / a.n);
        printf("\n");
    }
    return 0;
}
