// q3_node.hip — the 8 GPUs of one node behind the C ABI (include/q3tts.h q3tts_node_*).
//
// The reference runs one utterance at a time on one device (n_seq_max = 1, /root/reference/src/models/llama/mod.rs:413; one engine,
// `&mut self`, src/tts/engine.rs:390). Utterances share nothing but read-only weights, so the path shards over independent units:
// one engine (full weight replica) and one host thread per device, request i of a batch goes to device i mod G in order, and the
// only collective is the gather of the finished PCM to device 0 — as 16-bit samples, the form the reference saves audio in
// (src/utils/audio.rs:30-46): every device packs its utterances into one i16 buffer, the per-utterance sample counts travel in one
// ncclAllGather, the buffers in one group of ncclSend / ncclRecv over xGMI (RCCL has no gatherv), and device 0 copies the lot to the
// host once. RCCL is resolved at run time (dlopen): the library has no link-time dependency on it and loads on a machine without one.
#include <dlfcn.h>

#include <chrono>
#include <cstring>
#include <thread>

#include "q3_engine.h"

// ---- the nine RCCL entry points used, through dlsym (signatures: /opt/rocm/include/rccl/rccl.h) ------------------------------------
typedef struct ncclComm* nccl_comm_t;
typedef int nccl_result_t;                 // ncclSuccess = 0
enum { Q3_NCCL_INT8 = 0, Q3_NCCL_INT32 = 2 };  // ncclInt8 = 0, ncclInt32 = 2 (rccl.h ncclDataType_t)
struct Rccl {
    void* lib = nullptr;
    nccl_result_t (*CommInitAll)(nccl_comm_t*, int, const int*) = nullptr;
    nccl_result_t (*CommDestroy)(nccl_comm_t) = nullptr;
    const char* (*GetErrorString)(nccl_result_t) = nullptr;
    nccl_result_t (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    nccl_result_t (*Send)(const void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    nccl_result_t (*Recv)(void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    nccl_result_t (*GroupStart)() = nullptr;
    nccl_result_t (*GroupEnd)() = nullptr;
    std::string open() {
        if (lib) return "";
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) return "RCCL not found (librccl.so.1): the multi-GPU gather needs it";
#define SYM(field, name) do { *(void**)&field = dlsym(lib, name); if (!field) return std::string("RCCL symbol missing: ") + name; } while (0)
        SYM(CommInitAll, "ncclCommInitAll"); SYM(CommDestroy, "ncclCommDestroy"); SYM(GetErrorString, "ncclGetErrorString");
        SYM(AllGather, "ncclAllGather"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd");
#undef SYM
        return "";
    }
};
static Rccl g_rccl;

// f32 -> i16 exactly as the reference writes WAV samples (src/utils/audio.rs:35-37): (x * 32767).clamp(-32768, 32767) as i16 — `as`
// truncates toward zero. Rows of the engine's packed device PCM -> one contiguous i16 buffer (utterance j at off[j]).
__global__ __launch_bounds__(256) void k_pcm_pack_i16(const float* src, size_t stride, const int* n_samples, const long long* off, int16_t* dst) {
    const int j = blockIdx.y, n = n_samples[j];
    const float* s = src + (size_t)j * stride;
    int16_t* d = dst + off[j];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float v = s[i] * 32767.0f;
        v = fminf(fmaxf(v, -32768.0f), 32767.0f);
        d[i] = (int16_t)(int)truncf(v);
    }
}

struct NodeDev {
    int device = 0;
    q3tts_engine* eng = nullptr;
    hipStream_t stream = nullptr;              // collective stream of this device
    nccl_comm_t comm = nullptr;
    std::vector<int> idx;                      // global request indices of the current call, in order
    std::vector<q3tts_request> reqs;
    std::vector<q3tts_result> outs;
    int rc = 0; std::string err;
    int16_t* pack = nullptr; size_t pack_cap = 0;   // this device's utterances as i16, back to back
    int* lens_dev = nullptr; long long* off_dev = nullptr; int* all_lens_dev = nullptr; int lens_cap = 0;
    double gen_ms = 0;
};

struct q3tts_node {
    std::vector<NodeDev> dev;
    std::string err;
    int16_t* root = nullptr; size_t root_cap = 0;      // device 0: every device's pack buffer, rank after rank
    int16_t* host = nullptr; size_t host_cap = 0;      // pinned landing buffer of the gathered PCM
    q3tts_node_timings tm{};
    bool comms = false;
};
static thread_local std::string g_node_err;
static int node_err(q3tts_node* n, int code, const std::string& m) { if (n) n->err = m; g_node_err = m; return code; }
extern "C" const char* q3tts_node_last_error(const q3tts_node* n) { return n ? n->err.c_str() : g_node_err.c_str(); }

// global request indices of rank r of G, in order: {i : i mod G == r} — the partition BASELINE.json's north_star and SURVEY.md §8e name
extern "C" int32_t q3tts_node_shard(int32_t n_total, int32_t world, int32_t rank, int32_t* idx, int32_t cap) {
    if (n_total < 0 || world < 1 || rank < 0 || rank >= world) return -1;
    int32_t k = 0;
    for (int32_t i = rank; i < n_total; i += world) { if (idx && k < cap) idx[k] = i; ++k; }
    return k;
}

extern "C" void q3tts_node_destroy(q3tts_node* n) {
    if (!n) return;
    for (auto& d : n->dev) {
        hipSetDevice(d.device);
        if (d.comm && g_rccl.CommDestroy) g_rccl.CommDestroy(d.comm);
        if (d.eng) q3tts_engine_destroy(d.eng);
        hipSetDevice(d.device);
        hipFree(d.pack); hipFree(d.lens_dev); hipFree(d.off_dev); hipFree(d.all_lens_dev);
        if (d.stream) hipStreamDestroy(d.stream);
    }
    if (!n->dev.empty()) hipSetDevice(n->dev[0].device);
    hipFree(n->root);
    if (n->host) hipHostFree(n->host);
    delete n;
}

extern "C" int q3tts_node_create(const q3tts_engine_config* cfg, const int32_t* devices, int32_t n_devices, q3tts_node** out) {
    if (!cfg || !devices || !out || n_devices < 1 || n_devices > 64) return node_err(nullptr, Q3TTS_ERR_INVALID, "node: null argument or device count outside 1..64");
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return node_err(nullptr, Q3TTS_ERR_INVALID, "node: a device is listed twice (one engine and one RCCL rank per GPU)");
    q3tts_node* n = new q3tts_node();
    n->dev.resize(n_devices);
    // one engine per device, created by one thread per device (weights are generated / uploaded in parallel)
    std::vector<std::thread> th;
    for (int r = 0; r < n_devices; ++r) {
        n->dev[r].device = devices[r];
        th.emplace_back([n, r, cfg]() {
            NodeDev& d = n->dev[r];
            q3tts_engine_config c = *cfg;
            c.device = d.device;
            d.rc = q3tts_engine_create(&c, &d.eng);
            if (d.rc != Q3TTS_OK) { d.err = q3tts_last_error(nullptr); return; }
            d.rc = q3tts_set_device_pcm(d.eng, 1);
            if (hipSetDevice(d.device) != hipSuccess || hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess) { d.rc = Q3TTS_ERR_DEVICE; d.err = "node: stream creation failed"; }
        });
    }
    for (auto& t : th) t.join();
    for (int r = 0; r < n_devices; ++r)
        if (n->dev[r].rc != Q3TTS_OK) {
            const int rc = n->dev[r].rc; const std::string m = "device " + std::to_string(devices[r]) + ": " + n->dev[r].err;
            q3tts_node_destroy(n);
            return node_err(nullptr, rc, m);
        }
    *out = n;
    return Q3TTS_OK;
}

// communicators are created on the first gathering call (a codes-only or host-PCM node never touches RCCL)
static int node_comms(q3tts_node* n) {
    if (n->comms) return Q3TTS_OK;
    const std::string e = g_rccl.open();
    if (!e.empty()) return node_err(n, Q3TTS_ERR_UNSUPPORTED, e);
    const int G = (int)n->dev.size();
    std::vector<nccl_comm_t> comms(G); std::vector<int> devs(G);
    for (int r = 0; r < G; ++r) devs[r] = n->dev[r].device;
    const nccl_result_t rc = g_rccl.CommInitAll(comms.data(), G, devs.data());
    if (rc != 0) return node_err(n, Q3TTS_ERR_DEVICE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc));
    for (int r = 0; r < G; ++r) n->dev[r].comm = comms[r];
    n->comms = true;
    return Q3TTS_OK;
}

static double node_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define NHIP(n, call) do { hipError_t er__ = (call); if (er__ != hipSuccess) return node_err((n), Q3TTS_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(er__)); } while (0)
#define NNCCL(n, call) do { nccl_result_t rc__ = (call); if (rc__ != 0) return node_err((n), Q3TTS_ERR_DEVICE, std::string(#call) + ": " + g_rccl.GetErrorString(rc__)); } while (0)

// An open RCCL group is closed on every way out: an early return between ncclGroupStart and ncclGroupEnd would leave the group open and
// the next collective of the process queued behind it for good.
struct NcclGroup {
    bool open = false;
    nccl_result_t start() { const nccl_result_t rc = g_rccl.GroupStart(); open = rc == 0; return rc; }
    nccl_result_t end() { open = false; return g_rccl.GroupEnd(); }
    ~NcclGroup() { if (open) g_rccl.GroupEnd(); }
};

// the one collective: i16 PCM of every utterance to device 0, then one buffer per utterance (q3tts_free) at its global index
static int node_gather(q3tts_node* n, q3tts_result* outs, int16_t** pcm_i16) {
    const int G = (int)n->dev.size();
    int maxn = 1;
    for (int r = 0; r < G; ++r) maxn = std::max(maxn, (int)n->dev[r].idx.size());
    std::vector<size_t> pack_samples(G, 0);
    for (int r = 0; r < G; ++r) {  // pack: f32 rows of the engine's device PCM -> one i16 buffer per device
        NodeDev& d = n->dev[r];
        NHIP(n, hipSetDevice(d.device));
        const int cnt = (int)d.idx.size();
        std::vector<int> lens(maxn, 0); std::vector<long long> off(maxn, 0);
        size_t tot = 0;
        for (int j = 0; j < cnt; ++j) { lens[j] = (d.reqs[j].want_pcm && d.outs[j].status == Q3TTS_OK) ? d.outs[j].n_samples : 0; off[j] = (long long)tot; tot += (size_t)lens[j]; }
        pack_samples[r] = tot;
        if (d.lens_cap < maxn) {
            hipFree(d.lens_dev); hipFree(d.off_dev); hipFree(d.all_lens_dev); d.lens_dev = nullptr; d.off_dev = nullptr; d.all_lens_dev = nullptr; d.lens_cap = 0;
            NHIP(n, hipMalloc((void**)&d.lens_dev, sizeof(int) * maxn)); NHIP(n, hipMalloc((void**)&d.off_dev, sizeof(long long) * maxn));
            NHIP(n, hipMalloc((void**)&d.all_lens_dev, sizeof(int) * (size_t)maxn * G));
            d.lens_cap = maxn;
        }
        if (d.pack_cap < tot + 1) { hipFree(d.pack); d.pack = nullptr; d.pack_cap = 0; NHIP(n, hipMalloc((void**)&d.pack, sizeof(int16_t) * (tot + 1))); d.pack_cap = tot + 1; }
        NHIP(n, hipMemcpyAsync(d.lens_dev, lens.data(), sizeof(int) * maxn, hipMemcpyHostToDevice, d.stream));
        NHIP(n, hipMemcpyAsync(d.off_dev, off.data(), sizeof(long long) * maxn, hipMemcpyHostToDevice, d.stream));
        NHIP(n, hipStreamSynchronize(d.stream));  // (lens / off are locals)
        float* base = nullptr; int64_t stride = 0; int32_t rows = 0;
        if (cnt > 0 && tot > 0) {
            if (q3tts_get_device_pcm(d.eng, &base, &stride, &rows) != Q3TTS_OK || !base || rows < cnt) return node_err(n, Q3TTS_ERR_STATE, "node: the engine kept no device PCM");
            hipLaunchKernelGGL(k_pcm_pack_i16, dim3(64, cnt), dim3(256), 0, d.stream, (const float*)base, (size_t)stride, (const int*)d.lens_dev, (const long long*)d.off_dev, d.pack);
            NHIP(n, hipGetLastError());
        }
    }
    NcclGroup grp;
    // sample counts of every utterance of every device, on every device: one all-gather of maxn int32 per rank
    NNCCL(n, grp.start());
    for (int r = 0; r < G; ++r) {
        NodeDev& d = n->dev[r];
        NHIP(n, hipSetDevice(d.device));
        NNCCL(n, g_rccl.AllGather(d.lens_dev, d.all_lens_dev, (size_t)maxn, Q3_NCCL_INT32, d.comm, d.stream));
    }
    NNCCL(n, grp.end());
    NodeDev& d0 = n->dev[0];
    NHIP(n, hipSetDevice(d0.device));
    std::vector<int> all_lens((size_t)maxn * G);
    NHIP(n, hipMemcpyAsync(all_lens.data(), d0.all_lens_dev, sizeof(int) * all_lens.size(), hipMemcpyDeviceToHost, d0.stream));
    NHIP(n, hipStreamSynchronize(d0.stream));
    std::vector<size_t> rank_off(G + 1, 0);  // from the GATHERED counts: what device 0 would know in a multi-process job
    for (int r = 0; r < G; ++r) { size_t t = 0; for (int j = 0; j < maxn; ++j) t += (size_t)all_lens[(size_t)r * maxn + j]; rank_off[r + 1] = rank_off[r] + t; }
    for (int r = 0; r < G; ++r)
        if (rank_off[r + 1] - rank_off[r] != pack_samples[r]) return node_err(n, Q3TTS_ERR_DEVICE, "node: gathered sample counts disagree with the packed buffers");
    const size_t total = rank_off[G];
    if (n->root_cap < total + 1) { hipFree(n->root); n->root = nullptr; n->root_cap = 0; NHIP(n, hipMalloc((void**)&n->root, sizeof(int16_t) * (total + 1))); n->root_cap = total + 1; }
    if (n->host_cap < total + 1) { if (n->host) hipHostFree(n->host); n->host = nullptr; n->host_cap = 0; NHIP(n, hipHostMalloc((void**)&n->host, sizeof(int16_t) * (total + 1), hipHostMallocDefault)); n->host_cap = total + 1; }
    // PCM: one group of point-to-point transfers over xGMI, every peer straight into its place in device 0's buffer
    NNCCL(n, grp.start());
    for (int r = 1; r < G; ++r) {
        const size_t bytes = (rank_off[r + 1] - rank_off[r]) * sizeof(int16_t);
        if (!bytes) continue;
        NHIP(n, hipSetDevice(n->dev[r].device));
        NNCCL(n, g_rccl.Send(n->dev[r].pack, bytes, Q3_NCCL_INT8, 0, n->dev[r].comm, n->dev[r].stream));
        NHIP(n, hipSetDevice(d0.device));
        NNCCL(n, g_rccl.Recv(n->root + rank_off[r], bytes, Q3_NCCL_INT8, r, d0.comm, d0.stream));
    }
    NNCCL(n, grp.end());
    NHIP(n, hipSetDevice(d0.device));
    if (rank_off[1] > 0) NHIP(n, hipMemcpyAsync(n->root, d0.pack, rank_off[1] * sizeof(int16_t), hipMemcpyDeviceToDevice, d0.stream));
    if (total > 0) NHIP(n, hipMemcpyAsync(n->host, n->root, total * sizeof(int16_t), hipMemcpyDeviceToHost, d0.stream));
    NHIP(n, hipStreamSynchronize(d0.stream));
    for (int r = 1; r < G; ++r) { NHIP(n, hipSetDevice(n->dev[r].device)); NHIP(n, hipStreamSynchronize(n->dev[r].stream)); }
    // hand every utterance its own buffer (q3tts_free), at its global index
    for (int r = 0; r < G; ++r) {
        size_t o = rank_off[r];
        for (size_t j = 0; j < n->dev[r].idx.size(); ++j) {
            const int len = all_lens[(size_t)r * maxn + j], gi = n->dev[r].idx[j];
            if (len > 0) {
                int16_t* p = (int16_t*)malloc(sizeof(int16_t) * (size_t)len);
                if (!p) return node_err(n, Q3TTS_ERR_OOM, "malloc");
                memcpy(p, n->host + o, sizeof(int16_t) * (size_t)len);
                pcm_i16[gi] = p;
            }
            outs[gi].n_samples = len;
            o += (size_t)len;
        }
    }
    n->tm.gathered_bytes = (int64_t)(total * sizeof(int16_t));
    return Q3TTS_OK;
}

// Error contract: whatever the return code, every outs[i] is valid to pass to q3tts_result_free (the results of the devices that
// succeeded are handed over even when another device failed; a request that failed by itself carries its own status and the call
// still returns OK), and on a non-OK return no pcm_i16[i] is left allocated (all NULL).
extern "C" int q3tts_node_generate_batch(q3tts_node* n, const q3tts_request* reqs, int32_t n_reqs, q3tts_result* outs, int16_t** pcm_i16) {
    if (!n || !reqs || !outs || n_reqs <= 0) return node_err(n, Q3TTS_ERR_INVALID, "node: null/empty argument");
    const int G = (int)n->dev.size();
    const bool gather = pcm_i16 != nullptr;
    for (int i = 0; i < n_reqs; ++i) { memset(&outs[i], 0, sizeof(outs[i])); outs[i].status = Q3TTS_ERR_STATE; if (gather) pcm_i16[i] = nullptr; }
    if (gather) { const int rc = node_comms(n); if (rc != Q3TTS_OK) return rc; }
    const double t0 = node_now_ms();
    // ---- generation: device r runs requests {i : i mod G == r} through its own continuous-batching engine, no exchange
    std::vector<std::thread> th;
    for (int r = 0; r < G; ++r) {
        NodeDev& d = n->dev[r];
        d.idx.clear(); d.reqs.clear();
        for (int i = r; i < n_reqs; i += G) { d.idx.push_back(i); d.reqs.push_back(reqs[i]); if (gather && reqs[i].want_pcm) d.reqs.back().want_pcm = 2; }
        d.outs.assign(d.reqs.size(), q3tts_result{});
        for (auto& o : d.outs) o.status = Q3TTS_ERR_STATE;
        d.rc = Q3TTS_OK; d.err.clear(); d.gen_ms = 0;
        if (d.reqs.empty()) continue;
        th.emplace_back([&d]() {
            const double a = node_now_ms();
            d.rc = q3tts_generate_batch(d.eng, d.reqs.data(), (int)d.reqs.size(), d.outs.data());
            if (d.rc != Q3TTS_OK) d.err = q3tts_last_error(d.eng);
            d.gen_ms = node_now_ms() - a;
        });
    }
    for (auto& t : th) t.join();
    double gen_ms = 0;
    int first_bad = -1;
    for (int r = 0; r < G; ++r) {  // every device's results go to the caller first — also those of the devices beside a failing one
        NodeDev& d = n->dev[r];
        for (size_t j = 0; j < d.idx.size(); ++j) outs[d.idx[j]] = d.outs[j];
        if (d.rc != Q3TTS_OK && first_bad < 0) first_bad = r;
        gen_ms = std::max(gen_ms, d.gen_ms);
    }
    n->tm.generate_ms = (float)gen_ms; n->tm.gather_ms = 0; n->tm.gathered_bytes = 0; n->tm.n_devices = G;
    if (first_bad >= 0) return node_err(n, n->dev[first_bad].rc, "device " + std::to_string(n->dev[first_bad].device) + ": " + n->dev[first_bad].err);
    if (!gather) { n->tm.total_ms = (float)(node_now_ms() - t0); return Q3TTS_OK; }
    const double tg = node_now_ms();
    const int grc = node_gather(n, outs, pcm_i16);
    if (grc != Q3TTS_OK) {  // no half-delivered gather: what was handed out so far is taken back
        for (int i = 0; i < n_reqs; ++i) { free(pcm_i16[i]); pcm_i16[i] = nullptr; }
        return grc;
    }
    n->tm.gather_ms = (float)(node_now_ms() - tg);
    n->tm.total_ms = (float)(node_now_ms() - t0);
    return Q3TTS_OK;
}

extern "C" int q3tts_node_get_timings(const q3tts_node* n, q3tts_node_timings* out) {
    if (!n || !out) return Q3TTS_ERR_INVALID;
    *out = n->tm;
    return Q3TTS_OK;
}
extern "C" q3tts_engine* q3tts_node_engine(q3tts_node* n, int32_t rank) { return (n && rank >= 0 && rank < (int)n->dev.size()) ? n->dev[rank].eng : nullptr; }
extern "C" int32_t q3tts_node_size(const q3tts_node* n) { return n ? (int32_t)n->dev.size() : 0; }
