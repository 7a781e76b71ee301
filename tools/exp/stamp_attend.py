"""Experiment helper: writes stamped copies (-DQ3_STAMPS) of the attention kernel / engine into tools/exp/src and builds
tools/exp/libq3tts_stamps.so. Usage: python tools/exp/stamp_attend.py (from the repo root)."""
import os, shutil, subprocess
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(R, "qwen3-tts-rust_amd", "csrc"); dst = os.path.join(R, "tools", "exp", "src")
shutil.rmtree(dst, ignore_errors=True); shutil.copytree(src, dst, ignore=shutil.ignore_patterns("build", "*.so"))
for f in os.listdir(dst):
    if f.endswith((".h", ".hip", ".cpp")):
        t = open(os.path.join(dst, f)).read().replace('"../../include/q3tts.h"', '"../../../include/q3tts.h"'); open(os.path.join(dst, f), "w").write(t)
p = os.path.join(dst, "q3_kernels.h"); s = open(p).read()
s = s.replace("    Q3QkPrep prep;    // used when fused\n};", "    Q3QkPrep prep;    // used when fused\n    unsigned long long* dbg;\n};"); open(p, "w").write(s)
p = os.path.join(dst, "q3_kernels.hip"); s = open(p).read()
a = s.index("template <int R, bool FUSED>\n__global__ __launch_bounds__(R * 256) void k_attend(Q3Attend a) {"); b = s.index("void q3_launch_attend(const Q3Attend& a, hipStream_t s) {")
k = s[a:b]
k = k.replace("    extern __shared__ __attribute__((aligned(16))) float smem[];\n    const int g = blockIdx.x, row = blockIdx.y;", "#define ASTAMP(i) do { if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)\n    extern __shared__ __attribute__((aligned(16))) float smem[];\n    const int g = blockIdx.x, row = blockIdx.y;\n    ASTAMP(0);", 1)
k = k.replace("    const int slot = a.row_slot[row];\n    const int T = pos + 1,", "    const int slot = a.row_slot[row];\n    ASTAMP(1);\n    const int T = pos + 1,", 1)
k = k.replace("    __syncthreads();\n    const uint16_t* kb = a.kc + hb * hd;", "    ASTAMP(2);\n    __syncthreads();\n    ASTAMP(3);\n    const uint16_t* kb = a.kc + hb * hd;", 1)
k = k.replace("    mloc = wave_max(mloc);\n    if (lane == 0) mw[hh * 4 + sw] = mloc;\n    __syncthreads();", "    ASTAMP(4);\n    mloc = wave_max(mloc);\n    if (lane == 0) mw[hh * 4 + sw] = mloc;\n    __syncthreads();\n    ASTAMP(5);", 1)
k = k.replace("    if (lane == 0) lw[hh * 4 + sw] = lsum;\n    __syncthreads();", "    if (lane == 0) lw[hh * 4 + sw] = lsum;\n    __syncthreads();\n    ASTAMP(6);", 1)
k = k.replace("    __syncthreads();\n    for (int i = tid; i < R * hd; i += R * 256) {\n        const int h2 = i / hd, d = i - h2 * hd;", "    ASTAMP(7);\n    __syncthreads();\n    ASTAMP(8);\n    for (int i = tid; i < R * hd; i += R * 256) {\n        const int h2 = i / hd, d = i - h2 * hd;", 1)
tail = "        a.out[(size_t)row * a.ldo + (size_t)(g * R + h2) * hd + d] = ov / l;\n    }\n}"
i = k.rindex(tail); k = k[:i] + tail[:-1] + "    ASTAMP(9);\n}" + k[i + len(tail):]
s = s[:a] + k + s[b:]; open(p, "w").write(s)
p = os.path.join(dst, "q3_engine.hip"); s = open(p).read()
s = s.replace("        at.fused = fused; at.prep = qp;\n        q3_launch_attend(at, s);", '''        at.fused = fused; at.prep = qp;
        {
            static unsigned long long* abuf = nullptr;
            static const char* which = getenv("Q3_ATT_STAMP");
            if (!abuf) { hipMalloc((void**)&abuf, 256); hipMemset(abuf, 0, 256); }
            at.dbg = nullptr;
            if (which && fused && l == t.L - 1 && ((which[0] == 'P') == (&t == &e->P))) {
                at.dbg = abuf;
                unsigned long long st[16]; hipMemcpy(st, abuf, 128, hipMemcpyDeviceToHost);
                fprintf(stderr, "attend stamps: pos/slot %lld | prep done %lld | barrier1 %lld | scores %lld | barrier2 %lld | barrier3 %lld | PV %lld | barrier4 %lld | end %lld\\n",
                        (long long)(st[1] - st[0]), (long long)(st[2] - st[0]), (long long)(st[3] - st[0]), (long long)(st[4] - st[0]), (long long)(st[5] - st[0]), (long long)(st[6] - st[0]), (long long)(st[7] - st[0]), (long long)(st[8] - st[0]), (long long)(st[9] - st[0]));
            }
        }
        q3_launch_attend(at, s);''', 1)
s = s.replace("    Q3Attend at{}; at.qkv = (const float*)dq.p;", "    Q3Attend at{}; at.dbg = nullptr; at.qkv = (const float*)dq.p;")
open(p, "w").write(s)
objs = []
for f in ["q3_kernels.hip", "q3_gemm.hip", "q3_engine.hip", "q3_vocoder.hip", "q3_mel.hip", "q3_rng.cpp", "q3_gguf.cpp"]:
    o = os.path.join(R, "tools", "exp", "bs", "att_" + f + ".o"); objs.append(o)
    if True:
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17", "-w", "-x", "hip", "-c", os.path.join(dst, f), "-o", o])
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(R, "tools", "exp", "libq3tts_stamps.so")] + objs)
print("built")
