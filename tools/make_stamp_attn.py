"""Diagnostic build of attention.hip with s_memtime stamps in attn_bwd_dkv_kernel."""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "quadruplet-sentence-transformer_amd", "csrc")
s = open(os.path.join(CS, "attention.hip")).read()
s = s.replace("namespace {\n", 'namespace {\n#define STAMP(i) do { unsigned long long _t; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); tstamp[i] = _t; } while (0)\n', 1)
a = s.index("__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs a) {")
head, body = s[:a], s[a:]
body = body.replace("    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, fr = lane & 31;", "    unsigned long long tstamp[8]; unsigned long long tloop = 0;\n    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, fr = lane & 31;\n    STAMP(0);", 1)
body = body.replace("        Stager<D> sg;\n", "        STAMP(1);\n        Stager<D> sg;\n", 1)
body = body.replace("        sg.template store<false>(0, qimg, tid);", "        STAMP(2);\n        sg.template store<false>(0, qimg, tid);", 1)
body = body.replace("        __syncthreads();\n        if (!active) continue;\n        for (int it = 0; it < rows / 32; ++it) {", "        __syncthreads();\n        STAMP(3);\n        if (!active) continue;\n        for (int it = 0; it < rows / 32; ++it) {", 1)
body = body.replace("    if (!active) return;\n    bf16* krow = a.dqkv", "    STAMP(4);\n    if (!active) return;\n    bf16* krow = a.dqkv", 1)
# end of kernel: before "template <typename K>\nint set_lds"
e = body.index("template <typename K>\nint set_lds")
k_end = body.rfind("}\n", 0, e)
body = body[:k_end] + "    STAMP(5);\n    if (lane == 0 && a.drel == nullptr && a.lse_out) { unsigned long long* dbg = (unsigned long long*)a.lse_out + ((size_t)blockIdx.x * 4 + wave) * 8; for (int i = 0; i < 6; ++i) dbg[i] = tstamp[i]; }\n" + body[k_end:]
body = body.replace("a.rel = rel_bias; a.dqkv = (bf16*)dqkv; a.drel = drel; a.delta = delta_scratch;", "a.rel = rel_bias; a.dqkv = (bf16*)dqkv; a.drel = nullptr; a.delta = delta_scratch; a.lse_out = drel;")
s = head + body
open(os.path.join(CS, "_attn_stamp.hip"), "w").write(s)
objs = [os.path.join(CS, f) for f in ("qst_api.o", "loss.o", "gemm.o", "rowops.o", "optim.o", "x3.o")]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", os.path.join(CS, "_attn_stamp.hip"), "-o", os.path.join(CS, "_attn_stamp.o")], check=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "quadruplet-sentence-transformer_amd", "libqst_stamp.so"), os.path.join(CS, "_attn_stamp.o")] + objs, check=True)
os.remove(os.path.join(CS, "_attn_stamp.hip")); os.remove(os.path.join(CS, "_attn_stamp.o"))
print("built libqst_stamp.so (attention)")
