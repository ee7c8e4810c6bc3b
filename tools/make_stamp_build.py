"""Diagnostic build: gemm_nt_kernel with s_memtime stamps (prologue / per-stage wait, barrier, compute / epilogue).
Writes quadruplet-sentence-transformer_amd/libqst_stamp.so (never loaded by the product)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "quadruplet-sentence-transformer_amd", "csrc")
s = open(os.path.join(CS, "gemm.hip")).read()
s = s.replace("namespace {\n", 'namespace {\n#define STAMP(i) do { unsigned long long _t; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); tstamp[i] = _t; } while (0)\n', 1)
s = s.replace("    const int nk = g.K / NBK;\n    const int fr = lane & 31, fh = lane >> 5;\n    issue(0);",
              "    const int nk = g.K / NBK;\n    const int fr = lane & 31, fh = lane >> 5;\n    unsigned long long tstamp[8];\n    unsigned long long twait = 0, tbar = 0, tcomp = 0;\n    STAMP(0);\n    issue(0);\n    STAMP(1);")
s = s.replace("        wait_vmcnt<0>();                              // stage kt has landed for this wave's DMAs\n        __builtin_amdgcn_s_barrier();                 // ... for everyone's; and everyone is done reading slot (kt-1)&1\n",
              "        STAMP(2);\n        wait_vmcnt<0>();\n        STAMP(3);\n        __builtin_amdgcn_s_barrier();\n        STAMP(4);\n        twait += tstamp[3] - tstamp[2]; tbar += tstamp[4] - tstamp[3];\n")
s = s.replace("                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // D rows = n, col = m\n        }\n    }\n",
              "                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);   // D rows = n, col = m\n        }\n        STAMP(5);\n        tcomp += tstamp[5] - tstamp[4];\n    }\n")
s = s.replace("    __builtin_amdgcn_s_barrier();                     // all waves done with the ring before the epilogue reuses it\n",
              "    __builtin_amdgcn_s_barrier();\n    STAMP(6);\n    unsigned long long te0 = 0, te1 = 0;\n", 1)
s = s.replace("#pragma unroll\n    for (int i = 0; i < 2; ++i) {\n        f32x4 rv[12];", "    unsigned long long te0 = 0, te1 = 0, te2 = 0;\n#pragma unroll\n    for (int i = 0; i < 2; ++i) {\n        STAMP(2);\n        f32x4 rv[12];")
s = s.replace("#pragma unroll\n        for (int j = 0; j < 3; ++j)\n#pragma unroll\n            for (int g4 = 0; g4 < 4; ++g4) {\n                f32x4 v;\n#pragma unroll\n                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g4 + e];\n                *(f32x4*)(stg + fr * NT_STG_LD", "        STAMP(3);\n#pragma unroll\n        for (int j = 0; j < 3; ++j)\n#pragma unroll\n            for (int g4 = 0; g4 < 4; ++g4) {\n                f32x4 v;\n#pragma unroll\n                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g4 + e];\n                *(f32x4*)(stg + fr * NT_STG_LD")
s = s.replace("        // the same wave reads back what it wrote (wave-private region): no workgroup barrier needed\n#pragma unroll\n        for (int t = 0; t < 12; ++t) {\n            const int idx = t * 64 + lane;", "        STAMP(4);\n        te0 += tstamp[3] - tstamp[2]; te1 += tstamp[4] - tstamp[3];\n#pragma unroll\n        for (int t = 0; t < 12; ++t) {\n            const int idx = t * 64 + lane;")
a = s.index("// ---------------------------------------------------------------- TN")
b = s.rfind("}\n", 0, a)
s = s[:b] + "    STAMP(7);\n    if (lane == 0 && g.colsum) { unsigned long long* dbg = (unsigned long long*)g.colsum + ((size_t)blockIdx.x * 8 + wave) * 8; dbg[0] = tstamp[1] - tstamp[0]; dbg[1] = twait; dbg[2] = tbar; dbg[3] = tcomp; dbg[4] = tstamp[7] - tstamp[6]; dbg[5] = tstamp[7] - tstamp[0]; dbg[6] = te0; dbg[7] = te1; }\n" + s[b:]
open(os.path.join(CS, "_gemm_stamp.hip"), "w").write(s)
# ---- grouped wgrad stamps (task-list kernel): totals over all tasks of an MFMA wave
s = s.replace("#pragma unroll 1\n    for (int task = 0; task < ntasks; ++task) {", "    unsigned long long tstamp[8]; unsigned long long tbar = 0, tcomp = 0, tepi = 0, nst = 0; STAMP(0);\n#pragma unroll 1\n    for (int task = 0; task < ntasks; ++task) {")
s = s.replace("            __builtin_amdgcn_s_barrier();                  // stage mt landed (the loaders waited for it before arriving)\n", "            STAMP(3);\n            __builtin_amdgcn_s_barrier();\n            STAMP(4);\n            tbar += tstamp[4] - tstamp[3]; ++nst;\n")
s = s.replace("                        for (int e = 0; e < 8; ++e) bsum[i] += (float)fa[i][e];\n                }\n            }\n        }\n", "                        for (int e = 0; e < 8; ++e) bsum[i] += (float)fa[i][e];\n                }\n            }\n            STAMP(5);\n            tcomp += tstamp[5] - tstamp[4];\n        }\n")
s = s.replace("        __builtin_amdgcn_s_barrier();                      // end of piece: the loaders may refill the ring while we flush\n", "        __builtin_amdgcn_s_barrier();\n        STAMP(6);\n")
s = s.replace("                if (fh == 0 && n < g.N) atomicAdd(&g.colsum[n], t);\n            }\n        }\n    }\n}", "                if (fh == 0 && n < g.N) atomicAdd(&g.colsum[n], t);\n            }\n        }\n        STAMP(7);\n        tepi += tstamp[7] - tstamp[6];\n    }\n    if (wave < 4 && lane == 0 && grp.prob[7].A) { STAMP(1); unsigned long long* dbg = (unsigned long long*)grp.prob[7].A + ((size_t)blockIdx.x * 4 + wave) * 8; dbg[0] = nst; dbg[1] = 0; dbg[2] = tbar; dbg[3] = tcomp; dbg[4] = tepi; dbg[5] = tstamp[1] - tstamp[0]; }\n}")
open(os.path.join(CS, "_gemm_stamp.hip"), "w").write(s)
objs = [os.path.join(CS, f) for f in ("qst_api.o", "loss.o", "attention.o", "rowops.o", "optim.o", "x3.o")]
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", os.path.join(CS, "_gemm_stamp.hip"), "-o", os.path.join(CS, "_gemm_stamp.o")], check=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "quadruplet-sentence-transformer_amd", "libqst_stamp.so"), os.path.join(CS, "_gemm_stamp.o")] + objs, check=True)
os.remove(os.path.join(CS, "_gemm_stamp.o"))
print("built libqst_stamp.so")
