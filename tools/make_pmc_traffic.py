#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from a pmc_summary.json (tools/pmc_summary.py over the separate FETCH_SIZE / WRITE_SIZE
passes of tools/profile_round.sh): HBM bytes per launch of the two kernels bench.py's roofline objects describe, stamped with
the sha256 of the gemm.hip they were measured on -- bench.py reports `traffic` only while that hash matches the source.

    python tools/make_pmc_traffic.py gpurun_out/prof_r02/pmc_summary.json profiles/r02_pmc_traffic.json"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pick(d, prefix):
    ks = [k for k in d if k.startswith(prefix) and "hbm_bytes" in d[k]]
    if not ks:
        raise SystemExit(f"no kernel starting with {prefix!r} (with FETCH_SIZE and WRITE_SIZE) in the summary")
    k = max(ks, key=lambda k: d[k]["dispatches_seen"])
    return k, d[k]


def main(src, dst):
    d = json.load(open(src))
    gsrc = os.path.join(ROOT, "quadruplet-sentence-transformer_amd", "csrc", "gemm.hip")
    kw, w = pick(d, "gemm_tn_group_kernel")
    kf, f = pick(d, "gemm_nt_kernel<2, 2, 2, 2>")
    M, H, I = 32768, 384, 1536
    nparam = H * I + I * H + H * H + 3 * H * H
    out = {
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 2 "
                  "--kernel-reps 1 --no-cpu-baseline --no-extras` (in-step launches, MiniLM c2); gfx950 correction: FETCH_SIZE "
                  "counts 64 B per 128-B request for 16-B/lane streaming reads -> doubled (MI355X_MICROARCH.md, HBM); "
                  "WRITE_SIZE exact (also for float atomics); averaged per dispatch by tools/pmc_summary.py",
        "gemm_hip_sha256": hashlib.sha256(open(gsrc, "rb").read()).hexdigest(),
        "gemm_tn_group_kernel": {
            "kernel": kw + " (all four wgrads of one MiniLM layer, M = 32768, one M-range per XCD)",
            "dispatches": w["dispatches_seen"], "FETCH_SIZE_KiB_raw": round(w["FETCH_SIZE"], 1),
            "WRITE_SIZE_KiB": round(w["WRITE_SIZE"], 1), "hbm_bytes_per_launch": int(w["hbm_bytes"]),
            "algorithmic_bytes_per_launch": int(2 * M * (8 * H + 2 * I) + 4 * nparam),
            "atomic_flush_bytes_per_launch": int(8 * 4 * nparam),
            "algorithmic_note": "each dY and X row read once (bf16) + the 1,769,472 fp32 weight gradients once; the launch in fact "
                                "adds 8 fp32 partial sums per gradient with float atomics (atomic_flush_bytes_per_launch), which is "
                                "traffic of the method, not of the problem",
            "avg_duration_us_under_pmc": round(w["avg_duration_us_under_pmc"], 1)},
        "gemm_nt_kernel<2>": {
            "kernel": kf + " (FFN1 forward: [32768,384] x [1536,384]^T + bias, GELU -> gelu'(u), h bf16)",
            "dispatches": f["dispatches_seen"], "FETCH_SIZE_KiB_raw": round(f["FETCH_SIZE"], 1),
            "WRITE_SIZE_KiB": round(f["WRITE_SIZE"], 1), "hbm_bytes_per_launch": int(f["hbm_bytes"]),
            "algorithmic_bytes_per_launch": int(2 * M * H + 2 * I * H + 4 * M * I),
            "avg_duration_us_under_pmc": round(f["avg_duration_us_under_pmc"], 1)},
        "gemm_nt_kernel<2>_hbm_bytes_per_launch": int(f["hbm_bytes"]),
    }
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
