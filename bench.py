#!/usr/bin/env python3
"""Headline benchmark: quadruplets/sec of one full fine-tuning step (forward + quadruplet loss +
backward + clip + AdamW) on synthetic all-MiniLM-L6-v2-shaped batches, seq_len=128, 64 quadruplets
per GPU (BASELINE.json configs[1]; configs[3] when launched with N > 1 ranks).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement): value = whole-job quadruplets/s with the
inputs resident in HBM; `roofline` = the dominant kernel's achieved TFLOP/s (its algorithmic FLOPs per
launch / its average launch duration measured here with HIP events) against the dense bf16 MFMA peak;
`cpu_baseline` = the CPU oracle (oracle/torch_ref.py) timed on this box's host cores, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md chip table)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="all-MiniLM-L6-v2")
    ap.add_argument("--batch", type=int, default=64, help="quadruplets per GPU")
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--dropout", type=float, default=0.1,
                    help="hidden + attention-probability dropout of the timed training step (the reference's fit() trains "
                         "HF modules in train() mode: 0.1 / 0.1); 0 = off. The step without it is reported beside it.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--kernel-reps", type=int, default=10)
    ap.add_argument("--graph", action="store_true",
                    help="single GPU: replay the step from a captured HIP graph (device-side LR schedule)")
    ap.add_argument("--train-precision", choices=["bf16", "f16", "f16w", "fp8", "bf16x3"], default="bf16",
                    help="precision of the TIMED training step: bf16 (BASELINE configs[1]); fp8 = BASELINE configs[4]'s 'fp8 "
                         "MFMA GEMMs': every forward Linear on the fp8 matrix cores, bf16 backward; bf16x3 = the fp32-class "
                         "parity path")
    ap.add_argument("--force-dp", action="store_true",
                    help="single GPU: run the TIMED step through the data-parallel path -- init_process_group('nccl', "
                         "world_size=1), staged backward, the seven async RCCL all-reduces of the gradient buckets -- so that "
                         "the collectives' launch cost is in the headline number (config.parallelism says so)")
    ap.add_argument("--profile", action="store_true",
                    help="profiled runs: with --no-extras, also skip the dropout-off side steps, so that every launch in the "
                         "trace belongs to the headline step (or is one of the --kernel-reps direct launches)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the forward-only and loss-kernel side measurements (profiled runs: keeps per-step kernel counts clean)")
    return ap.parse_args()


CONFIG1 = dict(n_quadruplets=256, batch=8, seq_len=32, seed=14)      # BASELINE.json configs[0]; SURVEY.md 8d
LOSS_KW = dict(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5)   # training/main.py:211-218


def _c1_batches(cfg):
    from quadruplet_sentence_transformer_amd.synthetic import synthetic_quadruplets
    n = CONFIG1["n_quadruplets"] // CONFIG1["batch"]
    return [synthetic_quadruplets(cfg, CONFIG1["batch"], CONFIG1["seq_len"], seed=CONFIG1["seed"], ragged=True, step=i)
            for i in range(n)]


def cpu_baseline(cfg, arena, seq_len, batch, budget_s=25.0):
    """The CPU oracle (oracle/torch_ref.py: fp32 torch restatement of the reference path, pinned by tests/golden) timed
    on this box's host cores -- the baseline SURVEY.md 8d asks for, not a target:
      * config 1 EXACTLY (BASELINE.json configs[0]): MiniLM dims, 256 synthetic quadruplets, seq_len 32 (ragged lengths
        U{4..32}), batches of 8, seed 14 -- forward-only encode+loss over all 32 batches with the per-batch and mean loss
        recorded, then the full training step (forward + loss + autograd backward + clip_grad_norm_ + AdamW with ST's two
        parameter groups) on the same batches, bounded by the time budget;
      * the headline shape (configs[1]: `batch` quadruplets x `seq_len`), one full training step after one warm-up
        forward, for a like-for-like ratio.
    `value` is the config-1 full-training-step rate; `cores` = torch's intra-op thread count, chosen by a short
    calibration (4 forward batches at 8 / 16 / 32 / all threads): at these sizes all 128 threads of the box's host are
    several times SLOWER than 8-16 (measured 9.8 vs 36 q/s), and the baseline should be the CPU's best."""
    import torch
    from oracle import torch_ref as R
    from quadruplet_sentence_transformer_amd.config import build_layout
    from quadruplet_sentence_transformer_amd.synthetic import synthetic_quadruplets
    t_begin = time.perf_counter()
    batches = [[torch.from_numpy(x) for x in b] for b in _c1_batches(cfg)]
    P = R.arena_to_dict(arena, cfg)
    max_threads = torch.get_num_threads()
    tried = {}
    for nt in sorted({min(max_threads, t) for t in (8, 16, 32, max_threads)}):
        torch.set_num_threads(nt)
        with torch.no_grad():
            R.quadruplet_step(P, cfg, *batches[0], LOSS_KW)
            t0 = time.perf_counter()
            for b in batches[:4]:
                R.quadruplet_step(P, cfg, *b, LOSS_KW)
            tried[nt] = time.perf_counter() - t0
    threads = min(tried, key=tried.get)
    torch.set_num_threads(threads)
    # ---- config 1, forward only
    losses = []
    with torch.no_grad():
        R.quadruplet_step(P, cfg, *batches[0], LOSS_KW)                       # thread-pool / allocator warm-up
        t0 = time.perf_counter()
        for b in batches:
            loss, _ = R.quadruplet_step(P, cfg, *b, LOSS_KW)
            losses.append(float(loss))
        t_fwd = time.perf_counter() - t0
    # ---- config 1, full training step (what SentenceTransformer.fit runs per batch)
    P = R.arena_to_dict(arena, cfg, requires_grad=True)
    segs, _ = build_layout(cfg)
    opt = torch.optim.AdamW([{"params": [P[s.name] for s in segs if s.decay], "weight_decay": 0.01},
                             {"params": [P[s.name] for s in segs if not s.decay], "weight_decay": 0.0}],
                            lr=2e-5, betas=(0.9, 0.999), eps=1e-8)
    allp = [p for g in opt.param_groups for p in g["params"]]
    n_train, t_train = 0, 0.0
    for i, b in enumerate(batches):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, _ = R.quadruplet_step(P, cfg, *b, LOSS_KW)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(allp, 1.0)
        opt.step()
        dt = time.perf_counter() - t0
        if i > 0:                                                            # first step pays autograd start-up
            n_train += CONFIG1["batch"]
            t_train += dt
        if time.perf_counter() - t_begin > 0.6 * budget_s and i >= 4:
            break
    # ---- headline shape, one full step
    P = R.arena_to_dict(arena, cfg, requires_grad=True)
    # the headline's own batch (64 quadruplets: 0.75 s per 16 on this box's host, so ~3 s per step and ~12 s in all; VERDICT r04
    # asked for the like-for-like size); larger batches stay bounded at 64
    batch = min(batch, 64)
    ids, mask, types = [torch.from_numpy(x) for x in synthetic_quadruplets(cfg, batch, seq_len, seed=14, step=1000)]
    wl, _ = R.quadruplet_step(P, cfg, ids[:, :2], mask[:, :2], types[:, :2], LOSS_KW)     # warm-up: forward AND backward
    wl.backward()
    for t_ in P.values():
        t_.grad = None
    # one untimed full-size step (allocator / thread pool at this size), then at least three timed ones, bounded in time
    loss, _ = R.quadruplet_step(P, cfg, ids, mask, types, LOSS_KW)
    loss.backward()
    for t_ in P.values():
        t_.grad = None
    n_c2, t0 = 0, time.perf_counter()
    while n_c2 < 3 or (n_c2 < 6 and time.perf_counter() - t0 < 9.0):
        loss, _ = R.quadruplet_step(P, cfg, ids, mask, types, LOSS_KW)
        loss.backward()
        for t_ in P.values():
            t_.grad = None
        n_c2 += 1
    t_c2 = (time.perf_counter() - t0) / n_c2
    nq = CONFIG1["n_quadruplets"]
    return {"value": round(n_train / t_train, 2), "unit": "quadruplets/s", "cores": threads, "kind": "port",
            "sample": f"BASELINE configs[0] exactly: MiniLM dims, {nq} quadruplets, seq_len {CONFIG1['seq_len']} ragged, "
                      f"batch {CONFIG1['batch']}, seed {CONFIG1['seed']}; value = full training steps (fwd+loss+bwd+clip+AdamW) "
                      f"over {n_train} quadruplets in {t_train:.1f} s; fp32 torch CPU oracle (oracle/torch_ref.py), "
                      f"{threads} threads of {os.cpu_count()} logical CPUs (fastest of "
                      + ", ".join(f"{k}: {4 * CONFIG1['batch'] / v:.0f} q/s fwd" for k, v in sorted(tried.items())) + ")",
            "config1_fwd_only": {"value": round(nq / t_fwd, 2), "unit": "quadruplets/s", "seconds": round(t_fwd, 2),
                                 "mean_loss": round(float(sum(losses) / len(losses)), 6),
                                 "per_batch_loss": [round(x, 6) for x in losses]},
            "headline_shape_train_step": {"value": round(batch / t_c2, 3), "unit": "quadruplets/s",
                                          "what": f"fwd+loss+bwd of {batch} quadruplets x seq_len {seq_len}: mean of {n_c2} steps after an "
                                                  "untimed step of the same size",
                                          "seconds_per_step": round(t_c2, 2)}}, losses


def config1_on_gpu(cfg, arena, cpu_losses=None):
    """The same 32 config-1 batches through the HIP path (forward + fused loss; then timed full training steps on a
    separate trainer): per-batch losses next to the CPU oracle's -- the north star's 'loss matching CPU reference within
    1e-3' on BASELINE configs[0], measured in the run."""
    import torch
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer
    tr = QuadrupletTrainer(cfg, arena=arena, device=f"cuda:{torch.cuda.current_device()}", lr=2e-5, weight_decay=0.01, max_grad_norm=1.0)
    batches = [[torch.from_numpy(x).cuda() for x in b] for b in _c1_batches(cfg)]
    out = {}
    for prec in ("bf16", "bf16x3"):
        losses = [float(tr.forward_loss(*b, precision=prec)[0].item()) for b in batches]
        out[prec] = {"mean_loss": round(sum(losses) / len(losses), 6)}
        if cpu_losses:
            out[prec]["max_abs_loss_diff_vs_cpu"] = float(f"{max(abs(a - b) for a, b in zip(losses, cpu_losses)):.3e}")
    for b in batches[:4]:
        tr.step(*b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches:
        tr.step(*b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["train_quadruplets_per_s"] = round(CONFIG1["n_quadruplets"] / dt, 1)
    return out


def golden_parity(golden_path):
    """max |emb - golden| of the HIP forward against the committed fp32 HF vectors (tests/golden) for the inference precisions
    -- the tolerance the north star states (rtol 1e-3 / atol 1e-4, element by element) next to the rate of each configuration.
    Two full-dims MiniLM cases: minilm_l128 (HF-init weights, 2 quadruplets x seq_len 128 ragged; the top-level entries, as in
    earlier rounds) and minilm_c1 (BASELINE configs[0]'s shape, 8 x 32, trained-like weights: the harder one)."""
    import numpy as np
    import torch
    from quadruplet_sentence_transformer_amd.config import PRESETS
    from quadruplet_sentence_transformer_amd.encoder import HipEncoder
    from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets
    if not os.path.exists(golden_path):
        return None
    g = np.load(golden_path)
    cfg = PRESETS["all-MiniLM-L6-v2"]
    out = {}
    for key, B, L, wkw in (("minilm_l128", 2, 128, dict(std=0.02)),
                           ("minilm_c1", 8, 32, dict(std=0.04, bias_std=0.02, ln_jitter=0.05))):
        enc = HipEncoder(cfg)
        enc.load_arena(synthetic_params(cfg, seed=14, **wkw))
        ids, mask, types = synthetic_quadruplets(cfg, B, L, seed=14, ragged=True)
        dev = [torch.from_numpy(x).view(4 * B, L).cuda() for x in (ids, mask, types)]
        ref = g[key + "_emb"].reshape(4 * B, -1)
        res = {}
        for prec in ("bf16", "f16", "f16w", "bf16x3"):
            emb = enc.forward(*dev, precision=prec)[0].cpu().numpy()
            d = np.abs(emb - ref)
            bad = d > 1e-4 + 1e-3 * np.abs(ref)
            res[prec] = {"max_abs_emb_diff": float(f"{d.max():.3e}"), "within_rtol1e-3_atol1e-4": bool(not bad.any()),
                         "elements_outside": int(bad.sum()), "elements": int(bad.size)}
        if key == "minilm_l128":
            out.update(res)
        else:
            out[key] = res
        del enc
    return out


def time_kernels(trainer, n, L, reps, batches):
    """Average launch duration, HIP events on the launch stream, of
      * the grouped wgrad kernel (gemm_tn_group_kernel: all four weight gradients of one layer, the largest
        per-launch kernel of the step; 6 launches per step), and
      * the FFN-1 forward GEMM (gemm_nt_kernel<QST_EPI_GELU>: [M,H] x [I,H]^T, bias + GELU epilogue)
    at the step's shapes. Each timed launch follows a full training step, so caches are in the state the kernels
    see inside the step (back-to-back launches re-read a warm Infinity Cache and run ~15% faster)."""
    import torch
    from quadruplet_sentence_transformer_amd import _lib
    cfg = trainer.cfg
    M, H, I = n * L, cfg.hidden_size, cfg.intermediate_size
    dev = trainer.enc.device
    bf = torch.bfloat16
    lib = trainer.enc.lib
    st = _lib.current_stream_ptr()
    # FFN1 forward
    A = torch.randn(M, H, device=dev).to(bf)
    W = (torch.randn(I, H, device=dev) * 0.02).to(bf)
    bias = torch.zeros(I, device=dev)
    U = torch.empty(M, I, dtype=bf, device=dev)
    Hh = torch.empty(M, I, dtype=bf, device=dev)
    g = _lib.QstGemmArgs()
    g.A, g.B, g.C, g.C2, g.bias = A.data_ptr(), W.data_ptr(), U.data_ptr(), Hh.data_ptr(), bias.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc = M, I, H, H, H, I
    # grouped wgrad of one layer
    grp = _lib.QstTnGroup()
    grp.nprob, grp.splits = 4, 0
    keep = []
    for i, (N, K) in enumerate([(H, I), (I, H), (H, H), (3 * H, H)]):
        dY = torch.randn(M, N, device=dev).to(bf)
        X = torch.randn(M, K, device=dev).to(bf)
        C = torch.zeros(N, K, device=dev)
        cs = torch.zeros(N, device=dev)
        q = grp.prob[i]
        q.A, q.B, q.C, q.colsum = dY.data_ptr(), X.data_ptr(), C.data_ptr(), cs.data_ptr()
        q.M, q.N, q.K, q.lda, q.ldb, q.ldc = M, N, K, N, K, K
        keep += [dY, X, C, cs]
    # the feed-forward block as one kernel (csrc/ffn.hip, inference variant: what encode() / the forward-only figure run)
    chain = None
    if lib.qst_ffn_chain_supported(H, I) and M >= 16384:
        W2 = (torch.randn(H, I, device=dev) * 0.02).to(bf)
        b2, gamma, beta = torch.zeros(H, device=dev), torch.ones(H, device=dev), torch.zeros(H, device=dev)
        resid = torch.randn(M, H, device=dev)
        y, yb, xh = torch.empty(M, H, device=dev), torch.empty(M, H, dtype=bf, device=dev), torch.empty(M, H, dtype=bf, device=dev)
        rs = torch.empty(M, device=dev)
        fa, le = _lib.QstFfnArgs(), _lib.QstLnEpi()
        fa.A, fa.B1, fa.B2, fa.bias1, fa.bias2, fa.resid = (A.data_ptr(), W.data_ptr(), W2.data_ptr(), bias.data_ptr(),
                                                           b2.data_ptr(), resid.data_ptr())
        fa.C, fa.C2, fa.M, fa.H, fa.I = y.data_ptr(), yb.data_ptr(), M, H, I
        le.gamma, le.beta, le.eps, le.xhat, le.rstd = gamma.data_ptr(), beta.data_ptr(), 1e-12, xh.data_ptr(), rs.data_ptr()
        chain = (fa, le)
        keep += [W2, b2, gamma, beta, resid, y, yb, xh, rs]
    t_ffn1 = t_wgrad = t_chain = 0.0
    for i in range(reps):
        trainer.step(*batches[i % len(batches)])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        _lib.check(lib.qst_gemm_tn_group(grp, st))
        ev[1].record()
        _lib.check(lib.qst_gemm_nt(g, 2, st))
        ev[2].record()
        if chain is not None:
            _lib.check(lib.qst_ffn_chain(chain[0], chain[1], 0, st))
        ev[3].record()
        torch.cuda.synchronize()
        t_wgrad += ev[0].elapsed_time(ev[1])
        t_ffn1 += ev[1].elapsed_time(ev[2])
        t_chain += ev[2].elapsed_time(ev[3])
    wflops = 2.0 * M * (H * I + I * H + H * H + 3 * H * H)
    # algorithmic HBM bytes per launch (finding 7): every operand once, every output once
    nparam = H * I + I * H + H * H + 3 * H * H
    wbytes = 2.0 * M * (8 * H + 2 * I) + 4.0 * nparam                # every bf16 dY / X row once + the fp32 gradients once
    wflush = 8 * 4.0 * nparam                                        # what the launch really adds into memory: 8 M-ranges of fp32 atomics
    f1bytes = 2.0 * M * H + 2.0 * I * H + 2 * 2.0 * M * I            # A, W, then gelu'(u) and h
    chbytes = 2.0 * M * H + 4.0 * I * H + 4.0 * M * H + M * H * (4 + 2 + 2)     # A, W1+W2, resid; y fp32, y bf16, xhat
    third = None if chain is None else {
        "kernel": "ffn_chain_kernel<0, false> (FFN-1 + GELU + FFN-2 + LayerNorm in one launch; inference forward)",
        "ms": t_chain / reps, "flops_per_launch": 4.0 * M * I * H, "bytes_per_launch": chbytes, "shape": [M, I, H]}
    return ({"kernel": "gemm_tn_group_kernel (all 4 wgrads of one layer: dW2, dW1, dWo, dWqkv + bias grads)",
             "ms": t_wgrad / reps, "flops_per_launch": wflops, "bytes_per_launch": wbytes, "atomic_flush_bytes": wflush,
             "shape": [M, H, I]},
            {"kernel": "gemm_nt_kernel<2, 2, 2> (FFN1 fwd, bias+GELU epilogue)", "ms": t_ffn1 / reps,
             "flops_per_launch": 2.0 * M * I * H, "bytes_per_launch": f1bytes, "shape": [M, I, H]}, third)


def hbm_side(dk):
    """The same launch against the HBM roofline: these kernels' arithmetic intensity (170-290 FLOP/B) is below the chip's
    balance point (2.5 PF / 8 TB/s = 312), so the byte floor is the longer one."""
    gbs = dk["bytes_per_launch"] / (dk["ms"] * 1e-3) / 1e9
    out = {"algorithmic_bytes": int(dk["bytes_per_launch"]), "floor_us": round(dk["bytes_per_launch"] / (PEAK_HBM_GBS * 1e9) * 1e6, 1),
           "achieved_GBps": round(gbs, 1), "peak_GBps": PEAK_HBM_GBS, "frac": round(gbs / PEAK_HBM_GBS, 4),
           "flop_per_byte": round(dk["flops_per_launch"] / dk["bytes_per_launch"], 1)}
    if "atomic_flush_bytes" in dk:
        out["atomic_flush_bytes_not_in_algorithmic"] = int(dk["atomic_flush_bytes"])
    return out


def time_fwd_only(trainer, batches, steps, precision="bf16"):
    """Forward-only throughput (inference forward of the 4 columns as one [4B, L] pass + fused loss, no saved
    activations) -- SURVEY.md section 8d's second figure."""
    import torch
    for i in range(3):
        trainer.forward_loss(*batches[i % len(batches)], precision=precision)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        trainer.forward_loss(*batches[i % len(batches)], precision=precision)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def time_loss_kernel(D, reps=10, rows=262144):
    """HBM evidence for the fused loss kernel at a size where it is not launch-latency-bound:
    rows x D fp32, forward + gradients = 8*rows*D*4 algorithmic bytes."""
    import torch
    from quadruplet_sentence_transformer_amd.encoder import quadruplet_loss_raw
    x = [torch.nn.functional.normalize(torch.randn(rows, D, device="cuda"), dim=1) for _ in range(4)]
    args = (0.6, 1.0, 0.5, 0.5, 2.0, False, 1)
    for _ in range(2):
        quadruplet_loss_raw(*x, *args, want_grads=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = grads = None
    scratch = torch.empty(rows, dtype=torch.float32, device="cuda")
    from quadruplet_sentence_transformer_amd import _lib
    lib = _lib.load()
    out = torch.empty(1, dtype=torch.float32, device="cuda")
    grads = [torch.empty_like(x[0]) for _ in range(4)]
    st = _lib.current_stream_ptr()
    e0.record()
    for _ in range(reps):
        _lib.check(lib.qst_quadruplet_loss(x[0].data_ptr(), x[1].data_ptr(), x[2].data_ptr(), x[3].data_ptr(), rows, D,
                                           *[float(a) for a in args[:5]], 0, 1, out.data_ptr(), None,
                                           *[g_.data_ptr() for g_ in grads], scratch.data_ptr(), st))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = 8.0 * rows * D * 4
    return {"bound": "hbm", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
            "kernel": "quad_loss_kernel fwd+grads", "avg_launch_ms": round(ms, 4), "rows": rows, "D": D}


def baseline_config_name(model, B, L, world, precision="bf16"):
    """Which entry of BASELINE.json `configs` a (model, quadruplets per GPU, seq_len, GPUs) run corresponds to."""
    if model == "all-MiniLM-L6-v2" and L == 128 and B == 64:
        return "BASELINE.json configs[1]" if world == 1 else "BASELINE.json configs[1] per GPU" + (" = configs[3]" if world == 8 else "")
    if model == "all-mpnet-base-v2" and L == 256 and B == 32 and world == 1:
        return "BASELINE.json configs[2]"
    if model == "bert-base-uncased" and L == 384 and B == 128 and world == 1:
        if precision == "fp8":
            return "BASELINE.json configs[4]: fp8 MFMA GEMMs in the forward (MXFP8 weights and activations), bf16 backward"
        return ("BASELINE.json configs[4] shape, trained with bf16 operands (--train-precision fp8 runs the forward on the fp8 "
                "matrix cores)")
    return "not a BASELINE.json configuration"


def distinct_gpus_or_exit(rank, world, dev_index):
    """Ranks of an RCCL job must sit on different physical GPUs. A launcher may pin one GPU per rank (every rank then sees
    ONE device, index 0), so device indices alone cannot tell. Before init_process_group("nccl") every rank publishes
    (device identity, device index, the *_VISIBLE_DEVICES it was started with) over a TCP store; two ranks for which ALL
    THREE are equal are on the same card -- nothing distinguishes them -- and every rank exits non-zero (RCCL would end in a
    duplicate-GPU error or a stall). Ranks a launcher pinned differ in the environment part and are never flagged, whatever
    the identity strings say. The check is advisory: if the store cannot be set up it prints why and lets RCCL decide."""
    import datetime
    import torch
    from torch.distributed import TCPStore
    try:
        props = torch.cuda.get_device_properties(dev_index)
        ident = "|".join(str(getattr(props, k, "")) for k in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id"))
        env = "|".join(os.environ.get(k, "") for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
        me = f"{ident}#{dev_index}#{env}"
        host = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29500")) + 1
        store = TCPStore(host, port, world, rank == 0, timeout=datetime.timedelta(seconds=60))
        store.set(f"gpu{rank}", me)
        recs = [store.get(f"gpu{r}").decode() for r in range(world)]
        # every rank holds the same records and reaches the same verdict below; rank 0 hosts the store and keeps it alive
        # until every rank has read them (a store torn down early left slower ranks "skipping" the check and then blocked
        # in init_process_group while rank 0 had already exited)
        store.set(f"done{rank}", "1")
        if rank == 0:
            for r in range(world):
                store.get(f"done{r}")
    except Exception as ex:                               # a busy port, a torch without TCPStore options, ...
        print(f"bench.py rank {rank}: GPU-identity check skipped ({type(ex).__name__}: {ex})", file=sys.stderr)
        return
    if len(set(recs)) != world:
        print(f"bench.py rank {rank}: {world} ranks but only {len(set(recs))} distinct (GPU, index, visibility) records "
              f"({recs}): RCCL needs one GPU per rank (QST_DIST_BACKEND=gloo rehearses the step on a shared card)", file=sys.stderr)
        raise SystemExit(3)


def time_fp8_training(trainer, cfg, batches, steps, B, ms_bf16_nodrop):
    """BASELINE configs[4] as a training configuration: the full step with the forward's Linears on the fp8 matrix cores
    (MXFP8 weights and activations) and the bf16 backward over what that forward kept; timed without dropout next to the
    bf16 step without dropout (--train-precision fp8 times it as the headline step, dropout on)."""
    import torch
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer
    try:
        trainer.enc.set_dropout(0.0, 0.0)
        tr = QuadrupletTrainer(cfg, encoder=trainer.enc, lr=2e-5, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=10000,
                               total_steps=1000000, precision="fp8")
        for i in range(3):
            tr.step(*batches[i % len(batches)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(*batches[i % len(batches)])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out = {"value": round(B / (ms * 1e-3), 1), "unit": "quadruplets/s", "ms_per_step": round(ms, 4),
               "what": "the training step with every forward Linear on the fp8 matrix cores (QST_PREC_FP8 training forward) "
                       "and the bf16 backward; dropout off"}
        if ms_bf16_nodrop:
            out["speedup_vs_bf16_step_without_dropout"] = round(ms_bf16_nodrop / ms, 3)
            if ms_bf16_nodrop / ms < 1.0:
                out["note"] = ("SLOWER than the bf16 step at this shape: the fp8 forward pays for the bf16 copies the bf16 backward reads, "
                               "and at K = 384 its GEMMs gain nothing (three K stages). A precision mode here, not a speed path.")
        return out
    except Exception as ex:                               # a side figure must not take the bench line down
        return {"error": f"{type(ex).__name__}: {ex}"}


def time_x3_training(trainer, cfg, batches, steps, B):
    """The training step at PARITY precision (QuadrupletTrainer(precision="bf16x3"): fp32 activations, split-bf16 x3 products,
    fp32-class gradients) -- the only configuration that trains inside the north star's rtol 1e-3 / atol 1e-4 of the
    reference's fp32 step (training/main.py:142); dropout off, next to `step_without_dropout`."""
    import torch
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer
    try:
        trainer.enc.set_dropout(0.0, 0.0)
        tr = QuadrupletTrainer(cfg, encoder=trainer.enc, lr=2e-5, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=10000,
                               total_steps=1000000, precision="bf16x3")
        for i in range(2):
            tr.step(*batches[i % len(batches)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(*batches[i % len(batches)])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        return {"value": round(B / (ms * 1e-3), 1), "unit": "quadruplets/s", "ms_per_step": round(ms, 4),
                "what": "the full training step at parity precision (bf16x3: fp32 activations, three split-bf16 MFMAs per product in "
                        "every GEMM and in attention forward and backward, fp32 softmax / LayerNorm / GELU); dropout off"}
    except Exception as ex:                               # a side figure must not take the bench line down
        return {"error": f"{type(ex).__name__}: {ex}"}


def time_f16_training(trainer, cfg, batches, steps, B, dropout, seed, precision="f16"):
    """The training step on IEEE-half operands (QuadrupletTrainer(precision="f16"): the bf16 kernels compiled on f16, dynamic
    loss scale on the device -- the reference's use_amp=True, training/main.py:142) on the headline's shape, with the headline's
    dropout setting and without; the arenas it trains on are the headline trainer's (restored by the caller)."""
    import torch
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer
    out = {}
    try:
        for label, p in (("with_dropout", dropout), ("dropout_off", 0.0)):
            if label == "with_dropout" and not dropout > 0:
                continue
            trainer.enc.set_dropout(p, p, seed)
            tr = QuadrupletTrainer(cfg, encoder=trainer.enc, lr=2e-5, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=10000,
                                   total_steps=1000000, precision=precision)
            for i in range(5):
                tr.step(*batches[i % len(batches)])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                loss = tr.step(*batches[i % len(batches)])
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            out[label] = {"value": round(B / (ms * 1e-3), 1), "unit": "quadruplets/s", "ms_per_step": round(ms, 4),
                          "loss": round(float(loss.item()), 6)}
        sc = trainer.enc.amp_scaler.cpu().tolist()
        out["loss_scale"] = sc[0]
        out["skipped_steps"] = int(sc[3])
        out["what"] = ("the full training step with IEEE-half matrix-core operands (v_mfma_f32_32x32x16_f16: the bf16 step's "
                       "kernels compiled on the other 16-bit type, same bytes) under GradScaler's rules on the device; embeddings "
                       "inside rtol 1e-3 / atol 1e-4 of the fp32 reference where bf16's are not (golden_parity)"
                       + ("; f16w: every forward Linear multiplies by hi + lo of its weight (split-f16 weights, a second pass over K)"
                          if precision == "f16w" else ""))
        return out
    except Exception as ex:                               # a side figure must not take the bench line down
        return {"error": f"{type(ex).__name__}: {ex}"}


class _TrainState:
    """Parameters, Adam moments, step counters and dropout setting of the headline trainer, saved around the side figures that
    train on its arenas (they used to leave a mutated encoder behind)."""

    def __init__(self, trainer, dropout, seed):
        e = trainer.enc
        self.t, self.drop, self.seed = trainer, dropout, seed
        self.params, self.m, self.v = e.params.clone(), e.exp_avg.clone(), e.exp_avg_sq.clone()
        self.opt_step, self.sched_step, self.dstep = e.opt_step, trainer.sched_step, e.dropout_step

    def restore(self):
        e = self.t.enc
        e.params.copy_(self.params); e.exp_avg.copy_(self.m); e.exp_avg_sq.copy_(self.v)
        e.grads.zero_()
        e.opt_step, self.t.sched_step = self.opt_step, self.sched_step
        e.shadow_stale = e.shadow_mx_stale = e.shadow_f16_stale = True
        if e._step_dev is not None:
            e._step_dev.fill_(e.opt_step)
        e.set_dropout(self.drop, self.drop, self.seed) if self.drop > 0 else e.set_dropout(0.0, 0.0)
        e.set_dropout_step(self.dstep)


def time_dp_rccl_ws1(trainer, cfg, batches, steps, B, ms_dp1):
    """The data-parallel step on the ONE GPU a test box has: init_process_group("nccl", world_size=1) and the staged backward
    with its seven asynchronous RCCL all-reduces (one per layer bucket + the embedding bucket) in place -- what every rank of
    configs[3] executes, minus peers. Reports the step time beside the plain single-GPU step."""
    import torch
    import torch.distributed as dist
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer
    try:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev = torch.device(trainer.enc.device)
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        except TypeError:
            dist.init_process_group("nccl", rank=0, world_size=1)
        tr = QuadrupletTrainer(cfg, encoder=trainer.enc, lr=2e-5, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=10000,
                               total_steps=1000000, world_size=1, overlap=True, force_dp=True)
        for i in range(3):
            tr.step(*batches[i % len(batches)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            tr.step(*batches[i % len(batches)])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out = {"value": round(B / ms * 1e3, 1), "unit": "quadruplets/s", "ms_per_step": round(ms, 4),
               "overhead_vs_dp1_ms": round(ms - ms_dp1, 4),
               "what": "the same step through the data-parallel path: staged backward + 7 async RCCL all-reduces "
                       "(world_size 1 on this GPU; the 1 -> 8 GPU curve itself needs an 8-GPU node)"}
        # the other training precisions through the same path (f16: scaled gradients are exchanged, GradScaler decides after the
        # exchange; bf16x3: staged since round 5) -- a handful of steps each, against their own single-process step
        for prec, nst in (("f16", steps), ("bf16x3", max(3, steps // 5))):
            try:
                t1 = QuadrupletTrainer(cfg, encoder=trainer.enc, lr=2e-5, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=10000,
                                       total_steps=1000000, precision=prec)
                t2 = QuadrupletTrainer(cfg, encoder=trainer.enc, lr=2e-5, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=10000,
                                       total_steps=1000000, world_size=1, overlap=True, force_dp=True, precision=prec)
                res = []
                for t in (t1, t2):
                    for i in range(2):
                        t.step(*batches[i % len(batches)])
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(nst):
                        t.step(*batches[i % len(batches)])
                    torch.cuda.synchronize()
                    res.append((time.perf_counter() - t0) / nst * 1e3)
                out[prec] = {"ms_per_step_single": round(res[0], 4), "ms_per_step_dp_ws1": round(res[1], 4),
                             "overhead_ms": round(res[1] - res[0], 4)}
            except Exception as e:
                out[prec] = {"error": f"{type(e).__name__}: {e}"[:200]}
        # the exchange itself on the gradient arena: fp32 as the step does it, and as bf16 (QST_COMM_BF16: half the bytes on the
        # wire, paid for with two conversion passes over the arena and 8-bit gradient sums -- DESIGN.md section 5)
        try:
            g = trainer.enc.grads
            gb = torch.empty_like(g, dtype=torch.bfloat16)
            def t_of(fn, n=10):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n * 1e3
            def x32():
                dist.all_reduce(g)
            def x16():
                gb.copy_(g); dist.all_reduce(gb); g.copy_(gb)
            out["exchange_whole_arena_ms"] = {"fp32": round(t_of(x32), 4), "bf16_with_conversions": round(t_of(x16), 4),
                                              "mb_fp32": round(g.numel() * 4 / 1e6, 1)}
        except Exception as e:
            out["exchange_whole_arena_ms"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        dist.destroy_process_group()
        return out
    except Exception as e:                           # RCCL unavailable on this box: say so, do not fail the bench line
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def main():
    args = parse()
    # The contract is ONE line on stdout. Libraries write there too (RCCL prints a five-line version banner when its first
    # communicator comes up): file descriptor 1 is pointed at stderr for the whole run and the JSON line goes to the saved one.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1 (one process per GPU)")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(1, ndev)          # rehearsal on a 1-GPU box: several ranks share the card (gloo only)
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("QST_DIST_BACKEND", "nccl")      # "nccl" = RCCL over xGMI; "gloo" only to rehearse
        if backend == "nccl":
            distinct_gpus_or_exit(rank, world, dev_index)
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
            except TypeError:                       # a torch without the device_id keyword
                dist.init_process_group("nccl", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif args.force_dp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev_index))
        except TypeError:
            dist.init_process_group("nccl", rank=0, world_size=1)

    from quadruplet_sentence_transformer_amd.config import PRESETS, forward_flops_per_sequence
    from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer

    cfg = PRESETS[args.model]
    B, L = args.batch, args.seq_len
    if args.train_precision not in ("bf16", "f16", "f16w"):
        args.graph = False
    arena = synthetic_params(cfg, seed=14)          # same replica on every rank
    use_graph = bool(args.graph and world == 1 and not args.force_dp)
    trainer = QuadrupletTrainer(cfg, arena=arena, device=f"cuda:{dev_index}", lr=2e-5, weight_decay=0.01,
                                max_grad_norm=1.0, warmup_steps=10000, total_steps=1000000,
                                process_group=None, world_size=world, overlap=not args.no_overlap,
                                use_graph=use_graph, force_dp=args.force_dp,
                                precision=args.train_precision,
                                dropout=(args.dropout if args.dropout > 0 else None), dropout_seed=14 + rank)
    # a few distinct synthetic batches, resident in HBM before the timed region (rank-offset streams)
    nb = 4
    batches = []
    for i in range(nb):
        ids, mask, types = synthetic_quadruplets(cfg, B, L, seed=14, step=i, rank=rank)
        batches.append(tuple(torch.from_numpy(x).cuda() for x in (ids, mask, types)))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    for i in range(args.warmup):
        loss = trainer.step(*batches[i % nb])
    barrier()
    # per-step boundaries as events on the launch stream (no host sync inside the region): the median beside the mean
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = trainer.step(*batches[i % nb])
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ms_median = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    final_loss = float(loss.item())
    # side figure: the same step with dropout off (what rounds 1 measured; eager path only -- a captured graph holds its masks' launches)
    no_drop = None
    if args.dropout > 0 and not args.graph and not args.profile:
        trainer.enc.set_dropout(0.0, 0.0)
        for i in range(3):
            trainer.step(*batches[i % nb])
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            trainer.step(*batches[i % nb])
        barrier()
        dt1 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt1], device="cuda", dtype=torch.float32)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt1 = float(t.item())
        no_drop = {"value": round(B * world * args.steps / dt1, 1), "unit": "quadruplets/s",
                   "ms_per_step": round(dt1 / args.steps * 1e3, 4), "what": "the same training step with dropout off"}
        trainer.enc.set_dropout(args.dropout, args.dropout, 14 + rank)

    out = None
    # every rank runs the kernel timing: its interleaved training steps are collective (gradient all-reduce)
    dk, dk2, dk3 = time_kernels(trainer, 4 * B, L, args.kernel_reps, batches)
    if rank == 0:
        fwd_flops_q = 4.0 * forward_flops_per_sequence(cfg, L)
        train_flops_q = 3.0 * fwd_flops_q
        step_tflops = value * train_flops_q / 1e12
        achieved = dk["flops_per_launch"] / (dk["ms"] * 1e-3) / 1e12
        achieved2 = dk2["flops_per_launch"] / (dk2["ms"] * 1e-3) / 1e12
        # HBM bytes per launch come from a separate rocprofv3 --pmc run (they cannot be collected inside a timed run);
        # the profile file names the kernel source it was measured on, and a figure for other code is not reported
        traffic = traffic2 = None
        traffic_src = None
        import glob
        profs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
        prof = profs[-1] if profs else ""
        prof_name = "profiles/" + os.path.basename(prof)
        if os.path.exists(prof) and args.model == "all-MiniLM-L6-v2" and B == 64 and L == 128:
            try:
                import hashlib
                pj = json.load(open(prof))
                src = os.path.join(ROOT, "quadruplet-sentence-transformer_amd", "csrc", "gemm.hip")
                if pj.get("gemm_hip_sha256") == hashlib.sha256(open(src, "rb").read()).hexdigest():
                    traffic = pj["gemm_tn_group_kernel"]["hbm_bytes_per_launch"]
                    traffic2 = pj.get("gemm_nt_kernel<2>_hbm_bytes_per_launch")
                    traffic_src = prof_name + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, same gemm.hip)"
                else:
                    traffic_src = "stale: " + prof_name + " was measured on a different gemm.hip"
            except Exception:
                traffic = traffic2 = None
        out = {
            "metric": f"quadruplets/sec (seq_len={L}, {args.model}) training step", "value": round(value, 1),
            "unit": "quadruplets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "ms_per_step_median": round(ms_median, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "f16": "f16", "f16w": "f16 (split-f16 weights in the forward)", "fp8": "fp8 (MXFP8 forward GEMMs, bf16 backward)", "bf16x3": "bf16x3 (fp32-class)"}[args.train_precision],
            "data": "synthetic",
            "config": {"workload": f"{args.model} dims (random-init), {B} quadruplets/GPU x {world} GPU, seq_len={L}, "
                                   "fwd + gamma-quadruplet loss + bwd + clip + AdamW, "
                                   + (f"dropout {args.dropout:g} on hidden states and attention probabilities as the reference's "
                                      "train() mode" if args.dropout > 0 else "dropout off") + f" ({baseline_config_name(args.model, B, L, world, args.train_precision)})",
                       "global_batch": B * world, "seq_len": L,
                       "parallelism": f"dp{world}" + (" through RCCL (world_size 1: staged backward + 7 async all-reduces)" if args.force_dp else ""),
                       "precision": {"bf16": "bf16 MFMA operands, fp32 accumulate/residual/LN/softmax/loss/optimizer",
                                     "f16": "IEEE-half MFMA operands under a device-side dynamic loss scale, fp32 accumulate/residual/LN/softmax/loss/optimizer",
                                     "f16w": "as f16, every forward Linear against hi + lo of its weight (two f16 values per weight)",
                                     "fp8": "forward Linears on the fp8 matrix cores (MXFP8 weights and activations), bf16 attention "
                                            "and backward, fp32 accumulate/residual/LN/softmax/loss/optimizer",
                                     "bf16x3": "split-bf16 x3 MFMA products on fp32 operands (fp32-class), fp32 everywhere else"}[args.train_precision],
                       "launch": "hip graph replay" if use_graph else "eager"},
            "loss": round(final_loss, 6),
            "step_without_dropout": no_drop,
            "step_tflops": round(step_tflops, 2),
            "step_mfma_frac": round(step_tflops / (PEAK_BF16_TFLOPS * world), 4),
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": dk["kernel"], "avg_launch_ms": round(dk["ms"], 5), "shape_M_H_I": dk["shape"],
                         "hbm_side": hbm_side(dk)},
            "roofline_ffn1_fwd": {"bound": "mfma", "achieved": round(achieved2, 2), "peak": PEAK_BF16_TFLOPS,
                                  "unit": "TFLOP/s", "frac": round(achieved2 / PEAK_BF16_TFLOPS, 4), "traffic": traffic2,
                                  "kernel": dk2["kernel"], "avg_launch_ms": round(dk2["ms"], 5), "shape_MNK": dk2["shape"],
                                  "hbm_side": hbm_side(dk2)},
        }
        if dk3 is not None:
            a3 = dk3["flops_per_launch"] / (dk3["ms"] * 1e-3) / 1e12
            out["roofline_ffn_inference"] = {"bound": "mfma", "achieved": round(a3, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                             "frac": round(a3 / PEAK_BF16_TFLOPS, 4), "traffic": None, "kernel": dk3["kernel"],
                                             "avg_launch_ms": round(dk3["ms"], 5), "shape_MNK": dk3["shape"]}
        if world == 1 and not args.no_extras:
            snap = _TrainState(trainer, args.dropout, 14 + rank)
            t_f = time_fwd_only(trainer, batches, max(5, args.steps // 2))
            out["fwd_only"] = {"value": round(B / t_f, 1), "unit": "quadruplets/s", "ms_per_step": round(t_f * 1e3, 4),
                               "what": "encode 4 columns + loss forward, no backward / saved activations",
                               "mfma_frac": round(B / t_f * fwd_flops_q / 1e12 / PEAK_BF16_TFLOPS, 4)}
            if cfg.hidden_size % 128 == 0 and cfg.intermediate_size % 128 == 0:
                t_m = time_fwd_only(trainer, batches, max(5, args.steps // 2), precision="fp8")
                out["fwd_only_fp8"] = {"value": round(B / t_m, 1), "unit": "quadruplets/s", "ms_per_step": round(t_m * 1e3, 4),
                                       "speedup_vs_bf16": round(t_f / t_m, 3),
                                       **({"note": "no faster than bf16 at this shape (K = 384: three K stages per GEMM); the fp8 matrix "
                                                   "cores pay from K >= 768 (configs[4]: 1.2x)"} if t_f / t_m < 1.03 else {}),
                                       "what": "same on the fp8 matrix cores: MXFP8 weights AND activations, block-scaled MFMA "
                                               "(QST_PREC_FP8, inference; BASELINE configs[4])",
                                       "mfma_frac_of_fp8_peak": round(B / t_m * fwd_flops_q / 1e12 / (2 * PEAK_BF16_TFLOPS), 4)}
                out["train_step_fp8_forward"] = time_fp8_training(trainer, cfg, batches, max(5, args.steps // 2), B,
                                                                  no_drop["ms_per_step"] if no_drop else None)
            t_h = time_fwd_only(trainer, batches, max(5, args.steps // 2), precision="f16")
            out["fwd_only_f16"] = {"value": round(B / t_h, 1), "unit": "quadruplets/s", "ms_per_step": round(t_h * 1e3, 4),
                                   "speedup_vs_bf16": round(t_f / t_h, 3),
                                   "what": "same on IEEE-half operands (QST_PREC_F16: the bf16 kernels compiled on f16)"}
            out["train_step_f16"] = time_f16_training(trainer, cfg, batches, max(5, args.steps // 2), B, args.dropout, 14 + rank)
            snap.restore()
            t_w = time_fwd_only(trainer, batches, max(5, args.steps // 2), precision="f16w")
            out["fwd_only_f16w"] = {"value": round(B / t_w, 1), "unit": "quadruplets/s", "ms_per_step": round(t_w * 1e3, 4),
                                    "speedup_vs_bf16": round(t_f / t_w, 3),
                                    "what": "same with split-f16 weights in every Linear (QST_PREC_F16W)"}
            out["train_step_f16w"] = time_f16_training(trainer, cfg, batches, max(5, args.steps // 2), B, args.dropout, 14 + rank,
                                                       precision="f16w")
            snap.restore()
            t_3 = time_fwd_only(trainer, batches, max(5, args.steps // 2), precision="bf16x3")
            out["fwd_only_bf16x3"] = {"value": round(B / t_3, 1), "unit": "quadruplets/s", "ms_per_step": round(t_3 * 1e3, 4),
                                      "what": "same, parity precision (split-bf16 x3 MFMA, fp32 activations): the "
                                              "configuration that meets rtol 1e-3 / atol 1e-4 on embeddings"}
            out["train_step_bf16x3"] = time_x3_training(trainer, cfg, batches, max(3, args.steps // 5), B)
            snap.restore()
            out["golden_parity"] = golden_parity(os.path.join(ROOT, "tests", "golden", "encoder_golden.npz"))
            ms_ref = no_drop["ms_per_step"] if no_drop else ms_per_step           # (the RCCL rehearsal runs with dropout off)
            if args.dropout > 0 and not args.graph:
                trainer.enc.set_dropout(0.0, 0.0)
            if not args.force_dp:                       # (with --force-dp the headline step itself is that path)
                out["dp_rccl_ws1"] = time_dp_rccl_ws1(trainer, cfg, batches, max(5, args.steps // 2), B, ms_ref)
            if args.dropout > 0 and not args.graph:
                trainer.enc.set_dropout(args.dropout, args.dropout, 14 + rank)
            out["roofline_loss_kernel"] = time_loss_kernel(cfg.hidden_size)
        if world == 1 and not args.no_cpu_baseline:
            cpu_losses = None
            c1cfg = PRESETS["all-MiniLM-L6-v2"]
            c1arena = arena if args.model == "all-MiniLM-L6-v2" else synthetic_params(c1cfg, seed=14)
            out["cpu_baseline"], cpu_losses = cpu_baseline(c1cfg, c1arena, L, B)
            out["config1_hip"] = config1_on_gpu(c1cfg, c1arena, cpu_losses)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
    if world > 1 or args.force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
