"""GPU: several full optimisation steps (forward, quadruplet loss, backward, clip_grad_norm_, AdamW with ST's two
parameter groups, WarmupLinear schedule) against the CPU oracle running torch autograd + torch.optim.AdamW -- the
loop SentenceTransformer.fit runs for the reference (/root/reference/training/main.py:128-148)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd.config import PRESETS, build_layout, hf_param_views  # noqa: E402
from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets  # noqa: E402
from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer, warmup_linear_lr  # noqa: E402
from oracle import torch_ref as R  # noqa: E402

LOSS_KW = dict(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5, p=2.0, swap=False)


@pytest.mark.parametrize("name", ["tiny-bert", "tiny-mpnet"])
def test_training_steps_track_the_cpu_reference(name):
    cfg = PRESETS[name]
    B, L, steps, lr, warmup, total = 6, 32, 6, 2e-3, 2, 20
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    # CPU reference: leaf tensors per segment, decay groups by the segment's decay flag (== ST's name filter)
    P = R.arena_to_dict(arena, cfg, requires_grad=True)
    segs, _ = build_layout(cfg)
    groups = [{"params": [P[s.name] for s in segs if s.decay], "weight_decay": 0.01},
              {"params": [P[s.name] for s in segs if not s.decay], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.999), eps=1e-8)
    tr = QuadrupletTrainer(cfg, arena=arena, device="cuda:0", lr=lr, weight_decay=0.01, max_grad_norm=1.0,
                           warmup_steps=warmup, total_steps=total, **LOSS_KW)
    ref_losses, hip_losses = [], []
    for step in range(steps):
        # the same batch every step: random quadruplets carry no signal, so only over-fitting one batch moves the loss
        ids, mask, types = synthetic_quadruplets(cfg, B, L, seed=14, ragged=True, step=0)
        t = [torch.from_numpy(x) for x in (ids, mask, types)]
        for g in opt.param_groups:
            g["lr"] = warmup_linear_lr(lr, step, warmup, total)
        opt.zero_grad()
        loss, _ = R.quadruplet_step(P, cfg, *t, LOSS_KW, bf16_operands=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"]], 1.0)
        opt.step()
        ref_losses.append(loss.item())
        hip_losses.append(tr.step(*[x.cuda() for x in t]).item())
    ref_losses, hip_losses = np.array(ref_losses), np.array(hip_losses)
    assert ref_losses[-1] < ref_losses[0] - 0.05, "reference did not train"
    np.testing.assert_allclose(hip_losses, ref_losses, rtol=0, atol=3e-3)
    assert abs(hip_losses[0] - ref_losses[0]) < 2e-4
    # parameters after the steps: Adam normalises update magnitudes, so compare against how far training moved them
    got = tr.enc.params.cpu().numpy()
    worst = 0.0
    for s in segs:
        a = got[s.offset:s.offset + s.numel]
        b = P[s.name].detach().numpy().reshape(-1)
        w0 = arena[s.offset:s.offset + s.numel]
        if s.name.endswith("b_qkv"):
            # the key bias has a mathematically zero gradient (softmax shift invariance): Adam turns its rounding
            # noise into +-lr steps of random sign in BOTH implementations, so only the q and v thirds are comparable
            H = cfg.hidden_size
            keep = np.r_[0:H, 2 * H:3 * H]
            a, b, w0 = a[keep], b[keep], w0[keep]
        moved = np.abs(b - w0).mean()
        err = np.abs(a - b).mean()
        worst = max(worst, err / max(moved, 1e-12))
        assert err <= 0.15 * moved + 1e-7, f"{s.name}: mean |diff| {err:.3e} vs mean |update| {moved:.3e}"
    assert worst > 0   # something was compared


@pytest.mark.parametrize("name", ["tiny-bert", "tiny-mpnet"])
def test_graph_replayed_steps_match_eager_steps(name):
    """use_graph=True: the step is captured once per shape into a HIP graph and replayed, with the WarmupLinear
    schedule and the Adam step counter on the device (qst_clip_adamw_step_sched). Losses and parameters must follow
    the eager trainer (whose schedule is computed on the host) over warm-up and decay, across two batch shapes."""
    cfg = PRESETS[name]
    lr, warmup, total = 2e-3, 3, 12
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    kw = dict(arena=arena, device="cuda:0", lr=lr, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=warmup,
              total_steps=total, **LOSS_KW)
    eager = QuadrupletTrainer(cfg, **kw)
    graph = QuadrupletTrainer(cfg, use_graph=True, **kw)
    shapes = [(6, 32), (4, 64)]
    le, lg = [], []
    for step in range(10):
        B, L = shapes[step % 2]
        ids, mask, types = synthetic_quadruplets(cfg, B, L, seed=14, ragged=True, step=step % 4)
        t = [torch.from_numpy(x).cuda() for x in (ids, mask, types)]
        le.append(eager.step(*t).item())
        lg.append(graph.step(*t).item())          # .item() before the next replay overwrites the static loss
    assert len(graph._graphs) == 2
    np.testing.assert_allclose(lg, le, rtol=0, atol=2e-3)
    assert abs(lg[0] - le[0]) < 1e-6
    assert int(graph.enc._step_dev.item()) == 10 and graph.enc.opt_step == 10
    pe, pg, p0 = eager.enc.params.cpu().numpy(), graph.enc.params.cpu().numpy(), np.asarray(arena)
    moved = np.abs(pe - p0).mean()
    assert moved > 0
    # float atomics in the weight-gradient kernel make two runs of the SAME code differ at rounding level, and Adam
    # amplifies that on near-zero gradients; the two trainers must agree far inside the distance training moved them
    assert np.abs(pg - pe).mean() < 0.05 * moved
    with pytest.raises(ValueError):
        QuadrupletTrainer(cfg, use_graph=True, world_size=2, **kw)


@pytest.mark.parametrize("name,drop", [("tiny-bert", None), ("tiny-mpnet", None), ("tiny-bert", 0.1), ("tiny-mpnet", 0.1)])
def test_parity_precision_training_tracks_the_fp32_reference(name, drop):
    """QuadrupletTrainer(precision="bf16x3"): six optimisation steps on the split-bf16 x3 path against the fp32 oracle
    (torch autograd + torch.optim.AdamW, no bf16 emulation) -- the loss trajectory within 1e-4 of the reference's at every
    step (the bf16 path is held to 3e-3) and the parameters within 2% of the distance training moved them. drop = 0.1: the same
    in train() mode, as the reference's fit() runs (fp32, HF dropout 0.1 / 0.1) -- the oracle takes step k's masks from
    oracle/dropout_ref.py."""
    cfg = PRESETS[name]
    B, L, steps, lr, warmup, total = 6, 32, 6, 2e-3, 2, 20
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    P = R.arena_to_dict(arena, cfg, requires_grad=True)
    segs, _ = build_layout(cfg)
    groups = [{"params": [P[s.name] for s in segs if s.decay], "weight_decay": 0.01},
              {"params": [P[s.name] for s in segs if not s.decay], "weight_decay": 0.0}]
    opt = torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.999), eps=1e-8)
    tr = QuadrupletTrainer(cfg, arena=arena, device="cuda:0", lr=lr, weight_decay=0.01, max_grad_norm=1.0,
                           warmup_steps=warmup, total_steps=total, precision="bf16x3", dropout=drop, dropout_seed=31, **LOSS_KW)
    ref_losses, hip_losses = [], []
    for step in range(steps):
        ids, mask, types = synthetic_quadruplets(cfg, B, L, seed=14, ragged=True, step=0)
        t = [torch.from_numpy(x) for x in (ids, mask, types)]
        for g in opt.param_groups:
            g["lr"] = warmup_linear_lr(lr, step, warmup, total)
        opt.zero_grad()
        masks = None
        if drop:
            from oracle.dropout_ref import Masks
            masks = Masks(31, step + 1, drop, drop)
        loss, _ = R.quadruplet_step(P, cfg, *t, LOSS_KW, bf16_operands=False, dropout=masks)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for g in opt.param_groups for p in g["params"]], 1.0)
        opt.step()
        ref_losses.append(loss.item())
        hip_losses.append(tr.step(*[x.cuda() for x in t]).item())
    ref_losses, hip_losses = np.array(ref_losses), np.array(hip_losses)
    assert ref_losses[-1] < ref_losses[0] - 0.03, "reference did not train"
    np.testing.assert_allclose(hip_losses, ref_losses, rtol=0, atol=1e-4)
    got = tr.enc.params.cpu().numpy()
    for s in segs:
        a = got[s.offset:s.offset + s.numel]
        b = P[s.name].detach().numpy().reshape(-1)
        w0 = arena[s.offset:s.offset + s.numel]
        if s.name.endswith("b_qkv"):
            H = cfg.hidden_size
            keep = np.r_[0:H, 2 * H:3 * H]          # (the key third: a zero gradient, Adam steps of random sign in both)
            a, b, w0 = a[keep], b[keep], w0[keep]
        moved = np.abs(b - w0).mean()
        err = np.abs(a - b).mean()
        assert err <= 0.02 * moved + 1e-7, f"{s.name}: mean |diff| {err:.3e} vs mean |update| {moved:.3e}"
