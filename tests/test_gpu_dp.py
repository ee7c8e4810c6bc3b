"""GPU, 2 ranks sharing the one test card (gloo transport: RCCL needs one GPU per rank): the data-parallel step
(staged backward + per-layer all-reduce + clip + AdamW with grad_scale = 1/world) must leave every rank with the same
parameters as ONE rank stepping on the concatenated batch -- quadruplets are independent and the loss is a mean."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

import quadruplet_sentence_transformer_amd  # noqa: E402,F401


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dp_case(name):
    """(config, quadruplets per rank, seq_len). "minilm-c4": the code path BASELINE configs[3] runs on every rank --
    MiniLM layer dimensions with M = 4*32*128 = 16384 token rows per rank, i.e. the fused GEMM+LayerNorm kernels (forward
    and backward, whose gamma/beta partials of layer l-1 are produced during layer l's stage), the single-workgroup
    attention backward and the 8-range grouped wgrad -- on 2 layers and a small vocabulary to keep the test short."""
    from dataclasses import replace
    from quadruplet_sentence_transformer_amd.config import PRESETS
    if name == "minilm-c4":
        return replace(PRESETS["all-MiniLM-L6-v2"], num_layers=2, vocab_size=4096), 32, 128
    if name == "h768-fused":      # bert-base layer dims, the GEMM + LayerNorm launches of configs[4] forced (set_ln_fusion(1)) on 1,536 rows
        return replace(PRESETS["bert-base-uncased"], num_layers=2, vocab_size=4096), 1, 384
    return PRESETS[name], 4, 32


def _worker(rank, world, port, overlap, out_dir, case="tiny-bert", dropout=None, precision="bf16"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer
    torch.cuda.set_device(0)
    cfg, B, L = _dp_case(case)
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    tr = QuadrupletTrainer(cfg, arena=arena, device="cuda:0", lr=1e-3, world_size=world, overlap=overlap,
                           dropout=dropout, dropout_seed=100 + rank, precision=precision)   # every rank its own mask stream
    for step in range(2):
        ids, mask, types = synthetic_quadruplets(cfg, world * B, L, seed=14, ragged=True, step=step)
        sl = slice(rank * B, (rank + 1) * B)                  # this rank's shard of the global batch
        tr.step(*[torch.from_numpy(x[:, sl].copy()).cuda() for x in (ids, mask, types)])
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"params_{rank}.npy"), tr.enc.params.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap,case", [(True, "tiny-bert"), (False, "tiny-bert"), (True, "minilm-c4")])
def test_two_rank_step_equals_single_rank_on_global_batch(tmp_path, overlap, case):
    from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), overlap, str(tmp_path), case), nprocs=world, join=True)
    p0 = np.load(tmp_path / "params_0.npy")
    p1 = np.load(tmp_path / "params_1.npy")
    np.testing.assert_array_equal(p0, p1)                   # replicas stay bit-identical
    cfg, B, L = _dp_case(case)
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    tr = QuadrupletTrainer(cfg, arena=arena, device="cuda:0", lr=1e-3, world_size=1)
    for step in range(2):
        ids, mask, types = synthetic_quadruplets(cfg, world * B, L, seed=14, ragged=True, step=step)
        tr.step(*[torch.from_numpy(x).cuda() for x in (ids, mask, types)])
    ref = tr.enc.params.cpu().numpy()
    moved = np.abs(ref - arena).max()
    assert moved > 1e-4
    # same math up to fp32 summation order (atomics, different tile shapes for M = 4B*L vs 8B*L) and Adam's 1/sqrt(v)
    # Parameters whose gradient is mathematically zero (attention key biases) get an Adam update of +-lr whose sign is
    # rounding noise, so a handful of elements may differ by up to 2*lr; everything else must agree closely.
    bad = np.abs(p0 - ref) > 0.05 * moved
    assert bad.mean() < 1e-3, f"{bad.sum()} of {bad.size} parameters differ"
    assert np.abs(p0 - ref).max() <= 2.1e-3 and np.abs(p0 - ref).mean() < 2e-3 * moved


def test_parity_precision_trains_data_parallel(tmp_path):
    """precision="bf16x3" (the path that meets the north-star tolerance) under two ranks: replicas bit-identical, and equal to
    the single-process bf16x3 step on the global batch up to fp32 summation order -- its gradients are fp32-class, so the
    agreement is two orders tighter than the bf16 path's."""
    from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), True, str(tmp_path), "tiny-bert", None, "bf16x3"), nprocs=world, join=True)
    p0, p1 = np.load(tmp_path / "params_0.npy"), np.load(tmp_path / "params_1.npy")
    np.testing.assert_array_equal(p0, p1)
    cfg, B, L = _dp_case("tiny-bert")
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    tr = QuadrupletTrainer(cfg, arena=arena, device="cuda:0", lr=1e-3, world_size=1, precision="bf16x3")
    for step in range(2):
        ids, mask, types = synthetic_quadruplets(cfg, world * B, L, seed=14, ragged=True, step=step)
        tr.step(*[torch.from_numpy(x).cuda() for x in (ids, mask, types)])
    ref = tr.enc.params.cpu().numpy()
    moved = np.abs(ref - arena).max()
    assert moved > 1e-4
    bad = np.abs(p0 - ref) > 0.02 * moved            # (zero-gradient parameters get an Adam update whose sign is rounding noise)
    assert bad.mean() < 1e-3, f"{bad.sum()} of {bad.size} parameters differ"
    assert np.abs(p0 - ref).mean() < 2e-4 * moved


@pytest.mark.parametrize("prec", ["f16", "f16w"])
def test_f16_trains_data_parallel_under_one_loss_scale(tmp_path, prec):
    """precision="f16" / "f16w" under two ranks (round 5): the staged backward with its overlapped all-reduces runs on the
    SCALED gradients, every rank takes GradScaler's decision from the same reduced norm (qst_clip_adamw_step_amp after the
    exchange), so replicas stay bit-identical -- parameters and loss scale -- and equal the single-process step on the global
    batch up to fp32 summation order."""
    from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), True, str(tmp_path), "tiny-bert", None, prec), nprocs=world, join=True)
    p0, p1 = np.load(tmp_path / "params_0.npy"), np.load(tmp_path / "params_1.npy")
    np.testing.assert_array_equal(p0, p1)
    cfg, B, L = _dp_case("tiny-bert")
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    tr = QuadrupletTrainer(cfg, arena=arena, device="cuda:0", lr=1e-3, world_size=1, precision=prec)
    for step in range(2):
        ids, mask, types = synthetic_quadruplets(cfg, world * B, L, seed=14, ragged=True, step=step)
        tr.step(*[torch.from_numpy(x).cuda() for x in (ids, mask, types)])
    assert tr.enc.amp_scaler[3].item() == 0.0
    ref = tr.enc.params.cpu().numpy()
    moved = np.abs(ref - arena).max()
    assert moved > 1e-4
    bad = np.abs(p0 - ref) > 0.05 * moved            # (zero-gradient parameters get an Adam update whose sign is rounding noise)
    assert bad.mean() < 1e-3, f"{bad.sum()} of {bad.size} parameters differ"
    assert np.abs(p0 - ref).mean() < 1e-3 * moved


def test_replicas_stay_identical_with_per_rank_dropout(tmp_path):
    """With dropout every rank draws its own masks (seed + rank), so the ranks' local gradients differ by more than their
    data -- the reduced gradient, the global-norm clip and the update are still the same everywhere: replicas stay
    bit-identical, and the masks did something (the result differs from the dropout-free run)."""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), True, str(tmp_path), "tiny-bert", 0.1), nprocs=world, join=True)
    p0, p1 = np.load(tmp_path / "params_0.npy"), np.load(tmp_path / "params_1.npy")
    np.testing.assert_array_equal(p0, p1)
    assert np.isfinite(p0).all()
    mp.spawn(_worker, args=(world, _free_port(), True, str(tmp_path), "tiny-bert", None), nprocs=world, join=True)
    assert np.abs(np.load(tmp_path / "params_0.npy") - p0).max() > 1e-5


def _grads_after(cfg, arena, batch, mode, group=None):
    """Gradient arena of one forward + loss + backward on `batch`: mode "oneshot" = qst_encoder_backward in one call,
    "staged" = trainer.staged_backward's call sequence (per-layer stages, layer 0 without its weight gradients, the
    embedding stage, then the postponed weight-gradient launch) with a one-member process group standing in for the
    exchange."""
    from quadruplet_sentence_transformer_amd.encoder import stacked
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer, gradient_buckets, staged_backward
    tr = QuadrupletTrainer(cfg, arena=arena, device="cuda:0")
    enc = tr.enc
    if cfg.hidden_size == 768:
        enc.set_ln_fusion(1)
    enc.grads.zero_()
    loss, _, g, saved, (ids, mask, types) = tr.forward_loss(*batch, training=True, want_grads=True)
    if mode == "oneshot":
        enc.backward(ids, mask, types, stacked(g), saved)
    else:
        for w in staged_backward(enc, ids, mask, types, stacked(g), saved, None, gradient_buckets(cfg), group, True):
            w.wait()
    torch.cuda.synchronize()
    return enc.grads.clone(), float(loss.item())


def _one_rank_group_worker(rank, world, port, out_dir, backend="gloo", case="minilm-c4"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":                     # = RCCL on ROCm: the transport configs[3] runs on
        torch.cuda.set_device(0)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        except TypeError:
            dist.init_process_group("nccl", rank=rank, world_size=world)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets
    torch.cuda.set_device(0)
    cfg, B, L = _dp_case(case)
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    batch = [torch.from_numpy(x).cuda() for x in synthetic_quadruplets(cfg, B, L, seed=14, ragged=True)]
    g1, l1 = _grads_after(cfg, arena, batch, "oneshot")
    g2, l2 = _grads_after(cfg, arena, batch, "staged")
    np.save(os.path.join(out_dir, "g_oneshot.npy"), g1.cpu().numpy())
    np.save(os.path.join(out_dir, "g_staged.npy"), g2.cpu().numpy())
    np.save(os.path.join(out_dir, "loss.npy"), np.array([l1, l2]))
    dist.destroy_process_group()


@pytest.mark.parametrize("backend,case", [("gloo", "minilm-c4"), ("nccl", "minilm-c4"), ("nccl", "h768-fused")])
def test_staged_backward_matches_one_call_on_the_fused_path(tmp_path, backend, case):
    """backend "nccl": the same through REAL RCCL works on the one GPU a test box has (world_size 1): dist.all_reduce(
    async_op=True) on RCCL's own stream, ordered against the kernels the C-ABI enqueues on torch's current stream, waited
    before the comparison -- the mechanism of configs[3], which gloo (a blocking host-staged reduce) never exercises.
    The staged call order configs[3] relies on -- layer l's stage writes layer l-1's LayerNorm-2 gamma/beta partials and
    reduces them in that call; layer 0 runs without its weight gradients, the embedding stage follows, the postponed
    grouped wgrad comes last -- must produce the gradients of the single call. fp32 atomics make the weight gradients
    order-dependent in the last bits, so equality is per-tensor relative L2 < 1e-5 (two one-call runs differ as much)."""
    from quadruplet_sentence_transformer_amd.config import build_layout
    mp.spawn(_one_rank_group_worker, args=(1, _free_port(), str(tmp_path), backend, case), nprocs=1, join=True)
    g1, g2 = np.load(tmp_path / "g_oneshot.npy"), np.load(tmp_path / "g_staged.npy")
    l = np.load(tmp_path / "loss.npy")
    assert l[0] == l[1] and np.isfinite(g1).all() and np.isfinite(g2).all()
    cfg, _, _ = _dp_case(case)
    segs, _ = build_layout(cfg)
    gnorm = float(np.linalg.norm(g1))
    for s in segs:
        a, b = g1[s.offset:s.offset + s.numel], g2[s.offset:s.offset + s.numel]
        na = float(np.linalg.norm(a))
        assert na > 0 or "b_qkv" in s.name, s.name
        if na <= 1e-6 * gnorm:          # zero by symmetry (bare bert-base: the last LayerNorm's beta shifts every embedding alike): rounding noise
            assert float(np.linalg.norm(b)) <= 1e-5 * gnorm, s.name
            continue
        assert float(np.linalg.norm(a - b)) <= 1e-5 * max(na, 1e-12) + 1e-9, s.name


def test_native_communicator_behind_the_c_abi():
    """include/qst.h qst_comm_unique_id / qst_comm_init / qst_allreduce_bucket (csrc/comm.hip: RCCL bound at run time) on the
    one GPU a test box has, world size 1: (a) an in-place sum all-reduce of fp32 and bf16 buffers on the communicator's own
    stream, ordered against the compute stream by events, leaves the data as it was -- after a kernel that was still
    writing it when the collective was enqueued; (b) the staged backward at the c4 shape with NativeComm as its group
    equals the one-shot backward; (c) QuadrupletTrainer steps through it track the single-process steps."""
    from quadruplet_sentence_transformer_amd.comm import NativeComm
    from quadruplet_sentence_transformer_amd.config import build_layout
    from quadruplet_sentence_transformer_amd.synthetic import synthetic_params, synthetic_quadruplets
    from quadruplet_sentence_transformer_amd.trainer import QuadrupletTrainer
    torch.cuda.set_device(0)
    comm = NativeComm(0, 1, NativeComm.unique_id())
    assert comm.lib.qst_comm_rank(comm.handle) == 0 and comm.lib.qst_comm_world(comm.handle) == 1
    x = torch.zeros(1 << 22, device="cuda")
    for dt in (torch.float32, torch.bfloat16):
        y = torch.zeros(1 << 22, device="cuda", dtype=dt)
        y += 3                                               # still running when the collective is enqueued
        w = comm.all_reduce(y, async_op=True)
        w.wait()
        assert float(y.float().min()) == 3.0 and float(y.float().max()) == 3.0
    assert comm.lib.qst_allreduce_bucket(comm.handle, x.data_ptr(), 4, 7, 0) == -1          # unknown dtype
    cfg, B, L = _dp_case("minilm-c4")
    arena = synthetic_params(cfg, seed=14, std=0.05, bias_std=0.02, ln_jitter=0.05)
    batch = [torch.from_numpy(t).cuda() for t in synthetic_quadruplets(cfg, B, L, seed=14, ragged=True)]
    g1, l1 = _grads_after(cfg, arena, batch, "oneshot")
    g2, l2 = _grads_after(cfg, arena, batch, "staged", group=comm)
    assert l1 == l2
    segs, _ = build_layout(cfg)
    for s_ in segs:
        a, b = g1[s_.offset:s_.offset + s_.numel], g2[s_.offset:s_.offset + s_.numel]
        assert float((a - b).norm()) <= 1e-5 * max(float(a.norm()), 1e-12) + 1e-9, s_.name
    tcfg = _dp_case("tiny-bert")[0]
    tarena = synthetic_params(tcfg, seed=3, std=0.05)
    tb = [torch.from_numpy(t).cuda() for t in synthetic_quadruplets(tcfg, 6, 32, seed=3, ragged=True)]
    kw = dict(arena=tarena, device="cuda:0", lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    t_ref, t_dp = QuadrupletTrainer(tcfg, **kw), QuadrupletTrainer(tcfg, process_group=comm, world_size=1, force_dp=True, **kw)
    for _ in range(3):
        la, lb = t_ref.step(*tb), t_dp.step(*tb)
    torch.cuda.synchronize()
    # (AdamW's first updates are lr * g / |g|: the staged path's other atomic order flips that for gradients near zero,
    #  so parameters agree to a few lr, not to the last bit)
    assert abs(float(la) - float(lb)) < 1e-3
    diff = (t_dp.enc.params - t_ref.enc.params).abs()
    assert float(diff.max()) <= 6.5e-3 and float(diff.mean()) < 1e-4
    # (d) the same through the fp8 training forward in train() mode (dropout on): the staged stages run on the handle whose
    # forward filled the arena, so both trainers rebuild the same masks (tiny-mpnet: H = 128, I = 256 suit the fp8 stages)
    mcfg = _dp_case("tiny-mpnet")[0]
    marena = synthetic_params(mcfg, seed=3, std=0.05)
    mb = [torch.from_numpy(t).cuda() for t in synthetic_quadruplets(mcfg, 6, 32, seed=3, ragged=True)]
    kw = dict(arena=marena, device="cuda:0", lr=1e-3, weight_decay=0.01, max_grad_norm=1.0, precision="fp8", dropout=0.1,
              dropout_seed=9)
    f_ref, f_dp = QuadrupletTrainer(mcfg, **kw), QuadrupletTrainer(mcfg, process_group=comm, world_size=1, force_dp=True, **kw)
    for _ in range(3):
        la, lb = f_ref.step(*mb), f_dp.step(*mb)
    torch.cuda.synchronize()
    assert abs(float(la) - float(lb)) < 2e-3
    diff = (f_dp.enc.params - f_ref.enc.params).abs()
    assert float(diff.max()) <= 6.5e-3 and float(diff.mean()) < 1e-4
    comm.close()


class _FileWritingEvaluator:
    """Writes under output_path unconditionally, as the reference's QuadrupletLossEvaluator does
    (/root/reference/models/evaluators.py:81: os.path.join(output_path, "_quadruplet_loss_eval.json"))."""

    def __call__(self, model, output_path=None, epoch=-1, steps=-1):
        import json
        with open(os.path.join(output_path, "_quadruplet_loss_eval.json"), "a") as f:
            json.dump({"epoch": epoch, "steps": steps}, f)
        return float(-steps)


def _fit_worker(rank, world, port, out_dir, n_examples=24, evaluator=None, tag=""):
    """The reference's training call (training/main.py:128-148) on the drop-in, one process per rank."""
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch.utils.data import DataLoader
    from quadruplet_sentence_transformer_amd.losses import GammaQuadrupletLoss
    from quadruplet_sentence_transformer_amd.quadruplet_model import QuadrupletSentenceTransformerLossModel
    from quadruplet_sentence_transformer_amd.sentence_transformer import InputExample, SentenceTransformer
    torch.cuda.set_device(0)
    words = "a man rides red horse two dogs play in park woman eats green apple near old bridge small cat sleeps".split()
    rng = np.random.RandomState(3)
    examples = [InputExample(texts=[" ".join(rng.choice(words, size=rng.randint(3, 9))) for _ in range(4)])
                for _ in range(n_examples)]
    model = SentenceTransformer("tiny-bert", device="cuda:0")
    loss = GammaQuadrupletLoss(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5)
    lm = QuadrupletSentenceTransformerLossModel(st_model=model, quadruplet_loss=loss)
    dl = DataLoader(examples, batch_size=6, shuffle=False)             # 6 rows over 2 ranks: 3 + 3; over 4: 2+2+1+1
    model.fit(train_objectives=[(dl, lm)], evaluator=evaluator, epochs=1, steps_per_epoch=None, scheduler="WarmupLinear",
              warmup_steps=2, optimizer_class=torch.optim.AdamW, optimizer_params={"lr": 1e-3}, weight_decay=0.01,
              evaluation_steps=2 if evaluator is not None else 0, output_path=os.path.join(out_dir, f"model{tag}_w{world}"),
              save_best_model=True, max_grad_norm=1.0, use_amp=False, callback=None, show_progress_bar=False,
              checkpoint_path=os.path.join(out_dir, f"ckpt{tag}_w{world}"), checkpoint_save_steps=2,
              checkpoint_save_total_limit=1, dropout=0)      # (masks are per rank: only the dropout-free run is rank-count invariant)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"fit{tag}_w{world}_r{rank}.npy"), model._enc.params.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_fit_is_data_parallel_under_torch_distributed(tmp_path):
    """SentenceTransformer.fit -- the only training entry point the reference uses -- all-reduces when torch.distributed
    is initialised: two ranks splitting every batch end where one process on the whole batch ends, replicas identical,
    files written by rank 0 only."""
    mp.spawn(_fit_worker, args=(1, 0, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_fit_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    single = np.load(tmp_path / "fit_w1_r0.npy")
    r0, r1 = np.load(tmp_path / "fit_w2_r0.npy"), np.load(tmp_path / "fit_w2_r1.npy")
    np.testing.assert_array_equal(r0, r1)
    from quadruplet_sentence_transformer_amd.config import PRESETS
    from quadruplet_sentence_transformer_amd.synthetic import synthetic_params
    moved = np.abs(single - synthetic_params(PRESETS["tiny-bert"], seed=14)).max()
    assert moved > 1e-4
    bad = np.abs(r0 - single) > 0.05 * moved
    assert bad.mean() < 2e-3, f"{bad.sum()} of {bad.size} parameters differ"
    assert os.path.isfile(tmp_path / "model_w2" / "model.safetensors")          # evaluator=None: rank 0 saved at the end
    assert sorted(os.listdir(tmp_path / "ckpt_w2")) == ["4"]                     # 4 steps, limit 1


def test_fit_dp_with_an_empty_shard_and_a_file_writing_evaluator(tmp_path):
    """25 examples in batches of 6: the last batch has ONE row, so under two ranks rank 1's shard of it is empty. That rank
    must still issue the per-layer all-reduces its peer issues from inside the staged backward (it issued one
    arena-wide reduce instead -- RCCL hangs on that, gloo raises a size mismatch). And the evaluator, which runs on every
    rank, gets a real directory on the ranks that do not own output_path (the reference's evaluators write there
    unconditionally); rank 0 alone writes under output_path."""
    ev = _FileWritingEvaluator()
    mp.spawn(_fit_worker, args=(1, 0, str(tmp_path), 25, ev, "_e"), nprocs=1, join=True)
    mp.spawn(_fit_worker, args=(2, _free_port(), str(tmp_path), 25, ev, "_e"), nprocs=2, join=True)
    single = np.load(tmp_path / "fit_e_w1_r0.npy")
    r0, r1 = np.load(tmp_path / "fit_e_w2_r0.npy"), np.load(tmp_path / "fit_e_w2_r1.npy")
    np.testing.assert_array_equal(r0, r1)
    from quadruplet_sentence_transformer_amd.config import PRESETS
    from quadruplet_sentence_transformer_amd.synthetic import synthetic_params
    moved = np.abs(single - synthetic_params(PRESETS["tiny-bert"], seed=14)).max()
    bad = np.abs(r0 - single) > 0.05 * moved
    assert bad.mean() < 2e-3, f"{bad.sum()} of {bad.size} parameters differ"
    assert os.path.isfile(tmp_path / "model_e_w2" / "eval" / "_quadruplet_loss_eval.json")       # rank 0's evaluator output
