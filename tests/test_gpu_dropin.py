"""GPU: the SentenceTransformer-compatible surface (encode / __call__ / fit / save+load) and the reference's
loss-model call pattern, all executing through libqst.so."""
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

pytestmark = pytest.mark.gpu

import quadruplet_sentence_transformer_amd  # noqa: E402,F401
from quadruplet_sentence_transformer_amd.evaluation import SentenceEvaluator  # noqa: E402
from quadruplet_sentence_transformer_amd.losses import GammaQuadrupletLoss  # noqa: E402
from quadruplet_sentence_transformer_amd.quadruplet_model import (NEG_EXAMPLES, PART_POS_EXAMPLES, POS_EXAMPLES,  # noqa: E402
                                                                  REFERENCE_EXAMPLE, QuadrupletSentenceTransformerLossModel,
                                                                  to_input_example)
from quadruplet_sentence_transformer_amd.sentence_transformer import InputExample, SentenceTransformer  # noqa: E402

WORDS = "a man rides red horse two dogs play in park woman eats green apple near old bridge small cat sleeps".split()


def sent(i, n):
    rng = np.random.RandomState(i)
    return " ".join(rng.choice(WORDS, size=n))


def quad(i):
    return {REFERENCE_EXAMPLE: sent(i, 9), POS_EXAMPLES: [sent(i, 9) + " today", sent(i, 9) + " now"],
            PART_POS_EXAMPLES: sent(i, 4), NEG_EXAMPLES: sent(1000 + i, 11)}


@pytest.fixture(scope="module")
def model():
    return SentenceTransformer("tiny-bert", device="cuda")


def test_encode_shapes_order_and_determinism(model):
    texts = [sent(i, 3 + i % 7) for i in range(11)]
    e = model.encode(texts, batch_size=4)
    assert isinstance(e, np.ndarray) and e.shape == (11, 64) and np.isfinite(e).all()
    np.testing.assert_allclose(np.linalg.norm(e, axis=1), 1.0, rtol=1e-4)        # Normalize module
    one = model.encode(texts[5])
    assert one.shape == (64,)
    np.testing.assert_allclose(one, e[5], rtol=0, atol=2e-3)    # batch composition changes padding only
    t = model.encode(texts, convert_to_tensor=True)
    assert torch.is_tensor(t) and t.is_cuda and t.shape == (11, 64)
    np.testing.assert_allclose(t.cpu().numpy(), model.encode(texts, batch_size=32), rtol=0, atol=2e-3)


def test_encode_normalize_embeddings_runs_the_library_kernel():
    """encode(normalize_embeddings=True) on a model WITHOUT the Normalize module (ST auto-wrapped bert-base style):
    qst_normalize_rows == torch.nn.functional.normalize(p=2, dim=1)."""
    from dataclasses import replace
    from quadruplet_sentence_transformer_amd.config import PRESETS
    m = SentenceTransformer(config=replace(PRESETS["tiny-bert"], normalize=False), device="cuda")
    texts = [sent(i, 3 + i % 5) for i in range(7)]
    raw = m.encode(texts, convert_to_tensor=True)
    assert (raw.norm(dim=1) - 1).abs().max() > 1e-2
    got = m.encode(texts, normalize_embeddings=True, convert_to_tensor=True)
    torch.testing.assert_close(got, torch.nn.functional.normalize(raw, p=2, dim=1), rtol=1e-6, atol=1e-7)


def test_train_mode_drops_with_or_without_autograd():
    """HF modules drop whenever module.training is set: the reference's QuadrupletLossEvaluator takes its validation loss
    inside fit() under torch.no_grad() without calling eval() (models/evaluators.py:80-95) -- with dropout. eval() and
    encode() never drop; outside fit() (no dropout configured) train() mode is deterministic."""
    m = SentenceTransformer("tiny-bert", device="cuda")
    feats = m.tokenize([sent(i, 4 + i % 5) for i in range(6)])
    m.train()
    with torch.no_grad():
        a = m(dict(feats))["sentence_embedding"].clone()
        b = m(dict(feats))["sentence_embedding"].clone()
    assert torch.equal(a, b)                                        # no dropout configured
    m._enc.set_dropout(0.1, 0.1, 5)
    try:
        with torch.no_grad():
            c = m(dict(feats))["sentence_embedding"].clone()
            d = m(dict(feats))["sentence_embedding"].clone()
        assert not torch.equal(c, d) and (c - a).abs().max() > 1e-3   # fresh masks per pass
        assert m._live_graphs == 0
        m.eval()
        with torch.no_grad():
            e = m(dict(feats))["sentence_embedding"].clone()
        torch.testing.assert_close(e, a, rtol=0, atol=2e-5)
    finally:
        m._enc.set_dropout(0.0, 0.0)


def test_encode_parity_precision(model):
    texts = [sent(i, 3 + i % 7) for i in range(9)]
    fast = model.encode(texts)
    exact = model.encode(texts, precision="bf16x3")
    assert np.abs(fast - exact).max() < 3e-3 and np.abs(fast - exact).max() > 0      # bf16 floor vs fp32-class
    assert model.inference_precision == "bf16"
    np.testing.assert_allclose(model.encode(texts, precision="bf16x3", batch_size=2), exact, rtol=1e-3, atol=1e-4)


def test_four_call_pattern_equals_fused_pass(model):
    """models/quadruplet_sentence_transformer.py:42-75 runs 4 encoder calls; the fused [4B, L] pass must agree."""
    loss = GammaQuadrupletLoss(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5, p=2.0)
    lm4 = QuadrupletSentenceTransformerLossModel(model, loss, fused=False)
    lm1 = QuadrupletSentenceTransformerLossModel(model, loss, fused=True)
    batch = [to_input_example(quad(i)) for i in range(6)]
    feats, labels = model.smart_batching_collate(batch)
    assert len(feats) == 4 and labels.shape == (6,)
    model.train()
    enc = model._enc
    enc.grads.zero_()
    l4 = lm4([dict(f) for f in feats], labels)
    l4.backward()
    g4 = enc.grads.clone()
    enc.grads.zero_()
    l1 = lm1([dict(f) for f in feats], labels)
    l1.backward()
    g1 = enc.grads.clone()
    assert abs(l4.item() - l1.item()) < 2e-4
    assert (g4 - g1).norm().item() <= 2e-2 * g1.norm().item()
    # dict-keyed features (quadruplet_sentence_transformer.py:24-28)
    keyed = {k: dict(f) for k, f in zip((REFERENCE_EXAMPLE, POS_EXAMPLES, PART_POS_EXAMPLES, NEG_EXAMPLES), feats)}
    assert abs(lm1(keyed).item() - l1.item()) < 1e-6
    enc.grads.zero_()


class CountingEvaluator(SentenceEvaluator):
    def __init__(self, loss_model, batch):
        self.calls, self.loss_model, self.batch = [], loss_model, batch

    def __call__(self, model, output_path=None, epoch=-1, steps=-1):
        feats, labels = model.smart_batching_collate(self.batch)
        with torch.no_grad():
            v = self.loss_model(feats, labels).item()
        self.calls.append((epoch, steps, v))
        return -v            # higher is better for save_best_model


class Stop(BaseException):
    pass


def test_fit_trains_saves_and_reloads(model, tmp_path):
    loss = GammaQuadrupletLoss(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5, p=2.0)
    lm = QuadrupletSentenceTransformerLossModel(model, loss)
    data = [to_input_example(quad(i)) for i in range(32)]
    dl = DataLoader(data, batch_size=8, shuffle=True, num_workers=0)
    ev = CountingEvaluator(lm, data[:8])
    before = model._enc.params.clone()
    seen = []
    model.fit(train_objectives=[(dl, lm)], evaluator=ev, epochs=3, steps_per_epoch=None, scheduler="warmuplinear",
              warmup_steps=2, optimizer_class=torch.optim.AdamW, optimizer_params={"lr": 2e-3}, weight_decay=0.01,
              evaluation_steps=2, output_path=str(tmp_path / "out"), save_best_model=True, max_grad_norm=1.0,
              use_amp=False, callback=lambda score, epoch, steps: seen.append((score, epoch, steps)),
              show_progress_bar=False, checkpoint_path=str(tmp_path / "ckpt"), checkpoint_save_steps=4,
              checkpoint_save_total_limit=2)
    assert not torch.equal(before, model._enc.params)
    assert len(ev.calls) == 3 * (2 + 1) and len(seen) == len(ev.calls)      # every 2 steps + end of each epoch
    assert ev.calls[-1][2] < ev.calls[0][2], "validation loss did not go down"
    assert sorted(os.listdir(tmp_path / "ckpt")) == ["12", "8"]
    assert os.path.exists(tmp_path / "out" / "model.safetensors") and os.path.exists(tmp_path / "out" / "modules.json")
    # reload the saved best model from its directory (ir_evauation_script.py:128) and compare embeddings
    best = SentenceTransformer(str(tmp_path / "ckpt" / "12"), device="cuda")
    texts = [sent(i, 6) for i in range(5)]
    np.testing.assert_allclose(best.encode(texts), model.encode(texts), rtol=0, atol=1e-6)
    # a callback raising a BaseException subclass must propagate out of fit (training/callbacks.py:47, main.py:149)
    def boom(score, epoch, steps):
        raise Stop()
    with pytest.raises(Stop):
        model.fit(train_objectives=[(dl, lm)], evaluator=ev, epochs=1, evaluation_steps=1, callback=boom,
                  optimizer_params={"lr": 1e-4}, output_path=str(tmp_path / "out2"), show_progress_bar=False)


def test_fit_at_parity_precision(tmp_path):
    """fit(precision="bf16x3"): the reference's training call on the fp32-class path (forward AND backward as split-bf16 x3
    products; the reference trains in fp32, training/main.py:142). Same data, same seed, same schedule as a bf16 fit: both
    train, and the two end close to each other (they differ by the bf16 path's operand rounding, not in kind); a further
    epoch in train() mode (dropout 0.1, as the reference's fit() runs) goes through the same path."""
    def run(prec):
        torch.manual_seed(0)
        m = SentenceTransformer("tiny-bert", device="cuda")
        loss = GammaQuadrupletLoss(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5, p=2.0)
        lm = QuadrupletSentenceTransformerLossModel(m, loss)
        data = [to_input_example(quad(i)) for i in range(16)]
        dl = DataLoader(data, batch_size=8, shuffle=False, num_workers=0)
        ev = CountingEvaluator(lm, data[:8])
        m.fit(train_objectives=[(dl, lm)], evaluator=ev, epochs=3, scheduler="warmuplinear", warmup_steps=2,
              optimizer_params={"lr": 2e-3}, evaluation_steps=0, output_path=str(tmp_path / prec), show_progress_bar=False,
              dropout=0, precision=prec)
        return m, ev
    m3, ev3 = run("bf16x3")
    m1, ev1 = run("bf16")
    assert ev3.calls[-1][2] < ev3.calls[0][2], ("validation loss did not go down on the parity path", ev3.calls)
    assert m3.training_precision == "bf16"                      # restored after fit
    d = (m3._enc.params - m1._enc.params).abs()
    moved = (m3._enc.params - SentenceTransformer("tiny-bert", device="cuda")._enc.params).abs().mean()
    assert float(d.mean()) < 0.25 * float(moved), (float(d.mean()), float(moved))
    before = m3._enc.params.clone()
    m3.fit(train_objectives=[(DataLoader([to_input_example(quad(i)) for i in range(8)], batch_size=8),
                              QuadrupletSentenceTransformerLossModel(m3, GammaQuadrupletLoss(gamma=0.6)))],
           epochs=2, warmup_steps=0, show_progress_bar=False, dropout=0.1, dropout_seed=2, precision="bf16x3",
           optimizer_params={"lr": 1e-3})
    assert int(m3._enc.drop_state[2]) >= 2 and torch.isfinite(m3._enc.params).all() and not torch.equal(before, m3._enc.params)


def test_fit_with_use_amp_trains_on_f16_operands_under_the_device_side_scaler(tmp_path):
    """fit(use_amp=True) as the reference can call it (training/main.py:142 passes the flag; ST wraps the step in fp16 autocast +
    GradScaler): here precision "f16" -- IEEE-half operands, the loss scaled by a device-resident GradScaler, unscale / clip /
    step / update in qst_clip_adamw_step_amp. Same data, seed and schedule as the bf16 fit and as precision="f16w": all train
    and end close to each other; the scaler saw no overflow; the model leaves fit() on its default precision."""
    def run(**kw):
        torch.manual_seed(0)
        m = SentenceTransformer("tiny-bert", device="cuda")
        loss = GammaQuadrupletLoss(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5, p=2.0)
        lm = QuadrupletSentenceTransformerLossModel(m, loss)
        data = [to_input_example(quad(i)) for i in range(16)]
        dl = DataLoader(data, batch_size=8, shuffle=False, num_workers=0)
        ev = CountingEvaluator(lm, data[:8])
        m.fit(train_objectives=[(dl, lm)], evaluator=ev, epochs=3, scheduler="warmuplinear", warmup_steps=2,
              optimizer_params={"lr": 2e-3}, evaluation_steps=0, output_path=str(tmp_path / "o"), show_progress_bar=False,
              dropout=0, **kw)
        return m, ev
    ma, eva = run(use_amp=True)
    mw, evw = run(precision="f16w")
    m1, ev1 = run()
    for ev in (eva, evw):
        assert ev.calls[-1][2] < ev.calls[0][2], ("validation loss did not go down", ev.calls)
    assert ma.training_precision == "bf16"
    sc = ma._enc.amp_scaler.cpu().tolist()
    assert sc[0] == 65536.0 and sc[3] == 0.0                       # GradScaler's init_scale, no skipped step
    cnt = ma._enc._step2_dev.cpu().tolist()
    assert cnt == [6, 6]                                           # six optimiser steps, six scheduler steps
    moved = (m1._enc.params - SentenceTransformer("tiny-bert", device="cuda")._enc.params).abs().mean()
    for m in (ma, mw):
        d = (m._enc.params - m1._enc.params).abs()
        assert float(d.mean()) < 0.25 * float(moved), (float(d.mean()), float(moved))
    # encode() on the f16 precisions, against the parity path
    texts = [sent(i, 3 + i % 7) for i in range(9)]
    exact = ma.encode(texts, precision="bf16x3")
    np.testing.assert_allclose(ma.encode(texts, precision="f16w"), exact, rtol=1e-3, atol=1e-4)
    assert np.abs(ma.encode(texts, precision="f16") - exact).max() < 3e-4


def test_fit_on_the_fp8_matrix_cores_in_train_mode(tmp_path):
    """fit(precision="fp8") exactly as the reference calls fit (training/main.py:128-148: train() mode, HF dropout 0.1 from
    the config): every forward Linear on the fp8 matrix cores, the bf16 backward, the same counter-based dropout masks as the
    bf16 path. Same data, seed and schedule as a bf16 fit: both train and end close to each other."""
    def run(prec):
        torch.manual_seed(0)
        m = SentenceTransformer("tiny-mpnet", device="cuda")               # H = 128, I = 256: multiples of the 128-deep fp8 stage
        loss = GammaQuadrupletLoss(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5, p=2.0)
        lm = QuadrupletSentenceTransformerLossModel(m, loss)
        data = [to_input_example(quad(i)) for i in range(16)]
        dl = DataLoader(data, batch_size=8, shuffle=False, num_workers=0)
        ev = CountingEvaluator(lm, data[:8])
        m.fit(train_objectives=[(dl, lm)], evaluator=ev, epochs=3, scheduler="warmuplinear", warmup_steps=2,
              optimizer_params={"lr": 2e-3}, evaluation_steps=0, output_path=str(tmp_path / prec), show_progress_bar=False,
              dropout=0.1, dropout_seed=3, precision=prec)
        return m, ev
    m8, ev8 = run("fp8")
    m1, ev1 = run("bf16")
    assert int(m8._enc.drop_state[2]) >= 6                      # the device-side mask counter: six training forwards dropped
    assert ev8.calls[-1][2] < ev8.calls[0][2], "validation loss did not go down with the forward on the fp8 matrix cores"
    assert m8.training_precision == "bf16"                      # restored after fit
    d = (m8._enc.params - m1._enc.params).abs()
    moved = (m8._enc.params - SentenceTransformer("tiny-mpnet", device="cuda")._enc.params).abs().mean()
    assert float(d.mean()) < 0.35 * float(moved)


def test_named_parameters_are_views_with_hf_names(model):
    names = dict(model.named_parameters())
    assert "0.auto_model.embeddings.word_embeddings.weight" in names
    assert "0.auto_model.encoder.layer.1.attention.self.query.weight" in names
    w = names["0.auto_model.encoder.layer.0.output.LayerNorm.weight"]
    assert w.data_ptr() >= model._enc.params.data_ptr()
    assert w.data_ptr() < model._enc.params.data_ptr() + model._enc.params.numel() * 4


def test_fit_checkpoints_carry_optimizer_state_and_resume(tmp_path):
    """fit() checkpoints hold the Adam moments and step counters next to the ST model files, and
    fit(resume_from_checkpoint=...) continues the interrupted run: 6 uninterrupted steps == 3 steps, checkpoint, a NEW
    model object resumed from it, 3 more steps (same data order), up to the float-atomics noise of two identical runs."""
    examples = [to_input_example(quad(i)) for i in range(24)]

    def run(model, **kw):
        loader = DataLoader(examples, batch_size=4, shuffle=False)
        lm = QuadrupletSentenceTransformerLossModel(model, GammaQuadrupletLoss(gamma=0.6, margin_pos_neg=1.0,
                                                                               margin_pos_part=0.5, margin_part_neg=0.5))
        model.fit(train_objectives=[(loader, lm)], epochs=1, steps_per_epoch=6, warmup_steps=2,
                  optimizer_params={"lr": 1e-3}, show_progress_bar=False, **kw)

    full = SentenceTransformer("tiny-bert", device="cuda")
    w0 = full._enc.params.clone()
    run(full)
    part = SentenceTransformer("tiny-bert", device="cuda")
    ck = str(tmp_path / "ck")
    # stop during step 4: an evaluator callback that raises is how the reference stops early (callbacks.py:47)
    class Stop(BaseException):
        pass

    class Ev(SentenceEvaluator):
        def __call__(self, model, output_path=None, epoch=-1, steps=-1):
            return 0.0

    def cb(score, epoch, steps):
        if steps == 4:                   # the step-3 checkpoint has been written by then
            raise Stop()

    with pytest.raises(Stop):
        run(part, checkpoint_path=ck, checkpoint_save_steps=3, evaluator=Ev(), evaluation_steps=1, callback=cb)
    assert sorted(os.listdir(ck)) == ["3"]
    assert {"model.safetensors", "training_state.safetensors", "training_state.json"} <= set(os.listdir(os.path.join(ck, "3")))
    resumed = SentenceTransformer(os.path.join(ck, "3"), device="cuda")        # the checkpoint is a loadable ST directory
    run(resumed, resume_from_checkpoint=os.path.join(ck, "3"))
    assert resumed._enc.opt_step == 6
    moved = (full._enc.params - w0).abs().mean().item()
    assert moved > 0
    assert (resumed._enc.params - full._enc.params).abs().mean().item() < 0.05 * moved
    # and the moments really were restored: resuming WITHOUT them lands measurably elsewhere
    cold = SentenceTransformer(os.path.join(ck, "3"), device="cuda")
    run(cold)            # 6 fresh steps from the step-3 weights: different trajectory
    assert (cold._enc.params - full._enc.params).abs().mean().item() > 0.2 * moved
    with pytest.raises(FileNotFoundError):
        full.save(str(tmp_path / "plain"))
        run(SentenceTransformer("tiny-bert", device="cuda"), resume_from_checkpoint=str(tmp_path / "plain"))


def test_real_tokenizer_round_trip_and_bucketed_fit(tmp_path):
    """a model directory WITH a vocabulary goes through transformers.AutoTokenizer
    (tests/golden/tiny_vocab.txt, WordPiece): save -> SentenceTransformer(path) -> tokenize / smart_batching_collate /
    encode / fit, the fit() batches drawn by the length-bucketing sampler; the synthetic hashing tokenizer is not involved."""
    import shutil
    from quadruplet_sentence_transformer_amd.data import LengthBucketBatchSampler, padded_tokens
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = SentenceTransformer("tiny-bert", device="cuda")
    d = str(tmp_path / "tiny-bert-vocab")
    src.save(d)
    shutil.copy(os.path.join(root, "tests", "golden", "tiny_vocab.txt"), os.path.join(d, "vocab.txt"))
    model = SentenceTransformer(d, device="cuda")
    assert model.tokenizer is not None and type(model.tokenizer).__name__.startswith("BertTokenizer")
    vocab = {w.strip(): i for i, w in enumerate(open(os.path.join(d, "vocab.txt")))}
    feats = model.tokenize(["A man rides horses!", "two dogs"])
    assert feats["input_ids"][0].tolist() == [vocab[t] for t in "[CLS] a man rides horse ##s ! [SEP]".split()]
    assert feats["attention_mask"].tolist() == [[1] * 8, [1] * 4 + [0] * 4]
    # same weights, same ids -> same embeddings whichever object encodes them (the tokenizer is the only difference)
    emb = model.encode(["a man rides a horse", "two dogs play in the park"], convert_to_tensor=True)
    ids = model.tokenize(["a man rides a horse", "two dogs play in the park"])
    direct = src({k: v.cuda() for k, v in ids.items()})["sentence_embedding"]
    torch.testing.assert_close(emb, direct.detach(), rtol=0, atol=1e-6)
    # a saved copy keeps the tokenizer (save_pretrained) -> the next load tokenizes identically
    d2 = str(tmp_path / "resaved")
    model.save(d2)
    again = SentenceTransformer(d2, device="cuda")
    assert again.tokenize(["A man rides horses!"])["input_ids"].tolist() == feats["input_ids"][:1].tolist()

    # real-text fit(): quadruplets of very different lengths, batches from the length-bucketing sampler
    rng = np.random.RandomState(1)
    words = [w for w in vocab if w.isalpha()]
    examples = [InputExample(texts=[" ".join(rng.choice(words, size=n)) for _ in range(4)])
                for n in rng.choice([3, 5, 9, 14, 22, 40, 55], size=48)]
    lengths = model.token_lengths(examples)
    assert min(lengths) >= 5 and max(lengths) <= model.max_seq_length
    sampler = LengthBucketBatchSampler(lengths, 8, shuffle=True, seed=14, pool_batches=6)
    plain = [list(range(i, i + 8)) for i in range(0, 48, 8)]
    assert padded_tokens(lengths, list(sampler)) < padded_tokens(lengths, plain)
    dl = DataLoader(examples, batch_sampler=sampler)
    loss = GammaQuadrupletLoss(gamma=0.6, margin_pos_neg=1.0, margin_pos_part=0.5, margin_part_neg=0.5)
    lm = QuadrupletSentenceTransformerLossModel(st_model=model, quadruplet_loss=loss)
    before = model._enc.params.clone()
    model.fit(train_objectives=[(dl, lm)], epochs=1, warmup_steps=2, optimizer_params={"lr": 1e-3}, show_progress_bar=False)
    assert torch.isfinite(model._enc.params).all() and (model._enc.params - before).abs().max() > 1e-4
