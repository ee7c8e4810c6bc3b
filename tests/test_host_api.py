"""CPU: host-side mirror of the reference interface (validation, schedules, collate, buckets, drop-in imports)."""
import os
import sys

import numpy as np
import pytest
import torch

import quadruplet_sentence_transformer_amd  # noqa: F401
from quadruplet_sentence_transformer_amd import _lib
from quadruplet_sentence_transformer_amd.config import PRESETS, build_layout, forward_flops_per_sequence
from quadruplet_sentence_transformer_amd.losses import (DEFAULT_GAMMA, REDUCTIONS, GammaQuadrupletLoss, QuadrupletLoss,
                                                        gamma_quadruplet_loss)
from quadruplet_sentence_transformer_amd.sentence_transformer import InputExample, SyntheticTokenizer
from quadruplet_sentence_transformer_amd.trainer import gradient_buckets, warmup_linear_lr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_loss_defaults_match_reference():
    l = GammaQuadrupletLoss()
    assert (l.gamma, l.margin_pos_neg, l.margin_pos_part, l.margin_part_neg, l.p, l.swap, l.reduction) == \
           (0.6, 1.0, 1.0, 1.0, 2.0, False, "mean")          # losses.py:241-249
    assert DEFAULT_GAMMA == 0.6 and REDUCTIONS == frozenset(["mean", "sum", "none"])
    import inspect
    sig = inspect.signature(gamma_quadruplet_loss)
    assert [p.default for p in list(sig.parameters.values())[4:]] == [0.6, 1.0, 0.5, 0.5, 2.0, False, "mean"]   # :13-19
    assert list(sig.parameters)[:4] == ["x_anchor", "x_pos", "x_part", "x_neg"]


@pytest.mark.parametrize("kw,msg", [
    (dict(gamma=1.5), "gamma must be between 0 and 1, 1.5 given"),
    (dict(gamma=-0.1), "gamma must be between 0 and 1, -0.1 given"),
    (dict(margin_pos_neg=0), "margin_pos_neg must be positive, 0 given"),
    (dict(margin_pos_part=-1.0), "margin_pos_part must be positive, -1.0 given"),
    (dict(margin_part_neg=0.0), "margin_part_neg must be positive, 0.0 given"),
    (dict(p=0), "p must be positive, 0 given"),
])
def test_loss_validation_messages(kw, msg):
    with pytest.raises(ValueError) as e:
        GammaQuadrupletLoss(**kw)
    assert str(e.value) == msg
    x = torch.zeros(2, 4)
    with pytest.raises(ValueError) as e:
        gamma_quadruplet_loss(x, x, x, x, **kw)
    assert str(e.value) == msg


def test_loss_reduction_validation_and_setters():
    with pytest.raises(ValueError) as e:
        GammaQuadrupletLoss(reduction="avg")
    assert "reduction must be one of" in str(e.value) and "avg given" in str(e.value)
    l = GammaQuadrupletLoss()
    l.gamma, l.margin_part_neg, l.p, l.swap, l.reduction = 0.3, 0.25, 1.0, True, "sum"
    assert (l.gamma, l.margin_part_neg, l.p, l.swap, l.reduction) == (0.3, 0.25, 1.0, True, "sum")
    for attr, bad in (("gamma", 2), ("margin_pos_neg", 0), ("margin_pos_part", -1), ("margin_part_neg", 0), ("p", 0),
                      ("reduction", "x")):
        with pytest.raises(ValueError):
            setattr(l, attr, bad)
    assert issubclass(GammaQuadrupletLoss, QuadrupletLoss) and isinstance(l, torch.nn.Module)
    with pytest.raises(TypeError):
        QuadrupletLoss()            # abstract


def test_loss_refuses_cpu_tensors_loudly():
    x = torch.zeros(2, 4)
    with pytest.raises(_lib.QstError):
        gamma_quadruplet_loss(x, x, x, x)       # valid hyper-parameters, but no CPU path exists


def test_warmup_linear_schedule_matches_transformers():
    from transformers import get_linear_schedule_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=2e-5)
    sch = get_linear_schedule_with_warmup(opt, num_warmup_steps=7, num_training_steps=40)
    for step in range(45):
        assert abs(opt.param_groups[0]["lr"] - warmup_linear_lr(2e-5, step, 7, 40)) < 1e-12
        opt.step()
        sch.step()


def test_gradient_buckets_partition_the_arena():
    for name, cfg in PRESETS.items():
        b = gradient_buckets(cfg)
        _, total = build_layout(cfg)
        assert len(b) == cfg.num_layers + 1
        cover = np.zeros(total, np.int32)
        for lo, hi in b:
            cover[lo:hi] += 1
        assert (cover == 1).all()
        assert b[-1][0] == 0 and b[0][1] == total      # embeddings last, top layer first


def test_flops_formula_matches_survey():
    cfg = PRESETS["all-MiniLM-L6-v2"]
    assert abs(forward_flops_per_sequence(cfg, 128) / 1e9 - 2.869) < 2e-3
    assert abs(4 * forward_flops_per_sequence(cfg, 128) * 3 / 1e9 - 34.427) < 2e-2  # train GF / quadruplet


def test_synthetic_tokenizer_and_collate_shapes():
    cfg = PRESETS["tiny-bert"]
    tok = SyntheticTokenizer(cfg)
    out = tok(["a quick brown fox", "hello"], max_length=16)
    assert out["input_ids"].shape == out["attention_mask"].shape == (2, 6)
    assert out["attention_mask"].sum(1).tolist() == [6, 3]
    assert out["input_ids"][1, 3:].tolist() == [0, 0, 0] and "token_type_ids" in out
    out = tok(["x " * 50], max_length=8)
    assert out["input_ids"].shape == (1, 8)
    ex = InputExample(texts=["a", "b", "c", "d"])
    assert ex.label == 0 and str(ex).startswith("<InputExample>")


def test_dropin_namespaces_resolve_to_this_build():
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    try:
        for m in [k for k in sys.modules if k == "models" or k.startswith("models.") or k.startswith("sentence_transformers")]:
            del sys.modules[m]
        from sentence_transformers import InputExample as IE, SentenceTransformer as ST            # training/main.py:5
        from sentence_transformers.util import cos_sim, dot_score                                  # training/main.py:6
        from sentence_transformers.evaluation import SentenceEvaluator, SequentialEvaluator, SimilarityFunction, TripletEvaluator
        from models.losses import GammaQuadrupletLoss as G                                         # training/main.py:12
        from models.losses.losses import DEFAULT_GAMMA as DG, QuadrupletLoss as Q                  # :13, quadruplet_sentence_transformer.py:6
        assert G is GammaQuadrupletLoss and DG == 0.6 and Q is QuadrupletLoss
        assert ST.__module__.startswith("quadruplet_sentence_transformer_amd")
        from quadruplet_sentence_transformer_amd import util as U
        assert cos_sim is U.cos_sim and dot_score is U.dot_score        # libqst-backed (numerics: tests/test_gpu_retrieval.py)
        assert SequentialEvaluator([lambda m, o, e, s: 1.0, lambda m, o, e, s: 2.0])(None) == 2.0
    finally:
        sys.path.remove(os.path.join(ROOT, "dropin"))


def test_model_requires_a_device_and_fails_loudly():
    from quadruplet_sentence_transformer_amd.sentence_transformer import SentenceTransformer
    with pytest.raises(_lib.QstError):
        SentenceTransformer("tiny-bert", device="cpu")
    if not torch.cuda.is_available():
        with pytest.raises(_lib.QstError):
            SentenceTransformer("tiny-bert")
