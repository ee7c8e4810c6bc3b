"""CPU: the real-tokenizer branch (transformers.AutoTokenizer over a local vocabulary -- tests/golden/tiny_vocab.txt, a
WordPiece vocabulary written for these tests) and length bucketing."""
import json
import os
import shutil

import numpy as np
import pytest
import torch

import quadruplet_sentence_transformer_amd  # noqa: F401
from quadruplet_sentence_transformer_amd.config import PRESETS
from quadruplet_sentence_transformer_amd.data import LengthBucketBatchSampler, padded_tokens
from quadruplet_sentence_transformer_amd.sentence_transformer import (SyntheticTokenizer, count_tokens, load_tokenizer,
                                                                      tokenize_texts)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def vocab_dir(tmp_path, model_type="bert"):
    d = tmp_path / "model"
    d.mkdir()
    shutil.copy(os.path.join(ROOT, "tests", "golden", "tiny_vocab.txt"), d / "vocab.txt")
    json.dump({"model_type": model_type}, open(d / "config.json", "w"))
    return str(d)


def test_model_directory_tokenizer_is_used(tmp_path):
    cfg = PRESETS["tiny-bert"]
    tok = load_tokenizer(vocab_dir(tmp_path), cfg)
    assert tok is not None and type(tok).__name__.startswith("BertTokenizer") and len(tok) <= cfg.vocab_size
    vocab = [w.strip() for w in open(os.path.join(ROOT, "tests", "golden", "tiny_vocab.txt"))]
    ix = {w: i for i, w in enumerate(vocab)}
    out = tokenize_texts(tok, cfg, ["  A man rides horses!  ", "two dogs playing", "zebra"], 16)
    want = [[ix["[CLS]"], ix["a"], ix["man"], ix["rides"], ix["horse"], ix["##s"], ix["!"], ix["[SEP]"]],
            [ix["[CLS]"], ix["two"], ix["dogs"], ix["playing"], ix["[SEP]"]],
            [ix["[CLS]"], ix["[UNK]"], ix["[SEP]"]]]
    L = max(map(len, want))
    assert out["input_ids"].tolist() == [w + [ix["[PAD]"]] * (L - len(w)) for w in want]
    assert out["attention_mask"].sum(1).tolist() == [len(w) for w in want]
    assert out["token_type_ids"].shape == out["input_ids"].shape and int(out["token_type_ids"].sum()) == 0
    # truncation='longest_first' to max_length, special tokens kept
    cut = tokenize_texts(tok, cfg, ["a man rides a red horse in the park near the old bridge"], 6)
    assert cut["input_ids"].shape == (1, 6) and cut["input_ids"][0, 0] == ix["[CLS]"] and cut["input_ids"][0, -1] == ix["[SEP]"]
    assert count_tokens(tok, "A man rides horses!", 16) == 8 and count_tokens(tok, "a " * 40, 6) == 6


def test_no_vocabulary_means_no_tokenizer_and_oversized_vocabulary_raises(tmp_path):
    cfg = PRESETS["tiny-bert"]
    empty = tmp_path / "empty"
    empty.mkdir()
    json.dump({"model_type": "bert"}, open(empty / "config.json", "w"))
    assert load_tokenizer(str(empty), cfg) is None and load_tokenizer(None, cfg) is None
    from dataclasses import replace
    with pytest.raises(ValueError):
        load_tokenizer(vocab_dir(tmp_path), replace(cfg, vocab_size=64))      # ids would index past the embedding table
    syn = SyntheticTokenizer(cfg)
    assert count_tokens(syn, "a quick brown fox", 16) == 6 and count_tokens(syn, "x " * 50, 8) == 8


def test_length_bucketing_covers_every_example_once_and_cuts_padding():
    rng = np.random.default_rng(0)
    lengths = np.clip(rng.lognormal(2.5, 0.6, size=1003).astype(int) + 3, 3, 128)
    B = 16
    s = LengthBucketBatchSampler(lengths, B, shuffle=True, seed=3, pool_batches=10)
    e0 = list(s)
    assert len(e0) == len(s) == (1003 + B - 1) // B
    assert sorted(i for b in e0 for i in b) == list(range(1003))
    assert all(len(b) == B for b in e0 if b is not min(e0, key=len)) and sum(len(b) != B for b in e0) <= 1
    e1 = list(s)                                                       # the next epoch regroups and reorders
    assert sorted(i for b in e1 for i in b) == list(range(1003)) and e1 != e0
    plain = [list(range(i, min(i + B, 1003))) for i in range(0, 1003, B)]
    assert padded_tokens(lengths, e0) < 0.8 * padded_tokens(lengths, plain)
    # deterministic given (seed, epoch); drop_last drops the ragged batch
    s2 = LengthBucketBatchSampler(lengths, B, shuffle=True, seed=3, pool_batches=10)
    assert list(s2) == e0
    d = LengthBucketBatchSampler(lengths, B, drop_last=True)
    assert len(list(d)) == len(d) == 1003 // B
    ordered = list(LengthBucketBatchSampler(lengths, B, shuffle=False, pool_batches=1000))
    flat = [lengths[i] for b in ordered for i in b]
    assert flat == sorted(flat)                                        # one pool, no shuffle: a plain length sort
    loader = torch.utils.data.DataLoader(list(range(1003)), batch_sampler=s)
    assert sum(len(b) for b in loader) == 1003
