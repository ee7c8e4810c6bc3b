"""Host-side mirror of the reference's loss-model glue
(/root/reference/models/quadruplet_sentence_transformer.py:9-97): same class name, constructor and
forward(features, labels) contract, so `fit(train_objectives=[(dataloader, loss_model)])` is used the same way.
The reference's own module works unchanged on top of the drop-in `sentence_transformers` package; this mirror
exists because that module imports `dataset.*` (nltk downloads at import) and so cannot be
imported offline. `fused=True` (default) runs the four columns as ONE [4B, L] encoder pass instead of four
(mathematically identical, a quarter of the launches)."""
from __future__ import annotations

from typing import Dict, List, Optional, Union

import torch

from .sentence_transformer import InputExample, SentenceTransformer

# dataset/constants.py keys of a quadruplet instance
REFERENCE_EXAMPLE = "reference"
POS_EXAMPLES = "positive"
PART_POS_EXAMPLES = "part_positive"
NEG_EXAMPLES = "negative"
_KEYS = (REFERENCE_EXAMPLE, POS_EXAMPLES, PART_POS_EXAMPLES, NEG_EXAMPLES)


class QuadrupletSentenceTransformerLossModel(torch.nn.Module):
    def __init__(self, st_model: SentenceTransformer, quadruplet_loss: torch.nn.Module,
                 additional_model_kwargs: Optional[List[str]] = None,
                 additional_loss_kwargs: Optional[List[str]] = None, fused: bool = True):
        super().__init__()
        self._st_model = st_model
        self._quadruplet_loss = quadruplet_loss
        self._model_kw = additional_model_kwargs
        self._loss_kw = additional_loss_kwargs
        self.fused = fused

    def forward(self, features: Union[Dict, List[Dict]], labels=None) -> torch.Tensor:
        cols = [features[k] for k in _KEYS] if isinstance(features, dict) else [features[i] for i in range(4)]
        model_kw = {k: features[k] for k in (self._model_kw or [])}
        loss_kw = {k: features[k] for k in (self._loss_kw or [])}
        if self.fused and not model_kw:
            embs = self._encode_fused(cols)
        else:
            embs = [self._st_model(c, **model_kw)["sentence_embedding"] for c in cols]
        return self._quadruplet_loss(x_anchor=embs[0], x_pos=embs[1], x_part=embs[2], x_neg=embs[3], **loss_kw)

    def _encode_fused(self, cols):
        """Pad the four separately-padded columns to one length and run a single encoder pass."""
        L = max(c["input_ids"].shape[1] for c in cols)
        pad_id = self._st_model.cfg.pad_token_id

        def cat(key, fill):
            return torch.cat([torch.nn.functional.pad(c[key], (0, L - c[key].shape[1]), value=fill) for c in cols], 0)

        feats = {"input_ids": cat("input_ids", pad_id), "attention_mask": cat("attention_mask", 0)}
        if all("token_type_ids" in c for c in cols):
            feats["token_type_ids"] = cat("token_type_ids", 0)
        emb = self._st_model(feats)["sentence_embedding"]
        return list(emb.split([c["input_ids"].shape[0] for c in cols], 0))


def to_input_example(instance) -> InputExample:
    """Quadruplet dict -> InputExample(texts=[reference, positive, part_positive, negative]); lists pick one entry."""
    import random
    if isinstance(instance, tuple):
        instance = instance[0]
    texts = []
    for k in _KEYS:
        v = instance[k]
        texts.append(v[random.randrange(len(v))] if isinstance(v, list) else v)
    return InputExample(texts=texts)
