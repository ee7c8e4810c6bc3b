"""`SentenceTransformer`-compatible front end over the HIP encoder (host-side mirror of the
sentence-transformers 2.2.2 surface the reference uses).

What the reference's unchanged callers need and get here:
  * training/main.py:114-148  -> SentenceTransformer(model_name_or_path=, device=), .fit(<20 kwargs>)
  * models/quadruplet_sentence_transformer.py:42-60 -> model(features)['sentence_embedding'] with autograd
  * models/evaluators.py:68-96 -> model.smart_batching_collate, model.device, no_grad forward
  * ir_evauation_script.py:127-131, ST evaluators -> .encode(...)

The module executes nothing in torch: __call__ goes through an autograd.Function whose forward/backward
call libqst.so; fit() steps the fused clip+AdamW kernel on the flat arena. torch.nn.Parameter objects exist
only as HF-named VIEWS of the arena (so `named_parameters()` and name-based decay filters see what ST shows).
"""
from __future__ import annotations

import json
import logging
import math
import os
import shutil
import tempfile
import zlib
from collections import OrderedDict
from typing import Callable, Dict, Iterable, List, Optional, Tuple, Type, Union

import numpy as np
import torch
from torch import nn

from . import _lib
from .config import ARCH_BERT, ARCH_MPNET, PRESETS, EncoderConfig, hf_param_views
from .encoder import HipEncoder
from .synthetic import synthetic_params
from .trainer import allreduce_ranges, gradient_buckets, staged_backward, staged_reduce_order, warmup_linear_lr

logger = logging.getLogger(__name__)


class InputExample:
    """sentence_transformers.InputExample (used by models/quadruplet_sentence_transformer.py:83-97)."""

    def __init__(self, guid: str = "", texts: Optional[List[str]] = None, label: Union[int, float] = 0):
        self.guid = guid
        self.texts = texts
        self.label = label

    def __str__(self):
        return "<InputExample> label: {}, texts: {}".format(str(self.label), "; ".join(self.texts))


def batch_to_device(batch, target_device):
    """sentence_transformers.util.batch_to_device: move every tensor value of a dict."""
    for key in batch:
        if isinstance(batch[key], torch.Tensor):
            batch[key] = batch[key].to(target_device)
    return batch


# ------------------------------------------------------------------------------------------------ tokenizer
class SyntheticTokenizer:
    """Stand-in used ONLY when a model directory has no vocabulary (there are no vocab files offline):
    lower-cased whitespace/punctuation split, token id = stable hash into [1000, V). Real model
    directories with tokenizer files go through transformers.AutoTokenizer instead."""

    def __init__(self, cfg: EncoderConfig):
        self.cfg = cfg
        if cfg.arch == ARCH_MPNET:
            self.cls_id, self.sep_id, self.pad_id = 0, 2, 1
        else:
            self.cls_id, self.sep_id, self.pad_id = 101 % cfg.vocab_size, 102 % cfg.vocab_size, 0

    def _ids(self, text: str) -> List[int]:
        import re
        lo = min(1000, self.cfg.vocab_size // 4)
        toks = re.findall(r"\w+|[^\w\s]", str(text).lower())
        return [lo + zlib.crc32(t.encode("utf-8")) % (self.cfg.vocab_size - lo) for t in toks]

    def __call__(self, texts: List[str], max_length: int):
        rows = []
        for t in texts:
            ids = self._ids(t)[: max(0, max_length - 2)]
            rows.append([self.cls_id] + ids + [self.sep_id])
        L = max(len(r) for r in rows) if rows else 1
        input_ids = torch.full((len(rows), L), self.pad_id, dtype=torch.int64)
        mask = torch.zeros((len(rows), L), dtype=torch.int64)
        for i, r in enumerate(rows):
            input_ids[i, : len(r)] = torch.tensor(r, dtype=torch.int64)
            mask[i, : len(r)] = 1
        out = {"input_ids": input_ids, "attention_mask": mask}
        if self.cfg.type_vocab_size > 0:
            out["token_type_ids"] = torch.zeros_like(input_ids)
        return out


def load_tokenizer(tok_dir: Optional[str], cfg: EncoderConfig):
    """The model directory's own tokenizer (transformers.AutoTokenizer over vocab.txt / tokenizer.json / ..., strictly
    offline) or None when the directory holds no vocabulary. A vocabulary whose ids do not fit the embedding table is an
    error, not a fallback: silently hashing words instead would train a different model."""
    if tok_dir is None or not any(os.path.exists(os.path.join(tok_dir, f))
                                  for f in ("tokenizer.json", "vocab.txt", "vocab.json", "sentencepiece.bpe.model")):
        return None
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(tok_dir, local_files_only=True)
    if len(tok) > cfg.vocab_size:
        raise ValueError(f"{tok_dir}: tokenizer has {len(tok)} entries but the embedding table only {cfg.vocab_size} rows")
    return tok


def tokenize_texts(tokenizer, cfg: EncoderConfig, texts, max_len: int) -> Dict[str, torch.Tensor]:
    """sentence_transformers.models.Transformer.tokenize as ST 2.2.2 runs it for a list of strings:
    strip, then padding=True, truncation='longest_first', max_length, return_tensors='pt'."""
    texts = [str(t).strip() for t in texts]
    if isinstance(tokenizer, SyntheticTokenizer):
        return tokenizer(texts, max_len)
    out = tokenizer(texts, padding=True, truncation="longest_first", return_tensors="pt", max_length=max_len)
    out = {k: v for k, v in out.items()}
    if cfg.type_vocab_size > 0 and "token_type_ids" not in out:
        out["token_type_ids"] = torch.zeros_like(out["input_ids"])
    if cfg.type_vocab_size == 0:
        out.pop("token_type_ids", None)
    return out


def count_tokens(tokenizer, text: str, max_len: int) -> int:
    if isinstance(tokenizer, SyntheticTokenizer):
        return min(max_len, len(tokenizer._ids(str(text).strip())) + 2)
    return len(tokenizer(str(text).strip(), truncation="longest_first", max_length=max_len)["input_ids"])


# ------------------------------------------------------------------------------------------------ autograd bridge
class _EncodeFn(torch.autograd.Function):
    """forward: qst_encoder_forward; backward: qst_encoder_backward accumulating straight into the gradient
    arena (the HF-named parameters' .grad are views of it), so no per-parameter tensors cross autograd."""

    @staticmethod
    def forward(ctx, anchor, model, ids, mask, types, training):
        enc: HipEncoder = model._enc
        if not training and model.inference_precision != "bf16":
            emb, _, _ = enc.forward(ids, mask, types, training=False, precision=model.inference_precision)
            return emb
        saved = None
        prec = model.training_precision if training == 1 else "bf16"
        if training == 1:
            n, L = ids.shape
            nbytes = enc.lib.qst_encoder_saved_bytes(enc._handle_for(prec), n, L, 1)
            saved = torch.empty(nbytes, dtype=torch.uint8, device=enc.device)   # one arena per live graph
        # training == 2: a train()-mode pass without autograd (dropout on, nothing to keep): the shared activation arena
        emb, _, saved = enc.forward(ids, mask, types, training=bool(training), saved=saved, precision=prec)
        ctx.model, ctx.saved, ctx.inputs, ctx.prec = model, saved, (ids, mask, types), prec
        if training == 1:
            model._live_graphs += 1
        return emb

    @staticmethod
    def backward(ctx, grad_emb):
        model = ctx.model
        ids, mask, types = ctx.inputs
        grad_emb = grad_emb.to(torch.float32)
        model._live_graphs = max(0, model._live_graphs - 1)
        dp = model._dp
        if dp is not None and model._live_graphs == 0:
            # data-parallel fit(): the LAST outstanding encoder pass of the step (the only one on the fused [4B, L]
            # path) runs in stages with each finished layer's gradients handed to the all-reduce; gradients of earlier
            # passes of the same step are already in the arena and travel with them
            model._dp_works += staged_backward(model._enc, ids, mask, types, grad_emb, ctx.saved, None, dp["buckets"],
                                               dp["group"], dp["overlap"], precision=ctx.prec)
            model._dp_reduced = True
        else:
            model._enc.backward(ids, mask, types, grad_emb, ctx.saved, precision=ctx.prec)
        ctx.saved = None
        return None, None, None, None, None, None


class _Holder(nn.Module):
    """Plain container used to build the HF-style dotted parameter names."""


# ------------------------------------------------------------------------------------------------ the model
class SentenceTransformer(nn.Module):
    def __init__(self, model_name_or_path: Optional[str] = None, modules=None, device=None,
                 cache_folder: Optional[str] = None, config: Optional[EncoderConfig] = None, seed: int = 14,
                 allow_random_init: Optional[bool] = None):
        """model_name_or_path: a local ST/HF model directory, or a model NAME, which is resolved offline the way
        sentence-transformers 2.2.2 resolves it -- `cache_folder` (or $SENTENCE_TRANSFORMERS_HOME), then the Hugging Face
        hub cache. A name with no checkpoint on disk RAISES: fine-tuning or evaluating random weights under a
        pretrained model's name is never what a caller of the reference scripts wants. Random-init weights of a named
        architecture are available on request only: `allow_random_init=True` (or QST_ALLOW_RANDOM_INIT=1), or an
        explicit `config=`; the `tiny-*` presets are synthetic test architectures and always random-init."""
        super().__init__()
        if modules is not None:
            raise NotImplementedError("custom module lists are outside the hot path this build covers")
        self._model_card_name = model_name_or_path
        arena = None
        tok_dir = None
        self._dropout_config = (0.1, 0.1)        # HF defaults; a loaded config.json overrides (fit(dropout="config"))
        if config is not None:
            cfg = config
        elif model_name_or_path is not None and os.path.isdir(str(model_name_or_path)):
            cfg, arena, tok_dir = _load_model_dir(str(model_name_or_path))
            self._dropout_config = _dropout_from_config(str(model_name_or_path))
        else:
            full = str(model_name_or_path or "all-MiniLM-L6-v2")
            name = full.split("/")[-1]
            found = _find_cached_model(full, cache_folder)
            if found is not None:
                cfg, arena, tok_dir = _load_model_dir(found)
                self._dropout_config = _dropout_from_config(found)
            else:
                if name not in PRESETS:
                    raise ValueError(f"unknown model '{model_name_or_path}': pass a local model directory or one of "
                                     f"{sorted(PRESETS)} (no network access to fetch checkpoints)")
                if allow_random_init is None:
                    allow_random_init = name.startswith("tiny-") or os.environ.get("QST_ALLOW_RANDOM_INIT", "") == "1"
                if not allow_random_init:
                    raise FileNotFoundError(
                        f"no checkpoint for '{full}' on this machine (looked in cache_folder={cache_folder!r}, "
                        "$SENTENCE_TRANSFORMERS_HOME and the Hugging Face hub cache; there is no network access). Pass a "
                        "local model directory, or allow_random_init=True / QST_ALLOW_RANDOM_INIT=1 for seeded "
                        "random-init weights of that architecture.")
                cfg = PRESETS[name]
                logger.warning("'%s': seeded random-init weights of that architecture (seed %d), as requested.", name, seed)
        self.cfg = cfg
        if device is None:
            device = "cuda"
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.QstError(f"SentenceTransformer needs a HIP device (got '{device}'): this build has no CPU path")
        if not torch.cuda.is_available():
            raise _lib.QstError("SentenceTransformer needs a HIP device and none is visible: this build has no CPU path")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self._target_device = dev
        self._enc = HipEncoder(cfg, device=dev)
        self._enc.load_arena(arena if arena is not None else synthetic_params(cfg, seed=seed))
        self.max_seq_length = cfg.max_seq_length
        self.tokenizer = load_tokenizer(tok_dir, cfg)
        if self.tokenizer is None:
            self._synthetic_tokenizer = SyntheticTokenizer(cfg)
        # "bf16" (throughput) or "bf16x3" (fp32-class parity path) for no-grad forwards: encode() and evaluators
        self.inference_precision = "bf16"
        # "bf16", "bf16x3" or "fp8" for forwards that keep a graph (fit(precision=...), reset when fit() returns or raises): the
        # parity path "bf16x3" trains with fp32-class gradients as the reference's fp32 run does (training/main.py:142)
        self.training_precision = "bf16"
        self._live_graphs = 0          # training forwards whose backward has not run yet
        self._dp = None                # data-parallel state of a running fit(): {"group", "buckets", "overlap"}
        self._dp_works, self._dp_reduced = [], False
        self._anchor = nn.Parameter(torch.zeros((), device=dev))      # makes the Function's output require grad
        self._build_named_parameters()
        self.best_score = -9999999

    # ---- parameters as views of the arena
    def _build_named_parameters(self):
        root = _Holder()
        auto = _Holder()
        root.add_module("auto_model", auto)
        self.add_module("0", root)
        self._enc.ensure_train_state()
        views, gviews = self._enc.named_views(), self._enc.grad_views()
        for name, view in views.items():
            parts = name.split(".")
            mod = auto
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _Holder())
                mod = mod._modules[p]
            par = nn.Parameter(view, requires_grad=True)
            par.grad = gviews[name]
            mod.register_parameter(parts[-1], par)

    @property
    def device(self) -> torch.device:
        return self._target_device

    def to(self, *args, **kwargs):
        dev = None
        if args and isinstance(args[0], (str, torch.device)):
            dev = torch.device(args[0])
        dev = torch.device(kwargs["device"]) if "device" in kwargs else dev
        if dev is not None and (dev.type != "cuda" or (dev.index is not None and dev.index != self._target_device.index)):
            raise _lib.QstError(f"this model lives on {self._target_device}; moving the arena to {dev} is not supported")
        return self

    def get_sentence_embedding_dimension(self) -> int:
        return self.cfg.hidden_size

    def get_max_seq_length(self) -> int:
        return self.max_seq_length

    # ---- tokenisation
    def tokenize(self, texts: Union[List[str], List[Dict], List[Tuple[str, str]]]):
        return tokenize_texts(self.tokenizer if self.tokenizer is not None else self._synthetic_tokenizer, self.cfg,
                              texts, min(self.max_seq_length, 512))

    def token_lengths(self, examples: Iterable) -> List[int]:
        """Tokens the fused [4B, L] pass will spend on each example: the longest of its texts after truncation, special
        tokens included (input of data.LengthBucketBatchSampler). `examples`: InputExample objects or lists of texts."""
        max_len = min(self.max_seq_length, 512)
        tok = self.tokenizer if self.tokenizer is not None else self._synthetic_tokenizer
        out = []
        for ex in examples:
            texts = ex.texts if hasattr(ex, "texts") else ([ex] if isinstance(ex, str) else list(ex))
            out.append(max(count_tokens(tok, t, max_len) for t in texts))
        return out

    def smart_batching_collate(self, batch):
        """[InputExample] -> (list of per-column tokenised dicts, labels tensor) as ST 2.2.2 does."""
        num_texts = len(batch[0].texts)
        texts = [[] for _ in range(num_texts)]
        labels = []
        for example in batch:
            for idx, text in enumerate(example.texts):
                texts[idx].append(text)
            labels.append(example.label)
        labels = torch.tensor(labels)
        return [self.tokenize(col) for col in texts], labels

    # ---- forward
    def forward(self, features: Dict[str, torch.Tensor], **kwargs) -> Dict[str, torch.Tensor]:
        ids = features["input_ids"]
        mask = features["attention_mask"]
        types = features.get("token_type_ids") if self.cfg.type_vocab_size > 0 else None
        dev = self._target_device
        ids = ids.to(dev, torch.int64)
        mask = mask.to(dev, torch.int64)
        types = types.to(dev, torch.int64) if types is not None else None
        ids, mask, types, _ = HipEncoder.pad_inputs(ids, mask, types, self.cfg.pad_token_id)
        # train() mode = the HF modules drop (while fit() has dropout on), with or without autograd: the reference's
        # QuadrupletLossEvaluator computes its validation loss inside fit() under torch.no_grad() but never calls eval()
        # (/root/reference/models/evaluators.py:80-95), so that loss is taken WITH dropout there, and here. 2 = such a pass.
        training = (1 if torch.is_grad_enabled() else (2 if self._enc.dropout is not None else 0)) if self.training else 0
        with torch.cuda.device(dev):
            emb = _EncodeFn.apply(self._anchor, self, ids, mask, types, training)
        features.update({"sentence_embedding": emb})
        return features

    # ---- encode
    def encode(self, sentences: Union[str, List[str]], batch_size: int = 32, show_progress_bar: Optional[bool] = None,
               output_value: str = "sentence_embedding", convert_to_numpy: bool = True,
               convert_to_tensor: bool = False, device: Optional[str] = None, normalize_embeddings: bool = False,
               precision: Optional[str] = None):
        """precision: None = self.inference_precision; "bf16x3" = embeddings within rtol 1e-3/atol 1e-4 of fp32."""
        if output_value != "sentence_embedding":
            raise NotImplementedError("only output_value='sentence_embedding' is on the accelerated path")
        was_training, was_prec = self.training, self.inference_precision
        if precision is not None:
            self.inference_precision = precision
        self.eval()
        if convert_to_tensor:
            convert_to_numpy = False
        single = isinstance(sentences, str) or not hasattr(sentences, "__len__")
        if single:
            sentences = [sentences]
        order = np.argsort([-len(str(s)) for s in sentences])
        sorted_s = [sentences[i] for i in order]
        chunks = []
        with torch.no_grad():
            for start in range(0, len(sorted_s), batch_size):
                feats = self.tokenize(sorted_s[start:start + batch_size])
                emb = self.forward(feats)["sentence_embedding"].detach()
                if normalize_embeddings and emb.shape[0] > 0:
                    emb = emb.contiguous()
                    _lib.check(self._enc.lib.qst_normalize_rows(emb.data_ptr(), emb.shape[0], emb.shape[1], emb.data_ptr(),
                                                                _lib.current_stream_ptr()), "qst_normalize_rows")
                chunks.append(emb.cpu() if convert_to_numpy else emb)
        allemb = torch.cat(chunks, 0) if chunks else torch.zeros(0, self.cfg.hidden_size)
        inv = np.argsort(order)
        allemb = allemb[torch.as_tensor(inv, device=allemb.device)] if len(inv) else allemb
        if convert_to_numpy:
            allemb = allemb.numpy()
        if single:
            allemb = allemb[0]
        self.train(was_training)
        self.inference_precision = was_prec
        return allemb

    # ---- fit (kwargs = training/main.py:128-148)
    def fit(self, train_objectives: Iterable[Tuple[object, nn.Module]], evaluator=None, epochs: int = 1,
            steps_per_epoch=None, scheduler: str = "WarmupLinear", warmup_steps: int = 10000,
            optimizer_class: Type = torch.optim.AdamW, optimizer_params: Dict[str, object] = None,
            weight_decay: float = 0.01, evaluation_steps: int = 0, output_path: str = None,
            save_best_model: bool = True, max_grad_norm: float = 1, use_amp: bool = False,
            callback: Callable[[float, int, int], None] = None, show_progress_bar: bool = True,
            checkpoint_path: str = None, checkpoint_save_steps: int = 500, checkpoint_save_total_limit: int = 0,
            resume_from_checkpoint: str = None, data_parallel: Optional[str] = None, process_group=None,
            overlap_grad_reduce: bool = True, dropout="config", dropout_seed: int = 0, precision: str = "bf16"):
        """Same keyword set as sentence-transformers 2.2.2 `fit` (the reference passes all of them,
        training/main.py:128-148) plus `resume_from_checkpoint`: a checkpoint directory written by this method
        (weights + Adam moments + step counters; the reference's checkpoints hold weights only)
        from which training continues with the schedule where it stopped; and `precision`: "bf16" (default), "f16" (what
        use_amp=True selects: IEEE-half operands under a device-side GradScaler) or "bf16x3", the
        parity path -- forward and backward as split-bf16 x3 products with fp32 activations, gradients within 1e-4 of fp32
        autograd (the reference trains in fp32, training/main.py:142), dropout as on the bf16 path; single process; or "fp8"
        (BASELINE configs[4]): the forward's Linears on the fp8 matrix cores, the bf16 backward (dropout as in bf16).

        Data parallelism (one process per GPU, e.g. the unchanged training script under
        `python -m torch.distributed.run`): when torch.distributed is initialised with more than one rank, every step's
        gradients are summed over the ranks (RCCL all-reduce of the arena, per layer, overlapped with the rest of the
        backward: trainer.staged_backward), clipped on the GLOBAL norm and applied with 1/world, so replicas stay
        bit-identical. `data_parallel`:
          "split_batch" (default under torch.distributed) -- every rank's dataloader yields the SAME batch (same seed,
              as the reference's script sets it) and rank r trains on rows r::world of it: the run is the single-GPU run,
              step for step, with the batch spread over the GPUs;
          "per_rank_batches" -- every rank feeds its own batches (a DistributedSampler-style loader): the global batch
              is world x larger;
          "off" -- no exchange (every rank trains alone).
        Files (evaluator CSVs, best model, checkpoints) are written by rank 0 only; the evaluator itself runs on every
        rank so that score-driven control flow (early stopping) stays in step.

        Dropout: the reference's fit() runs the HF modules in train() mode, i.e. with the checkpoint config's
        hidden_dropout_prob / attention_probs_dropout_prob (HF default 0.1 / 0.1; training/main.py:128). `dropout`:
        "config" (default) = those values (0.1 / 0.1 when the model was built without a config.json), a float p or a pair
        (p_hidden, p_attn), 0 / None = off. Masks come from the library's counter-based generator (include/qst_kernels.h:
        QstDrop; same distribution as torch's, another random stream), seeded with dropout_seed + rank."""
        optimizer_params = dict(optimizer_params or {"lr": 2e-5})
        if optimizer_class not in (torch.optim.AdamW,):
            raise NotImplementedError(f"fit() drives the fused HIP AdamW; optimizer_class={optimizer_class} is not supported")
        lr = float(optimizer_params.get("lr", 2e-5))
        betas = tuple(optimizer_params.get("betas", (0.9, 0.999)))
        eps = float(optimizer_params.get("eps", 1e-8))
        # use_amp: the reference's reduced precision is fp16 autocast + GradScaler (training/main.py:142 passes the flag on); here
        # that is precision="f16" -- IEEE-half matrix-core operands, fp32 accumulation / residual stream / LayerNorm / softmax /
        # loss, and GradScaler's scale / unscale / skip / grow rules on the device (qst_clip_adamw_step_amp)
        if use_amp and precision == "bf16":
            precision = "f16"
        dataloaders = [dl for dl, _ in train_objectives]
        loss_models = [lm for _, lm in train_objectives]
        for dl in dataloaders:
            dl.collate_fn = self.smart_batching_collate
        for lm in loss_models:
            lm.to(self._target_device)
        self.best_score = -9999999
        if steps_per_epoch is None or steps_per_epoch == 0:
            steps_per_epoch = min(len(dl) for dl in dataloaders)
        t_total = int(steps_per_epoch * epochs)
        sched = scheduler.lower()
        if sched not in ("warmuplinear", "constantlr", "warmupconstant"):
            raise ValueError(f"Unknown scheduler {scheduler}")

        def lr_at(step):
            if sched == "constantlr":
                return lr
            if sched == "warmupconstant":
                return lr * min(1.0, float(step) / float(max(1, warmup_steps)))
            return warmup_linear_lr(lr, step, warmup_steps, t_total)

        if precision not in ("bf16", "f16", "f16w", "bf16x3", "fp8"):
            raise ValueError("fit(precision=...) is 'bf16', 'f16', 'f16w', 'bf16x3' or 'fp8'")
        amp = precision in ("f16", "f16w")
        if amp and sched != "warmuplinear" and sched != "constantlr":
            raise ValueError("precision='f16' / use_amp drives the device-side schedule: 'WarmupLinear' or 'constantlr'")
        if data_parallel not in (None, "off", "split_batch", "per_rank_batches"):
            raise ValueError(f"data_parallel={data_parallel!r}: expected 'split_batch', 'per_rank_batches' or 'off'")
        enc = self._enc
        enc.ensure_train_state()
        enc.grads.zero_()
        if dropout == "config":
            dropout = getattr(self, "_dropout_config", (0.1, 0.1))
        if dropout is None or dropout == 0:
            dropout = (0.0, 0.0)
        p_hidden, p_attn = (float(dropout), float(dropout)) if isinstance(dropout, (int, float)) else map(float, dropout)
        # ---- data parallelism
        import torch.distributed as dist
        world, rank = 1, 0
        if data_parallel != "off" and dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        dp_mode = (data_parallel or "split_batch") if world > 1 else "off"
        self._dp = None if world == 1 else {"group": process_group, "buckets": gradient_buckets(self.cfg),
                                            "overlap": bool(overlap_grad_reduce)}
        self._dp_works, self._dp_reduced, self._live_graphs = [], False, 0
        is_main = rank == 0
        # The evaluator runs on every rank (score-driven control flow must stay in step) and the reference's evaluators write
        # under output_path unconditionally (models/evaluators.py:81 joins it with a file name): ranks other than 0 get a
        # scratch directory of their own, removed when fit() returns -- never None.
        scratch_dir = None
        eval_out = output_path
        if not is_main and evaluator is not None and output_path is not None:
            scratch_dir = tempfile.mkdtemp(prefix=f"qst_eval_rank{rank}_")
            eval_out = scratch_dir
        self.training_precision = precision
        try:
            enc.set_dropout(p_hidden, p_attn, int(dropout_seed) + rank)
            if amp:
                enc.ensure_amp_scaler()
            global_step = 0
            if resume_from_checkpoint is not None:
                global_step = self._load_training_state(resume_from_checkpoint)
                enc.set_dropout_step(self._resume_dropout_step)     # the mask stream continues where the checkpoint left it
            self._fit_meta = {"scheduler": sched, "lr": lr, "warmup_steps": int(warmup_steps), "t_total": t_total}
            iters = [iter(dl) for dl in dataloaders]
            if global_step > 0:
                # resumed run: consume the batches the interrupted run already trained on (exact for unshuffled loaders)
                for i, dl in enumerate(dataloaders):
                    for _ in range(global_step % max(1, len(dl))):
                        next(iters[i])
            first_epoch, skip_steps = divmod(global_step, steps_per_epoch)
            for epoch in range(first_epoch, epochs):
                training_steps = skip_steps if epoch == first_epoch else 0
                for lm in loss_models:
                    lm.train()
                self._rebind_grads()
                for _ in range(steps_per_epoch - training_steps):
                    for idx, lm in enumerate(loss_models):
                        try:
                            data = next(iters[idx])
                        except StopIteration:
                            iters[idx] = iter(dataloaders[idx])
                            data = next(iters[idx])
                        features, labels = data
                        weight = 1.0
                        if dp_mode == "split_batch":
                            features, labels, n_total, n_mine = _shard_batch(features, labels, rank, world)
                            weight = float(n_mine) * world / float(max(1, n_total))   # mean over the GLOBAL batch after the 1/world
                        if _loss_reduction(lm) == "sum":
                            weight = float(world)
                        if labels.numel() > 0:
                            labels = labels.to(self._target_device)
                            features = [batch_to_device(f, self._target_device) for f in features]
                            loss_value = lm(features, labels)
                            if amp:                         # GradScaler.scale(loss): the scale is a device scalar
                                (loss_value * (enc.amp_scaler[0] * weight)).backward()
                            else:
                                (loss_value if weight == 1.0 else loss_value * weight).backward()
                        if world > 1:
                            if not self._dp_reduced:
                                # empty shard (the last batch had fewer rows than ranks), or a loss model that bypassed
                                # _EncodeFn: take part in the SAME sequence of all-reduces the other ranks issue from inside
                                # staged_backward -- per-layer slices in its order, or one arena-wide reduce without overlap
                                self._dp_works += allreduce_ranges(
                                    enc.grads, staged_reduce_order(self._dp["buckets"], enc.total, self._dp["overlap"]),
                                    process_group, async_op=self._dp["overlap"])
                            for w in self._dp_works:
                                w.wait()
                            self._dp_works, self._dp_reduced, self._live_graphs = [], False, 0
                        # clip_grad_norm_ + AdamW.step + zero_grad, one pass over the arena, norm stays on the device
                        if amp:
                            # scaler.unscale_ + clip + scaler.step + scaler.update; the learning rate follows the device's
                            # scheduler count (ST skips scheduler.step() whenever the scale changed)
                            enc.adamw_step_amp(lr, int(warmup_steps), t_total if sched == "warmuplinear" else 0, betas, eps,
                                               weight_decay, float(max_grad_norm), 1.0 / world)
                        else:
                            enc.adamw_step(lr_at(global_step), betas, eps, weight_decay, float(max_grad_norm), 1.0 / world)
                    training_steps += 1
                    global_step += 1
                    if evaluation_steps > 0 and training_steps % evaluation_steps == 0:
                        self._eval_during_training(evaluator, eval_out, save_best_model and is_main, epoch, training_steps, callback)
                        for lm in loss_models:
                            lm.train()
                    if checkpoint_path is not None and checkpoint_save_steps is not None and checkpoint_save_steps > 0 \
                            and global_step % checkpoint_save_steps == 0 and is_main:
                        self._save_checkpoint(checkpoint_path, checkpoint_save_total_limit, global_step)
                self._eval_during_training(evaluator, eval_out, save_best_model and is_main, epoch, -1, callback)
        finally:
            # whatever ended the loop (EarlyStoppingException derives from BaseException and must pass through untouched,
            # training/main.py:149): the scratch directory goes, and the model leaves train()-time state behind -- a later
            # forward must not silently run with this fit()'s dropout, precision or data-parallel group
            if scratch_dir is not None:
                shutil.rmtree(scratch_dir, ignore_errors=True)
            self._dp = None
            self._dp_works, self._dp_reduced, self._live_graphs = [], False, 0
            enc.set_dropout(0.0, 0.0)
            self.training_precision = "bf16"
        if evaluator is None and output_path is not None and is_main:
            self.save(output_path)
        if checkpoint_path is not None and is_main:
            self._save_checkpoint(checkpoint_path, checkpoint_save_total_limit, global_step)

    def _rebind_grads(self):
        gv = self._enc.grad_views()
        auto = self._modules["0"]._modules["auto_model"]
        for name, p in auto.named_parameters():
            if p.grad is None or p.grad.data_ptr() != gv[name].data_ptr():
                p.grad = gv[name]

    def _eval_during_training(self, evaluator, output_path, save_best_model, epoch, steps, callback):
        eval_path = output_path
        if output_path is not None:
            os.makedirs(output_path, exist_ok=True)
            eval_path = os.path.join(output_path, "eval")
            os.makedirs(eval_path, exist_ok=True)
        if evaluator is not None:
            score = evaluator(self, output_path=eval_path, epoch=epoch, steps=steps)
            if callback is not None:
                callback(score, epoch, steps)          # EarlyStoppingException(BaseException) propagates (callbacks.py:47)
            if score > self.best_score:
                self.best_score = score
                if save_best_model:
                    self.save(output_path)

    def _save_checkpoint(self, checkpoint_path, checkpoint_save_total_limit, step):
        path = os.path.join(checkpoint_path, str(step))
        self.save(path)
        # training state next to the ST model files: Adam moments (flat arenas) + counters, so a run can resume
        from safetensors.torch import save_file
        st = self._enc.optimizer_state()
        save_file({"exp_avg": st["exp_avg"].detach().cpu().contiguous(),
                   "exp_avg_sq": st["exp_avg_sq"].detach().cpu().contiguous()},
                  os.path.join(path, "training_state.safetensors"))
        meta = dict(getattr(self, "_fit_meta", {}))
        meta.update({"global_step": int(step), "opt_step": int(self._enc.opt_step), "best_score": float(self.best_score),
                     "dropout_step": int(self._enc.dropout_step)})
        with open(os.path.join(path, "training_state.json"), "w") as f:
            json.dump(meta, f, indent=2)
        if checkpoint_save_total_limit is not None and checkpoint_save_total_limit > 0:
            old = sorted(int(d) for d in os.listdir(checkpoint_path) if d.isdigit())
            for s in old[:-checkpoint_save_total_limit]:
                shutil.rmtree(os.path.join(checkpoint_path, str(s)), ignore_errors=True)

    def _load_training_state(self, path: str) -> int:
        """Weights, Adam moments and counters from a checkpoint directory written by fit(); returns its global step."""
        from safetensors.torch import load_file
        state_file = os.path.join(path, "training_state.safetensors")
        if not os.path.isfile(state_file):
            raise FileNotFoundError(f"{path} holds no training_state.safetensors (weights-only checkpoint: cannot resume)")
        with open(os.path.join(path, "training_state.json")) as f:
            meta = json.load(f)
        cfg, arena, _ = _load_model_dir(path)
        if (cfg.hidden_size, cfg.num_layers, cfg.vocab_size, cfg.arch) != (self.cfg.hidden_size, self.cfg.num_layers,
                                                                            self.cfg.vocab_size, self.cfg.arch):
            raise ValueError(f"{path}: checkpoint architecture differs from this model")
        self._enc.load_arena(arena)
        st = load_file(state_file)
        st["opt_step"] = torch.tensor([int(meta["opt_step"])], dtype=torch.int64)
        self._enc.load_optimizer_state(st)
        self.best_score = float(meta.get("best_score", self.best_score))
        self._resume_dropout_step = int(meta.get("dropout_step", 0))
        return int(meta["global_step"])

    # ---- save / load: ST model-directory layout (modules.json, config.json, model.safetensors, 1_Pooling/...)
    def save(self, path: str, model_name: Optional[str] = None, create_model_card: bool = False, **kwargs):
        if path is None:
            return
        os.makedirs(path, exist_ok=True)
        cfg = self.cfg
        from safetensors.torch import save_file
        tensors = OrderedDict((k, v.detach().cpu().contiguous().clone()) for k, v in self._enc.named_views().items())
        save_file(tensors, os.path.join(path, "model.safetensors"))
        hf = {"model_type": "bert" if cfg.arch == ARCH_BERT else "mpnet", "vocab_size": cfg.vocab_size,
              "hidden_size": cfg.hidden_size, "num_hidden_layers": cfg.num_layers,
              "num_attention_heads": cfg.num_heads, "intermediate_size": cfg.intermediate_size,
              "max_position_embeddings": cfg.max_position, "type_vocab_size": cfg.type_vocab_size,
              "layer_norm_eps": cfg.layer_norm_eps, "hidden_act": "gelu", "pad_token_id": cfg.pad_token_id,
              "relative_attention_num_buckets": cfg.rel_buckets,
              "hidden_dropout_prob": self._dropout_config[0], "attention_probs_dropout_prob": self._dropout_config[1]}
        json.dump(hf, open(os.path.join(path, "config.json"), "w"), indent=2)
        json.dump({"max_seq_length": self.max_seq_length, "do_lower_case": False},
                  open(os.path.join(path, "sentence_bert_config.json"), "w"), indent=2)
        modules = [{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
                   {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]
        os.makedirs(os.path.join(path, "1_Pooling"), exist_ok=True)
        json.dump({"word_embedding_dimension": cfg.hidden_size, "pooling_mode_cls_token": False,
                   "pooling_mode_mean_tokens": True, "pooling_mode_max_tokens": False,
                   "pooling_mode_mean_sqrt_len_tokens": False},
                  open(os.path.join(path, "1_Pooling", "config.json"), "w"), indent=2)
        if cfg.normalize:
            modules.append({"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"})
            os.makedirs(os.path.join(path, "2_Normalize"), exist_ok=True)
        json.dump(modules, open(os.path.join(path, "modules.json"), "w"), indent=2)
        if self.tokenizer is not None:
            self.tokenizer.save_pretrained(path)


def _shard_batch(features, labels, rank: int, world: int):
    """Rows rank::world of a collated batch ([dict of [B, L_k] tensors] x columns, labels [B]); a dict-of-columns batch
    (the reference's glue also accepts {'reference': ..., ...}, quadruplet_sentence_transformer.py:24-33) likewise."""
    n_total = int(labels.shape[0])
    sel = torch.arange(n_total)[rank::world]

    def cut(f):
        return {k: (v[sel] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n_total else v)
                for k, v in f.items()}
    if isinstance(features, dict):
        feats = {k: (cut(v) if isinstance(v, dict) else v) for k, v in features.items()}
    else:
        feats = [cut(f) for f in features]
    return feats, labels[sel], n_total, int(sel.numel())


def _loss_reduction(loss_model) -> str:
    for obj in (loss_model, getattr(loss_model, "_quadruplet_loss", None)):
        red = getattr(obj, "reduction", None)
        if isinstance(red, str):
            return red
    return "mean"


def _find_cached_model(name: str, cache_folder: Optional[str] = None) -> Optional[str]:
    """Directory of a named model among the places sentence-transformers 2.2.2 / huggingface_hub leave one:
    <cache>/<org>_<name> (ST's snapshot layout, org 'sentence-transformers' when the name has none), <cache>/<name>,
    and <hub cache>/models--<org>--<name>/snapshots/<rev>. None if there is no config.json + weights anywhere."""
    org, _, short = name.rpartition("/")
    orgs = [org] if org else ["sentence-transformers", ""]
    roots = [cache_folder, os.environ.get("SENTENCE_TRANSFORMERS_HOME"),
             os.path.join(os.environ.get("TORCH_HOME", os.path.join(os.path.expanduser("~"), ".cache", "torch")),
                          "sentence_transformers")]
    cands = []
    for root in roots:
        if not root:
            continue
        for o in orgs:
            cands.append(os.path.join(root, f"{o}_{short}" if o else short))
            if o:
                cands.append(os.path.join(root, o, short))
    hub = os.environ.get("HF_HUB_CACHE") or os.path.join(
        os.environ.get("HF_HOME", os.path.join(os.path.expanduser("~"), ".cache", "huggingface")), "hub")
    for root in [hub, cache_folder]:
        if not root:
            continue
        for o in orgs:
            snaps = os.path.join(root, f"models--{o}--{short}" if o else f"models--{short}", "snapshots")
            if os.path.isdir(snaps):
                cands += sorted((os.path.join(snaps, d) for d in os.listdir(snaps)), key=os.path.getmtime, reverse=True)
    for c in cands:
        if os.path.isfile(os.path.join(c, "config.json")) and any(
                os.path.isfile(os.path.join(c, f)) for f in ("model.safetensors", "pytorch_model.bin")):
            return c
    return None


def _dropout_from_config(path: str):
    """(hidden_dropout_prob, attention_probs_dropout_prob) of a model directory's config.json, HF defaults 0.1 / 0.1."""
    try:
        hf = json.load(open(os.path.join(path, "config.json")))
    except (OSError, ValueError):
        return (0.1, 0.1)
    return (float(hf.get("hidden_dropout_prob", 0.1)), float(hf.get("attention_probs_dropout_prob", 0.1)))


def _load_model_dir(path: str):
    """Read an ST/HF model directory (as written by save() above or by sentence-transformers)."""
    cfg_path = os.path.join(path, "config.json")
    if not os.path.exists(cfg_path):
        raise FileNotFoundError(f"{path} has no config.json")
    hf = json.load(open(cfg_path))
    mt = hf.get("model_type", "bert")
    if mt not in ("bert", "mpnet"):
        raise NotImplementedError(f"model_type '{mt}' is not on the accelerated path (bert, mpnet)")
    normalize, max_seq = False, min(512, int(hf.get("max_position_embeddings", 512)))
    mj = os.path.join(path, "modules.json")
    if os.path.exists(mj):
        mods = json.load(open(mj))
        normalize = any(m.get("type", "").endswith("Normalize") for m in mods)
        for m in mods:
            kind = m.get("type", "").rsplit(".", 1)[-1]
            if kind not in ("Transformer", "Pooling", "Normalize"):
                raise NotImplementedError(f"{path}: module '{m.get('type')}' is not on the accelerated path "
                                          "(Transformer -> Pooling(mean) -> [Normalize])")
            if kind == "Pooling":
                pc = os.path.join(path, m.get("path", "1_Pooling"), "config.json")
                if os.path.exists(pc):
                    pool = json.load(open(pc))
                    on = sorted(k for k, v in pool.items() if k.startswith("pooling_mode_") and v)
                    if on != ["pooling_mode_mean_tokens"]:
                        raise NotImplementedError(f"{path}: pooling {on} -- only mean-token pooling is implemented "
                                                  "(pool_norm_fwd/bwd); this checkpoint would be pooled wrongly")
    sb = os.path.join(path, "sentence_bert_config.json")
    if os.path.exists(sb):
        max_seq = int(json.load(open(sb)).get("max_seq_length", max_seq))
    arch = ARCH_BERT if mt == "bert" else ARCH_MPNET
    cfg = EncoderConfig(arch=arch, vocab_size=hf["vocab_size"], hidden_size=hf["hidden_size"],
                        num_layers=hf["num_hidden_layers"], num_heads=hf["num_attention_heads"],
                        intermediate_size=hf["intermediate_size"], max_position=hf["max_position_embeddings"],
                        type_vocab_size=0 if arch == ARCH_MPNET else hf.get("type_vocab_size", 2),
                        layer_norm_eps=hf.get("layer_norm_eps", 1e-12), normalize=normalize, max_seq_length=max_seq,
                        rel_buckets=hf.get("relative_attention_num_buckets", 32),
                        pad_token_id=hf.get("pad_token_id", 1 if arch == ARCH_MPNET else 0))
    st_path = os.path.join(path, "model.safetensors")
    if os.path.exists(st_path):
        from safetensors.torch import load_file
        sd = load_file(st_path)
    elif os.path.exists(os.path.join(path, "pytorch_model.bin")):
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    else:
        raise FileNotFoundError(f"{path} has neither model.safetensors nor pytorch_model.bin")
    sd = {k[len("bert."):] if k.startswith("bert.") else (k[len("mpnet."):] if k.startswith("mpnet.") else k): v
          for k, v in sd.items()}
    from .config import build_layout
    segs, total = build_layout(cfg)
    so = {s.name: s for s in segs}
    arena = np.zeros(total, np.float32)
    missing = []
    for name, seg, off, shape in hf_param_views(cfg):
        if name not in sd:
            missing.append(name)
            continue
        s = so[seg]
        n = int(np.prod(shape))
        arena[s.offset + off:s.offset + off + n] = sd[name].to(torch.float32).numpy().reshape(-1)
    if missing:
        raise KeyError(f"{path}: checkpoint lacks {len(missing)} tensors, e.g. {missing[:3]}")
    return cfg, arena, path
