"""Batch construction for the real-text fit() driver.

The fused quadruplet pass pads all 4*B sequences of a batch to the longest one (rounded up to a multiple of 32), so a
batch costs 4 * B * L_max tokens of encoder work whatever the other lengths are. LengthBucketBatchSampler groups
examples of similar length: the reference's `DataLoader(dataset, shuffle=True, batch_size=B)`
(/root/reference/training/main.py:42-44) becomes `DataLoader(dataset, batch_sampler=LengthBucketBatchSampler(...))`,
with fit() and smart_batching_collate unchanged."""
from __future__ import annotations

from typing import Iterator, List, Sequence

import numpy as np


class LengthBucketBatchSampler:
    """Yields lists of dataset indices. Every epoch: shuffle, cut into pools of `pool_batches` batches, sort each pool by
    length, cut it into batches, then shuffle the order of all batches -- batches hold examples of similar length while
    their composition and order still change from epoch to epoch (a full sort would fix both).

    lengths: tokens per example (SentenceTransformer.token_lengths). drop_last as torch's BatchSampler."""

    def __init__(self, lengths: Sequence[int], batch_size: int, shuffle: bool = True, seed: int = 14, pool_batches: int = 50,
                 drop_last: bool = False):
        if batch_size <= 0 or pool_batches <= 0:
            raise ValueError("batch_size and pool_batches must be positive")
        self.lengths = np.asarray(lengths, dtype=np.int64)
        self.batch_size, self.shuffle, self.seed = int(batch_size), bool(shuffle), int(seed)
        self.pool_batches, self.drop_last = int(pool_batches), bool(drop_last)
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def __len__(self) -> int:
        n = len(self.lengths)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[List[int]]:
        n = len(self.lengths)
        rng = np.random.default_rng(self.seed + self.epoch)
        order = rng.permutation(n) if self.shuffle else np.arange(n)
        pool = self.batch_size * self.pool_batches
        batches = []
        for p0 in range(0, n, pool):
            idx = order[p0:p0 + pool]
            idx = idx[np.argsort(self.lengths[idx], kind="stable")]
            for b0 in range(0, len(idx), self.batch_size):
                b = idx[b0:b0 + self.batch_size]
                if len(b) == self.batch_size or not self.drop_last:
                    batches.append(b.tolist())
        if self.shuffle:
            batches = [batches[i] for i in rng.permutation(len(batches))]
        self.epoch += 1
        return iter(batches)


def padded_tokens(lengths: Sequence[int], batches: Sequence[Sequence[int]], multiple: int = 32) -> int:
    """Encoder token rows a list of batches costs: per batch, size x the longest example rounded up to `multiple`."""
    L = np.asarray(lengths)
    return int(sum(len(b) * (-(-int(L[list(b)].max()) // multiple) * multiple) for b in batches if len(b)))
