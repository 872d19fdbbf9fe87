"""Synthetic Amazon-toys-shaped interaction data (there is no network for the real blobs, and the
reference's own train/val/test .pth files are missing from its tree: /root/reference/.MISSING_LARGE_BLOBS).

Shape statistics are those of the shipped dataset/amazon-toys/toy/seq2pat_data.pth:
  N = 11 925 items incl. PAD id 0, U = 19 412 training rows, L = 50, post-padded with 0,
  train seqlen histogram below (mean 5.47, median 3, p90 11, p99 37, max 47 -> 10.95 % of B*L valid),
  Zipf-like item popularity (top 1 % of items = 8.8 % of interactions, top 10 % = 37.6 %).
Targets are the inputs shifted by one (seq2seq training rows, dataset/preprocess_amazon.ipynb cell 20).
"""
from __future__ import annotations

import numpy as np

TOYS_N_ITEMS = 11925
TOYS_N_ROWS = 19412
MAX_SEQ_LEN = 50
# number of training rows with seqlen == i (i = 0..47) in amazon-toys
TOYS_SEQLEN_HIST = [0, 0, 6579, 3822, 2272, 1516, 1035, 798, 548, 422, 323, 309, 233, 171, 157, 140, 113, 97, 69, 63,
                    55, 63, 39, 37, 44, 46, 39, 33, 27, 21, 24, 22, 18, 22, 22, 13, 15, 17, 9, 13, 5, 9, 9, 5, 8, 4, 7, 119]


def item_popularity(n_items: int, exponent: float = 0.66, shift: float = 50.0) -> np.ndarray:
    """P(item of popularity rank r) ~ (r + shift)^-exponent, r = 1..n_items-1 (shifted power law tuned
    to toys: top 1 % of items ~ 9 % of interactions, top 10 % ~ 37 %)."""
    r = np.arange(1, n_items, dtype=np.float64)
    p = (r + shift) ** (-exponent)
    return p / p.sum()


def make_rows(n_rows: int = TOYS_N_ROWS, n_items: int = TOYS_N_ITEMS, L: int = MAX_SEQ_LEN, seed: int = 2024,
              dense: bool = False, markov: float = 0.0):
    """Returns dict of int64 numpy arrays in the layout of SeparateDataset.unpack
    (/root/reference data/dataset.py:79-91): user_id[U], in_item_id[U,L], item_id[U,L], seqlen[U],
    label[U,L], domain_id[U,L]."""
    rng = np.random.default_rng(seed)
    if dense:
        seqlen = np.full(n_rows, L, dtype=np.int64)
    else:
        h = np.asarray(TOYS_SEQLEN_HIST, dtype=np.float64)
        seqlen = rng.choice(len(h), size=n_rows, p=h / h.sum()).astype(np.int64)
        seqlen = np.minimum(seqlen, L)
    pop = item_popularity(n_items)
    perm = rng.permutation(n_items - 1) + 1                     # popularity rank -> item id
    total = int(seqlen.sum()) + n_rows
    draws = perm[rng.choice(n_items - 1, size=total, p=pop)]
    in_item = np.zeros((n_rows, L), dtype=np.int64)
    tgt = np.zeros((n_rows, L), dtype=np.int64)
    label = np.zeros((n_rows, L), dtype=np.int64)
    # markov > 0: with that probability the next item is a fixed function of the current one (the same function for every split
    # of a dataset) — a signal a sequential recommender can learn, for end-to-end sanity checks; 0 = i.i.d. popularity draws
    succ = np.random.default_rng(12345).permutation(n_items - 1) + 1 if markov > 0 else None
    o = 0
    for u in range(n_rows):
        n = int(seqlen[u])
        s = draws[o:o + n + 1].copy()
        if succ is not None:
            follow = rng.random(n + 1) < markov
            for t in range(1, n + 1):
                if follow[t]:
                    s[t] = succ[s[t - 1] - 1]
        o += n + 1
        in_item[u, :n] = s[:n]
        tgt[u, :n] = s[1:n + 1]
        label[u, :n] = 1
    return {
        "user_id": np.arange(1, n_rows + 1, dtype=np.int64),
        "in_item_id": in_item,
        "item_id": tgt,
        "seqlen": seqlen,
        "label": label,
        "domain_id": np.zeros((n_rows, L), dtype=np.int64),
    }
