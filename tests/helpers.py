"""Shared test helpers: tiny model builders and synthetic batches (seeded)."""
import numpy as np


def tiny_groups(N, R, rng, overlap=True):
    """Ragged, possibly overlapping voxel groups (layers.py:13)."""
    perm = rng.permutation(N)
    cuts = np.sort(rng.choice(np.arange(1, N), size=R - 1, replace=False))
    groups = [np.sort(g) for g in np.split(perm, cuts)]
    if overlap:
        groups = [np.unique(np.concatenate([g, rng.choice(N, size=2)])) for g in groups]
    return groups


def synth_batch(B, N, T, V, U, rng, min_len=2, dtype=np.float32, zero_first=False):
    """Synthetic batch with the reference's tuple layout
    (data_generator_guse.py:156-171): ((betas, cap, a0, c0), target_ids)."""
    x = rng.standard_normal((B, N)).astype(dtype)
    cap = np.zeros((B, T), np.int32)
    for b in range(B):
        L = int(rng.integers(min_len, T))          # tokens incl. start, excl. end
        cap[b, 0] = 1
        cap[b, 1:L] = rng.integers(3, V, size=L - 1)
        if L < T:
            cap[b, L] = 2
    if zero_first:
        cap[0, 0] = 0
    tgt = np.zeros_like(cap)
    tgt[:, :-1] = cap[:, 1:]
    a0 = np.zeros((B, U), dtype)
    return (x, cap, a0, a0.copy()), tgt
