"""Shared helpers of the test-suite: seeded datasets in every storage kind and reference-built index images."""
from __future__ import annotations

import numpy as np

from oracle import oraclebind, refbind

NP_DTYPE = {"f32": np.float32, "f16": np.float16, "i8": np.int8, "b1": np.uint8}


def make_vectors(n: int, ndim: int, dtype: str, seed: int, clustered: bool = True) -> np.ndarray:
    """Seeded vectors in the storage kind. `clustered` = low-rank latent + noise so that HNSW has structure."""
    rng = np.random.default_rng(seed)
    if dtype == "b1":
        if clustered and ndim >= 16:
            latent = rng.standard_normal((n, 8)) @ rng.standard_normal((8, ndim))
            bits = (latent + 0.3 * rng.standard_normal((n, ndim))) > 0
        else:
            bits = rng.integers(0, 2, (n, ndim)).astype(bool)
        return np.packbits(bits, axis=1)  # MSB first, like cast_to_b1x8_gt (index_plugins.hpp:1139-1158)
    if clustered and ndim >= 8:
        rank = max(2, min(16, ndim // 4))
        x = rng.standard_normal((n, rank)) @ rng.standard_normal((rank, ndim)) + 0.05 * rng.standard_normal((n, ndim))
    else:
        x = rng.standard_normal((n, ndim))
    if dtype == "i8":
        x = x / np.abs(x).max() * 100.0
        return np.clip(np.rint(x), -127, 127).astype(np.int8)
    return x.astype(NP_DTYPE[dtype])


def build_image(n: int, ndim: int, metric: str, dtype: str, seed: int = 1, connectivity: int = 16,
                expansion_add: int = 128, threads: int = 1, clustered: bool = True, keys=None, remove=()):
    """Builds an index with the REAL reference (single-threaded ⇒ deterministic graph) and serializes it.
    → (image bytes as np.uint8, vectors, RefIndex)."""
    vectors = make_vectors(n, ndim, dtype, seed, clustered)
    index = refbind.RefIndex(ndim, metric, dtype, connectivity=connectivity, expansion_add=expansion_add)
    if keys is None:
        keys = np.arange(n, dtype=np.uint64) + 1000
    if n:
        added = index.add(keys, vectors, threads=threads)
        assert added == n
    for key in remove:
        index.remove(int(key))
    return index.save_buffer(), vectors, index


def same_float_bits(a: np.ndarray, b: np.ndarray) -> bool:
    return np.array_equal(np.asarray(a, dtype=np.float32).view(np.uint32), np.asarray(b, dtype=np.float32).view(np.uint32))


def oracle_search(image: np.ndarray, queries: np.ndarray, k: int, dtype: str, expansion: int = 64, lanes: int = 0,
                  exact: bool = False):
    return oraclebind.OracleIndex(image).search(queries, k, dtype=dtype, expansion=expansion, lanes=lanes, exact=exact)
