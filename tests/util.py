"""Shared helpers of the test-suite: seeded datasets in every storage kind and reference-built index images."""
from __future__ import annotations

import numpy as np

from oracle import oraclebind, refbind

NP_DTYPE = {"f32": np.float32, "f64": np.float64, "f16": np.float16, "i8": np.int8, "b1": np.uint8, "bf16": np.uint16}


def to_bf16(x: np.ndarray) -> np.ndarray:
    """float → bf16 bit patterns the way the reference narrows (truncation, index_plugins.hpp:453-469); numpy has no
    bfloat16, so bf16 rows are uint16 arrays."""
    return (np.ascontiguousarray(x, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def make_vectors(n: int, ndim: int, dtype: str, seed: int, clustered: bool = True, metric: str = "") -> np.ndarray:
    """Seeded vectors in the storage kind. `clustered` = low-rank latent + noise so that HNSW has structure.
    `metric` shapes the values where the metric has a domain: probability histograms for divergence, (latitude,
    longitude) in degrees for haversine."""
    rng = np.random.default_rng(seed)
    if metric == "haversine":
        assert ndim == 2
        x = np.stack([rng.uniform(-80, 80, n), rng.uniform(-180, 180, n)], axis=1)
        return x.astype(NP_DTYPE[dtype])
    if metric == "divergence":
        rank = max(2, min(8, ndim // 4))
        x = np.exp(0.7 * (rng.standard_normal((n, rank)) @ rng.standard_normal((rank, ndim))))
        x /= x.sum(axis=1, keepdims=True)
        return to_bf16(x) if dtype == "bf16" else x.astype(NP_DTYPE[dtype])
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
    return to_bf16(x) if dtype == "bf16" else x.astype(NP_DTYPE[dtype])


def build_image(n: int, ndim: int, metric: str, dtype: str, seed: int = 1, connectivity: int = 16,
                expansion_add: int = 128, threads: int = 1, clustered: bool = True, keys=None, remove=(), vectors=None):
    """Builds an index with the REAL reference (single-threaded ⇒ deterministic graph) and serializes it.
    → (image bytes as np.uint8, vectors, RefIndex)."""
    if vectors is None:
        vectors = make_vectors(n, ndim, dtype, seed, clustered, metric=metric)
    index = refbind.RefIndex(ndim, metric, dtype, connectivity=connectivity, expansion_add=expansion_add)
    if keys is None:
        keys = np.arange(n, dtype=np.uint64) + 1000
    if n:
        added = index.add(keys, vectors, threads=threads)
        assert added == n
    for key in remove:
        index.remove(int(key))
    return index.save_buffer(), vectors, index


def exact_pair(metric: str, dtype: str) -> bool:
    """Pairs whose distances are integer-valued, or one IEEE division of exact integers: the reference, the oracle and the
    GPU must agree bit for bit — keys, distances, counts and both traversal counters, ties included."""
    return dtype == "b1" or (dtype == "i8" and metric in ("l2sq", "ip"))


def layout_exact(metric: str) -> bool:
    """Metrics without a transcendental: the GPU result is bit-identical to the oracle run in the kernels' summation layout.
    divergence (log) and haversine (sin, cos, asin) go through two different math libraries — tolerance there."""
    return metric not in ("divergence", "haversine")


def tolerance(dtype: str) -> float:
    """|d - d_ref| <= tolerance · max(1, |d_ref|) for the float pairs (BASELINE.json north_star / SURVEY §8d)."""
    return 2e-3 if dtype in ("f16", "bf16") else 1e-5


def separated_positions(reference_distances: np.ndarray, reference_counts: np.ndarray, k: int, dtype: str) -> np.ndarray:
    """SURVEY §8(d)'s label rule made checkable. `reference_distances` are the reference's k + 1 nearest per query (one more
    than asked, so that the k-th result has a right-hand neighbour). → bool [Q, k]: position i is SEPARATED when the
    reference's own neighbouring distances (i - 1 and i + 1, as far as they exist) are farther from distance i than twice the
    stated tolerance — both sides' rounding together cannot reorder such a position, so its label must be identical."""
    rd = np.asarray(reference_distances, dtype=np.float64)
    q, width = rd.shape
    assert width >= k + 1 or width == k, "pass the reference's k + 1 nearest"
    found = np.arange(width)[None, :] < np.asarray(reference_counts)[:, None]
    gap = 2.0 * tolerance(dtype) * np.maximum(1.0, np.abs(np.where(found, rd, 0.0)))
    separated = found.copy()
    left = np.abs(rd[:, 1:] - rd[:, :-1])  # |d[i+1] - d[i]|
    both = found[:, 1:] & found[:, :-1]
    close = both & ~(left > np.maximum(gap[:, 1:], gap[:, :-1]))
    separated[:, 1:] &= ~close   # too close to its left neighbour
    separated[:, :-1] &= ~close  # too close to its right neighbour
    return separated[:, :k]


def assert_float_parity(got_keys, got_distances, got_counts, reference_search, queries, k: int, dtype: str, what: str = ""):
    """The float-pair bar against the REAL reference: counts equal; every distance within the stated tolerance; labels
    IDENTICAL at every separated position (see `separated_positions`). `reference_search(queries, k)` → the reference's
    (keys, distances, counts, …). Returns (share of positions that are separated, label agreement on ALL positions)."""
    rkeys1, rdists1, rcounts1 = reference_search(queries, k + 1)[:3]
    rkeys, rdists, rcounts = reference_search(queries, k)[:3]
    assert np.array_equal(got_counts, rcounts), f"{what}: counts differ from the reference's"
    found = np.arange(k)[None, :] < np.asarray(rcounts)[:, None]
    scale = np.maximum(1.0, np.abs(np.where(found, rdists, 0)))
    error = np.abs(np.where(found, np.asarray(got_distances, dtype=np.float64) - rdists, 0))
    assert np.all(error <= tolerance(dtype) * scale), f"{what}: a distance is off by {error.max():.3g}"
    separated = separated_positions(rdists1, rcounts1, k, dtype) & found
    # the reference's k and k + 1 searches agree on the first k wherever THEY are separated (same traversal, same ef >= k + 1
    # or not — if they did not, the position says nothing about the GPU)
    separated &= rkeys1[:, :k] == rkeys
    wrong = separated & (np.asarray(got_keys) != rkeys)
    assert not wrong.any(), (f"{what}: {int(wrong.sum())} of {int(separated.sum())} separated positions carry another label than "
                             f"the reference's; first at query {np.argwhere(wrong)[0][0]} position {np.argwhere(wrong)[0][1]}")
    return float(separated.sum() / max(1, found.sum())), float(((np.asarray(got_keys) == rkeys) | ~found).mean())


def same_float_bits(a: np.ndarray, b: np.ndarray) -> bool:
    return np.array_equal(np.asarray(a, dtype=np.float32).view(np.uint32), np.asarray(b, dtype=np.float32).view(np.uint32))


def oracle_search(image: np.ndarray, queries: np.ndarray, k: int, dtype: str, expansion: int = 64, lanes: int = 0,
                  exact: bool = False, frontier_in_top: bool = False, threads: int = 1):
    """`frontier_in_top`: restate the engine's heap-less frontier (kernels.hpp frontier_top_k) instead of the reference's heap;
    GPU tests pass what the engine reports it ran (`stats.frontier == 2`). `threads` > 1 splits the batch over that many host
    threads (queries are independent; the oracle's mode switch is thread-local, its index read-only)."""
    oracle = oraclebind.OracleIndex(image)
    if threads <= 1 or len(queries) < 2 * threads:
        return oracle.search(queries, k, dtype=dtype, expansion=expansion, lanes=lanes, exact=exact, frontier_in_top=frontier_in_top)
    from concurrent.futures import ThreadPoolExecutor
    bounds = np.linspace(0, len(queries), threads + 1).astype(int)
    with ThreadPoolExecutor(threads) as pool:
        parts = list(pool.map(lambda i: oracle.search(queries[bounds[i]:bounds[i + 1]], k, dtype=dtype, expansion=expansion, lanes=lanes,
                                                      exact=exact, frontier_in_top=frontier_in_top), range(threads)))
    return tuple(np.concatenate([part[field] for part in parts]) for field in range(5))


def with_64_bit_dimensions(image: np.ndarray) -> np.ndarray:
    """The same index as the reference writes it under `serialization_config_t::use_64_bit_dimensions`
    (index_dense.hpp:1006-1024): the matrix announced by two u64 instead of two u32, everything else unchanged."""
    rows, cols = np.frombuffer(image[:8].tobytes(), dtype=np.uint32)
    head = np.frombuffer(np.array([rows, cols], dtype=np.uint64).tobytes(), dtype=np.uint8)
    return np.concatenate([head, image[8:]])


def without_vectors(image: np.ndarray) -> np.ndarray:
    """... and under `exclude_vectors` (index_dense.hpp:1004): the file starts with the 64-byte head."""
    rows, cols = np.frombuffer(image[:8].tobytes(), dtype=np.uint32)
    return image[8 + int(rows) * int(cols):].copy()


def with_40_bit_slots(image: np.ndarray) -> np.ndarray:
    """The same index as an `index_dense_gt<u64, uint40_t>` would write it: every neighbour slot on the node tapes as the 5-byte
    `uint40_t` of index.hpp:969-1031 (the compressed-slot kind of `index_dense_big_t`, index_dense.hpp:2230) and the head's
    slot-kind byte set to u40_k (index_plugins.hpp:142). Transcoded here because the reference only instantiates that slot type
    together with 128-bit keys."""
    rows, cols = (int(v) for v in np.frombuffer(image[:8].tobytes(), dtype=np.uint32))
    head = 8 + rows * cols
    out = [image[:head + 64].copy()]
    out[0][head + 16] = 2  # u40_k
    size, m, m0 = (int(v) for v in np.frombuffer(image[head + 64:head + 88].tobytes(), dtype=np.uint64))
    levels_at = head + 64 + 40
    out.append(image[head + 64:levels_at + 2 * size])
    levels = np.frombuffer(image[levels_at:levels_at + 2 * size].tobytes(), dtype=np.int16)
    at = levels_at + 2 * size
    raw = image.tobytes()
    pieces = []
    for level in levels:
        pieces.append(raw[at:at + 10])  # key, level
        at += 10
        for cells in [m0] + [m] * int(level):
            pieces.append(raw[at:at + 4])  # count
            slots = np.frombuffer(raw[at + 4:at + 4 + 4 * cells], dtype=np.uint32)
            wide = np.zeros((cells, 5), dtype=np.uint8)
            wide[:, :4] = slots.view(np.uint8).reshape(cells, 4)
            pieces.append(wide.tobytes())
            at += 4 + 4 * cells
    assert at == len(raw), "the image does not end with its last tape"
    out.append(np.frombuffer(b"".join(pieces), dtype=np.uint8))
    return np.concatenate(out)


def with_uuid_keys_announced(image: np.ndarray) -> np.ndarray:
    """The head of an `index_dense_big_t` file: key kind uuid_k (index_plugins.hpp:143). (Only the head: the loader must refuse it
    by name before it looks at a tape.)"""
    rows, cols = (int(v) for v in np.frombuffer(image[:8].tobytes(), dtype=np.uint32))
    out = image.copy()
    out[8 + rows * cols + 15] = 3
    return out


def mapped_hip_runtime() -> str:
    """Path of the HIP runtime this process has mapped (the engine's, shared with torch when torch is installed —
    usearch_amd/index.py `_share_hip_runtime`): a test that calls the runtime directly must not pull in a second copy."""
    import usearch_amd.index
    usearch_amd.index.library()
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return line.split()[-1]
    raise RuntimeError("no HIP runtime is mapped")
