#!/usr/bin/env python3
"""Where the wide exact tile differs from the bit-exact kernel on one of the parity configurations (diagnosis)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tests import util
from usearch_amd import Index
os.environ["USEARCH_AMD_EXACT_TILE"] = "256"
metric, dtype, ndim, n, k = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
removed = np.arange(5, n, 13) + 1000
image, vectors, _ = util.build_image(n, ndim, metric, dtype, seed=93, remove=removed[:200], expansion_add=16, connectivity=4)
queries = util.make_vectors(700, ndim, dtype, seed=94)
queries[:20] = vectors[:20]
index = Index.restore(image)
exact = index.search(queries, k, exact=True)
for attempt in range(3):
    tiled = index.search(queries, k, exact="tiled")
    bad = np.argwhere(exact.keys != tiled.keys)
    print(f"attempt {attempt}: {len(bad)} cells differ in {len(set(bad[:, 0]))} queries; counts equal {np.array_equal(exact.counts, tiled.counts)}")
    for q, p in bad[:12]:
        print(f"  query {q} (wave {q % 256 // 32}, tile {q // 256}) position {p}: exact key {exact.keys[q, p]} d {exact.distances[q, p]:.6f} | tiled key {tiled.keys[q, p]} d {tiled.distances[q, p]:.6f}")
    missing = [int(key) for q in sorted(set(bad[:, 0]))[:6] for key in exact.keys[q] if key not in tiled.keys[q]]
    print("  keys the tile misses:", missing[:12], "-> slots", [m - 1000 for m in missing[:12]])
