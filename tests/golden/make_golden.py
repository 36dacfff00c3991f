#!/usr/bin/env python3
"""tests/golden/make_golden.py — regenerates the committed golden fixtures from the REAL reference.

Run where /root/reference is mounted (it compiles `oracle/_ref/libusearch_ref.so` from the reference's own sources through
`oracle/Makefile` if needed):

    python tests/golden/make_golden.py [--all]

Each `tests/golden/<name>.npz` holds a small index image serialized by the reference (`usearch_save_buffer`), seeded
queries, and what the reference itself answers for them (`index_dense_gt::search`, single-threaded): keys, distances,
counts and the two traversal counters, plus its exact-search answer. The GPU box has no /root/reference — these files are
how the reference's behaviour travels there. `kat.json` restates the literal known-answer vectors of the reference's own
test-suites with their file:line.
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CASES = [
    # name, metric, dtype, ndim, n, connectivity, k, expansion, queries
    ("cos_f32_24", "cos", "f32", 24, 400, 16, 10, 64, 32),
    ("cos_f16_96", "cos", "f16", 96, 350, 16, 10, 64, 24),
    ("l2sq_f32_3", "l2sq", "f32", 3, 200, 3, 5, 16, 32),
    ("ip_f16_40", "ip", "f16", 40, 300, 8, 7, 32, 24),
    ("l2sq_i8_96", "l2sq", "i8", 96, 500, 16, 10, 64, 32),
    ("cos_i8_33", "cos", "i8", 33, 300, 16, 10, 64, 24),
    ("hamming_b1_128", "hamming", "b1", 128, 800, 16, 10, 64, 48),
    ("hamming_b1_72", "hamming", "b1", 72, 300, 4, 3, 8, 24),
    # the rest of the reference's metric x scalar dispatch table (index_plugins.hpp:1930-2008)
    ("cos_bf16_64", "cos", "bf16", 64, 300, 16, 10, 64, 24),
    ("l2sq_f64_16", "l2sq", "f64", 16, 300, 8, 5, 32, 24),
    ("pearson_f32_24", "pearson", "f32", 24, 300, 16, 10, 64, 24),
    ("pearson_i8_40", "pearson", "i8", 40, 300, 16, 10, 64, 24),
    ("divergence_f32_32", "divergence", "f32", 32, 300, 16, 10, 64, 24),
    ("haversine_f32_2", "haversine", "f32", 2, 400, 16, 10, 64, 32),
    ("tanimoto_b1_128", "tanimoto", "b1", 128, 500, 16, 10, 64, 32),
    ("sorensen_b1_72", "sorensen", "b1", 72, 300, 8, 5, 32, 24),
]


def main():
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libusearch_ref.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    from tests import util
    regenerate_all = "--all" in sys.argv  # default: only the fixtures that are not there yet
    for name, metric, dtype, ndim, n, connectivity, k, expansion, nq in CASES:
        if os.path.exists(os.path.join(HERE, f"{name}.npz")) and not regenerate_all:
            continue
        removed = np.arange(5, n, 7) + 1000 if name == "cos_f32_24" else ()
        image, vectors, index = util.build_image(n, ndim, metric, dtype, seed=101, connectivity=connectivity,
                                                 remove=removed)
        queries = util.make_vectors(nq, ndim, dtype, seed=202, metric=metric)
        queries[: nq // 4] = vectors[: nq // 4]
        index.expansion_search = expansion
        keys, distances, counts, visited, computed = index.search(queries, k, dtype=dtype, threads=1)
        exact_keys, exact_distances, exact_counts, *_ = index.search(queries, k, dtype=dtype, exact=True, threads=1)
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, image=image, queries=queries, keys=keys, distances=distances, counts=counts,
                            visited=visited, computed=computed, exact_keys=exact_keys,
                            exact_distances=exact_distances, exact_counts=exact_counts,
                            meta=np.array(json.dumps(dict(metric=metric, dtype=dtype, ndim=ndim, n=n,
                                                          connectivity=connectivity, k=k, expansion=expansion))))
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, image {len(image) / 1024:.0f} KiB")

    kat = {
        "comment": "Known-answer vectors held by the reference's own tests (paths relative to /root/reference)",
        "distance": [
            {"source": "golang/lib_test.go:835-877", "metric": "cos", "dtype": "f32", "ndim": 3,
             "a": [1, 0, 0], "b": [0, 1, 0], "expected": 1.0, "tolerance": 0.01},
            {"source": "golang/lib_test.go:835-877", "metric": "l2sq", "dtype": "f32", "ndim": 3,
             "a": [1, 0, 0], "b": [0, 1, 0], "expected": 2.0, "tolerance": 0.01},
            {"source": "golang/lib_test.go:835-877", "metric": "l2sq", "dtype": "i8", "ndim": 3,
             "a": [10, 0, 0], "b": [0, 10, 0], "expected": 200.0, "tolerance": 0.1},
            {"source": "javascript/usearch.test.js:228-267", "metric": "l2sq", "dtype": "f32", "ndim": 3,
             "a": [0.2, 0.6, 0.4], "b": [0.6, 0.6, 0.4], "expected": 0.16, "tolerance": 1e-6},
            {"source": "rust/lib.rs:1897-1924", "metric": "hamming", "dtype": "b1", "ndim": 8,
             "a": [0b01111000], "b": [0b11110000], "expected": 2.0, "tolerance": 0},
            {"source": "rust/lib.rs:1897-1924", "metric": "hamming", "dtype": "b1", "ndim": 8,
             "a": [0b01111000], "b": [0b00001111], "expected": 6.0, "tolerance": 0},
        ],
        "search": [
            {"source": "rust/lib.rs:1897-1924", "metric": "hamming", "dtype": "b1", "ndim": 8,
             "vectors": [[0b00001111], [0b11110000]], "keys": [42, 43], "query": [0b01111000], "k": 2,
             "expected_keys": [43, 42], "expected_distances": [2.0, 6.0]},
            {"source": "cpp/test.cpp:1045-1100 (test_replacing_update: 1-D l2sq, keys come back 42, 43, 44)",
             "metric": "l2sq", "dtype": "f32", "ndim": 1, "vectors": [[10.1], [10.2], [10.3]], "keys": [42, 43, 44],
             "query": [10.0], "k": 3, "expected_keys": [42, 43, 44], "expected_distances": None},
        ],
    }
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)


if __name__ == "__main__":
    main()
