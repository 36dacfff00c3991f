"""CPU: the synthetic-data helpers of bench.py (the GPU twin `synthetic_vectors_device` produces the same kinds in HBM)."""
import numpy as np

import bench
from tests import util


def test_synthetic_vectors_in_every_storage_kind():
    for dtype, columns in (("f32", 40), ("f16", 40), ("bf16", 40), ("f64", 40), ("i8", 40), ("b1", 5)):
        rows = bench.synthetic_vectors(300, 40, dtype, seed=1)
        assert rows.shape == (300, columns) and rows.dtype == bench.NUMPY_STORAGE[dtype]
    # brain floats are the upper halves of the f32 values (truncation, index_plugins.hpp:453-469)
    assert np.array_equal(bench.synthetic_vectors(64, 24, "bf16", seed=3), util.to_bf16(bench.synthetic_vectors(64, 24, "f32", seed=3)))
    # bits are packed MSB first like cast_to_b1x8_gt (index_plugins.hpp:1139-1158)
    bits = bench.synthetic_vectors(8, 16, "b1", seed=5)
    assert np.array_equal(bits, np.packbits(bench.synthetic_vectors(8, 16, "f32", seed=5) > 0, axis=1))


def test_host_core_count_respects_the_cgroup_quota():
    assert 1 <= bench.host_cores() <= (__import__("os").cpu_count() or 1)
