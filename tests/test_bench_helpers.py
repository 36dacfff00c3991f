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


def test_gpus_flag_relaunches_itself_as_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher becomes N ranks under torch.distributed.run; nothing the script accepts
    may collide with the launcher's own option abbreviations ("--n" reads as an ambiguous prefix of --nnodes / --nproc-per-node
    there, even behind the script name)."""
    import sys

    from torch.distributed.run import get_args_parser
    captured = {}
    monkeypatch.setattr(bench.os, "execv", lambda program, command: captured.update(program=program, command=command))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5", "--n", "1000", "--sharded",
                                      "--dtype", "b1", "--dim", "128", "--queries", "100000", "--no-cpu-baseline"])
    bench.relaunch_with_ranks(4)
    command = captured["command"]
    assert command[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in command
    parsed = get_args_parser().parse_args(command[3:])  # raises SystemExit on an ambiguous / unknown option
    assert parsed.training_script.endswith("bench.py") and parsed.master_addr == "127.0.0.1"
    assert parsed.training_script_args == ["--gpus", "4", "--steps", "20", "--warmup", "5", "--vectors", "1000", "--sharded",
                                           "--dtype", "b1", "--dim", "128", "--queries", "100000", "--no-cpu-baseline"]
    # every option bench.py itself defines survives the launcher's parser behind the script name
    import re
    options = sorted(set(re.findall(r'add_argument\("(--[a-z-]+)"', open(bench.__file__).read())))
    assert "--vectors" in options and len(options) > 15
    for option in options:
        get_args_parser().parse_args(["--nproc-per-node=2", bench.__file__, option, "1"])
