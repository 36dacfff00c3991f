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


def test_memory_plan_and_dry_run():
    """`bench.py --dry`: the per-rank memory plan (DESIGN.md §2's arrays) and the launcher command, without touching a GPU."""
    import json
    import os
    import subprocess
    import sys
    import bench
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    headline = bench.memory_plan(10_000_000, 768, "f16", 16, 10_000, 10)
    assert abs(headline["index_bytes"] / 1e9 - 16.8) < 0.1 and headline["rows_inline_with_lists"] == 0
    shard = bench.memory_plan(125_000_000, 128, "b1", 16, 100_000, 10)
    assert abs(shard["index_bytes"] / 1e9 - 84.0) < 1.0 and shard["rows_inline_with_lists"] == 125_000_000 * 32 * 16
    c4 = bench.memory_plan(100_000_000, 96, "i8", 16, 100_000, 10)
    assert c4["vectors"] == 100_000_000 * 128  # 96-byte rows at pitch 128
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--sharded", "--config", "c5", "--dry"],
                         capture_output=True, text=True, timeout=120)
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["dry_run"] and line["fits"] and line["vectors_in_all"] == 1_000_000_000
    assert "--nproc-per-node 8" in line["command"] and "--master-addr 127.0.0.1" in line["command"] and "--dry" not in line["command"]
