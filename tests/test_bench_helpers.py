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


def test_launcher_path_of_the_sharded_bench_two_ranks_on_the_cpu():
    """`BENCH_REHEARSAL=cpu python bench.py --gpus 2 --sharded`: the script re-executes itself under torch.distributed.run with two
    ranks, which join over gloo, follow rank 0's choice of the expansion, run warm-up and timed steps through the product's sharded
    entry point (host transport; the shard search is a stand-in — there is no search without a device) between barriers, and rank 0
    prints ONE JSON line. The control flow of an N-GPU run, exercised every round although no multi-GPU node ever was available."""
    import json
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_REHEARSAL="cpu", GLOO_SOCKET_IFNAME="lo")
    for name in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(name, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--sharded", "--config", "c5", "--steps", "4",
                          "--warmup", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [text for text in out.stdout.splitlines() if text.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 alone speaks
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert "REHEARSAL" in line["data"] and line["roofline"] is None and line["cpu_baseline"] is None
    assert line["config"]["preflight"]["world"] == 2 and line["config"]["parallelism"] == "shards2"
    assert line["config"]["expansion_search"] == 96  # what rank 0 decided
    exchange = line["config"]["exchange"]
    assert exchange["exchanges_per_step"] == 1 and exchange["gathered_bytes"] == 2 * exchange["block_bytes"]
    # a world that is not what --gpus said is refused before anything runs
    env3 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    assert subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--sharded", "--steps", "1", "--warmup", "0"],
                          capture_output=True, text=True, timeout=120, env=env3).returncode == 0
