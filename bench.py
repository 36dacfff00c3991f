#!/usr/bin/env python3
"""bench.py — batched HNSW search throughput on MI355X, one process per GPU.

    python bench.py [--gpus N --steps K --warmup W] [--vectors N --dim D --dtype f16 --metric cos --queries Q --k 10 ...]

A *step* is one pass of the hot path over one batch: `--queries` (default 10 000) queries searched against the
HBM-resident index, inputs and outputs already in HBM. Rank 0 prints ONE JSON line (contract in the task statement),
with two extra objects:
  "roofline":     achieved algorithmic GB/s of the search kernel (bytes from the reference's own per-query counters,
                  SURVEY §8d, ÷ HIP-event kernel time measured on the launch stream) against the 8 TB/s HBM peak;
  "cpu_baseline": the REAL reference (`oracle/_ref`, compiled from /root/reference's own headers) searching the same
                  index with the loop of cpp/bench.cpp:352-377 on all host cores, on a bounded sample (rank 0, N = 1).

Default workload = the configuration BASELINE.json's metric is quoted on: 10M x 768 f16 cosine, batch 10 000, k = 10.
The seeded synthetic vectors are generated in HBM and the index is built ON THE GPU (`usearch_amd.build`, ≈20 s for 10M;
`--builder reference` lets the reference build it on the host cores instead — minutes per million vectors, so only for
small `--vectors`). The reference is handed the very same index (`save_buffer` → `usearch_view_buffer`) for the CPU baseline.
`--gpus N` without a launcher re-executes itself under `torch.distributed.run` with N ranks (one per GPU); under a launcher
(RANK / WORLD_SIZE in the environment) it is one of the ranks. With N > 1 every rank builds (deterministically, so
identically) and holds a replica and searches its own batch: weak scaling, no collective on the data path, `value` = the
queries all ranks answered per second. `--sharded` is the capacity mode for indexes beyond one GPU: every rank builds and
holds its own shard of `--vectors` vectors EACH, the batch is broadcast, every rank searches its shard, per-shard top-k travel in
ONE packed RCCL all-gather and are merged (`usearch_amd_sharded_search_many`, usearch_amd/csrc/sharded.hip). Per-GPU work
is fixed as N grows (weak scaling of the index size): `value` = batch × shards ÷ time, "shard-queries/s" — the 1-GPU point
is one shard searched by the same batch — and `config.queries_per_second` is the rate against the whole N-shard index
(DESIGN.md §7).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (≈6.3 TB/s achievable on streaming copies)
DTYPE_BYTES = {"f32": 4.0, "f16": 2.0, "bf16": 2.0, "f64": 8.0, "i8": 1.0, "b1": 0.125}
NUMPY_STORAGE = {"f32": np.float32, "f16": np.float16, "bf16": np.uint16, "f64": np.float64, "i8": np.int8, "b1": np.uint8}


def host_cores() -> int:
    """Cores this process may really use: the cgroup CPU quota when there is one (the GPU boxes advertise 256 logical
    CPUs but grant a 16-CPU quota; oversubscribing the reference's OpenMP loops makes them slower), else all CPUs."""
    cpus = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cpus = max(1, min(cpus, int(round(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return cpus


def log(*args):
    print(*args, file=sys.stderr, flush=True)


def synthetic_vectors(count: int, dim: int, dtype: str, seed: int, basis_seed: int = 42) -> np.ndarray:
    """SURVEY §8d "LR-r": x = z·B + 0.05·ε with a fixed rank-r basis (r = 32 for d ≥ 256, else 16). i.i.d. Gaussian data has
    no neighbourhood structure at 768-d (recall@10 = 0.05 for the reference itself), so the recall target needs this."""
    rank = 32 if dim >= 256 else 16
    basis = np.random.default_rng(basis_seed).standard_normal((rank, dim)).astype(np.float32)
    rng = np.random.default_rng(seed)
    out = []
    for begin in range(0, count, 65536):
        rows = min(65536, count - begin)
        x = rng.standard_normal((rows, rank), dtype=np.float32) @ basis
        x += 0.05 * rng.standard_normal((rows, dim), dtype=np.float32)
        if dtype == "f32":
            out.append(x)
        elif dtype == "f16":
            out.append(x.astype(np.float16))
        elif dtype == "f64":
            out.append(x.astype(np.float64))
        elif dtype == "bf16":  # bit patterns; truncation like the reference's f32_to_bf16 (index_plugins.hpp:453-469)
            out.append((np.ascontiguousarray(x).view(np.uint32) >> 16).astype(np.uint16))
        elif dtype == "i8":
            out.append(np.clip(np.rint(x * (127.0 / 24.0)), -127, 127).astype(np.int8))
        elif dtype == "b1":
            out.append(np.packbits(x > 0, axis=1))
        else:
            raise ValueError(dtype)
    return np.concatenate(out) if out else np.zeros((0, dim))


def synthetic_vectors_device(count: int, dim: int, dtype: str, seed: int, device, basis_seed: int = 42):
    """The same "LR-r" distribution generated on the GPU (torch is plumbing here: device memory + RNG), as a
    [count, bytes_per_vector] uint8 tensor in the storage kind. 10M x 768 takes seconds instead of minutes on the host."""
    import torch
    rank = 32 if dim >= 256 else 16
    basis = torch.from_numpy(np.random.default_rng(basis_seed).standard_normal((rank, dim)).astype(np.float32)).to(device)
    generator = torch.Generator(device=device)
    generator.manual_seed(seed)
    row_bytes = int(dim * DTYPE_BYTES[dtype]) if dtype != "b1" else (dim + 7) // 8
    out = torch.empty((count, row_bytes), dtype=torch.uint8, device=device)
    weights = (2 ** torch.arange(7, -1, -1, device=device)).to(torch.uint8)
    for begin in range(0, count, 262144):
        rows = min(262144, count - begin)
        x = torch.randn((rows, rank), generator=generator, device=device) @ basis
        x += 0.05 * torch.randn((rows, dim), generator=generator, device=device)
        if dtype == "f32":
            block = x
        elif dtype == "f16":
            block = x.to(torch.float16)
        elif dtype == "f64":
            block = x.to(torch.float64)
        elif dtype == "bf16":  # truncation like the reference's f32_to_bf16 (index_plugins.hpp:453-469)
            block = (x.contiguous().view(torch.int32) >> 16).to(torch.int16)
        elif dtype == "i8":
            block = torch.clamp(torch.round(x * (127.0 / 24.0)), -127, 127).to(torch.int8)
        else:  # b1: sign bits, MSB first (cast_to_b1x8_gt, index_plugins.hpp:1139-1158)
            bits = (x > 0).to(torch.uint8)
            if dim % 8:
                bits = torch.nn.functional.pad(bits, (0, 8 - dim % 8))
            block = (bits.view(rows, -1, 8) * weights).sum(dim=2).to(torch.uint8)
        out[begin:begin + rows] = block.contiguous().view(torch.uint8).view(rows, row_bytes)
    return out


WALK_SOURCES = ("common.hpp", "engine.hpp", "engine.hip", "host_util.hpp", "image.hpp", "kernels.hpp", "launch_impl.hpp",
                "placement.hpp", "placement.hip")


def source_hash() -> str:
    """Names the build of the timed path: sha256 over the sources the graph walk is compiled from (the kernels, their launch
    dispatch, the engine that sizes and launches them, placement). A `roofline.traffic` measured by PMC passes is only attached
    to a line produced by the same sources (profiles/<round>/traffic.json); the builder, the exact-search kernels, the
    collective and the C surfaces can change without voiding it."""
    digest = hashlib.sha256()
    directory = os.path.join(ROOT, "usearch_amd", "csrc")
    for name in sorted(os.listdir(directory)):
        if name in WALK_SOURCES or (name.startswith("search_") and name.endswith(".hip")):
            digest.update(name.encode())
            digest.update(open(os.path.join(directory, name), "rb").read())
    return digest.hexdigest()[:16]


def relaunch_with_ranks(gpus: int) -> None:
    """`python bench.py --gpus N` on its own: become N ranks, one per GPU, under torch.distributed.run."""
    with socket.socket() as probe:
        probe.bind(("127.0.0.1", 0))
        port = probe.getsockname()[1]
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    command += ["--vectors" if argument == "--n" else argument.replace("--n=", "--vectors=", 1) if argument.startswith("--n=")
                else argument for argument in sys.argv[1:]]
    log(f"[bench] --gpus {gpus} without a launcher: " + " ".join(command))
    os.execv(sys.executable, command)


def flush_native_stdio() -> None:
    """What native libraries hold in C stdio buffers (RCCL prints a version banner through printf when a communicator is made)
    goes out NOW — on every rank, long before rank 0 prints the one JSON line — instead of at process exit, after it."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass


def recall_per_query(found: np.ndarray, truth: np.ndarray, k: int) -> np.ndarray:
    return np.array([len(np.intersect1d(found[i], truth[i])) / k for i in range(len(found))], dtype=np.float64)


def interval(values: np.ndarray):
    """Mean and the half-width of its 95 % confidence interval."""
    mean = float(np.mean(values))
    half = 1.96 * float(np.std(values, ddof=1)) / np.sqrt(len(values)) if len(values) > 1 else 0.0
    return mean, half


def stress_rows(args, metric: str, device, local_rank: int, expansion: int) -> dict:
    """The two secondary datasets SURVEY §8(d) wants next to the headline, on a smaller index (`--stress-n`): i.i.d.
    Gaussian vectors (no neighbourhood structure at 768-d: the stress case of any graph index) and the reference's own
    benchmark data, `eval.random_vectors` = uniform[0,1) with in-sample self-recall@1 (python/usearch/eval.py:55-62, 97-139)."""
    import torch

    import usearch_amd
    rows = {}
    row_bytes = int(args.dim * DTYPE_BYTES[args.dtype]) if args.dtype != "b1" else (args.dim + 7) // 8

    def to_storage(x):
        if args.dtype == "f32":
            block = x
        elif args.dtype == "f16":
            block = x.to(torch.float16)
        elif args.dtype == "f64":
            block = x.to(torch.float64)
        elif args.dtype == "bf16":
            block = (x.contiguous().view(torch.int32) >> 16).to(torch.int16)
        elif args.dtype == "i8":
            block = torch.clamp(torch.round(x * 100.0 if float(x.min()) >= 0 else x * (127.0 / 4.0)), -127, 127).to(torch.int8)
        else:
            bits = (x > (0.5 if float(x.min()) >= 0 else 0.0)).to(torch.uint8)
            if args.dim % 8:
                bits = torch.nn.functional.pad(bits, (0, 8 - args.dim % 8))
            weights = (2 ** torch.arange(7, -1, -1, device=device)).to(torch.uint8)
            block = (bits.view(len(x), -1, 8) * weights).sum(dim=2).to(torch.uint8)
        return block.contiguous().view(torch.uint8).view(len(x), row_bytes)

    def generate(count: int, seed: int, uniform: bool):
        generator = torch.Generator(device=device)
        generator.manual_seed(seed)
        out = torch.empty((count, row_bytes), dtype=torch.uint8, device=device)
        for begin in range(0, count, 262144):
            size = min(262144, count - begin)
            x = (torch.rand if uniform else torch.randn)((size, args.dim), generator=generator, device=device)
            out[begin:begin + size] = to_storage(x)
        return out

    n, q = args.stress_n, min(args.queries, 10_000)
    for name, uniform in (("iid_gaussian", False), ("uniform_self_recall", True)):
        t0 = time.time()
        data = generate(n, 7 + int(uniform), uniform)
        built = usearch_amd.build(None, metric, args.dtype, connectivity=args.connectivity, expansion_add=args.expansion_add,
                                  device=local_rank, device_pointer=data.data_ptr(), count=n, stride=data.stride(0),
                                  ndim=args.dim)
        index = built.index
        queries = data[:q].clone() if uniform else generate(q, 99, False)
        del data
        keys = torch.zeros((q, args.k), dtype=torch.int64, device=device)
        distances = torch.zeros((q, args.k), dtype=torch.float32, device=device)
        counters = [torch.zeros(q, dtype=torch.int64, device=device) for _ in range(3)]
        stream = torch.cuda.Stream(device)

        def run(ef: int, timed: bool):
            return index.search_device(queries.data_ptr(), q, queries.stride(0), args.k, ef, keys.data_ptr(),
                                       distances.data_ptr(), counters[0].data_ptr(), counters[1].data_ptr(),
                                       counters[2].data_ptr(), stream=stream.cuda_stream, timed=timed)
        row = {"vectors": n, "queries": q}
        if uniform:  # the reference's self_recall: every member must find itself first, default expansion
            run(64, False)
            row["self_recall_at_1_ef64"] = float((keys[:, 0].cpu().numpy() == np.arange(q)).mean())
            stats = run(64, True)
            row["qps_ef64"] = q / (stats.kernel_ms / 1e3)
            row["kernel_passes_ef64"] = int(stats.passes)  # 1 = every traversal fit the default scratch sizing
        else:
            sample = min(q, 1000)
            host = queries[:sample].cpu().numpy().view(NUMPY_STORAGE[args.dtype])
            truth = index.search(host, args.k, dtype=args.dtype, exact=True).keys
            for ef in sorted({64, expansion}):
                stats = run(ef, True)
                found = keys[:sample].cpu().numpy().astype(np.uint64)
                row[f"recall_at_{args.k}_ef{ef}"] = float(np.mean(recall_per_query(found, truth, args.k)))
                row[f"qps_ef{ef}"] = q / (stats.kernel_ms / 1e3)
                row[f"kernel_passes_ef{ef}"] = int(stats.passes)
        row["seconds"] = round(time.time() - t0, 1)
        rows[name] = row
        log(f"[bench] stress row {name}: {row}")
        del built, index, queries
        torch.cuda.empty_cache()
    return rows


# BASELINE.json `configs`, as single-GPU command lines (c5: one 125M shard of the 1B index; `--gpus 8 --sharded` makes it the whole)
PRESETS = {
    "c1": {"what": "100k x 128 f32 cos, k = 10, ef = 64 (the reference's own CPU-runnable case, cpp/bench.cpp)",
           "set": {"n": 100_000, "dim": 128, "dtype": "f32", "queries": 10_000, "expansion": 64}},
    "c2": {"what": "1M x 768 f32 cos, batch 10k", "set": {"n": 1_000_000, "dim": 768, "dtype": "f32", "queries": 10_000}},
    "c3": {"what": "10M x 768 f16 cos, batch 10k (the headline)", "set": {"n": 10_000_000, "dim": 768, "dtype": "f16", "queries": 10_000}},
    "c4": {"what": "100M x 96 i8 l2sq, batch 100k", "set": {"n": 100_000_000, "dim": 96, "dtype": "i8", "queries": 100_000}},
    "c5": {"what": "125M x 128 b1 hamming, batch 100k: one shard of the 1B index",
           "set": {"n": 125_000_000, "dim": 128, "dtype": "b1", "queries": 100_000}},
}


def secondary_lines(presets, timeout_seconds: float) -> list:
    """BASELINE.json's other configurations (`PRESETS`), each run by this script in a process of its own — `python bench.py --config cN
    --no-secondary …`: its own index at the configuration's full size, its own recall sweep, roofline and reference baseline — and
    summarised for `config.secondary` of the default line. A configuration that fails or overruns is reported as such, never hidden."""
    import subprocess
    out = []
    for name in presets:
        command = [sys.executable, os.path.abspath(__file__), "--config", name, "--no-secondary", "--no-stress-rows", "--steps", "10",
                   "--warmup", "2", "--cpu-seconds", "4", "--recall-queries", "2000", "--no-load-timing"]
        t0 = time.time()
        entry = {"preset": name, "what": PRESETS[name]["what"], "command": " ".join(["python", "bench.py"] + command[2:])}
        try:
            done = subprocess.run(command, capture_output=True, text=True, timeout=timeout_seconds, cwd=ROOT)
            lines = [text for text in done.stdout.splitlines() if text.startswith("{")]
            if done.returncode != 0 or not lines:
                raise RuntimeError(f"exit code {done.returncode}: {done.stderr.strip().splitlines()[-1] if done.stderr.strip() else 'no output'}")
            line = json.loads(lines[-1])
            roofline, cpu, config = line["roofline"], line.get("cpu_baseline") or {}, line["config"]
            entry.update({
                "workload": config["workload"], "value": line["value"], "unit": line["unit"], "ms_per_step": line["ms_per_step"],
                "steps": line["steps"], "dtype": line["dtype"], "expansion_search": config["expansion_search"],
                "recall_at_k": config["recall_at_k"], "recall_ci": config["recall_ci"], "recall_queries": config["recall_queries"],
                "index_build_seconds": config["index_build_seconds"], "kernel_passes": config["kernel_passes"],
                "visited_set_probe": config.get("visited_set_probe"),
                "roofline": {key: roofline.get(key) for key in ("bound", "achieved", "peak", "unit", "frac", "lines_touched_frac", "kernel",
                                                                "kernel_ms", "kernel_instantiation", "algorithmic_bytes_per_launch",
                                                                "kernel_ms_second_load", "traffic", "traffic_source")},
                "cpu_baseline": {key: cpu.get(key) for key in ("value", "unit", "cores", "kind", "label_agreement_with_the_gpu",
                                                               "ground_truth_check", "sample")}})
        except subprocess.TimeoutExpired:
            entry["error"] = f"no line within {timeout_seconds:.0f} s"
        except (RuntimeError, ValueError, KeyError, OSError) as error:
            entry["error"] = str(error)[:300]
        entry["wall_seconds"] = round(time.time() - t0, 1)
        log(f"[bench] secondary {name}: " + (entry.get("error") or f"{entry['value']:,.0f} {entry['unit']}, {entry['ms_per_step']:.2f} ms/step, "
            f"frac {entry['roofline']['frac']:.3f} (whole lines {entry['roofline']['lines_touched_frac']:.3f}), recall {entry['recall_at_k']}") +
            f" [{entry['wall_seconds']} s]")
        out.append(entry)
    return out


def memory_plan(n: int, dim: int, dtype: str, connectivity: int, queries: int, k: int) -> dict:
    """HBM one rank needs for its index of `n` vectors (the flat arrays of `snapshot_view_t`), the builder's transient
    peak, and the batch — what `--dry` prints and what an N-GPU run is checked against before anything is allocated."""
    row_bytes = (dim + 7) // 8 if dtype == "b1" else int(dim * DTYPE_BYTES[dtype])
    chunks = -(-row_bytes // 16)
    lanes = 1 if chunks == 1 else 2 if chunks <= 8 else 8
    pitch = 16 * lanes * -(-chunks // lanes)
    if 16 < row_bytes <= 128:  # rows of <= 128 bytes sit at a power-of-two pitch (no row straddles two lines)
        pitch = 1 << (row_bytes - 1).bit_length()
    m0 = 2 * connectivity
    upper_lists = n / (connectivity - 1)  # levels are geometric with ratio 1 / M: about n / (M - 1) upper-level lists in all
    plan = {"vectors": n * pitch, "level0_lists": n * m0 * 4, "upper_lists": int(upper_lists * connectivity * 4), "upper_refs": n * 4,
            "keys": n * 8, "rows_inline_with_lists": n * m0 * 16 if row_bytes <= 16 else 0}
    index_bytes = sum(plan.values())
    # the builder holds the caller's matrix next to the index while it links (and the serialized image only on the host)
    plan["index_bytes"] = index_bytes
    plan["build_peak_bytes"] = index_bytes + n * row_bytes + (2 << 30)
    plan["batch_bytes"] = queries * (row_bytes + k * 12 + 32) + (1 << 30)  # queries, results, counters + scratch slabs of a workspace
    plan["exact_ground_truth_bytes"] = 8 * 256 * queries * k * 12  # partial lists of the exact kernel (recall check)
    return plan


MFMA_PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0, "i8": 5000.0}  # MI355X_MICROARCH.md: dense f16 / bf16 2.5 PF; i8 at twice that rate


def run_exact(args) -> None:
    """`python bench.py --exact [--gpus N]`: a step is `search(…, exact = true)` of the whole batch against every stored vector
    (index.hpp:4252-4268 per query; `exact_search_t`, index_plugins.hpp:2071-2164, for a batch) through the matrix-unit kernel
    of csrc/exact_tiled.hip — SURVEY §8(f) rank 1, and where bench.py's own recall ground truth comes from. Every query meets
    every row, so this kernel IS bound by the matrix units: `roofline.bound` = "mfma", 2·Q·N·d operations per step. Same launch
    contract and JSON line as the default mode; replicas, one batch per GPU."""
    import torch
    import torch.distributed as dist

    import usearch_amd
    metric = args.metric or ("l2sq" if args.dtype == "i8" else "cos")
    if args.dtype not in MFMA_PEAK_TFLOPS:
        raise SystemExit("--exact times the matrix-unit kernel: --dtype f16, bf16 or i8")
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    cores = host_cores()

    data = synthetic_vectors_device(args.n, args.dim, args.dtype, 42, device)
    t0 = time.time()
    # the scan does not read the graph: the cheapest one the builder makes carries the rows into a snapshot
    built = usearch_amd.build(None, metric, args.dtype, connectivity=4, expansion_add=16, device=local_rank,
                              device_pointer=data.data_ptr(), count=args.n, stride=data.stride(0), ndim=args.dim)
    del data
    torch.cuda.empty_cache()
    index = built.index
    queries_dev = synthetic_vectors_device(args.queries, args.dim, args.dtype, 43 + 1000 * rank, device)
    queries_host = queries_dev.cpu().numpy().view(NUMPY_STORAGE[args.dtype])
    keys_dev = torch.zeros((args.queries, args.k), dtype=torch.int64, device=device)
    dist_dev = torch.zeros((args.queries, args.k), dtype=torch.float32, device=device)
    counts_dev = torch.zeros(args.queries, dtype=torch.int64, device=device)
    stream = torch.cuda.Stream(device)
    if rank == 0:
        log(f"[bench] {args.n}x{args.dim} {args.dtype} in HBM ({time.time() - t0:.1f}s), batch of {args.queries} resident")

    def step() -> float:
        return index.exact_search_device(queries_dev.data_ptr(), args.queries, queries_dev.stride(0), args.k, keys_dev.data_ptr(),
                                         dist_dev.data_ptr(), counts_dev.data_ptr(), stream=stream.cuda_stream, tiled=True)

    flush_native_stdio()
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.75:  # sustained clocks, as in the default mode
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kernel_ms.append(step())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        operations = 2.0 * args.queries * args.n * args.dim
        kernel_s = float(np.mean(kernel_ms)) / 1e3
        achieved = operations / kernel_s / 1e12
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        found_keys = keys_dev.cpu().numpy().astype(np.uint64)
        found_distances = dist_dev.cpu().numpy()
        # the bit-exact wave-per-query kernel on a sample of the batch (one dataset pass per query)
        checked = min(args.queries, 64)
        plain = index.search(queries_host[:checked], args.k, dtype=args.dtype, exact=True)
        same_keys = float(np.mean(plain.keys == found_keys[:checked]))
        worst = float(np.nanmax(np.abs(plain.distances - found_distances[:checked])))
        same_bits = float(np.mean(plain.distances.view(np.uint32) == found_distances[:checked].view(np.uint32)))
        log(f"[bench] exact step: kernel {kernel_s * 1e3:.1f} ms = {achieved:.0f} T(FL)OP/s; against the bit-exact kernel on {checked} queries: "
            f"keys equal {same_keys:.4f}, distance bits equal {same_bits:.4f}, max |diff| {worst:.3g}")
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import refbind
            image = built.save_buffer()
            reference = refbind.RefIndex.from_buffer(image, view=True, dtype=args.dtype)
            t1 = time.perf_counter()
            reference.search(queries_host[:cores], args.k, dtype=args.dtype, exact=True, threads=cores)
            pilot_seconds = time.perf_counter() - t1
            sample_q = int(min(args.queries, max(cores, cores * int(args.cpu_seconds / max(pilot_seconds, 1e-3)))))
            t1 = time.perf_counter()
            rkeys, *_ = reference.search(queries_host[:sample_q], args.k, dtype=args.dtype, exact=True, threads=cores)
            cpu_seconds = time.perf_counter() - t1
            cpu = {"value": sample_q / cpu_seconds, "unit": "queries/s", "cores": cores, "kind": "reference",
                   "sample": f"{sample_q} of the step's {args.queries} queries against the same {args.n} vectors, the reference's "
                             f"`search(…, exact = true)` (index.hpp:4252-4268) on {cores} threads, serial (auto-vectorised) metrics; "
                             f"{cpu_seconds:.1f}s; label agreement with the GPU {float(np.mean(rkeys == found_keys[:sample_q])):.4f}"}
            del reference, image
        workload = f"exact search, {args.n}x{args.dim} {args.dtype} {metric}, batch {args.queries}, k={args.k}"
        line = {
            "metric": f"exact-search QPS (search(..., exact=true) for a batch), {args.n}x{args.dim} {args.dtype} {metric}, batch={args.queries}",
            "value": args.queries * world * args.steps / elapsed, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic (seeded rank-32 latent + 0.05 noise, out-of-sample queries)",
            "config": {"workload": workload, "vectors": args.n, "dimensions": args.dim, "parallelism": f"replicas{world}",
                       "kernel": "exact_wide_kernel (256 queries x 256 rows per workgroup, LDS-DMA staging, one round of the chip) + row / query norms + partition merge",
                       "checked_against_bit_exact_kernel": {"queries": checked, "keys_equal": same_keys,
                                                            "distance_bits_equal": same_bits, "max_abs_difference": worst},
                       "sources": source_hash()},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s" if args.dtype != "i8" else "TOP/s",
                         "frac": achieved / peak, "traffic": None, "kernel": "exact_wide_kernel", "kernel_ms": kernel_s * 1e3,
                         "operations_per_launch": operations},
            "cpu_baseline": cpu,
        }
        flush_native_stdio()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def join_ranks(args, on_cpu: bool = False):
    """This process as one rank of the run: RANK / LOCAL_RANK / WORLD_SIZE from the launcher (one rank per GPU over RCCL), the process
    group, and the preflight of an N > 1 run — N ranks on N DIFFERENT devices, the launcher's world equal to `--gpus`.
    BENCH_REHEARSAL=1: every rank on cuda:0 with gloo collectives — the N > 1 control flow (decisions broadcast from rank 0,
    barriers, the max over ranks, the sharded step's exchange) on a box with ONE GPU; RCCL refuses two ranks on one device, so this
    is a rehearsal of the launcher path, never a measurement. `on_cpu` (BENCH_REHEARSAL=cpu, `rehearse_on_cpu`): no device at all."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    rehearsal = os.environ.get("BENCH_REHEARSAL") in ("1", "cpu")
    if rehearsal:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # every rank is on this box
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cpu") if on_cpu else torch.device("cuda", local_rank)
    if not on_cpu:
        torch.cuda.set_device(local_rank)
    preflight = {"world": world, "devices": 1}
    if world > 1:
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
        if on_cpu:
            mine = (local_rank, "cpu")
        else:
            identity = torch.cuda.get_device_properties(device)
            mine = (local_rank, str(getattr(identity, "uuid", "")) or f"{identity.name}#{local_rank}")
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        preflight["devices"] = len({uuid for _, uuid in everyone})
        preflight["local_ranks"] = [r for r, _ in everyone]
        if not rehearsal and preflight["devices"] != world:
            raise SystemExit(f"{world} ranks share {preflight['devices']} device(s): one process per GPU is the contract")
    return rank, world, local_rank, rehearsal, device, preflight


def rank_zero_decides(flag: bool, rank: int, world: int, device) -> bool:
    """Rank 0 decides, everybody follows (the recall sweep is made of collectives when the index is sharded)."""
    if world == 1:
        return flag
    import torch
    import torch.distributed as dist
    box = torch.tensor([1 if (flag and rank == 0) else 0], device=device)
    dist.broadcast(box, 0)
    return bool(box.item())


def timed_steps(step, steps: int, world: int, device) -> float:
    """EXACTLY `steps` calls of `step(i)` between a barrier + device synchronisation on both sides; the seconds of the slowest rank."""
    import torch
    import torch.distributed as dist
    on_gpu = device.type == "cuda"
    if on_gpu:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    if on_gpu:
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def rehearse_on_cpu(args) -> None:
    """BENCH_REHEARSAL=cpu — the launcher path of `bench.py --gpus N --sharded` WITHOUT a GPU (tests/test_bench_helpers.py runs it with
    two ranks every round): torch.distributed.run → `join_ranks` (gloo) → the expansion chosen by rank 0 and followed by all → warm-up →
    `timed_steps` (barriers, max over ranks) → ONE JSON line from rank 0. Every step goes through the product's own sharded entry point
    (`usearch_amd_sharded_search_many`: query broadcast, one packed all-gather, the merge) over the host transport; ONLY the local
    search of a shard is a seeded stand-in, because the search itself exists on the device alone — the line says so and is not a
    measurement of anything."""
    import torch
    import torch.distributed as dist

    rank, world, _, _, device, preflight = join_ranks(args, on_cpu=True)
    from usearch_amd.sharded import Communicator
    queries_count, wanted, stride = min(args.queries, 64), args.k, 16

    def local_search(queries: np.ndarray, count: int, k: int, expansion: int):
        rng = np.random.default_rng(int(queries.astype(np.int64).sum()) % 1000 + 17 * rank + expansion)
        distances = np.sort(rng.integers(0, 64, size=(count, k)).astype(np.float32), axis=1)
        keys = (rng.integers(0, args.n, size=(count, k)).astype(np.uint64) + np.uint64(rank * args.n))
        return keys, distances, np.full(count, k, dtype=np.uint64)

    communicator = Communicator.on_host(
        rank, world,
        lambda send, receive: dist.all_gather_into_tensor(torch.from_numpy(receive), torch.from_numpy(send)) if world > 1 else receive.__setitem__(slice(None), send),
        lambda buffer, root: dist.broadcast(torch.from_numpy(buffer), src=root) if world > 1 else None, local_search)
    queries = np.full((queries_count, stride), 1 + rank, dtype=np.uint8)
    keys = np.zeros((queries_count, wanted), dtype=np.uint64)
    distances = np.zeros((queries_count, wanted), dtype=np.float32)
    counts = np.zeros(queries_count, dtype=np.uint64)
    last = {}

    def search_step(expansion: int):
        last["stats"], last["step"] = communicator.search_raw(None, queries.ctypes.data, queries_count, stride, wanted, expansion, 0,
                                                              keys.ctypes.data, distances.ctypes.data, counts.ctypes.data, 0, 0)

    expansion = 0
    for candidate in (64, 96, 128):  # the sweep's shape: every rank steps, rank 0 scores and decides, everybody follows
        search_step(candidate)
        expansion = candidate
        if rank_zero_decides(candidate >= 96, rank, world, device):
            break
    for _ in range(args.warmup):
        search_step(expansion)
    elapsed = timed_steps(lambda i: search_step(expansion), args.steps, world, device)
    if rank == 0:
        flush_native_stdio()
        print(json.dumps({
            "metric": f"REHEARSAL of the launcher path, {world} rank(s) on the CPU over gloo", "value": queries_count * args.steps * world / elapsed,
            "unit": "shard-queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "REHEARSAL on the CPU: the local search of a shard is a seeded stand-in, nothing is measured",
            "config": {"workload": "launcher rehearsal", "parallelism": f"shards{world}", "preflight": preflight, "expansion_search": expansion,
                       "exchange": {"transport": communicator.kind, "block_bytes": int(last["step"].block_bytes),
                                    "gathered_bytes": int(last["step"].gathered_bytes), "exchanges_per_step": int(last["step"].exchanges)}},
            "roofline": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=5)
    parser.add_argument("--warmup", type=int, default=1)
    # (`--n` would collide with the launcher's own abbreviations — torch.distributed.run reads "--n" as an ambiguous prefix of
    # --nnodes / --nproc-per-node even behind the script name — hence the long name; `--n` stays as an alias for direct runs)
    parser.add_argument("--vectors", "--n", dest="n", type=int, default=int(os.environ.get("BENCH_N", 10_000_000)),
                        help="vectors in the index (per shard with --sharded)")
    parser.add_argument("--dim", type=int, default=768)
    parser.add_argument("--dtype", default="f16", choices=list(DTYPE_BYTES))
    parser.add_argument("--metric", default=None)
    parser.add_argument("--queries", type=int, default=10_000)
    parser.add_argument("--k", type=int, default=10)
    parser.add_argument("--expansion", type=int, default=0,
                        help="0 = smallest of the sweep (64 ... 1024) whose recall@k is >= 0.95 with 95 %% confidence")
    parser.add_argument("--connectivity", type=int, default=16)
    parser.add_argument("--expansion-add", type=int, default=128)
    parser.add_argument("--recall-queries", type=int, default=-1,
                        help="queries with exact ground truth (-1 = the batch, at most 10 000; sharded mode: 1000)")
    parser.add_argument("--cpu-seconds", type=float, default=12.0)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--no-stress-rows", action="store_true")
    parser.add_argument("--no-load-timing", action="store_true", help="skip timing the load of the serialized image")
    parser.add_argument("--no-host-api", action="store_true",
                        help="skip the host-buffer and single-query legs (profiled runs: every launch of the timed kernel is a timed batch)")
    parser.add_argument("--stress-n", type=int, default=1_000_000)
    parser.add_argument("--sharded", action="store_true")
    parser.add_argument("--transport", default="rccl", choices=["rccl", "torch"],
                        help="sharded mode: native RCCL communicator (default) or torch.distributed collectives")
    parser.add_argument("--builder", default=os.environ.get("BENCH_BUILDER", "gpu"), choices=["gpu", "reference"],
                        help="who links the index: the device builder (default) or the reference on the host cores")
    parser.add_argument("--build-threads", type=int, default=int(os.environ.get("BENCH_BUILD_THREADS", 0)),
                        help="threads the reference uses to build the index (0 = 2 x the cgroup CPU quota)")
    parser.add_argument("--traffic-json", default=os.environ.get("BENCH_TRAFFIC_JSON", ""),
                        help="PMC-derived HBM bytes per launch of this workload (scripts/pmc_traffic.py), copied into "
                             "roofline.traffic when it was measured with the same sources")
    parser.add_argument("--wave-clock", action="store_true", help="record the batch-tail telemetry of the timed steps")
    parser.add_argument("--no-placement-check", action="store_true",
                        help="skip loading the image a SECOND time after the timed steps and timing the batch on that copy "
                             "(roofline.kernel_ms_second_load: how reproducible the settled placement is)")
    parser.add_argument("--no-reload", action="store_true",
                        help="search the builder's own arrays instead of the saved image loaded back through the device loader "
                             "(the default walks what `usearch_load` would give a user: matrix placed after the settle window)")
    parser.add_argument("--no-tune", action="store_true",
                        help="do not place the matrix by trial on the batch before the timed steps (usearch_amd_snapshot_tune)")
    parser.add_argument("--no-secondary", action="store_true",
                        help="skip BASELINE.json's other configurations (c1, c2, c4, c5 — each in a process of its own, summarised "
                             "under config.secondary of the default line)")
    parser.add_argument("--secondary", default="c1,c2,c4,c5", help="which presets ride along with the default line")
    parser.add_argument("--secondary-timeout", type=float, default=420.0, help="seconds one secondary configuration may take")
    parser.add_argument("--exact", action="store_true",
                        help="time the exact (brute-force) search of the batch through the matrix-unit kernel instead of the graph walk")
    parser.add_argument("--max-batch", type=int, default=int(os.environ.get("BENCH_MAX_BATCH", 0)),
                        help="device builder: most nodes linked per batch (0 = the builder's default)")
    parser.add_argument("--dry", action="store_true",
                        help="no GPU work: print the per-rank memory plan of this configuration against 288 GB of HBM and the exact "
                             "command line an N-GPU run is launched with, then exit (what the 8 x 125M recipe is checked with)")
    parser.add_argument("--config", default=None, choices=sorted(PRESETS),
                        help="one of BASELINE.json's configurations by name: " + "; ".join(f"{k} = {v['what']}" for k, v in sorted(PRESETS.items())))
    args = parser.parse_args()
    if args.config:  # a preset fills in what the command line left at its default
        for name, value in PRESETS[args.config]["set"].items():
            if getattr(args, name) == parser.get_default(name):
                setattr(args, name, value)
    if args.dry:
        plan = memory_plan(args.n, args.dim, args.dtype, args.connectivity, args.queries, args.k)
        hbm = 288 * 10**9
        need = max(plan["build_peak_bytes"], plan["index_bytes"] + plan["batch_bytes"] + plan["exact_ground_truth_bytes"])
        forwarded = [a for a in sys.argv[1:] if a != "--dry"]
        command = (f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port 29500 "
                   f"bench.py {' '.join(forwarded)}") if args.gpus > 1 else f"python bench.py {' '.join(forwarded)}"
        print(json.dumps({"dry_run": True, "ranks": args.gpus, "mode": "shards" if args.sharded else "replicas",
                          "vectors_per_rank": args.n, "vectors_in_all": args.n * (args.gpus if args.sharded else 1),
                          "per_rank_gigabytes": {name: round(value / 1e9, 2) for name, value in plan.items()},
                          "per_rank_peak_gigabytes": round(need / 1e9, 1), "hbm_gigabytes": hbm / 1e9, "fits": need < 0.92 * hbm,
                          "exchange_bytes_per_rank_and_step": args.queries * args.k * 12 + args.queries * 8 + 8 if args.sharded else 0,
                          "command": command}))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_with_ranks(args.gpus)
    if os.environ.get("BENCH_REHEARSAL") == "cpu":
        return rehearse_on_cpu(args)
    if args.exact:
        return run_exact(args)
    metric = args.metric or ("hamming" if args.dtype == "b1" else "l2sq" if args.dtype == "i8" else "cos")
    cores = host_cores()
    if not args.build_threads:
        args.build_threads = 2 * cores

    import torch
    import torch.distributed as dist

    rank, world, local_rank, rehearsal, device, preflight = join_ranks(args)

    import usearch_amd

    # ---- the index. Replicas: every rank builds the same seeded data (the device build is deterministic). Shards: rank r
    #      builds its own `--vectors` vectors with keys offset by r * n.
    sharded = args.sharded
    data_seed = 42 + (rank if sharded else 0)
    key_base = rank * args.n if sharded else 0
    build_seconds, build_stats, ref_index, image, built, reload_seconds = 0.0, None, None, None, None, None
    t_generate = time.time()
    if args.builder == "gpu":
        data = synthetic_vectors_device(args.n, args.dim, args.dtype, data_seed, device)
        torch.cuda.synchronize()
        generate_seconds = time.time() - t_generate
        keys = np.arange(args.n, dtype=np.uint64) + key_base if key_base else None
        t0 = time.time()
        built = usearch_amd.build(None, metric, args.dtype, keys=keys, connectivity=args.connectivity,
                                  expansion_add=args.expansion_add, device=local_rank, device_pointer=data.data_ptr(),
                                  count=args.n, stride=data.stride(0), ndim=args.dim, max_batch=args.max_batch)
        build_seconds = time.time() - t0
        del data
        torch.cuda.empty_cache()
        index = built.index
        build_stats = built.stats.as_dict()
        if rank == 0:
            log(f"[bench] generated {args.n}x{args.dim} {args.dtype} in HBM in {generate_seconds:.1f}s; GPU build "
                f"{build_seconds:.1f}s ({args.n / build_seconds:,.0f} vectors/s; search {build_stats['seconds_search']:.1f}s, "
                f"link {build_stats['seconds_link']:.1f}s, {build_stats['batches']} batches, max level {build_stats['max_level']})")
        if not args.no_reload:
            # What a user of the reference's API walks is a LOADED index (`usearch_load` / `usearch_view`): the image the builder
            # saves goes back through the device loader — which places the matrix of stored rows after the settle window
            # (csrc/placement.hpp: the driver hands freed frames back late; an array allocated once they are back lands on the
            # frames it prefers, the fast ones, every time). The builder's own arrays were allocated next to 15 GB of generated data.
            t1 = time.time()
            image = built.save_buffer()
            index = None
            built.close()
            built = None
            torch.cuda.empty_cache()
            usearch_amd.note_device_free()  # torch's blocks went back through another allocator: the loader waits for them too
            t2 = time.time()
            index = usearch_amd.Index.restore(image, device=local_rank)
            torch.cuda.synchronize()
            reload_seconds = {"save_buffer": round(t2 - t1, 2), "settle_and_load": round(time.time() - t2, 2),
                              "settle_ms": index.placement["settle_ms"]}
            if rank == 0:
                log(f"[bench] image of {image.nbytes / 1e9:.1f} GB saved in {t2 - t1:.1f}s, loaded back in "
                    f"{time.time() - t2:.1f}s (of which {index.placement['settle_ms']:.0f} ms waiting for freed frames)")
            if world > 1 or (args.no_cpu_baseline and args.no_placement_check):
                image = None  # only the reference leg and the second load (rank 0, N = 1) need it again
    else:
        from oracle import refbind  # the reference builds the index on the host; never on the timed GPU path
        vectors = synthetic_vectors(args.n, args.dim, args.dtype, seed=data_seed)
        ref_index = refbind.RefIndex(args.dim, metric, args.dtype, args.connectivity, args.expansion_add, 64)
        t0 = time.time()
        ref_index.add(np.arange(args.n, dtype=np.uint64) + key_base, vectors,
                      threads=max(1, args.build_threads // (world if world > 1 else 1)))
        build_seconds = time.time() - t0
        if rank == 0:
            log(f"[bench] reference built {args.n}x{args.dim} {args.dtype} in {build_seconds:.1f}s")
        image = ref_index.save_buffer()
        del vectors
        index = usearch_amd.Index.restore(image, device=local_rank)
    bpv = index.bytes_per_vector
    if rank == 0:
        log(f"[bench] snapshot: {len(index)} vectors, {index.memory_usage / 1e9:.2f} GB HBM, lanes/row {index.lanes_per_row}, "
            f"row stride {index.row_stride}, max level {index.max_level}")

    # ---- queries: out-of-sample, seeded per rank (shards: one batch for all); resident in HBM before the timed region
    queries_dev = synthetic_vectors_device(args.queries, args.dim, args.dtype, 43 if sharded else 43 + 1000 * rank, device)
    queries_host = queries_dev.cpu().numpy().view(NUMPY_STORAGE[args.dtype])
    keys_dev = torch.zeros((args.queries, args.k), dtype=torch.int64, device=device)
    dist_dev = torch.zeros((args.queries, args.k), dtype=torch.float32, device=device)
    counts_dev = torch.zeros(args.queries, dtype=torch.int64, device=device)
    visited_dev = torch.zeros(args.queries, dtype=torch.int64, device=device)
    computed_dev = torch.zeros(args.queries, dtype=torch.int64, device=device)
    stream = torch.cuda.Stream(device)
    tuning = usearch_amd.Tuning(wave_clock=1) if args.wave_clock else None

    sharded_searcher, exchange_ms = None, []
    if sharded:
        from usearch_amd.sharded import gpu_searcher
        sharded_searcher = gpu_searcher(index, rank, world, local_rank, stream=stream.cuda_stream,
                                        prefer="gloo" if rehearsal else args.transport)
        if rank == 0:
            log(f"[bench] sharded step over the '{sharded_searcher.communicator.kind}' transport, {world} shard(s)")

    def search_step(expansion: int, timed: bool):
        if sharded_searcher is not None:
            # ONE native call: broadcast → local search on this rank's shard → packed all-gather (RCCL) → merge kernel
            *_, stats = sharded_searcher.search(queries_dev, args.k, expansion, broadcast_from=0, timed=timed, tuning=tuning,
                                                out=(keys_dev, dist_dev, counts_dev, visited_dev, computed_dev))
            if timed:
                exchange_ms.append(sharded_searcher.last_step.exchange_ms)
            return stats
        return index.search_device(queries_dev.data_ptr(), args.queries, queries_dev.stride(0), args.k, expansion,
                                   keys_dev.data_ptr(), dist_dev.data_ptr(), counts_dev.data_ptr(),
                                   visited_dev.data_ptr(), computed_dev.data_ptr(), stream=stream.cuda_stream,
                                   timed=timed, tuning=tuning)

    # ---- recall@k against EXACT search (the brute-force kernel, bit-checked against the reference's `exact = true` in
    #      tests/test_gpu_exact.py). Replicas: on the whole batch. Shards: every rank scans its shard for a sample, the
    #      per-shard truths are merged on rank 0 (distance ascending). The metric is quoted at the smallest expansion of the
    #      sweep whose recall is >= 0.95 with 95 % confidence (lower end of the interval).
    recall, recall_half, expansion = None, None, args.expansion
    sample = min(args.queries, 10_000) if args.recall_queries < 0 else min(args.recall_queries, args.queries)
    if sharded and args.recall_queries < 0:
        sample = min(1000, args.queries)
    by_distance = args.dtype in ("b1", "i8")
    truth, truth_distances = None, None
    if sample and (rank == 0 or sharded):
        t0 = time.time()
        # ground truth: the matrix-unit exact kernel where the pair has one (f16 / bf16 within float tolerance of the bit-exact
        # kernel, i8 identical to it; tests/test_gpu_exact.py), the wave-per-query exact kernel otherwise
        tiled = metric in ("cos", "ip", "l2sq") and args.dtype in ("f16", "bf16", "i8") and args.k <= 64
        exact = index.search(queries_host[:sample], args.k, dtype=args.dtype, exact="tiled" if tiled else True)
        truth, truth_distances = exact.keys, exact.distances
        exact_ms = exact.stats.kernel_ms
        if sharded and world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, (truth, truth_distances))
            all_keys = np.concatenate([g[0] for g in gathered], axis=1)
            all_distances = np.concatenate([g[1] for g in gathered], axis=1)
            order = np.argsort(np.nan_to_num(all_distances, nan=np.inf), axis=1, kind="stable")[:, :args.k]
            truth = np.take_along_axis(all_keys, order, axis=1)
            truth_distances = np.take_along_axis(all_distances, order, axis=1)
        if rank == 0:
            log(f"[bench] exact ground truth for {sample} queries in {time.time() - t0:.1f}s "
                f"({'matrix-unit' if tiled else 'wave-per-query'} kernel {exact_ms:.1f} ms)"
                + (" (recall counted by distance: ties)" if by_distance else ""))
    sweep = [args.expansion] if args.expansion else [64, 96, 128, 192, 256, 320, 384, 448, 512, 576, 640, 704, 768, 896, 1024]

    sweep_passes = {}

    def recall_at(ef: int):
        """Every rank runs the step (shards: it is a collective); rank 0 scores it."""
        sweep_passes[ef] = int(search_step(ef, False).passes)  # 1 = every traversal fit the default scratch
        if rank != 0 or not sample:
            return None, None
        found = keys_dev[:sample].cpu().numpy().astype(np.uint64)
        if by_distance:
            # integer-valued metrics (Hamming, i8) tie massively: any result at least as close as the exact k-th neighbour is
            # a correct one, whatever its key (the usual tie-aware recall); float metrics compare keys
            per_query = np.mean(dist_dev[:sample].cpu().numpy() <= truth_distances[:, -1:], axis=1)
        else:
            per_query = recall_per_query(found, truth, args.k)
        mean, half = interval(per_query)
        log(f"[bench] ef={ef}: recall@{args.k} = {mean:.4f} +- {half:.4f} on {sample} queries")
        return mean, half

    def agreed(flag: bool) -> bool:
        return rank_zero_decides(flag, rank, world, device)

    if sample:
        below = 0
        for ef in sweep:
            mean, half = recall_at(ef)
            recall, recall_half, expansion = mean, half, ef
            if agreed(rank == 0 and mean - half >= 0.95):
                break
            below = ef
        # the metric is quoted at the SMALLEST expansion that reaches the recall: walk the gap between the last
        # grid point that missed it and the first that made it in steps of 16
        reached = agreed(rank == 0 and recall is not None and recall - recall_half >= 0.95)
        if not args.expansion and reached and below:
            for ef in range(below + 16, expansion, 16):
                mean, half = recall_at(ef)
                if agreed(rank == 0 and mean - half >= 0.95):
                    recall, recall_half, expansion = mean, half, ef
                    break
    if world > 1:
        chosen = torch.tensor([expansion or 64], device=device)
        dist.broadcast(chosen, 0)
        expansion = int(chosen.item())
    expansion = expansion or 64

    # ---- placement: where the workspace's scratch block and the matrix land in HBM decides which of four speeds the walk runs at
    #      (csrc/placement.hpp). The ENGINE tries placements inside the launches that fill the chip — the sweep above was made of such
    #      launches — and lets the walk itself judge them on the caller's queries: nothing to do here but to report what it did
    #      (read again after the timed steps: trials may still be under way during the warm-up).
    # ---- placement: which frames of HBM the matrix of stored rows received decides which of several levels the walk runs at (44.4 …
    #      51.4 ms for this batch over the same bytes, per box and per what the process allocated before: profiles/r06_settled/), and only
    #      the walk itself tells placements apart. The loader placed the matrix after the settle window (reproducible inside the process);
    #      now the host does what a service does before it serves: hands the engine a sample of the batches to come — this batch, at the
    #      expansion just chosen — and lets it place the matrix by trial (`usearch_amd_snapshot_tune`: explicit, synchronous, never inside
    #      a search call). Nothing moves after this line; the timed steps assert it.
    placement_policy = ("the loader places the matrix once, after USEARCH_AMD_SETTLE_MS (1000) since the last big release; then "
                        "usearch_amd_snapshot_tune on the batch at the chosen expansion BEFORE the timed steps: up to 8 fresh device-to-device "
                        "copies timed against the incumbent on the batch's first queries, the faster stays, three wins of the incumbent in "
                        "a row end it; no trial inside any search call; the scratch block is drawn by run_ladder when a chip-filling launch "
                        "needs a new one (<= 8 candidates timed by the launch's first queries)")
    tuned = None

    def tune_placement():
        if args.no_tune or sharded_searcher is not None:
            return None
        t1 = time.perf_counter()
        trials = index.tune_device(queries_dev.data_ptr(), args.queries, queries_dev.stride(0), args.k, expansion)
        report = {"trials": trials, "seconds": round(time.perf_counter() - t1, 2), "moved": index.placement["kept"]}
        if rank == 0:
            log(f"[bench] matrix placed by trial on the batch at ef={expansion}: {trials} trials in {report['seconds']}s, {report['moved']} moved it "
                f"(candidate / incumbent ms: {list(zip(index.placement['judge_ms'], index.placement['incumbent_ms']))})")
        return report

    tuned = tune_placement()
    draws_before_timing = index.placement["draws"]

    # ---- warmup, then EXACTLY `steps` timed steps between barriers
    flush_native_stdio()
    # the device needs a few hundred milliseconds of load to reach its sustained clocks (scripts/variance_probe.py: the first five
    # 50-ms launches after a pause run 4 % slower than the next forty): load it first, then the contract's W warm-up steps
    t_ramp, ramp_steps = time.perf_counter(), 0
    while (ramp_steps < 12) if sharded else (time.perf_counter() - t_ramp < 0.75):
        search_step(expansion, False)
        torch.cuda.synchronize()
        ramp_steps += 1
    for _ in range(args.warmup):
        search_step(expansion, False)
    kernel_ms, tails, step_stats = [], [], []

    def timed_step(_):
        stats = search_step(expansion, True)
        kernel_ms.append(stats.kernel_ms)
        tails.append(stats.tail_idle)
        step_stats.append(stats)

    elapsed = timed_steps(timed_step, args.steps, world, device)
    stats = step_stats[-1]
    passes = max(int(each.passes) for each in step_stats)

    placement = {"matrix": index.placement, "policy": placement_policy, "reload": reload_seconds, "tune": tuned}
    # nothing may have moved the matrix (or timed copies of it) inside the timed steps
    assert placement["matrix"]["draws"] == draws_before_timing, "a placement trial ran inside the timed region"

    # ---- algorithmic bytes of one step from the per-query counters:
    #      B_q = computed·bpv + visited·(4·M0) + k·8 + bpv        (upper-level lists counted at the level-0 size)
    computed = computed_dev.cpu().numpy().astype(np.float64)
    visited = visited_dev.cpu().numpy().astype(np.float64)
    m0 = 2 * index.connectivity
    step_bytes = float(np.sum(computed * bpv + visited * 4 * m0 + args.k * 8 + bpv))
    kernel_s = float(np.mean(kernel_ms)) / 1e3
    achieved = step_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
    # lines touched: what the same accesses cost in whole 128-byte lines (short rows: a 96-byte row is one line, a 16-byte
    # row still one) — the bound that applies when rows are shorter than a line
    row_lines = -(-index.row_stride // 128) if index.row_stride >= 128 else 1
    list_bytes = 128 * -(-4 * m0 // 128)
    inline_rows, lanes_per_row = bool(index.inline_rows), index.lanes_per_row
    if inline_rows:  # the neighbours' rows lie next to the list: a hop is one contiguous block, whatever is fresh
        touched_bytes = float(np.sum(visited * (list_bytes + 128 * -(-m0 * 16 // 128)) + 128 + row_lines * 128))
    else:
        touched_bytes = float(np.sum(computed * row_lines * 128 + visited * list_bytes + 128 + row_lines * 128))

    # ---- the same batch through the HOST-buffer entry point (query upload + result download over PCIe included):
    #      reported for DESIGN.md, never as `value`
    host_api_qps, single_query_us = None, None
    if rank == 0 and world == 1 and not sharded and not args.no_host_api:
        index.expansion_search = expansion
        index.search(queries_host, args.k, dtype=args.dtype)
        t1 = time.perf_counter()
        index.search(queries_host, args.k, dtype=args.dtype)
        host_api_qps = args.queries / (time.perf_counter() - t1)
        # one query at a time, as a `usearch_search` loop would issue them (latency, not throughput)
        for i in range(8):
            index.search(queries_host[i], args.k, dtype=args.dtype)
        t1 = time.perf_counter()
        for i in range(64):
            index.search(queries_host[i], args.k, dtype=args.dtype)
        single_query_us = (time.perf_counter() - t1) / 64 * 1e6

    # ---- the reference on the host cores, same index, same queries, same ef (rank 0, N = 1 only). In sharded mode that is
    #      the reference's own `Indexes` shape over this one shard: search, then `merge_into` — which for a single shard is
    #      the search itself.
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import refbind
        if ref_index is None:  # hand the GPU-built index to the reference: serialize, then `usearch_view_buffer`
            t1 = time.time()
            if image is None:
                image = built.save_buffer()
            ref_index = refbind.RefIndex.from_buffer(image, view=True, dtype=args.dtype)
            log(f"[bench] serialized {image.nbytes / 1e9:.1f} GB for the reference in {time.time() - t1:.1f}s")
        ref_index.expansion_search = expansion
        threads = cores
        # the gate of the metric is recall against EXACT search: SURVEY §8(d) names the reference's `search(…, exact = true)`
        # (index.hpp:4252-4268) as the ground truth. The sweep above scored against the product's own exact kernel; a bounded sample of
        # it is held against the reference's here (about two exact queries a second on these cores for the headline index).
        truth_check = None
        if truth is not None and sample:
            checked = int(min(sample, 32, max(4, 2.5e11 / max(1.0, args.n * bpv))))
            t1 = time.perf_counter()
            tkeys, tdists, *_ = ref_index.search(queries_host[:checked], args.k, dtype=args.dtype, exact=True, threads=threads)
            if by_distance:  # integer-valued metrics tie: the k-th distances must agree, the keys need not
                same = float(np.mean(tdists[:, -1] == truth_distances[:checked, -1]))
            else:
                same = float(np.mean(tkeys == truth[:checked]))
            truth_check = {"queries": checked, "seconds": round(time.perf_counter() - t1, 1),
                           "agreement_of_the_products_exact_kernel_with_reference_exact": same,
                           "what": "k-th distances equal" if by_distance else "labels equal, position by position"}
            log(f"[bench] ground truth: the product's exact kernel against the reference's search(exact=true) on {checked} queries: "
                f"{same:.4f} ({truth_check['what']}; {truth_check['seconds']}s)")
        pilot = min(args.queries, 16 * threads)
        t1 = time.perf_counter()
        ref_index.search(queries_host[:pilot], args.k, dtype=args.dtype, threads=threads)
        rate = pilot / (time.perf_counter() - t1)
        sample_q = int(min(args.queries, max(pilot, rate * args.cpu_seconds)))
        t1 = time.perf_counter()
        rkeys, *_ = ref_index.search(queries_host[:sample_q], args.k, dtype=args.dtype, threads=threads)
        cpu_seconds = time.perf_counter() - t1
        found_keys = keys_dev[:sample_q].cpu().numpy().astype(np.uint64)
        if sharded:  # `Indexes` folds every shard's result with merge_into, the first one into an empty buffer: equal distances
            from oracle import oraclebind  # come out in reverse order (index.hpp:2650-2670); apply it to the reference's rows
            folded = min(sample_q, 2000)
            rkeys_all, rdists_all, rcounts_all, *_ = ref_index.search(queries_host[:folded], args.k, dtype=args.dtype, threads=threads)
            merged_keys = np.zeros_like(rkeys_all)
            for i in range(folded):
                row_keys, row_distances = np.zeros(args.k, dtype=np.uint64), np.zeros(args.k, dtype=np.float32)
                n = int(rcounts_all[i])
                oraclebind.merge_into(row_keys, row_distances, 0, rkeys_all[i, :n], rdists_all[i, :n], n)
                merged_keys[i] = row_keys
            agree = float(np.mean(found_keys[:folded] == merged_keys))
        else:
            agree = float(np.mean(found_keys == rkeys))
        # float-valued pairs walk with the frontier as the open cells of `top` (no heap): the same batch once more with the
        # reference's heap, so that the line says how often the two differ and how both agree with the reference
        frontier_check = None
        if stats.frontier == 2 and not sharded:
            default_counters = (computed_dev.cpu().numpy().copy(), visited_dev.cpu().numpy().copy())
            default_bits = dist_dev.cpu().numpy().view(np.uint32).copy()
            heap_tuning = usearch_amd.Tuning(frontier=1)
            index.search_device(queries_dev.data_ptr(), args.queries, queries_dev.stride(0), args.k, expansion, keys_dev.data_ptr(),
                                dist_dev.data_ptr(), counts_dev.data_ptr(), visited_dev.data_ptr(), computed_dev.data_ptr(),
                                stream=stream.cuda_stream, timed=False, tuning=heap_tuning)
            torch.cuda.synchronize()
            heap_keys = keys_dev.cpu().numpy().astype(np.uint64)
            other_counters = int(np.sum((computed_dev.cpu().numpy() != default_counters[0]) | (visited_dev.cpu().numpy() != default_counters[1])))
            other_keys = int(np.sum((heap_keys[:sample_q] != found_keys).any(axis=1)))
            other_bits = int(np.sum((dist_dev.cpu().numpy().view(np.uint32)[:sample_q] != default_bits[:sample_q]).any(axis=1)))
            frontier_check = {"queries": int(args.queries), "queries_whose_counters_differ_between_frontiers": other_counters,
                              "queries_whose_keys_differ": other_keys, "queries_whose_distance_bits_differ": other_bits,
                              "of_the_first": int(sample_q),
                              "label_agreement_with_reference": {"in_top_frontier": agree,
                                                                 "heap_frontier": float(np.mean(heap_keys[:sample_q] == rkeys))}}
            log(f"[bench] frontier check: {other_counters} of {args.queries} queries walk differently under the reference's heap "
                f"(counters), {other_keys} of the first {sample_q} return other keys; label agreement with the reference "
                f"{agree:.4f} (in-top) / {frontier_check['label_agreement_with_reference']['heap_frontier']:.4f} (heap)")
        # one query at a time on one core: what a `usearch_search` loop sees from the reference (latency, not throughput)
        t1 = time.perf_counter()
        for i in range(32):
            ref_index.search(queries_host[i:i + 1], args.k, dtype=args.dtype, threads=1)
        reference_single_us = (time.perf_counter() - t1) / 32 * 1e6
        cpu = {"value": sample_q / cpu_seconds, "unit": "shard-queries/s" if sharded else "queries/s", "cores": threads,
               "kind": "reference", "label_agreement_with_the_gpu": agree, "frontier_check": frontier_check,
               "ground_truth_check": truth_check,
               "sample": f"{sample_q} of the step's {args.queries} queries, same index, same ef={expansion}, "
                         f"OpenMP static,32 loop of cpp/bench.cpp:352-377; serial (auto-vectorised) metrics, SimSIMD "
                         f"unavailable offline; {cpu_seconds:.1f}s; label agreement with the GPU {agree:.4f}; one query at a "
                         f"time on one core: {reference_single_us:.0f} us"}
        # ---- loading the very same serialized image: the device flattener (upload of levels, tapes and vectors + prefix scans +
        #      scatter kernel) next to the reference's `usearch_load_buffer` (index_dense.hpp:1102-1227)
        load_seconds = None
        if not args.no_load_timing:
            del ref_index
            t1 = time.perf_counter()
            restored = usearch_amd.Index.restore(image, device=local_rank)
            torch.cuda.synchronize()
            ours = time.perf_counter() - t1
            assert len(restored) == args.n
            del restored
            torch.cuda.empty_cache()
            t1 = time.perf_counter()
            loaded = refbind.RefIndex.from_buffer(image, view=False, dtype=args.dtype)
            theirs = time.perf_counter() - t1
            del loaded
            load_seconds = {"image_bytes": int(image.nbytes), "device_flattener": round(ours, 2), "reference_load_buffer": round(theirs, 2)}
            log(f"[bench] loading the {image.nbytes / 1e9:.1f} GB image: {ours:.1f}s into HBM (device flattener), {theirs:.1f}s for the "
                f"reference's usearch_load_buffer")
            ref_index = None
        cpu["load_seconds"] = load_seconds

    # ---- how reproducible the placement is: the index is closed, the very same image loaded a SECOND time (settle window, tuning and
    #      all) and the batch timed on that copy. Outside the timed region; `roofline.kernel_ms_second_load`.
    second_load = None
    if rank == 0 and world == 1 and not sharded and not args.no_placement_check and (image is not None or built is not None):
        try:
            if image is None:
                image = built.save_buffer()
            ref_index = None
            index.close()
            if built is not None:
                built.close()
                built = None
            torch.cuda.empty_cache()
            usearch_amd.note_device_free()
            index = usearch_amd.Index.restore(image, device=local_rank)
            retuned = tune_placement()
            t_ramp = time.perf_counter()
            while time.perf_counter() - t_ramp < 0.75:
                search_step(expansion, False)
                torch.cuda.synchronize()
            again = [search_step(expansion, True).kernel_ms for _ in range(8)][2:]
            second_load = {"kernel_ms": float(np.mean(again)), "settle_ms": index.placement["settle_ms"], "tune": retuned,
                           "frac": step_bytes / (float(np.mean(again)) / 1e3) / 1e9 / HBM_PEAK_GBPS}
            log(f"[bench] the same image loaded a second time: kernel {second_load['kernel_ms']:.2f} ms against {kernel_s * 1e3:.2f} ms in the "
                f"timed steps ({(second_load['kernel_ms'] / (kernel_s * 1e3) - 1) * 100:+.1f} %)")
        except (RuntimeError, MemoryError) as error:
            log(f"[bench] no second-load check: {error}")
    image = None

    stress = None
    if rank == 0 and world == 1 and not sharded and not args.no_stress_rows:
        built = None
        torch.cuda.empty_cache()
        stress = stress_rows(args, metric, device, local_rank, expansion)

    # ---- BASELINE.json's other configurations, each at its full size in a process of its own (this one has released the GPU by then):
    #      the driver's run of the default line observes them all. Only the default workload carries them.
    secondary = None
    default_workload = (args.config in (None, "c3") and (args.n, args.dim, args.dtype, args.queries) == (10_000_000, 768, "f16", 10_000))
    if rank == 0 and world == 1 and not sharded and not args.no_secondary and default_workload:
        try:
            index.close()
        except Exception:
            pass
        index = built = None
        del queries_dev, keys_dev, dist_dev, counts_dev, visited_dev, computed_dev
        torch.cuda.empty_cache()
        secondary = secondary_lines([name for name in args.secondary.split(",") if name in PRESETS and name != "c3"],
                                    args.secondary_timeout)

    if rank == 0:
        workload = (f"{args.n}x{args.dim} {args.dtype} {metric}, batch {args.queries}, k={args.k}, "
                    f"M={args.connectivity}, ef_construction={args.expansion_add}, ef={expansion}")
        # roofline.traffic: HBM bytes per launch from rocprofv3 PMC passes (separate runs by necessity — scripts/profile_round.sh,
        # scripts/pmc_traffic.py). Attached only when it was measured on this very workload with the very sources this line's
        # library is built from; anything else stays null.
        sources = source_hash()
        traffic, traffic_source = None, None
        candidates = [args.traffic_json] if args.traffic_json else []
        for directory, _, files in sorted(os.walk(os.path.join(ROOT, "profiles")), reverse=True):
            if "traffic.json" in files:
                candidates.append(os.path.join(directory, "traffic.json"))
        for candidate in candidates:
            try:
                recorded = json.load(open(candidate))
            except (OSError, ValueError):
                continue
            if recorded.get("workload") == workload and recorded.get("sources") == sources and recorded.get("hbm_bytes_per_launch"):
                traffic = recorded["hbm_bytes_per_launch"]
                traffic_source = os.path.relpath(candidate, ROOT) + " (PMC passes of this workload, same sources)"
                break
        shards = world if sharded else 1
        replicas = 1 if sharded else world
        queries_per_second = args.queries * args.steps * replicas / elapsed  # against the whole index
        line = {
            "metric": f"QPS at recall@{args.k}>=0.95, {args.n * shards}x{args.dim} {args.dtype} {metric}, batch={args.queries}"
                      + (f", {shards} shard(s) of {args.n}" if sharded else ""),
            "value": queries_per_second * shards,
            "unit": "shard-queries/s" if sharded else "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (seeded rank-32 latent + 0.05 noise, out-of-sample queries)"
                    + (" — REHEARSAL: all ranks share one GPU, not a measurement" if rehearsal else ""),
            "config": {"workload": workload,
                       "vectors": args.n * shards, "dimensions": args.dim, "expansion_search": expansion,
                       "recall_at_k": recall, "recall_queries": sample, "recall_ci": [recall - recall_half, recall + recall_half]
                       if recall is not None else None,
                       "parallelism": ("shards" if sharded else "replicas") + str(world),
                       "placement": placement, "preset": args.config, "preflight": preflight,
                       "queries_per_second": queries_per_second,
                       "scaling_definition": ("weak: every GPU holds one shard of --vectors vectors and searches the whole batch; value = "
                                              "batch x shards / time (shard-queries/s), the 1-GPU point is one shard; "
                                              "queries_per_second is the rate against the whole index") if sharded else
                                             "weak: every GPU holds a replica and searches its own batch; value = all batches / time",
                       "exchange": ({"transport": sharded_searcher.communicator.kind,
                                     "block_bytes": int(sharded_searcher.last_step.block_bytes),
                                     "gathered_bytes": int(sharded_searcher.last_step.gathered_bytes),
                                     "exchange_ms": float(np.mean(exchange_ms)) if exchange_ms else None,
                                     "exchanges_per_step": int(sharded_searcher.last_step.exchanges)} if sharded else None),
                       "index_builder": args.builder, "index_build_seconds": round(build_seconds, 1),
                       "index_build": build_stats, "kernel_passes": passes,
                       "kernel_passes_by_expansion": {str(ef): n for ef, n in sorted(sweep_passes.items())},
                       "scratch_mode": {1: "lds", 2: "global-hash", 3: "global"}.get(stats.mode, "?"),
                       "frontier": {1: "heap", 2: "in-top"}.get(stats.frontier, "?"),
                       # usearch_amd_stats_t::variant: 1 / 2 / 3 = 4 / 8 / 12 row loads in flight per lane, 4 = two rows per lane group
                       # per round (2 x 12 loads), 5 = four waves per query (small batches)
                       "kernel_variant": stats.variant,
                       "persistent_waves": stats.grid, "lds_bytes_per_wave": stats.lds_bytes,
                       "rows_inline_with_lists": inline_rows,
                       "visited_set_probe": {"mode": {0: "compare-and-swap", 1: "load, then compare-and-swap", 2: "loads and plain stores, no atomic"}.get(stats.probe_mode),
                                             "seen_cells": stats.seen_cells, "claim_bits": stats.claim_bits,
                                             "build_cut_for_plain_batches": bool(stats.plain), "aside_cells": stats.aside_cells} if stats.mode == 2 else None,
                       "batch_tail_idle": float(np.mean(tails)) if args.wave_clock else None,
                       "host_buffer_api_qps_pcie_inclusive": host_api_qps,
                       "single_query_latency_us_host_api": single_query_us,
                       "sources": sources, "stress_rows": stress, "secondary": secondary},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "kernel_ms_second_load": second_load["kernel_ms"] if second_load else None,
                         "frac_second_load": second_load["frac"] if second_load else None,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "search_kernel", "kernel_ms": kernel_s * 1e3,
                         # template arguments of the timed instantiation as rocprofv3 prints them: metric and scalar codes, lanes
                         # per row, build, scratch mode, `top` cells per lane, frontier, the cut for plain batches (short rows)
                         "kernel_instantiation": (f"search_kernel<{ord({'tanimoto': 't', 'jaccard': 't'}.get(metric, {'cos': 'c', 'ip': 'i', 'l2sq': 'e', 'hamming': 'b', 'pearson': 'p', 'haversine': 'h', 'divergence': 'd', 'sorensen': 's'}.get(metric, '?')))}, "
                                                  f"{ {'b1': 1, 'bf16': 4, 'f64': 10, 'f32': 11, 'f16': 12, 'i8': 23}[args.dtype]}, {lanes_per_row}, "
                                                  f"{stats.variant - 1}, {stats.mode - 1}, {stats.top_cells}, {stats.frontier - 1}, {'true' if stats.plain else 'false'}>"),
                         "algorithmic_bytes_per_launch": step_bytes,
                         "lines_touched_bytes_per_launch": touched_bytes,
                         "lines_touched_frac": touched_bytes / kernel_s / 1e9 / HBM_PEAK_GBPS if kernel_s > 0 else None,
                         "distances_per_query": float(np.mean(computed)), "hops_per_query": float(np.mean(visited))},
            "cpu_baseline": cpu,
        }
        flush_native_stdio()  # the JSON is the last thing this process writes to stdout
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
