#!/usr/bin/env python3
"""bench.py — batched HNSW search throughput on MI355X, one process per GPU.

    python bench.py [--gpus N --steps K --warmup W] [--n N --dim D --dtype f16 --metric cos --queries Q --k 10 ...]

A *step* is one pass of the hot path over one batch: `--queries` (default 10 000) queries searched against the
HBM-resident index, inputs and outputs already in HBM. Rank 0 prints ONE JSON line (contract in the task statement),
with two extra objects:
  "roofline":     achieved algorithmic GB/s of the search kernel (bytes from the reference's own per-query counters,
                  SURVEY §8d, ÷ HIP-event kernel time measured on the launch stream) against the 8 TB/s HBM peak;
  "cpu_baseline": the REAL reference (`oracle/_ref`, compiled from /root/reference's own headers) searching the same
                  index with the loop of cpp/bench.cpp:352-377 on all host cores, on a bounded sample (rank 0, N = 1).

The index is BUILT by the reference (index construction is outside the GPU path): the reference library adds
the seeded synthetic vectors on the host cores, serialises the index (`usearch_save_buffer`) and that image is what the
engine uploads. With N > 1 every rank holds a replica and searches its own batch (weak scaling, no collective on the data
path). `--sharded` is the capacity mode: every rank builds and holds its own shard of `--n` vectors
EACH (per-GPU work fixed as N grows = weak scaling of the index size), the batch is broadcast, every rank
searches its shard, per-shard top-k are all-gathered over RCCL and merged (usearch_amd/sharded.py).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (≈6.3 TB/s achievable on streaming copies)
DTYPE_BYTES = {"f32": 4.0, "f16": 2.0, "i8": 1.0, "b1": 0.125}


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
        elif dtype == "i8":
            block = torch.clamp(torch.round(x * (127.0 / 24.0)), -127, 127).to(torch.int8)
        else:  # b1: sign bits, MSB first (cast_to_b1x8_gt, index_plugins.hpp:1139-1158)
            bits = (x > 0).to(torch.uint8)
            if dim % 8:
                bits = torch.nn.functional.pad(bits, (0, 8 - dim % 8))
            block = (bits.view(rows, -1, 8) * weights).sum(dim=2).to(torch.uint8)
        out[begin:begin + rows] = block.contiguous().view(torch.uint8).view(rows, row_bytes)
    return out


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=5)
    parser.add_argument("--warmup", type=int, default=1)
    parser.add_argument("--n", type=int, default=int(os.environ.get("BENCH_N", 200_000)), help="vectors in the index")
    parser.add_argument("--dim", type=int, default=768)
    parser.add_argument("--dtype", default="f16", choices=list(DTYPE_BYTES))
    parser.add_argument("--metric", default=None)
    parser.add_argument("--queries", type=int, default=10_000)
    parser.add_argument("--k", type=int, default=10)
    parser.add_argument("--expansion", type=int, default=0,
                        help="0 = smallest of 64/96/128/192/256/320/384/512/768/1024 with recall@k >= 0.95")
    parser.add_argument("--connectivity", type=int, default=16)
    parser.add_argument("--expansion-add", type=int, default=128)
    parser.add_argument("--recall-queries", type=int, default=1000)
    parser.add_argument("--cpu-seconds", type=float, default=12.0)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--sharded", action="store_true")
    parser.add_argument("--build-threads", type=int, default=int(os.environ.get("BENCH_BUILD_THREADS", 0)),
                        help="threads the reference uses to build the index (0 = 2 x the cgroup CPU quota)")
    parser.add_argument("--cache-dir", default=os.environ.get("BENCH_CACHE_DIR", ""),
                        help="keep the reference-built index image here between runs (e.g. /dev/shm)")
    args = parser.parse_args()
    metric = args.metric or ("hamming" if args.dtype == "b1" else "l2sq" if args.dtype == "i8" else "cos")
    cores = host_cores()
    if not args.build_threads:
        args.build_threads = 2 * cores

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import usearch_amd
    from oracle import refbind  # the reference builds the index and is the cpu_baseline; never on the timed GPU path

    cache_path = None
    if args.cache_dir:
        cache_path = os.path.join(args.cache_dir, f"usearch_amd_{args.n}x{args.dim}{args.dtype}_{metric}_m{args.connectivity}"
                                                  f"_efa{args.expansion_add}.usearch")
    # ---- the index: built once by the reference on the host cores (rank 0), shared with the other ranks through /dev/shm
    image_path = f"/dev/shm/usearch_amd_bench_{os.environ.get('MASTER_PORT', '0')}_{args.n}x{args.dim}{args.dtype}.usearch"
    build_seconds = 0.0
    ref_index = None
    shard_base = 0
    if args.sharded and world > 1:
        per_shard = args.n  # per-GPU shard size is fixed; the total index grows with the number of GPUs
        shard_base = rank * per_shard
        vectors = synthetic_vectors(per_shard, args.dim, args.dtype, seed=42 + rank)
        ref_index = refbind.RefIndex(args.dim, metric, args.dtype, args.connectivity, args.expansion_add, 64)
        t0 = time.time()
        ref_index.add(np.arange(per_shard, dtype=np.uint64) + shard_base, vectors,
                      threads=max(1, 2 * cores // world))
        build_seconds = time.time() - t0
        image = ref_index.save_buffer()
    else:
        if rank == 0 and cache_path and os.path.exists(cache_path):
            image = np.fromfile(cache_path, dtype=np.uint8)
            ref_index = refbind.RefIndex.from_buffer(image, view=False, dtype=args.dtype)
            log(f"[bench] reusing the reference-built index {cache_path}")
        elif rank == 0:
            vectors = synthetic_vectors(args.n, args.dim, args.dtype, seed=42)
            ref_index = refbind.RefIndex(args.dim, metric, args.dtype, args.connectivity, args.expansion_add, 64)
            t0 = time.time()
            ref_index.add(np.arange(args.n, dtype=np.uint64), vectors, threads=args.build_threads)
            build_seconds = time.time() - t0
            log(f"[bench] reference built {args.n}x{args.dim} {args.dtype} in {build_seconds:.1f}s "
                f"on {args.build_threads or refbind.max_threads()} threads")
            image = ref_index.save_buffer()
            if cache_path:
                image.tofile(cache_path)
        if rank == 0:
            if world > 1:
                image.tofile(image_path)
        if world > 1:
            dist.barrier()
            if rank != 0:
                image = np.fromfile(image_path, dtype=np.uint8)
            dist.barrier()
            if rank == 0:
                os.unlink(image_path)
    t0 = time.time()
    index = usearch_amd.Index.restore(image, device=local_rank)
    upload_seconds = time.time() - t0
    bpv = index.bytes_per_vector
    if rank == 0:
        log(f"[bench] snapshot: {len(index)} vectors, {index.memory_usage / 1e9:.2f} GB HBM, upload {upload_seconds:.1f}s, "
            f"lanes/row {index.lanes_per_row}, row stride {index.row_stride}")

    # ---- queries: out-of-sample, seeded per rank; resident in HBM before the timed region
    queries_host = synthetic_vectors(args.queries, args.dim, args.dtype, seed=43 if args.sharded else 43 + 1000 * rank)
    queries_dev = torch.from_numpy(queries_host.view(np.uint8).reshape(args.queries, -1)).to(device)
    keys_dev = torch.zeros((args.queries, args.k), dtype=torch.int64, device=device)
    dist_dev = torch.zeros((args.queries, args.k), dtype=torch.float32, device=device)
    counts_dev = torch.zeros(args.queries, dtype=torch.int64, device=device)
    visited_dev = torch.zeros(args.queries, dtype=torch.int64, device=device)
    computed_dev = torch.zeros(args.queries, dtype=torch.int64, device=device)
    stream = torch.cuda.Stream(device)

    sharded_searcher = None
    if args.sharded and world > 1:
        from usearch_amd.sharded import gpu_searcher
        sharded_searcher = gpu_searcher(index, stream=stream.cuda_stream)
        merged = {}

    def search_step(expansion: int, timed: bool):
        if sharded_searcher is not None:
            # broadcast the batch → local search on this rank's shard → all-gather (RCCL) → merge kernel
            merged["keys"], merged["distances"], merged["counts"] = sharded_searcher.search(queries_dev, args.k, expansion)
            visited_dev.copy_(sharded_searcher.local_search.last_visited)
            computed_dev.copy_(sharded_searcher.local_search.last_computed)
            return usearch_amd.Stats(passes=1)
        return index.search_device(queries_dev.data_ptr(), args.queries, queries_dev.stride(0), args.k, expansion,
                                   keys_dev.data_ptr(), dist_dev.data_ptr(), counts_dev.data_ptr(),
                                   visited_dev.data_ptr(), computed_dev.data_ptr(), stream=stream.cuda_stream,
                                   timed=timed)

    # ---- recall@k against the reference's own exact search (index.hpp:4252-4268) on a sample; pick ef
    recall, expansion = None, args.expansion
    sample = min(args.recall_queries, args.queries)
    if rank == 0 and not args.sharded and sample:
        truth, *_ = ref_index.search(queries_host[:sample], args.k, dtype=args.dtype, exact=True, threads=2 * cores)
        sweep = [args.expansion] if args.expansion else [64, 96, 128, 192, 256, 320, 384, 512, 768, 1024]
        for ef in sweep:
            search_step(ef, False)
            found = keys_dev[:sample].cpu().numpy().astype(np.uint64)
            recall = float(np.mean([len(np.intersect1d(found[i], truth[i])) / args.k for i in range(sample)]))
            expansion = ef
            log(f"[bench] ef={ef}: recall@{args.k} = {recall:.4f} on {sample} queries")
            if recall >= 0.95:
                break
    if world > 1:
        chosen = torch.tensor([expansion or 64], device=device)
        dist.broadcast(chosen, 0)
        expansion = int(chosen.item())
    expansion = expansion or (256 if args.sharded else 64)

    # ---- warmup, then EXACTLY `steps` timed steps between barriers
    for _ in range(args.warmup):
        search_step(expansion, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    kernel_ms, passes = [], 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = search_step(expansion, True)
        kernel_ms.append(stats.kernel_ms)
        passes = max(passes, stats.passes)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- algorithmic bytes of one step from the per-query counters:
    #      B_q = computed·bpv + visited·(4·M0) + k·8 + bpv        (upper-level lists counted at the level-0 size)
    computed = computed_dev.cpu().numpy().astype(np.float64)
    visited = visited_dev.cpu().numpy().astype(np.float64)
    m0 = 2 * index.connectivity
    step_bytes = float(np.sum(computed * bpv + visited * 4 * m0 + args.k * 8 + bpv))
    kernel_s = float(np.mean(kernel_ms)) / 1e3
    achieved = step_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0

    # ---- the same batch through the HOST-buffer entry point (query upload + result download over PCIe included):
    #      reported for DESIGN.md, never as `value`
    host_api_qps = None
    if rank == 0 and world == 1:
        index.expansion_search = expansion
        index.search(queries_host, args.k, dtype=args.dtype)
        t1 = time.perf_counter()
        index.search(queries_host, args.k, dtype=args.dtype)
        host_api_qps = args.queries / (time.perf_counter() - t1)

    # ---- the reference on the host cores, same index, same queries, same ef (rank 0, N = 1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ref_index.expansion_search = expansion
        threads = cores
        pilot = min(args.queries, 64 * threads)
        t1 = time.perf_counter()
        ref_index.search(queries_host[:pilot], args.k, dtype=args.dtype, threads=threads)
        rate = pilot / (time.perf_counter() - t1)
        sample_q = int(min(args.queries, max(pilot, rate * args.cpu_seconds)))
        t1 = time.perf_counter()
        rkeys, *_ = ref_index.search(queries_host[:sample_q], args.k, dtype=args.dtype, threads=threads)
        cpu_seconds = time.perf_counter() - t1
        agree = float(np.mean(keys_dev[:sample_q].cpu().numpy().astype(np.uint64) == rkeys))
        cpu = {"value": sample_q / cpu_seconds, "unit": "queries/s", "cores": threads, "kind": "reference",
               "sample": f"{sample_q} of the step's {args.queries} queries, same index, same ef={expansion}, "
                         f"OpenMP static,32 loop of cpp/bench.cpp:352-377; serial (auto-vectorised) metrics, SimSIMD "
                         f"unavailable offline; {cpu_seconds:.1f}s; label agreement with the GPU {agree:.4f}"}

    if rank == 0:
        total_queries = args.queries * args.steps * (1 if args.sharded else world)
        total_vectors = args.n * (world if args.sharded else 1)
        line = {
            "metric": f"QPS at recall@{args.k}>=0.95, {args.metric or metric} {args.dtype}, batch={args.queries}",
            "value": total_queries / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (seeded rank-32 latent + 0.05 noise, out-of-sample queries)",
            "config": {"workload": f"{args.n}x{args.dim} {args.dtype} {metric}, batch {args.queries}, k={args.k}, "
                                   f"M={args.connectivity}, ef_construction={args.expansion_add}, ef={expansion}",
                       "vectors": total_vectors, "dimensions": args.dim, "expansion_search": expansion,
                       "recall_at_k": recall, "parallelism": ("shards" if args.sharded else "replicas") + str(world),
                       "index_build_seconds": round(build_seconds, 1), "kernel_passes": passes,
                       "scratch_mode": {1: "lds", 2: "global-hash", 3: "global"}.get(stats.mode, "?"),
                       "persistent_waves": stats.grid, "lds_bytes_per_wave": stats.lds_bytes,
                       "host_buffer_api_qps_pcie_inclusive": host_api_qps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": "search_kernel", "kernel_ms": kernel_s * 1e3,
                         "algorithmic_bytes_per_launch": step_bytes,
                         "distances_per_query": float(np.mean(computed)), "hops_per_query": float(np.mean(visited))},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
