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

Default workload = the configuration BASELINE.json's metric is quoted on: 10M x 768 f16 cosine, batch 10 000, k = 10.
The seeded synthetic vectors are generated in HBM and the index is built ON THE GPU (`usearch_amd.build`, ≈20 s for 10M;
`--builder reference` lets the reference build it on the host cores instead — minutes per million vectors, so only for
small `--n`). The reference is handed the very same index (`save_buffer` → `usearch_view_buffer`) for the CPU baseline.
With N > 1 every rank builds (deterministically, so identically) and holds a replica and searches its own batch (weak
scaling, no collective on the data path). `--sharded` is the capacity mode: every rank builds and holds its own shard of
`--n` vectors EACH (per-GPU work fixed as N grows = weak scaling of the index size), the batch is broadcast, every rank
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


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=5)
    parser.add_argument("--warmup", type=int, default=1)
    parser.add_argument("--n", type=int, default=int(os.environ.get("BENCH_N", 10_000_000)), help="vectors in the index")
    parser.add_argument("--dim", type=int, default=768)
    parser.add_argument("--dtype", default="f16", choices=list(DTYPE_BYTES))
    parser.add_argument("--metric", default=None)
    parser.add_argument("--queries", type=int, default=10_000)
    parser.add_argument("--k", type=int, default=10)
    parser.add_argument("--expansion", type=int, default=0,
                        help="0 = smallest of the sweep (64 ... 1024) with recall@k >= 0.95")
    parser.add_argument("--connectivity", type=int, default=16)
    parser.add_argument("--expansion-add", type=int, default=128)
    parser.add_argument("--recall-queries", type=int, default=1000)
    parser.add_argument("--cpu-seconds", type=float, default=12.0)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--sharded", action="store_true")
    parser.add_argument("--builder", default=os.environ.get("BENCH_BUILDER", "gpu"), choices=["gpu", "reference"],
                        help="who links the index: the device builder (default) or the reference on the host cores")
    parser.add_argument("--build-threads", type=int, default=int(os.environ.get("BENCH_BUILD_THREADS", 0)),
                        help="threads the reference uses to build the index (0 = 2 x the cgroup CPU quota)")
    parser.add_argument("--traffic-json", default=os.environ.get("BENCH_TRAFFIC_JSON", ""),
                        help="PMC-derived HBM bytes per launch of this workload (scripts/pmc_traffic.py), copied into "
                             "roofline.traffic")
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

    # ---- the index. Replicas: every rank builds the same seeded data (the device build is deterministic). Shards: rank r
    #      builds its own `--n` vectors with keys offset by r * n.
    sharded = args.sharded and world > 1
    data_seed = 42 + (rank if sharded else 0)
    key_base = rank * args.n if sharded else 0
    build_seconds, build_stats, ref_index, image, built = 0.0, None, None, None, None
    t_generate = time.time()
    if args.builder == "gpu":
        data = synthetic_vectors_device(args.n, args.dim, args.dtype, data_seed, device)
        torch.cuda.synchronize()
        generate_seconds = time.time() - t_generate
        keys = np.arange(args.n, dtype=np.uint64) + key_base if key_base else None
        t0 = time.time()
        built = usearch_amd.build(None, metric, args.dtype, keys=keys, connectivity=args.connectivity,
                                  expansion_add=args.expansion_add, device=local_rank, device_pointer=data.data_ptr(),
                                  count=args.n, stride=data.stride(0), ndim=args.dim)
        build_seconds = time.time() - t0
        del data
        torch.cuda.empty_cache()
        index = built.index
        build_stats = built.stats.as_dict()
        if rank == 0:
            log(f"[bench] generated {args.n}x{args.dim} {args.dtype} in HBM in {generate_seconds:.1f}s; GPU build "
                f"{build_seconds:.1f}s ({args.n / build_seconds:,.0f} vectors/s; search {build_stats['seconds_search']:.1f}s, "
                f"link {build_stats['seconds_link']:.1f}s, {build_stats['batches']} batches, max level {build_stats['max_level']})")
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

    # ---- queries: out-of-sample, seeded per rank; resident in HBM before the timed region
    queries_dev = synthetic_vectors_device(args.queries, args.dim, args.dtype, 43 if sharded else 43 + 1000 * rank, device)
    queries_host = queries_dev.cpu().numpy().view(NUMPY_STORAGE[args.dtype])
    keys_dev = torch.zeros((args.queries, args.k), dtype=torch.int64, device=device)
    dist_dev = torch.zeros((args.queries, args.k), dtype=torch.float32, device=device)
    counts_dev = torch.zeros(args.queries, dtype=torch.int64, device=device)
    visited_dev = torch.zeros(args.queries, dtype=torch.int64, device=device)
    computed_dev = torch.zeros(args.queries, dtype=torch.int64, device=device)
    stream = torch.cuda.Stream(device)

    sharded_searcher = None
    if sharded:
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

    # ---- recall@k against EXACT search (the brute-force kernel, bit-checked against the reference's `exact = true` in
    #      tests/test_gpu_exact.py) on a sample; pick the smallest ef of the sweep that reaches 0.95
    recall, expansion = None, args.expansion
    sample = min(args.recall_queries, args.queries)
    if rank == 0 and not sharded and sample:
        t0 = time.time()
        exact = index.search(queries_host[:sample], args.k, dtype=args.dtype, exact=True)
        truth, truth_distances = exact.keys, exact.distances
        # integer-valued metrics (Hamming, i8) tie massively: any result at least as close as the exact k-th neighbour is a
        # correct one, whatever its key (the usual tie-aware recall); float metrics compare keys
        by_distance = args.dtype in ("b1", "i8")
        log(f"[bench] exact ground truth for {sample} queries in {time.time() - t0:.1f}s"
            + (" (recall counted by distance: ties)" if by_distance else ""))
        sweep = [args.expansion] if args.expansion else [64, 96, 128, 192, 256, 320, 384, 448, 512, 576, 640, 704, 768, 896, 1024]

        def recall_at(ef: int) -> float:
            search_step(ef, False)
            found = keys_dev[:sample].cpu().numpy().astype(np.uint64)
            if by_distance:
                found_distances = dist_dev[:sample].cpu().numpy()
                value = float(np.mean(found_distances <= truth_distances[:, -1:]))
            else:
                value = float(np.mean([len(np.intersect1d(found[i], truth[i])) / args.k for i in range(sample)]))
            log(f"[bench] ef={ef}: recall@{args.k} = {value:.4f} on {sample} queries")
            return value

        below = 0
        for ef in sweep:
            recall, expansion = recall_at(ef), ef
            if recall >= 0.95:
                break
            below = ef
        # the metric is quoted at the SMALLEST expansion that reaches the recall: walk the gap between the last
        # grid point that missed it and the first that made it in steps of 16
        if not args.expansion and recall >= 0.95 and below:
            for ef in range(below + 16, expansion, 16):
                finer = recall_at(ef)
                if finer >= 0.95:
                    recall, expansion = finer, ef
                    break
    if world > 1:
        chosen = torch.tensor([expansion or 64], device=device)
        dist.broadcast(chosen, 0)
        expansion = int(chosen.item())
    expansion = expansion or (256 if sharded else 64)

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
        from oracle import refbind
        if ref_index is None:  # hand the GPU-built index to the reference: serialize, then `usearch_view_buffer`
            t1 = time.time()
            image = built.save_buffer()
            ref_index = refbind.RefIndex.from_buffer(image, view=True, dtype=args.dtype)
            log(f"[bench] serialized {image.nbytes / 1e9:.1f} GB for the reference in {time.time() - t1:.1f}s")
        ref_index.expansion_search = expansion
        threads = cores
        pilot = min(args.queries, 16 * threads)
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
        workload = (f"{args.n}x{args.dim} {args.dtype} {metric}, batch {args.queries}, k={args.k}, "
                    f"M={args.connectivity}, ef_construction={args.expansion_add}, ef={expansion}")
        # roofline.traffic: HBM bytes per launch from rocprofv3 PMC passes (separate runs by necessity — scripts/profile_round.sh,
        # scripts/pmc_traffic.py). Either handed in (--traffic-json), or the committed measurement of this very workload.
        traffic, traffic_source = None, None
        if args.traffic_json and os.path.exists(args.traffic_json):
            traffic, traffic_source = json.load(open(args.traffic_json)).get("hbm_bytes_per_launch"), args.traffic_json
        else:
            for directory in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
                candidate = os.path.join(ROOT, "profiles", directory, "bench.json")
                try:
                    recorded = json.load(open(candidate))
                except (OSError, ValueError):
                    continue
                if recorded.get("config", {}).get("workload") == workload and recorded.get("roofline", {}).get("traffic"):
                    traffic, traffic_source = recorded["roofline"]["traffic"], f"profiles/{directory}/traffic.json (PMC passes of the same workload)"
                    break
        total_queries = args.queries * args.steps * (1 if sharded else world)
        total_vectors = args.n * (world if sharded else 1)
        line = {
            "metric": f"QPS at recall@{args.k}>=0.95, {args.n}x{args.dim} {args.dtype} {metric}, batch={args.queries}",
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
            "config": {"workload": workload,
                       "vectors": total_vectors, "dimensions": args.dim, "expansion_search": expansion,
                       "recall_at_k": recall, "parallelism": ("shards" if sharded else "replicas") + str(world),
                       "index_builder": args.builder, "index_build_seconds": round(build_seconds, 1),
                       "index_build": build_stats, "kernel_passes": passes,
                       "scratch_mode": {1: "lds", 2: "global-hash", 3: "global"}.get(stats.mode, "?"),
                       "persistent_waves": stats.grid, "lds_bytes_per_wave": stats.lds_bytes,
                       "host_buffer_api_qps_pcie_inclusive": host_api_qps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
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
