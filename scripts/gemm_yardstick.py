#!/usr/bin/env python3
"""scripts/gemm_yardstick.py — what this chip gives a LIBRARY GEMM at the exact search's shape: 10 000 queries × 768 dimensions against
slices of the 10M × 768 f16 matrix (torch.matmul → hipBLASLt / rocBLAS), f16 in, f32 out. The exact tile of csrc/exact_tiled.hip does the
same product AND the per-query top-k in one kernel (`bench.py --exact`: 2·10⁴·10⁷·768 FLOP per batch); the plain product alone is
the yardstick for what its inner loop could reach. Prints TFLOP/s per slice size.

    python scripts/gemm_yardstick.py [--queries 10000] [--dim 768] [--rows 1000000 2000000 4000000]"""
import argparse

import torch


def main() -> None:
    parser = argparse.ArgumentParser()
    parser.add_argument("--queries", type=int, default=10_000)
    parser.add_argument("--dim", type=int, default=768)
    parser.add_argument("--rows", type=int, nargs="+", default=[1_000_000, 2_000_000, 4_000_000])
    parser.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    args = parser.parse_args()
    device = torch.device("cuda", 0)
    dtype = torch.float16 if args.dtype == "f16" else torch.bfloat16
    queries = torch.randn((args.queries, args.dim), device=device, dtype=torch.float32).to(dtype)
    for rows in args.rows:
        matrix = torch.randn((rows, args.dim), device=device, dtype=torch.float32).to(dtype)
        out = torch.empty((args.queries, rows), device=device, dtype=dtype)  # (an f32 output of 10 000 × 4M would be 160 GB)
        for _ in range(3):
            torch.matmul(queries, matrix.T, out=out)
        torch.cuda.synchronize()
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        begin.record()
        steps = 10
        for _ in range(steps):
            torch.matmul(queries, matrix.T, out=out)
        end.record()
        torch.cuda.synchronize()
        ms = begin.elapsed_time(end) / steps
        flops = 2.0 * args.queries * rows * args.dim
        print(f"{args.queries} x {args.dim} @ {args.dim} x {rows} {args.dtype}: {ms:.3f} ms = {flops / ms / 1e9:.0f} TFLOP/s "
              f"(output {args.queries * rows * 2 / 1e9:.1f} GB written per product: {args.queries * rows * 2 / ms / 1e6:.0f} GB/s)", flush=True)
        del matrix, out
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
