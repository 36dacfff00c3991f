#!/usr/bin/env python3
"""scripts/box_check.py — is a slow box slow at everything? The headline batch runs in 44.5 … 45.1 ms on some boxes of the pool and in
51.3 ms on others whatever placement is tried (profiles/r06_settled/). Three yardsticks that have nothing to do with the walk: a
device-to-device stream copy (GB/s), the library GEMM at the exact search's shape (TFLOP/s; fast box: 1 024) and a dependency-free
random gather of 1.5-KB rows out of a 15-GB array (GB/s)."""
import time

import torch


def main() -> None:
    device = torch.device("cuda", 0)
    a = torch.empty(8 << 30, dtype=torch.uint8, device=device)
    b = torch.empty_like(a)
    a.fill_(1)
    for _ in range(3):
        b.copy_(a)
    torch.cuda.synchronize()
    begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    begin.record()
    for _ in range(10):
        b.copy_(a)
    end.record()
    torch.cuda.synchronize()
    ms = begin.elapsed_time(end) / 10
    print(f"stream copy of 8 GiB: {ms:.2f} ms = {2 * a.numel() / ms / 1e6:.0f} GB/s read + written", flush=True)
    del a, b
    queries = torch.randn((10_000, 768), device=device).half()
    matrix = torch.randn((1_000_000, 768), device=device).half()
    out = torch.empty((10_000, 1_000_000), device=device, dtype=torch.float16)
    for _ in range(3):
        torch.matmul(queries, matrix.T, out=out)
    torch.cuda.synchronize()
    begin.record()
    for _ in range(10):
        torch.matmul(queries, matrix.T, out=out)
    end.record()
    torch.cuda.synchronize()
    ms = begin.elapsed_time(end) / 10
    print(f"10000 x 768 @ 768 x 1000000 f16: {ms:.3f} ms = {2.0 * 10_000 * 1_000_000 * 768 / ms / 1e9:.0f} TFLOP/s", flush=True)
    del matrix, out, queries
    rows = torch.empty((10_000_000, 768), dtype=torch.float16, device=device)
    rows.fill_(0.5)
    picks = torch.randint(0, 10_000_000, (2_000_000,), device=device)
    for _ in range(3):
        gathered = rows[picks]
    torch.cuda.synchronize()
    begin.record()
    for _ in range(10):
        gathered = rows[picks]
    end.record()
    torch.cuda.synchronize()
    ms = begin.elapsed_time(end) / 10
    print(f"gather of 2M random 1536-byte rows out of 15.4 GB (torch index_select): {ms:.2f} ms = {picks.numel() * 1536 / ms / 1e6:.0f} GB/s read", flush=True)


if __name__ == "__main__":
    main()
