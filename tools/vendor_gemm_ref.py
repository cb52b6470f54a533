#!/usr/bin/env python3
"""Reference point for the projection GEMMs: what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) sustains on the encoder's GEMM shapes with random
16-bit operands on THIS device, next to this library's kernels measured in the same process (tools/one_shape.py-style per-kernel timing is in
profiles/rNN/encoder_forward_breakdown.txt).  Not a product path: the product never calls a vendor GEMM.  Usage: python tools/vendor_gemm_ref.py"""
import time
import torch

dev = torch.device("cuda:0")
shapes = [("QKV  (K=1024, N=3072)", 1024, 3072), ("out  (K=1024, N=1024)", 1024, 1024), ("FF1  (K=1024, N=4096)", 1024, 4096), ("FF2  (K=4096, N=1024)", 4096, 1024)]
for T in (32000, 131072):
    for dt in (torch.bfloat16, torch.float16):
        for name, K, N in shapes:
            a = torch.randn(T, K, device=dev, dtype=torch.float32).to(dt)
            w = (0.02 * torch.randn(N, K, device=dev, dtype=torch.float32)).to(dt)
            for _ in range(3):
                c = a @ w.t()
            torch.cuda.synchronize()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                c = a @ w.t()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            print(f"T={T:6d} {str(dt)[6:]:8s} {name}: {ms:7.3f} ms  {2.0 * T * K * N / ms / 1e9:7.1f} TFLOP/s  ({2.0 * T * K * N / ms / 1e9 / 2500:.3f} of 2.5 PF)", flush=True)
            del a, w, c
