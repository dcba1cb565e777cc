#!/usr/bin/env python3
"""Context number: plain bf16 / fp16 GEMM of the C2 shape through torch (hipBLASLt / rocBLAS) on this device.
Not part of the product path; quoted in DESIGN.md next to the fused W4A8 kernel's time."""
import torch

dev = torch.device("cuda:0")
for (M, K, N) in ((2048, 4096, 4096), (2048, 11008, 4096), (2048, 4096, 11008), (16384, 5120, 5120)):
    for dt in (torch.bfloat16, torch.float16):
        a = torch.randn(M, K, device=dev, dtype=dt)
        w = torch.randn(N, K, device=dev, dtype=dt)
        for _ in range(10):
            torch.nn.functional.linear(a, w)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                torch.nn.functional.linear(a, w)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        print(f"{M}x{K}x{N} {str(dt)[6:]:9s} {best * 1e3:8.1f} us  {2.0 * M * K * N / best / 1e9:8.1f} TFLOP/s")
