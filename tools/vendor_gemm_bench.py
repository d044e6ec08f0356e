"""Vendor-library reference point for the encoder GEMM shapes: torch.mm / F.linear in bf16 (hipBLASLt, then rocBLAS) at the four
shapes of one encoder layer at B = 8 and B = 64, timed with HIP events on operand copies that rotate through more memory than
the Infinity Cache holds.  Not part of the product or the tests: a yardstick for csrc/gemm.hip (profiles/r04_vendor_gemm.txt)."""
import sys, torch
import torch.nn.functional as F

dev = "cuda:0"
shapes = [("q/k/v", 3840, 1280, False), ("out-proj", 1280, 1280, False), ("fc1+bias+GELU", 5120, 1280, True), ("fc2", 1280, 5120, False)]
for lib in ("hipblaslt", "rocblas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:   # noqa
        print("cannot select", lib, e); continue
    for M in (12000, 96000):
        for name, N, K, gelu in shapes:
            nbuf = max(2, int(600e6 // (M * K * 2 + M * N * 2)) + 1)
            As = [torch.randn(M, K, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
            W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) / K ** 0.5
            b = torch.randn(N, device=dev, dtype=torch.bfloat16)
            outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
            def run(i):
                if gelu:
                    return torch._addmm_activation(b, As[i % nbuf], W.t(), use_gelu=True)
                return torch.mm(As[i % nbuf], W.t(), out=outs[i % nbuf])
            for i in range(5): run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 30
            e0.record()
            for i in range(reps): run(i)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            print(f"{lib:10s} M={M:6d} {name:14s} N={N:5d} K={K:5d}: {us:8.1f} us/launch  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
            sys.stdout.flush()
            del As, outs
