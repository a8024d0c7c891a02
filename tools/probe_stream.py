#!/usr/bin/env python3
"""What a plain streaming read / copy reaches on this box (torch kernels), to put the SpMM's GB/s in proportion."""
import torch

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for mb in (128, 512, 2048):
        a = torch.empty(mb * 1024 * 1024 // 8, dtype=torch.float64, device=dev).normal_()
        b = torch.empty_like(a)
        for name, fn, nbytes in (("sum (read)", lambda: a.sum(), a.numel() * 8), ("copy (read+write)", lambda: b.copy_(a), 2 * a.numel() * 8)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print(f"{mb:5d} MiB {name:18s}: {ms * 1e3:8.1f} us -> {nbytes / 1e9 / (ms / 1e3):6.0f} GB/s", flush=True)
