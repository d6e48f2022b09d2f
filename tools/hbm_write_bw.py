#!/usr/bin/env python3
"""Sustained write / copy bandwidth against the working-set size (the 256 MB Infinity Cache absorbs smaller ones)."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (32, 64, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    tf = t(lambda: x.fill_(1.0)); tc = t(lambda: y.copy_(x))
    print(f"{mb:5d} MB: fill {mb * 1.048576e6 / tf / 1e12:5.2f} TB/s written | copy {2 * mb * 1.048576e6 / tc / 1e12:5.2f} TB/s read+written")
