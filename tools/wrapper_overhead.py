#!/usr/bin/env python3
"""Host time of the pieces a Python op wrapper is made of (us per call)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch_robotics_amd._lib import lib
dev = torch.device("cuda:0")
x = torch.zeros(1024, 7, device=dev)
L = lib()
def t(name, fn, n=20000):
    for _ in range(200): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    print(f"{name:55s} {(time.perf_counter() - t0) / n * 1e6:6.2f} us")
def ctx():
    with torch.cuda.device(dev): pass
t("with torch.cuda.device(dev): pass", ctx)
t("torch.cuda.current_device()", lambda: torch.cuda.current_device())
t("torch.cuda.current_stream(dev).cuda_stream", lambda: torch.cuda.current_stream(dev).cuda_stream)
t("torch.empty((n, 3), device, dtype)", lambda: torch.empty((1024, 3), device=dev, dtype=torch.float32))
t("x.data_ptr()", lambda: x.data_ptr())
t("x.contiguous()", lambda: x.contiguous())
t("x.reshape(-1, 7)", lambda: x.reshape(-1, 7))
t("ctypes call, no arguments (trk_spec_count)", lambda: L.trk_spec_count())
t("x.is_cuda / dtype checks", lambda: (x.is_cuda, x.dtype != torch.float32))
