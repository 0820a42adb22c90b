#!/usr/bin/env python3
"""Raw pinned host <-> device copy rate on this box (the ceiling of the upload path)."""
import torch, time
for mb in (48, 256):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device="cuda")
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f"pinned H2D {mb} MiB: {mb * 1.048576e6 / dt / 1e9:.1f} GB/s")
    t = time.perf_counter()
    for _ in range(10): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f"pinned D2H {mb} MiB: {mb * 1.048576e6 / dt / 1e9:.1f} GB/s")
