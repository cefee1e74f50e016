#!/usr/bin/env python3
"""Host <-> device copy bandwidth on this box (pinned and pageable), for the PCIe-inclusive numbers of DESIGN.md section 8."""
import time, torch
dev = torch.device("cuda", 0)
for mb in (16, 256):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    for name, h in (("pinned", torch.empty(n, dtype=torch.uint8).pin_memory()), ("pageable", torch.empty(n, dtype=torch.uint8))):
        for direction in ("H2D", "D2H"):
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(8):
                    if direction == "H2D": d.copy_(h, non_blocking=True)
                    else: h.copy_(d, non_blocking=True)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print("%4d MB %-8s %s %6.1f GB/s" % (mb, name, direction, 8 * n / dt / 1e9), flush=True)
# two directions at once on two streams
n = 256 << 20
h1 = torch.empty(n, dtype=torch.uint8).pin_memory(); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d1 = torch.empty(n, dtype=torch.uint8, device=dev); d2 = torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8):
    with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("256 MB pinned H2D + D2H concurrently: %.1f GB/s each way" % (8 * n / dt / 1e9))
