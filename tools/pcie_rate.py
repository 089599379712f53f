"""PCIe-inclusive rate of the batch front-end: H2D copy of a batch of 640x480 gray (u8) + depth (u16) frames from pinned host
memory, alone and overlapped with the device step on a second stream (DESIGN.md section 4 quotes the result; never `value`)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    gray_h = torch.empty((B, 480, 640), dtype=torch.uint8).pin_memory()
    depth_h = torch.empty((B, 480, 640), dtype=torch.int16).pin_memory()
    gray_d = torch.empty_like(gray_h, device="cuda")
    depth_d = torch.empty_like(depth_h, device="cuda")
    for _ in range(2):
        gray_d.copy_(gray_h, non_blocking=True); depth_d.copy_(depth_h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        gray_d.copy_(gray_h, non_blocking=True); depth_d.copy_(depth_h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    mb = (gray_h.numel() + depth_h.numel() * 2) / 1e6
    print(f"batch {B}: {mb:.0f} MB host->device in {dt * 1e3:.2f} ms = {mb / dt / 1e3:.1f} GB/s -> {B / dt:.0f} frames/s if the copy were the only cost")


if __name__ == "__main__":
    main()
