"""Timing of the C_in = 1 convolution kernels at the bench batch (developer tool, GPU box only).   python tools/bench_c1.py [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from qea import ops  # noqa: E402
from bench_convt import timeit  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    H, W = 32, 128
    for Co in (32, 64):
        x = torch.rand(B, H, W, device="cuda")
        w = torch.randn(Co, 9, device="cuda")
        b = torch.randn(Co, device="cuda")
        y = torch.empty(B, H, W, Co, device="cuda")
        dy = torch.randn(B, H, W, Co, device="cuda")
        dw, db = torch.empty(Co, 9, device="cuda"), torch.empty(Co, device="cuda")
        dx = torch.empty(B, H, W, device="cuda")
        gb = B * H * W * Co * 4 / 1e9
        f = timeit(lambda: ops.conv_c1_fwd(x, w, b, y, Co, B, H, W, Co, relu=True))
        g = timeit(lambda: ops.conv_c1_wgrad(x, dy, Co, dw, db, B, H, W, Co))
        d = timeit(lambda: ops.conv_c1_dgrad(dy, Co, w, dx, B, H, W, Co))
        print(f"Co {Co}: fwd {f * 1e3:7.1f} us ({gb / f:.2f} TB/s)  wgrad {g * 1e3:7.1f} us ({gb / g:.2f} TB/s)  dgrad {d * 1e3:7.1f} us ({gb / d:.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
