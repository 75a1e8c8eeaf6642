"""Timing of the 1x1 GEMM launches of the CRNN head (LSTM input projections, their input gradient, Linear) under forced tiles."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from qea import ops  # noqa: E402
from bench_convt import timeit  # noqa: E402


def main():
    TB = int(sys.argv[1]) if len(sys.argv) > 1 else 63488
    tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 20, 21, 22, 25]
    shapes = ((1024, 512), (512, 2048), (96, 512), (512, 96))
    if os.environ.get("GEMM_SHAPES"):                     # e.g. GEMM_SHAPES=1024x512,512x2048
        shapes = tuple(tuple(int(v) for v in sh.split("x")) for sh in os.environ["GEMM_SHAPES"].split(","))
    for (N, K) in shapes:
        x = torch.randn(TB, K, device="cuda")
        w = torch.randn(N, K, device="cuda")
        y = torch.empty(TB, N, device="cuda")
        for tile in tiles:
            try:
                t = timeit(lambda: ops.conv_igemm(x, w, y, B=1, H=1, W=TB, Cin=K, OH=1, OW=TB, N=N, KH=1, KW=1, ldx=K, ldy=N, tile=tile))
            except Exception:
                t = float("nan")
            print(f"M{TB} N{N:5d} K{K:5d} tile{tile:3d} {t * 1e3:8.1f} us {2.0 * TB * N * K / t / 1e9:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
