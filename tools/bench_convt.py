"""Timing of the transposed-conv GEMMs (forward scatter / input gradient) under forced tiles (developer tool, GPU box only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"))
from qea import ops  # noqa: E402


def timeit(fn, n=5):
    for _ in range(3):
        fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 21, 22, 23, 25]
    for (h, w, cin, c) in ((16, 64, 64, 32), (8, 32, 128, 64), (4, 16, 256, 128), (2, 8, 512, 256)):
        dev = "cuda"
        x = torch.randn(B * h * w, cin, device=dev)
        wup = torch.randn(cin, 2, 2, c, device=dev)
        cat = torch.empty(B * 2 * h * 2 * w, 2 * c, device=dev)
        bias = torch.randn(c, device=dev)
        dcat = torch.randn(B * 2 * h * 2 * w, 2 * c, device=dev)
        dd = torch.empty(B * h * w, cin, device=dev)
        wT = ops.transposed(wup.view(cin, 4 * c), cin, 4 * c)
        for tile in tiles:
            try:
                f = timeit(lambda: ops.conv_igemm(x, wT, cat, B=B, H=h, W=w, Cin=cin, OH=h, OW=w, N=4 * c, KH=1, KW=1, ldx=cin, ldy=2 * c, bias=bias,
                                                  out_mode=ops.OUT_CONVT, tile=tile))
            except Exception as e:
                f = float("nan")
            try:
                g = timeit(lambda: ops.conv_igemm(dcat, wup, dd, B=B, H=2 * h, W=2 * w, Cin=c, OH=h, OW=w, N=cin, KH=2, KW=2, stride=(2, 2), ldx=2 * c,
                                                  ldy=cin, tile=tile))
            except Exception as e:
                g = float("nan")
            fl = 2.0 * B * h * w * cin * 4 * c
            print(f"h{h:3d} w{w:3d} {cin:4d}->{c:4d} tile{tile:3d}  fwd {f * 1e3:8.1f} us {fl / f / 1e9:6.1f} TF   dgrad {g * 1e3:8.1f} us {fl / g / 1e9:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
