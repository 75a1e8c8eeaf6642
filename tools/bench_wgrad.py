"""Per-shape timing of qea_conv_wgrad at the bench batch (developer tool, GPU box only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"))
from qea import ops  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from bench_conv import SHAPES  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    tot_t = tot_f = 0.0
    for (H, W, Cin, Cout), tile in [(sh, t) for sh in SHAPES for t in tiles]:
        if (tile in (1, 9, 12, 13) and min(Cin, Cout) < 128) or (tile in (7, 10) and (Cout < 128 or Cin != 64)) or (tile in (8, 11) and (Cin < 128 or Cout != 64)):
            continue
        x = torch.randn(B, H, W, Cin, device="cuda")
        dy = torch.randn(B, H, W, Cout, device="cuda")
        dw = torch.empty(Cout, 3, 3, Cin, device="cuda")
        kw = dict(B=B, PH=H, PW=W, QH=H, QW=W, R=Cout, Cc=Cin, KH=3, KW=3, pad=(1, 1), ldp=Cout, ldq=Cin, tile=tile)
        if ops.SPLIT_F16["on"]:                             # the abs-max scalars outside the timed loop (producers carry them in the product)
            kw.update(p_amax=ops.absmax(dy, Cout, B * H * W, Cout), q_amax=ops.absmax(x, Cin, B * H * W, Cin))
        try:
            for _ in range(2):
                ops.conv_wgrad(dy, x, dw, **kw)
        except Exception:                                   # a forced tile that does not take this shape
            print(f"H{H:3d} W{W:3d} Cin{Cin:4d} Cout{Cout:4d} tile{tile}   n/a", flush=True)
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            ops.conv_wgrad(dy, x, dw, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * B * H * W * Cout * 9 * Cin
        print(f"H{H:3d} W{W:3d} Cin{Cin:4d} Cout{Cout:4d} tile{tile} {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)
        tot_t += ms
        tot_f += fl
    print(f"TOTAL {tot_t:.2f} ms  {tot_f / tot_t / 1e9:.1f} TF")


if __name__ == "__main__":
    main()
