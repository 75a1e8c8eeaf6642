"""Per-step timing of the BiLSTM layer kernels (developer tool, GPU box only): python tools/bench_lstm.py [B ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"))
from qea import ops  # noqa: E402


def main():
    T = 31
    for B in [int(a) for a in sys.argv[1:]] or [512, 2048]:
        dev = "cuda"
        whf, whr = torch.randn(1024, 256, device=dev) / 16, torch.randn(1024, 256, device=dev) / 16
        g0 = torch.randn(T, B, 2048, device=dev)
        c, y, dy = torch.empty(T, B, 512, device=dev), torch.empty(T, B, 512, device=dev), torch.randn(T, B, 512, device=dev)
        dc = torch.empty(B, 512, device=dev)
        for mode in ("f32", "split_bf16"):
            prev = ops.set_mfma_mode(mode)
            pf, pb, split = ops.lstm_packs(whf, whr)
            res = []
            for what in ("fwd", "bwd"):
                best = 1e9
                for _rep in range(3):
                    gates = g0.clone()
                    if what == "bwd":
                        ops.lstm_layer_fwd_any(gates, c, y, pf, split, T, B)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    if what == "fwd":
                        ops.lstm_layer_fwd_any(gates, c, y, pf, split, T, B)
                    else:
                        ops.lstm_layer_bwd_any(gates, c, dy, pb, split, dc, T, B)
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) * 1e3 / T)
                res.append(best)
            ops.set_mfma_mode(prev)
            print(f"B {B:5d} {mode:10s} fwd {res[0]:7.1f} us/step  bwd {res[1]:7.1f} us/step", flush=True)


if __name__ == "__main__":
    main()
