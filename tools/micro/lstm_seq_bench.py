"""One BiLSTM layer pass: the one-launch kernels (csrc/lstm_seq.hip) against the per-step split-bf16 kernels, by batch size."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"))
from qea import ops  # noqa: E402


def run(T, B, mode, reps=20):
    dev = "cuda"
    g = torch.Generator().manual_seed(1)
    wf = (torch.randn(1024, 256, generator=g) / 16).to(dev)
    wr = (torch.randn(1024, 256, generator=g) / 16).to(dev)
    ops.LSTM_SEQ["on"] = mode == "seq"
    pf, pb, m = ops.lstm_packs(wf, wr)
    gx = (torch.randn(T, B, 2048, generator=g) * 0.5).to(dev)
    dy = torch.randn(T, B, 512, generator=g).to(dev)
    c, y, dc = torch.empty(T, B, 512, device=dev), torch.empty(T, B, 512, device=dev), torch.empty(B, 512, device=dev)
    out = {}
    for what in ("fwd", "bwd"):
        ts = []
        for i in range(reps + 3):
            gates = gx.clone()
            if what == "bwd":
                ops.lstm_layer_fwd_any(gates, c, y, pf, m, T, B)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if what == "fwd":
                ops.lstm_layer_fwd_any(gates, c, y, pf, m, T, B)
            else:
                ops.lstm_layer_bwd_any(gates, c, dy, pb, m, dc, T, B)
            e1.record()
            torch.cuda.synchronize()
            if i >= 3:
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        out[what] = ts[len(ts) // 2]
        out[what + "_res"] = (y.clone() if what == "fwd" else gates.clone())
    return out


if __name__ == "__main__":
    T = 31
    for B in (8, 32, 128, 512, 1024, 2048):
        a, b = run(T, B, "seq"), run(T, B, "bf3")
        ef = ((a["fwd_res"] - b["fwd_res"]).norm() / b["fwd_res"].norm()).item()
        eb = ((a["bwd_res"] - b["bwd_res"]).norm() / b["bwd_res"].norm()).item()
        print(f"T={T} B={B:5d}  fwd seq {a['fwd']:8.1f} us  steps {b['fwd']:8.1f} us | bwd seq {a['bwd']:8.1f} us  steps {b['bwd']:8.1f} us | "
              f"rel diff fwd {ef:.2e} bwd {eb:.2e} nan {bool(torch.isnan(a['fwd_res']).any() or torch.isnan(a['bwd_res']).any())}", flush=True)
