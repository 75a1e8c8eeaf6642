"""Is the error of a split-bf16 conv launch RANDOM (averages out in sums over pixels) or SYSTEMATIC (a scale / sign bias that
per-channel sums keep)?  One 3x3 conv per shape and input distribution, both MFMA modes, against fp64 on the CPU:
  l2      plain l2-relative error
  scale   <y - ref, ref> / <ref, ref>                 (a pure scale error e gives e)
  colsum  || sum_p (y - ref) || / || sum_p |ref| ||   (what a bias gradient sees, normalised by the abs-sum: no cancellation blow-up)
GPU box: python tools/split_bias_probe.py"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"))
from qea import ops  # noqa: E402

torch.set_num_threads(16)
SHAPES = [(4, 8, 32, 256, 128), (4, 16, 64, 128, 64), (4, 8, 32, 256, 256), (4, 4, 32, 512, 512), (4, 16, 64, 64, 128)]


def draw(kind, shape, g):
    x = torch.randn(shape, generator=g)
    if kind == "sparse":                       # a gradient behind ReLU / max-pool masks: 75 % exact zeros, heavy tail
        x = x * (torch.rand(shape, generator=g) < 0.25) * torch.exp(torch.randn(shape, generator=g))
    if kind == "positive":                     # post-ReLU activations
        x = x.clamp_min(0)
    if kind == "small":                        # gradient magnitudes of a mean-reduced loss
        x = x * 1e-5
    if kind == "wide":                         # 12 decades of dynamic range inside one tensor
        x = x * torch.exp(4 * torch.randn(shape, generator=g))
    return x


for (B, H, W, Ci, Co) in SHAPES:
    for kind in ("gauss", "sparse", "positive", "small", "wide"):
        g = torch.Generator().manual_seed(7)
        x = draw(kind, (B, H, W, Ci), g)
        w = torch.randn(Co, 3, 3, Ci, generator=g) / (9 * Ci) ** 0.5
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
        line = f"B{B} {H}x{W} {Ci}->{Co} {kind:9s}"
        for mode in ("split_bf16", "split_f16", "f32"):
            prev = ops.set_mfma_mode(mode)
            y = torch.empty(B * H * W, Co, device="cuda")
            ops.conv_igemm(x.cuda(), w.cuda(), y, B=B, H=H, W=W, Cin=Ci, OH=H, OW=W, N=Co, KH=3, KW=3, pad=(1, 1), ldx=Ci, ldy=Co)
            ops.set_mfma_mode(prev)
            e = y.cpu().double() - ref
            l2 = e.norm().item() / ref.norm().item()
            scale = (e * ref).sum().item() / (ref * ref).sum().item()
            colsum = e.sum(0).norm().item() / ref.abs().sum(0).norm().item()
            line += f" | {mode}: l2 {l2:.2e} scale {scale:+.2e} colsum {colsum:.2e}"
        print(line, flush=True)
