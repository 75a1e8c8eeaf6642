"""UNet fwd+bwd on N document-sized images [N,1,400,512] (the patch flow's unit is N = 1, train_nn_patch.py:237-242; N > 1 =
--docs_per_step with per-document BatchNorm groups): ms per pass and per document, the conv / wgrad classes' TFLOP/s and the tile
each 3x3 layer gets (developer tool, GPU box only).   python tools/bench_doc.py [H W [N]]"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"))
from models.model_unet import UNet  # noqa: E402
from qea import _lib, ops  # noqa: E402


def main():
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (400, 512)
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    torch.manual_seed(0)
    net = UNet().cuda().train()
    x = torch.rand(N, 1, H, W, device="cuda")
    ones = torch.ones_like(x)

    def step():
        net.zero_grad()
        y = net(x, bn_groups=N)
        torch.nn.functional.mse_loss(y, ones).backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    ops.set_overlap(False)
    for k in (ops.PROF_CONV_IGEMM, ops.PROF_CONV_WGRAD):
        ops.prof_enable(k, True)
    ops.prof_reset()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    out = {"image": [N, 1, H, W], "bn_groups": N, "ms_per_fwd_bwd": round(ms, 3), "ms_per_document": round(ms / N, 3)}
    for name, k in (("conv_igemm", ops.PROF_CONV_IGEMM), ("conv_wgrad", ops.PROF_CONV_WGRAD)):
        q = ops.prof_read(k)
        out[name] = {"tflops": round(q["flops"] / (q["ms"] * 1e-3) / 1e12, 1), "ms_per_pass": round(q["ms"] / n, 3), "launches_per_pass": q["launches"] / n,
                     "split_bf16_flop_fraction": round(q["flops_split_bf16"] / max(q["flops"], 1), 3)}
    L = _lib.lib()
    tiles = {}
    for lvl, (cin, cout) in enumerate(((32, 32), (64, 64), (128, 128), (256, 256), (512, 512))):
        h, w = H >> lvl, W >> lvl
        d = _lib.ConvDesc(x=None, w=None, y=None, scale=None, bias=None, mask=None, B=1, H=h, W=w, Cin=cin, OH=h, OW=w, N=cout, KH=3, KW=3,
                          pad_h=1, pad_w=1, stride_h=1, stride_w=1, ldx=cin, ldy=cout, ldmask=0, relu=0, accumulate=0, out_mode=0, tile=0,
                          x_planes=None, w_planes=None, stats=None, w_frag_planes=None)
        tiles[f"{h}x{w}x{cin}->{cout}"] = {"split_bf16": bool(L.qea_conv_igemm_uses_split_bf16(C.byref(d))),
                                           "lds_halo_kernel": bool(L.qea_conv_igemm_wants_frag_planes(C.byref(d)))}
    out["layers"] = tiles
    import json
    print(json.dumps(out))


if __name__ == "__main__":
    main()
