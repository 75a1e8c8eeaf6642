"""Per-shape timing of qea_conv_igemm at the bench batch (developer tool, GPU box only)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"))
from qea import _lib  # noqa: E402

SHAPES = [
    (32, 128, 32, 32), (32, 128, 64, 32), (16, 64, 32, 64), (16, 64, 64, 64), (16, 64, 128, 64),
    (8, 32, 64, 128), (8, 32, 128, 128), (8, 32, 256, 128), (4, 16, 128, 256), (4, 16, 256, 256),
    (4, 16, 512, 256), (2, 8, 256, 512), (2, 8, 512, 512),
    (16, 64, 64, 128), (8, 32, 128, 256), (8, 32, 256, 256), (4, 32, 256, 512), (4, 32, 512, 512),
    (4, 16, 256, 512),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
    L = _lib.lib()
    shapes = [sh for sh in SHAPES if sh[2] >= 64 and sh[3] >= 64] if len(sys.argv) > 3 and sys.argv[3] == "wide" else SHAPES
    if len(sys.argv) > 3 and sys.argv[3] == "narrow":
        shapes = [sh for sh in SHAPES if sh[2] <= 64 and sh[3] <= 64]
    if len(sys.argv) > 3 and sys.argv[3] == "mid":
        shapes = [(16, 64, 128, 64), (16, 64, 64, 128), (8, 32, 64, 128), (8, 32, 128, 128), (8, 32, 256, 128), (16, 64, 128, 128), (4, 32, 256, 128)]
    s = torch.cuda.current_stream().cuda_stream
    tot_t = tot_f = 0.0
    for (H, W, Cin, Cout) in shapes:
        x = torch.randn(B, H, W, Cin, device="cuda")
        w = torch.randn(Cout, 3, 3, Cin, device="cuda")
        y = torch.empty(B, H, W, Cout, device="cuda")
        xp = torch.empty(L.qea_split_planes_bytes(B * H * W, Cin), dtype=torch.uint8, device="cuda")
        wp = torch.empty(L.qea_split_planes_bytes(Cout, 9 * Cin), dtype=torch.uint8, device="cuda")
        _lib.check(L.qea_split_planes(x.data_ptr(), Cin, B * H * W, Cin, xp.data_ptr(), s))
        _lib.check(L.qea_split_planes(w.data_ptr(), 9 * Cin, Cout, 9 * Cin, wp.data_ptr(), s))
        narrow = (Cin == 32 or (Cin % 64 == 0 and Cin <= 512)) and (Cout in (32, 64) or Cout % 128 == 0) and (
            W % 32 == 0 or ((H, W) in ((4, 16), (2, 8)) and Cin % 64 == 0 and Cout % 128 == 0))
        fp = xmax = None
        f16 = os.environ.get("QEA_SPLIT", "") == "f16"           # two-way fp16 split of the halo kernel (ABI v6)
        if narrow and f16:
            from qea import ops
            fp = torch.empty(L.qea_pack_frag_planes_f16_bytes(Cout, Cin), dtype=torch.uint8, device="cuda")
            wmax = ops.absmax(w, 9 * Cin, Cout, 9 * Cin)
            _lib.check(L.qea_pack_frag_planes_f16(w.data_ptr(), Cout, Cin, wmax.data_ptr(), fp.data_ptr(), s))
            xmax = ops.absmax(x, Cin, B * H * W, Cin)
        elif narrow:
            fp = torch.empty(L.qea_pack_frag_planes_bytes(Cout, Cin), dtype=torch.uint8, device="cuda")
            _lib.check(L.qea_pack_frag_planes(w.data_ptr(), Cout, Cin, fp.data_ptr(), s))
        for tile in tiles:
            pre = 100 <= tile < 200                   # tile id + 100: the same tile on pre-split operands; + 200: pre-split FILTER only
            wonly = tile >= 200
            d = _lib.ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), scale=None, bias=None, mask=None,
                              B=B, H=H, W=W, Cin=Cin, OH=H, OW=W, N=Cout, KH=3, KW=3, pad_h=1, pad_w=1,
                              stride_h=1, stride_w=1, ldx=Cin, ldy=Cout, ldmask=0, relu=0, accumulate=0,
                              out_mode=0, tile=tile % 100, x_planes=xp.data_ptr() if pre else None, w_planes=wp.data_ptr() if (pre or wonly) else None,
                              stats=None, w_frag_planes=fp.data_ptr() if (fp is not None and tile % 100 == 24) else None,
                              x_absmax=xmax.data_ptr() if (xmax is not None and tile % 100 == 24) else None)
            if L.qea_conv_igemm(C.byref(d), s) != 0:     # a forced tile that does not take this shape
                print(f"H{H:3d} W{W:3d} Cin{Cin:4d} Cout{Cout:4d} tile{tile}   n/a", flush=True)
                continue
            for _ in range(4):
                _lib.check(L.qea_conv_igemm(C.byref(d), s))
            ms = 1e9
            for _rep in range(3):                      # best of three batches (the first launches after an allocation run slow)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 5
                e0.record()
                for _ in range(n):
                    L.qea_conv_igemm(C.byref(d), s)
                e1.record()
                torch.cuda.synchronize()
                ms = min(ms, e0.elapsed_time(e1) / n)
            fl = 2.0 * B * H * W * Cout * 9 * Cin
            print(f"H{H:3d} W{W:3d} Cin{Cin:4d} Cout{Cout:4d} tile{tile} {ms:8.3f} ms {fl / ms / 1e9:7.1f} TF", flush=True)
            if tile == tiles[0]:
                tot_t += ms
                tot_f += fl
    print(f"TOTAL {tot_t:.2f} ms  {tot_f / tot_t / 1e9:.1f} TF (first tile choice)")


if __name__ == "__main__":
    main()
