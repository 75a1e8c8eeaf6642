"""Developer tool: run ONE 3x3 conv shape repeatedly on a list of tiles (target for rocprofv3 --pmc passes).
argv: H W Cin Cout B tile[,tile...]   (tile + 100 = the same tile on pre-split operands)"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"))
from qea import _lib
H, W, Cin, Cout, B = (int(v) for v in sys.argv[1:6])
tiles = [int(t) for t in sys.argv[6].split(",")] if len(sys.argv) > 6 else [0]
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda"); y = torch.empty(B, H, W, Cout, device="cuda")
xp = torch.empty(L.qea_split_planes_bytes(B * H * W, Cin), dtype=torch.uint8, device="cuda")
wp = torch.empty(L.qea_split_planes_bytes(Cout, 9 * Cin), dtype=torch.uint8, device="cuda")
_lib.check(L.qea_split_planes(x.data_ptr(), Cin, B * H * W, Cin, xp.data_ptr(), s))
_lib.check(L.qea_split_planes(w.data_ptr(), 9 * Cin, Cout, 9 * Cin, wp.data_ptr(), s))
for tile in tiles:
    pre = tile >= 100
    d = _lib.ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), scale=None, bias=None, mask=None, B=B, H=H, W=W, Cin=Cin, OH=H, OW=W,
                      N=Cout, KH=3, KW=3, pad_h=1, pad_w=1, stride_h=1, stride_w=1, ldx=Cin, ldy=Cout, ldmask=0, relu=0, accumulate=0,
                      out_mode=0, tile=tile % 100, x_planes=xp.data_ptr() if pre else None, w_planes=wp.data_ptr() if pre else None)
    for _ in range(12):
        _lib.check(L.qea_conv_igemm(C.byref(d), s))
    torch.cuda.synchronize()
