"""Developer tool: run ONE 3x3 conv shape repeatedly (target for rocprofv3 --pmc passes).  argv: H W Cin Cout [B]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"))
from qea import ops
H, W, Cin, Cout = (int(v) for v in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 512
x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda"); y = torch.empty(B, H, W, Cout, device="cuda")
for _ in range(12):
    ops.conv_igemm(x, w, y, B=B, H=H, W=W, Cin=Cin, OH=H, OW=W, N=Cout, KH=3, KW=3, pad=(1, 1), ldx=Cin, ldy=Cout)
torch.cuda.synchronize()
