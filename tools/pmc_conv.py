"""Developer tool: run ONE big conv shape repeatedly (for rocprofv3 --pmc MFMA-busy / clock reads)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"))
from qea import ops
B, H, W, Cin, Cout = 512, 4, 32, 512, 512
x = torch.randn(B, H, W, Cin, device="cuda"); w = torch.randn(Cout, 3, 3, Cin, device="cuda"); y = torch.empty(B, H, W, Cout, device="cuda")
for _ in range(20):
    ops.conv_igemm(x, w, y, B=B, H=H, W=W, Cin=Cin, OH=H, OW=W, N=Cout, KH=3, KW=3, pad=(1, 1), ldx=Cin, ldy=Cout)
torch.cuda.synchronize()
