"""Developer tool: host enqueue time vs GPU time of one Phase-B step."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd")]
import bench
from models.model_crnn import CRNN
from models.model_unet import UNet
from qea.loss import CTCLoss
from qea.optim import FusedAdam
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda")
prep, crnn = UNet().to(dev), CRNN(95, False).to(dev)
crnn.register_backward_hook(crnn.backward_hook)
opt = FusedAdam(prep.parameters(), lr=5e-5)
x, y, lens = bench.synth_batch(B, 1000, dev)
ins = torch.full((B,), 31, dtype=torch.int32)
ctc = CTCLoss()
def step():
    prep.train(); crnn.train()
    for m in crnn.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm): m.eval()
    prep.zero_grad(); crnn.zero_grad()
    img = prep(x); lp = crnn(img)
    (ctc(lp, y, ins, lens) + torch.nn.functional.mse_loss(img, torch.ones_like(img))).backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B}: host enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
