"""Developer tool: per-parameter (error vs fp64 reference, reference's own fp32 deviation) for the CRNN golden fixture."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"), os.path.join(ROOT, "tests")]
import helpers as H
from models.model_crnn import CRNN
from oracle import model_oracle as mo
from qea.loss import CTCLoss
mode = sys.argv[1] if len(sys.argv) > 1 else "bn_train"
fx = H.golden("crnn_b3.npz")
y, ysz = H.encode([str(s) for s in fx["labels"]])
net = CRNN(95, False); net.load_state_dict(mo.seeded_state(mo.crnn_state_shapes(), 2)); net = net.cuda()
net.register_backward_hook(net.backward_hook); net.train()
if mode == "bn_eval":
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm): m.eval()
x = torch.from_numpy(fx["x"]).cuda().requires_grad_()
lp = net(x)
print("lp max err", np.abs(lp.detach().cpu().numpy() - fx[f"{mode}|lp64"]).max())
CTCLoss()(lp, y, torch.tensor([31] * 3, dtype=torch.int), ysz).backward()
rep = {}
try:
    H.check_grad_vs64(fx, f"{mode}|g|", ((k, p.grad) for k, p in net.named_parameters()), report=rep)
except AssertionError as e:
    print("FAIL")
for k, (err, dev) in rep.items():
    print(f"{k:36s} err {err:.2e} dev {dev:.2e}")
