"""Developer tool: locate the UNet backward precision gap on the B=2 golden fixture."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd"), os.path.join(ROOT, "tests")]
import helpers as H
from oracle import model_oracle as mo
from models.model_unet import UNet

fx = H.golden("unet_b2.npz")
x = torch.from_numpy(fx["x"]); r = torch.from_numpy(fx["r"])
su = mo.seeded_state(mo.unet_state_shapes(), 1)

def oracle(dtype):
    st = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in su.items()}
    P, Bf = mo.split_state(st)
    inter = {}
    # dec1 internals: re-run the forward by hand up to dec1 with retained tensors
    xi = x.to(dtype)
    e1 = mo._unet_block(xi, P, Bf, "encoder1", "enc1", True)
    e2 = mo._unet_block(F.max_pool2d(e1, 2, 2), P, Bf, "encoder2", "enc2", True)
    e3 = mo._unet_block(F.max_pool2d(e2, 2, 2), P, Bf, "encoder3", "enc3", True)
    e4 = mo._unet_block(F.max_pool2d(e3, 2, 2), P, Bf, "encoder4", "enc4", True)
    d = mo._unet_block(F.max_pool2d(e4, 2, 2), P, Bf, "bottleneck", "bottleneck", True)
    for lvl, skip in ((4, e4), (3, e3), (2, e2)):
        d = F.conv_transpose2d(d, P[f"upconv{lvl}.weight"], P[f"upconv{lvl}.bias"], stride=2)
        d = mo._unet_block(torch.cat((d, skip), 1), P, Bf, f"decoder{lvl}", f"dec{lvl}", True)
    d = F.conv_transpose2d(d, P["upconv1.weight"], P["upconv1.bias"], stride=2)
    cat = torch.cat((d, e1), 1)
    y1 = F.conv2d(cat, P["decoder1.dec1conv1.weight"], None, padding=1); y1.retain_grad()
    n1 = mo._bn(y1, P, Bf, "decoder1.dec1norm1", True); n1.retain_grad()
    a1 = F.relu(n1); a1.retain_grad()
    y2 = F.conv2d(a1, P["decoder1.dec1conv2.weight"], None, padding=1); y2.retain_grad()
    n2 = mo._bn(y2, P, Bf, "decoder1.dec1norm2", True); n2.retain_grad()
    a2 = F.relu(n2); a2.retain_grad()
    out = torch.sigmoid(F.conv2d(a2, P["conv.weight"], P["conv.bias"]))
    loss = F.mse_loss(out, torch.ones_like(out)) + (out * r.to(dtype)).sum() / out.numel()
    loss.backward()
    return P, dict(y1=y1, n1=n1, a1=a1, y2=y2, n2=n2, a2=a2, out=out)

P64, I64 = oracle(torch.float64)
P32, I32 = oracle(torch.float32)
net = UNet(); net.load_state_dict(su); net = net.cuda().train()
eng = net._engine()
xg = x.cuda()
y = net(xg)
loss = F.mse_loss(y, torch.ones_like(y)) + (y * r.cuda()).sum() / y.numel()
# grab the saved ctx through the autograd node
ctx = y.grad_fn.saved
blk = ctx["blocks"]["decoder1"]
def nchw(t, C): return t.reshape(2, 32, 128, C).permute(0, 3, 1, 2)
def rel(a, b): return ((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()
print("fwd y1  hip %.2e cpu32 %.2e" % (rel(nchw(blk["y1"], 32), I64["y1"]), rel(I32["y1"], I64["y1"])))
print("fwd a1  hip %.2e cpu32 %.2e" % (rel(nchw(blk["a1"], 32), I64["a1"]), rel(I32["a1"], I64["a1"])))
print("fwd y2  hip %.2e cpu32 %.2e" % (rel(nchw(blk["y2"], 32), I64["y2"]), rel(I32["y2"], I64["y2"])))
print("fwd a2  hip %.2e cpu32 %.2e" % (rel(nchw(blk["out"], 32), I64["a2"]), rel(I32["a2"], I64["a2"])))
m_h = (nchw(blk["a1"], 32) > 0).cpu(); m64 = I64["a1"] > 0; m32 = I32["a1"] > 0
print("mask a1 flips: hip", (m_h != m64).sum().item(), "cpu32", (m32 != m64).sum().item(), "of", m64.numel())
m_h = (nchw(blk["out"], 32) > 0).cpu(); m64b = I64["a2"] > 0
print("mask a2 flips: hip", (m_h != m64b).sum().item(), "cpu32", ((I32["a2"] > 0) != m64b).sum().item())
loss.backward()
for n in ["decoder1.dec1norm2.weight", "decoder1.dec1norm2.bias", "decoder1.dec1conv2.weight", "decoder1.dec1norm1.weight",
          "decoder1.dec1norm1.bias", "decoder1.dec1conv1.weight", "upconv1.bias", "upconv1.weight", "conv.weight", "conv.bias"]:
    p = dict(net.named_parameters())[n]
    print("%-28s hip %.2e cpu32 %.2e   |sum|/sum|.| of ref64 %.1e" % (n, rel(p.grad, P64[n].grad), rel(P32[n].grad, P64[n].grad), 0))
# per-channel view of dec1norm1.bias
g_h = dict(net.named_parameters())["decoder1.dec1norm1.bias"].grad.double().cpu(); g64 = P64["decoder1.dec1norm1.bias"].grad; g32 = P32["decoder1.dec1norm1.bias"].grad.double()
print("dbeta1 per channel: hip-ref64", (g_h - g64).abs().topk(5), "ref", g64[(g_h - g64).abs().topk(5).indices])
print("cpu32-ref64", (g32 - g64).abs().max().item())
# the exact dz1 from the oracle, to measure cancellation
dz1 = I64["a1"].grad * (I64["a1"] > 0)
print("cancellation sum|dz1| / |sum dz1| per channel (top)", (dz1.abs().sum((0, 2, 3)) / dz1.sum((0, 2, 3)).abs()).topk(5).values)
