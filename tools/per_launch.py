"""Developer tool: per-launch time / TFLOP/s of the MFMA kernel classes over ONE Phase-B step at the bench batch."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "query-efficient-approx-to-improve-ocr_amd")]
import bench
from models.model_crnn import CRNN
from models.model_unet import UNet
from qea import ops
from qea.loss import CTCLoss
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda")
prep, crnn = UNet().to(dev), CRNN(95, False).to(dev)
crnn.register_backward_hook(crnn.backward_hook)
x, y, lens = bench.synth_batch(B, 1000, dev)
ins = torch.full((B,), 31, dtype=torch.int32)
shapes = []
orig = {n: getattr(ops, n) for n in ("conv_igemm", "conv_wgrad")}
def wrap(name):
    f = orig[name]
    def g(*a, **k):
        if name == "conv_igemm":
            shapes.append(("igemm", f"M={k['B']*k['OH']*k['OW']} N={k['N']} K={k['KH']*k['KW']*k['Cin']} {k['KH']}x{k['KW']} s{k.get('stride',(1,1))[0]} mode{k.get('out_mode',0)}"))
        else:
            shapes.append(("wgrad", f"M={k['B']*k['PH']*k['PW']} R={k['R']} C={k['Cc']} {k['KH']}x{k['KW']}"))
        return f(*a, **k)
    return g
def step():
    prep.train(); crnn.train()
    for m in crnn.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm): m.eval()
    prep.zero_grad(); crnn.zero_grad()
    img = prep(x); lp = crnn(img)
    (CTCLoss()(lp, y, ins, lens) + torch.nn.functional.mse_loss(img, torch.ones_like(img))).backward()
ops.set_overlap(False)   # a launch's event-to-event time must not include a co-running kernel
step(); step()
for n in orig: setattr(ops, n, wrap(n))
import qea.unet_engine, qea.crnn_engine
ops.prof_enable(0, True); ops.prof_enable(1, True); ops.prof_reset()
step(); torch.cuda.synchronize()
for klass, tag in ((0, "igemm"), (1, "wgrad")):
    ms, fl = ops.prof_read_launches(klass)
    names = [s for t, s in shapes if t == tag]
    print(f"== {tag}: {len(ms)} launches, {ms.sum():.2f} ms, {fl.sum()/ms.sum()/1e9:.1f} TF")
    order = sorted(range(len(ms)), key=lambda i: -ms[i])
    for i in order[:int(os.environ.get('TOP', '40'))]:
        print(f"  {ms[i]*1e3:8.1f} us {fl[i]/ms[i]/1e9:7.1f} TF  {names[i] if i < len(names) else '?'}")
