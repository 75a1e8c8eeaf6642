"""Runs ONE candidate of a cond_b*.npz pack through the HIP path (both phases) and returns its errors against the CPU
oracle evaluated in fp64 (full tensors) and against the reference's fp64 samples stored in the fixture.  Shared by
tests/test_conditioned_gpu.py and tools/qualify_fixtures.py.  TEST INFRASTRUCTURE: imports the oracle."""
import numpy as np
import torch
import torch.nn.functional as F

import helpers as H

ZERO_GRAD = ("convo.conv5.bias", "convo.conv6.bias")     # exactly zero in Phase A (a bias in front of a batch-statistics BN)
_oracle_cache = {}


def bn_eval(m):
    for x in m.modules():
        if isinstance(x, torch.nn.modules.batchnorm._BatchNorm):
            x.eval()


def hip_models(ws):
    from models.model_crnn import CRNN
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    prep = UNet()
    prep.load_state_dict(mo.default_init_state(mo.unet_state_shapes(), ws))
    crnn = CRNN(95, False)
    crnn.load_state_dict(mo.default_init_state(mo.crnn_state_shapes(), ws + 1))
    prep, crnn = prep.cuda(), crnn.cuda()
    crnn.register_backward_hook(crnn.backward_hook)
    return prep, crnn


def oracle(case, fx, c):
    key = (case, c)
    if key not in _oracle_cache:
        _oracle_cache[key] = H.oracle_cond_case(fx, c)
    return _oracle_cache[key]


def run_candidate(case, fx, c):
    """-> dict: loss_B / loss_A (relative), img / lp (max abs), buf (max relative), zero (ZERO_GRAD tensors, relative to the
    conv6 weight gradient), tensor {tag: plain ||g - g64|| / ||g64|| on the FULL tensor}, direct {tag: the same against the
    fixture's fp64 samples / norms of the REFERENCE run}."""
    from qea.loss import CTCLoss
    rB, rA = oracle(case, fx, c)
    ws = int(fx["ws"])
    x = torch.from_numpy(fx[c + "x"]).cuda()
    labels, labels_a = [str(s) for s in fx[c + "labels"]], [str(s) for s in fx[c + "labels_a"]]
    Bn, T = x.shape[0], x.shape[-1] // 4 - 1
    ins = torch.full((Bn,), T, dtype=torch.int)
    res = dict(tensor={}, direct={}, buf=0.0, zero=0.0)

    def record(tag, got, ref64, fx_prefix=None):
        res["tensor"][tag] = H.full_rel_err(got, ref64)
        if fx_prefix is not None:
            s64 = torch.from_numpy(fx[fx_prefix + "|s64"]).double()
            g = got.detach().double().flatten().cpu()[H.sample_index_small(got.numel())]
            l264 = float(fx[fx_prefix + "|l264"])
            res["direct"][tag] = max((g - s64).norm().item() / max(s64.norm().item(), l264 * (s64.numel() / got.numel()) ** 0.5),
                                     abs(got.double().norm().item() - l264) / l264)

    # ---- Phase B: UNet(train) -> CRNN(train, BN eval) -> CTC + MSE -> backward (train_nn_area.py:277-287)
    prep, crnn = hip_models(ws)
    prep.train(); crnn.train(); bn_eval(crnn)
    prep.zero_grad(); crnn.zero_grad()
    img = prep(x)
    lp = crnn(img)
    y, ysz = H.encode(labels)
    loss = CTCLoss()(lp, y, ins, ysz) + F.mse_loss(img, torch.ones_like(img))
    loss.backward()
    res["loss_B"] = abs(loss.item() - float(fx[c + "B|loss64"])) / abs(float(fx[c + "B|loss64"]))
    res["img"] = (img.detach().cpu().double() - torch.from_numpy(fx[c + "B|img64"])).abs().max().item()
    res["lp"] = (lp.detach().cpu().double() - torch.from_numpy(fx[c + "B|lp64"])).abs().max().item()
    for name, p in prep.named_parameters():
        record("B|prep|" + name, p.grad, rB["g_prep"][name], c + "B|g|prep|" + name)
    for name, p in crnn.named_parameters():
        record("B|crnn|" + name, p.grad, rB["g_crnn"][name], c + "B|g|crnn|" + name)
    for name, b in prep.named_buffers():
        if b.is_floating_point():
            ref = torch.from_numpy(fx[c + "B|buf|" + name])
            res["buf"] = max(res["buf"], (b.cpu().double() - ref).abs().max().item() / max(1.0, ref.abs().max().item()))
    # ---- Phase A: CRNN(train-mode BN) -> CTC -> backward, gradient wrt the input too (train_nn_area.py:262-271)
    _, crnn = hip_models(ws)
    crnn.train(); crnn.zero_grad()
    xa = x.clone().requires_grad_()
    lpa = crnn(xa)
    ya, ysa = H.encode(labels_a)
    la = CTCLoss()(lpa, ya, ins, ysa)
    la.backward()
    res["loss_A"] = abs(la.item() - float(fx[c + "A|loss64"])) / abs(float(fx[c + "A|loss64"]))
    res["lp"] = max(res["lp"], (lpa.detach().cpu().double() - torch.from_numpy(fx[c + "A|lp64"])).abs().max().item())
    record("A|dx", xa.grad, torch.from_numpy(fx[c + "A|dx64"]))
    scale = max(rA["g_crnn"]["convo.conv6.weight"].abs().max().item(), 1e-30)
    for name, p in crnn.named_parameters():
        if name in ZERO_GRAD:
            res["zero"] = max(res["zero"], p.grad.abs().max().item() / scale)
            continue
        record("A|crnn|" + name, p.grad, rA["g_crnn"][name], c + "A|g|" + name)
    for name, b in crnn.named_buffers():
        if b.is_floating_point():
            ref = torch.from_numpy(fx[c + "A|buf|" + name])
            res["buf"] = max(res["buf"], (b.cpu().double() - ref).abs().max().item() / max(1.0, ref.abs().max().item()))
    torch.cuda.synchronize()
    v = sorted(res["tensor"].values())
    res["worst"], res["median"] = v[-1], v[len(v) // 2]
    res["worst_tag"] = max(res["tensor"].items(), key=lambda kv: kv[1])[0]
    res["worst_direct"] = max(res["direct"].values())
    return res
