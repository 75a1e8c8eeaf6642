"""Runs ONE candidate of a cond_b*.npz pack through the HIP path (both phases) and returns its errors against the CPU
oracle evaluated in fp64 under the HIP forward's own ReLU / max-pool decisions (full tensors; tests/decisions.py), the
decisions that differ from the free-running oracle's, and the per-layer forward errors.  Used by
tests/test_conditioned_gpu.py and tools/forward_ladder.py.  TEST INFRASTRUCTURE: imports the oracle."""
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
    """the FREE-running fp64 oracle on candidate c (cached): results + Trace records of both phases"""
    key = (case, c)
    if key not in _oracle_cache:
        _oracle_cache[key] = H.oracle_cond_case(fx, c, record=True)
    return _oracle_cache[key]


def run_candidate(case, fx, c, report=None):
    """One candidate through the HIP path, both phases, against the fp64 oracle evaluated UNDER THE DECISIONS THE HIP FORWARD
    TOOK (tests/decisions.py).  -> dict: loss_B / loss_A (relative, against the reference's fp64 loss in the fixture),
    img / lp (max abs against the fixture), buf (max relative), zero (ZERO_GRAD tensors relative to the conv6 weight gradient),
    tensor {tag: plain ||g - g64|| / ||g64|| on the FULL tensor, g64 = decision-conditioned oracle gradient},
    free {tag: the same against the free-running oracle = the reference's fp64 gradient},
    flips {phase|site: (flips, decisions, worst margin in rounding units)}, ladder {phase|site: forward l2-relative error}."""
    import decisions as D
    from qea.loss import CTCLoss
    rB, rA = oracle(case, fx, c)
    ws = int(fx["ws"])
    x = torch.as_tensor(fx[c + "x"]).cuda()
    labels, labels_a = [str(s) for s in fx[c + "labels"]], [str(s) for s in fx[c + "labels_a"]]
    Bn, T = x.shape[0], x.shape[-1] // 4 - 1
    ins = torch.full((Bn,), T, dtype=torch.int)
    res = dict(tensor={}, free={}, flips={}, ladder={}, buf=0.0, zero=0.0)

    # ---- Phase B: UNet(train) -> CRNN(train, BN eval) -> CTC + MSE -> backward (train_nn_area.py:277-287)
    prep, crnn = hip_models(ws)
    prep.train(); crnn.train(); bn_eval(crnn)
    prep.zero_grad(); crnn.zero_grad()
    img = prep(x)
    lp = crnn(img)
    fu, tu = D.hip_unet_trace(img.grad_fn.saved)              # read BEFORE the backward (it overwrites the gate buffers in place)
    fc, tc = D.hip_crnn_trace(lp.grad_fn.saved)
    y, ysz = H.encode(labels)
    loss = CTCLoss()(lp, y, ins, ysz) + F.mse_loss(img, torch.ones_like(img))
    loss.backward()
    force = {**fu, **fc}
    cB = H.oracle_cond_phase_b(fx, c, force)
    # forward values against the reference's fp64 run stored in the fixture (the free-running oracle, pinned to it to 1e-10 by
    # tests/test_conditioned_cpu.py, stands in for packs that store inputs only)
    ref = lambda key, alt: torch.as_tensor(fx[c + key]).double() if (c + key) in fx else torch.as_tensor(alt).double()
    lB = float(ref("B|loss64", rB["loss"]))
    res["loss_B"] = abs(loss.item() - lB) / abs(lB)
    res["img"] = (img.detach().cpu().double() - ref("B|img64", rB["img"])).abs().max().item()
    res["lp"] = (lp.detach().cpu().double() - ref("B|lp64", rB["lp"])).abs().max().item()
    for tag, net, key in (("B|prep|", prep, "g_prep"), ("B|crnn|", crnn, "g_crnn")):
        for name, p in net.named_parameters():
            res["tensor"][tag + name] = H.full_rel_err(p.grad, cB[key][name])
            res["free"][tag + name] = H.full_rel_err(p.grad, rB[key][name])
    res["flips"].update({"B|" + k: v for k, v in D.flip_report(force, rB["rec"]).items()})
    res["ladder"].update({"B|" + k: v for k, v in D.ladder({**tu, **tc}, rB["rec"]).items()})
    for name, b in prep.named_buffers():
        if b.is_floating_point():
            rb = ref("B|buf|" + name, rB["buf_prep"][name])
            res["buf"] = max(res["buf"], (b.cpu().double() - rb).abs().max().item() / max(1.0, rb.abs().max().item()))
    # ---- Phase A: CRNN(train-mode BN) -> CTC -> backward, gradient wrt the input too (train_nn_area.py:262-271)
    _, crnn = hip_models(ws)
    crnn.train(); crnn.zero_grad()
    xa = x.clone().requires_grad_()
    lpa = crnn(xa)
    fa, ta = D.hip_crnn_trace(lpa.grad_fn.saved)
    ya, ysa = H.encode(labels_a)
    la = CTCLoss()(lpa, ya, ins, ysa)
    la.backward()
    cA = H.oracle_cond_phase_a(fx, c, fa)
    lA = float(ref("A|loss64", rA["loss"]))
    res["loss_A"] = abs(la.item() - lA) / abs(lA)
    res["lp"] = max(res["lp"], (lpa.detach().cpu().double() - ref("A|lp64", rA["lp"])).abs().max().item())
    res["tensor"]["A|dx"] = H.full_rel_err(xa.grad, cA["dx"])
    res["free"]["A|dx"] = H.full_rel_err(xa.grad, rA["dx"])
    scale = max(rA["g_crnn"]["convo.conv6.weight"].abs().max().item(), 1e-30)
    for name, p in crnn.named_parameters():
        if name in ZERO_GRAD:
            res["zero"] = max(res["zero"], p.grad.abs().max().item() / scale)
            continue
        res["tensor"]["A|crnn|" + name] = H.full_rel_err(p.grad, cA["g_crnn"][name])
        res["free"]["A|crnn|" + name] = H.full_rel_err(p.grad, rA["g_crnn"][name])
    res["flips"].update({"A|" + k: v for k, v in D.flip_report(fa, rA["rec"]).items()})
    res["ladder"].update({"A|" + k: v for k, v in D.ladder(ta, rA["rec"]).items()})
    for name, b in crnn.named_buffers():
        if b.is_floating_point():
            rb = ref("A|buf|" + name, rA["buf_crnn"][name])
            res["buf"] = max(res["buf"], (b.cpu().double() - rb).abs().max().item() / max(1.0, rb.abs().max().item()))
    torch.cuda.synchronize()
    v = sorted(res["tensor"].values())
    res["worst"], res["median"] = v[-1], v[len(v) // 2]
    res["worst_tag"] = max(res["tensor"].items(), key=lambda kv: kv[1])[0]
    res["worst_free"] = max(res["free"].values())
    res["n_flips"] = sum(f[0] for f in res["flips"].values())
    res["worst_flip_units"] = max((f[2] for f in res["flips"].values()), default=0.0)
    return res


def unselected_pack(ent):
    """An 'unselected' entry of tests/golden/ladder.json ({B, W, ws, seeds}) as a fixture-like dict: inputs drawn exactly as
    tests/golden/make_golden.py::make_conditioned draws them for image seed xs, WITHOUT the knife-edge-free selection."""
    B, W, T = ent["B"], ent["W"], ent["W"] // 4 - 1
    fx = {"ws": ent["ws"], "n_candidates": len(ent["seeds"])}
    for i, xs in enumerate(ent["seeds"]):
        c = f"c{i}|"
        fx[c + "x"] = torch.rand(B, 1, 32, W, generator=torch.Generator().manual_seed(xs))
        fx[c + "labels"] = H.synth_labels(B, xs, 1, max(1, T // 2))
        fx[c + "labels_a"] = H.synth_labels(B, xs + 100, 1, max(1, T // 2))
    return fx
