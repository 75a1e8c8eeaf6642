"""Whole-network parity of the HIP UNet / CRNN / CTC / Adam path against (a) the golden fixtures
generated from the reference and (b) the CPU oracle on larger seeded batches."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import helpers as H

pytestmark = pytest.mark.gpu


def _load(module, shapes_fn, seed):
    from oracle import model_oracle as mo
    module.load_state_dict(mo.seeded_state(shapes_fn(), seed))
    return module.cuda()


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def test_unet_golden():
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    fx = H.golden("unet_b2.npz")
    x = torch.from_numpy(fx["x"]).cuda()
    net = _load(UNet(), mo.unet_state_shapes, 1).eval()
    with torch.no_grad():
        y = net(x)
    assert np.abs(y.cpu().numpy() - fx["y_eval"]).max() < 2e-5
    net = _load(UNet(), mo.unet_state_shapes, 1).train()
    y = net(x)
    r = torch.from_numpy(fx["r"]).cuda()
    loss = F.mse_loss(y, torch.ones_like(y)) + (y * r).sum() / y.numel()
    loss.backward()
    assert np.abs(y.detach().cpu().numpy() - fx["y_train64"]).max() < 2e-5
    assert abs(loss.item() - float(fx["loss64"])) < 1e-5
    H.check_grad_vs64(fx, "g|", ((k, p.grad) for k, p in net.named_parameters()))
    H.check_tensor_summary(fx, "buf|", ((k, v) for k, v in net.state_dict().items() if mo.is_buffer(k)), rtol=1e-5)


@pytest.mark.parametrize("mode", ["bn_train", "bn_eval"])
def test_crnn_golden(mode):
    from models.model_crnn import CRNN
    from oracle import model_oracle as mo
    from qea.loss import CTCLoss
    fx = H.golden("crnn_b3.npz")
    labels = [str(s) for s in fx["labels"]]
    y, ysz = H.encode(labels)
    net = _load(CRNN(95, False), mo.crnn_state_shapes, 2)
    net.register_backward_hook(net.backward_hook)
    net.train()
    if mode == "bn_eval":
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_()
    lp = net(x)
    assert lp.shape == (31, 3, 95)
    assert np.abs(lp.detach().cpu().numpy() - fx[f"{mode}|lp64"]).max() < 2e-4
    insz = torch.tensor([31] * 3, dtype=torch.int)
    per = CTCLoss(reduction="none")(lp.detach(), y, insz, ysz).cpu().numpy()
    ref = fx[f"{mode}|nll"]
    assert (np.isinf(per) == np.isinf(ref)).all() and np.allclose(per[np.isfinite(ref)], ref[np.isfinite(ref)], rtol=1e-4)
    loss = CTCLoss()(lp, y, insz, ysz)
    assert np.isinf(loss.item())
    loss.backward()
    dx = x.grad.cpu().numpy()
    refdx, dx32 = fx[f"{mode}|dx64"], fx[f"{mode}|dx"]
    assert np.isfinite(dx).all()
    if mode == "bn_eval":
        assert np.abs(dx[1]).max() == 0.0
    dev = max(np.linalg.norm(dx32 - refdx) / np.linalg.norm(refdx), float(fx[f"{mode}|dxcond"]))
    assert np.linalg.norm(dx - refdx) <= max(1e-4, 3 * dev) * np.linalg.norm(refdx)
    H.check_grad_vs64(fx, f"{mode}|g|", ((k, p.grad) for k, p in net.named_parameters()))
    H.check_tensor_summary(fx, f"{mode}|buf|", ((k, v) for k, v in net.state_dict().items() if mo.is_buffer(k)), rtol=1e-4)


def _oracle_phase_b(x, labels, su, sc, dtype):
    from oracle import model_oracle as mo
    from oracle import step_oracle as so
    cast = lambda st: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in st.items()}
    tr = so.OracleTrainer(cast(su), cast(sc), H.C2I)
    tr.zero()
    img = mo.unet_forward(tr.Pu, tr.Bu, x.to(dtype), training=True)
    lp = mo.crnn_forward(tr.Pc, tr.Bc, img, bn_training=False)
    y, ysz = H.encode(labels)
    loss = so.ctc_mean(lp, y, ysz) + F.mse_loss(img, torch.ones_like(img))
    loss.backward()
    return tr, img.detach(), lp.detach(), loss.item()


@pytest.mark.parametrize("style,B", [("pos", 8), ("uniform", 6)])
def test_phase_b_vs_oracle(style, B):
    """UNet(train) -> CRNN(BN eval) -> CTC + MSE -> backward: EVERY gradient tensor against the CPU
    oracle evaluated in fp64, with the oracle's own fp32 deviation as the conditioning yard-stick."""
    from models.model_crnn import CRNN
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    from qea.loss import CTCLoss
    x = H.synth_images(B, 77) if style == "pos" else torch.rand(B, 1, 32, 128, generator=torch.Generator().manual_seed(79))
    labels = H.synth_labels(B, 78, 1, 10)
    su, sc = mo.seeded_state(mo.unet_state_shapes(), 11), mo.seeded_state(mo.crnn_state_shapes(), 12)
    t64, img64, lp64, loss64 = _oracle_phase_b(x, labels, su, sc, torch.float64)
    t32, img32, _, _ = _oracle_phase_b(x, labels, su, sc, torch.float32)
    # conditioning probe: exact gradient under a 4e-6 relative input perturbation (see helpers.check_grad_vs64)
    xp = x * (1 + 4e-6 * torch.randn(x.shape, generator=torch.Generator().manual_seed(99)))
    t64p, _, _, _ = _oracle_phase_b(xp.double(), labels, su, sc, torch.float64)

    prep = UNet()
    prep.load_state_dict(su)
    prep = prep.cuda().train()
    crnn = CRNN(95, False)
    crnn.load_state_dict(sc)
    crnn = crnn.cuda().train()
    crnn.register_backward_hook(crnn.backward_hook)
    for m in crnn.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    prep.zero_grad()
    crnn.zero_grad()
    img = prep(x.cuda())
    lp = crnn(img)
    y, ysz = H.encode(labels)
    loss = CTCLoss()(lp, y, torch.tensor([31] * B, dtype=torch.int), ysz) + F.mse_loss(img, torch.ones_like(img))
    loss.backward()
    assert _rel(img.detach(), img64) < max(1e-5, 3 * _rel(img32, img64))
    assert (lp.detach().cpu().double() - lp64).abs().max().item() < 2e-4
    assert abs(loss.item() - loss64) < 1e-4 * abs(loss64)                      # north_star: CTC loss within 1e-4
    bad = {}
    for name, p in list(prep.named_parameters()) + list(crnn.named_parameters()):
        src64 = t64.Pu if name in t64.Pu else t64.Pc
        src32 = t32.Pu if name in t32.Pu else t32.Pc
        r64 = src64[name].grad.double()
        dev = (src32[name].grad.double() - r64).norm().item() / max(r64.norm().item(), 1e-300)
        srcp = t64p.Pu if name in t64p.Pu else t64p.Pc
        cond = (srcp[name].grad.double() - r64).norm().item() / max(r64.norm().item(), 1e-300)
        err, worst = H.robust_rel_err(p.grad, r64)
        if err > max(1e-4, 3 * max(dev, cond)):
            bad[name] = (err, dev, cond)
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1][0])[:8]
    for name, b in list(prep.named_buffers()):
        if b.dtype == torch.float32:
            assert _rel(b, t64.Bu[name]) < 1e-5, name


def test_step_golden_with_fused_adam():
    from models.model_crnn import CRNN
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    from qea.loss import CTCLoss
    from qea.optim import FusedAdam
    fx = H.golden("step_area_b4.npz")
    x = torch.from_numpy(fx["x"]).cuda()
    labels = [str(s) for s in fx["labels"]]
    prep = _load(UNet(), mo.unet_state_shapes, 3)
    crnn = _load(CRNN(95, False), mo.crnn_state_shapes, 4)
    crnn.register_backward_hook(crnn.backward_hook)
    opt_c = FusedAdam(crnn.parameters(), lr=1e-4, weight_decay=0)
    ctc = CTCLoss()
    # Phase A (train_nn_area.py:212-275) with the fixture's selection, noise and labels
    crnn.train(); prep.eval(); prep.zero_grad(); crnn.zero_grad()
    with torch.no_grad():
        preds_all = prep(x)
    idx = torch.from_numpy(fx["A|idx"])
    preds = preds_all[idx.cuda()]
    losses = []
    for i in range(2):
        noisy = (preds - torch.from_numpy(fx[f"A|noise{i}"]).cuda()).clamp(0, 1)
        lp = crnn(noisy)
        y, ysz = H.encode([labels[j][::-1] for j in idx.tolist()])
        loss = ctc(lp, y, torch.tensor([31] * len(idx), dtype=torch.int), ysz)
        losses.append(loss.item())
    loss.backward()
    assert np.allclose(losses, fx["A|losses"], rtol=2e-5)
    H.check_grad_vs64(fx, "A|g|", ((k, p.grad) for k, p in crnn.named_parameters()))
    opt_c.step()
    noise_driven = ("convo.conv5.bias", "convo.conv6.bias")
    H.check_tensor_summary(fx, "A|crnn|", ((k, v) for k, v in crnn.state_dict().items() if k not in noise_driven), rtol=1e-4)
    # Phase B from the freshly seeded CRNN of the fixture
    crnn = _load(CRNN(95, False), mo.crnn_state_shapes, 6)
    crnn.register_backward_hook(crnn.backward_hook)
    opt_p = FusedAdam(prep.parameters(), lr=5e-5, weight_decay=0)
    prep.train(); crnn.train()
    for m in crnn.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    prep.zero_grad(); crnn.zero_grad()
    img = prep(x)
    lp = crnn(img)
    y, ysz = H.encode(labels)
    lossB = ctc(lp, y, torch.tensor([31] * 4, dtype=torch.int), ysz) + F.mse_loss(img, torch.ones_like(img)) * 1.0
    lossB.backward()
    assert abs(lossB.item() - float(fx["B|loss64"])) < 1e-4 * abs(lossB.item())
    assert np.abs(img.detach().cpu().numpy() - fx["B|img64"]).max() < 2e-5
    H.check_grad_vs64(fx, "B|g|prep|", ((k, p.grad) for k, p in prep.named_parameters()))
    H.check_grad_vs64(fx, "B|g|crnn|", ((k, p.grad) for k, p in crnn.named_parameters()))
    opt_p.step()
    # Adam's first step is lr*sign(g): insensitive to the gradient's conditioning except where |g| ~ 0
    # (a gradient element whose sign is within the conditioning noise moves its weight by up to 2*lr)
    H.check_tensor_summary(fx, "B|prep|", prep.state_dict().items(), rtol=2e-4, atol=2.1 * 5e-5)


def test_checkpoint_roundtrip(tmp_path):
    """whole-module pickle (reference checkpoint format) survives save -> load -> forward."""
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    net = _load(UNet(), mo.unet_state_shapes, 1).eval()
    x = H.synth_images(2, 5).cuda()
    with torch.no_grad():
        y0 = net(x)
    path = tmp_path / "Prep_model_0_12.34"
    torch.save(net, path)
    net2 = torch.load(path, weights_only=False).cuda().eval()
    with torch.no_grad():
        y1 = net2(x)
    assert torch.equal(y0, y1)
    assert list(net2.state_dict().keys()) == list(mo.unet_state_shapes().keys())


@pytest.mark.parametrize("W", [64, 256, 384, 512])
def test_other_widths_whole_path_vs_oracle(W):
    """The kernels are width-agnostic (W % 16 == 0, T = W/4 - 1): BASELINE configs[4] as reinterpreted in SURVEY F8 (variable-width
    lines up to 32x512 in width buckets, datasets/bucketing.py).  The WHOLE Phase-B path — UNet(train BN) -> CRNN(BN eval) ->
    CTC(mean, targets up to T/2 characters: up to 127 frames x 127 CTC states) + MSE -> backward — at every bucket width against
    the fp64 oracle: loss 1e-4, log-probs, and every gradient tensor at the plain 1e-4 under the HIP forward's decisions
    (tests/decisions.py), as at W = 128."""
    import decisions as D
    from models.model_crnn import CRNN
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    from qea.loss import CTCLoss
    B, T, ws = 3, W // 4 - 1, 33
    x = torch.rand(B, 1, 32, W, generator=torch.Generator().manual_seed(W))
    labels = H.synth_labels(B, W, 1, max(1, T // 2))
    y, ysz = H.encode(labels)
    ins = torch.full((B,), T, dtype=torch.int)
    prep, net = UNet(), CRNN(95, False)
    prep.load_state_dict(mo.default_init_state(mo.unet_state_shapes(), ws))
    net.load_state_dict(mo.default_init_state(mo.crnn_state_shapes(), ws + 1))
    prep, net = prep.cuda().train(), net.cuda().train()
    net.register_backward_hook(net.backward_hook)
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    img = prep(x.cuda())
    lp = net(img)
    assert lp.shape == (T, B, 95)
    force = {**D.hip_unet_trace(D.saved_of(img))[0], **D.hip_crnn_trace(D.saved_of(lp))[0]}
    loss = CTCLoss()(lp, y, ins, ysz) + F.mse_loss(img, torch.ones_like(img))
    loss.backward()
    Pu, Bu = mo.split_state(H._state64(mo.unet_state_shapes(), ws))
    Pc, Bc = mo.split_state(H._state64(mo.crnn_state_shapes(), ws + 1))
    tr = mo.Trace(force, record=False)
    img_r = mo.unet_forward(Pu, Bu, x.double(), training=True, trace=tr)
    lp_r = mo.crnn_forward(Pc, Bc, img_r, bn_training=False, trace=tr)
    assert lp_r.shape[0] == T
    loss_r = F.ctc_loss(lp_r, y, ins, ysz) + F.mse_loss(img_r, torch.ones_like(img_r))
    loss_r.backward()
    assert abs(loss.item() - loss_r.item()) <= 1e-4 * abs(loss_r.item())
    assert (lp.detach().cpu().double() - lp_r.detach()).abs().max().item() < 1e-5
    errs = {n: H.full_rel_err(p.grad, (Pu[n] if n in Pu else Pc[n]).grad) for n, p in list(prep.named_parameters()) + list(net.named_parameters())}
    bad = {n: f"{v:.2e}" for n, v in errs.items() if not v <= 1e-4}
    assert not bad, bad
    print(f"\n[width {W}] T = {T}: loss rel {abs(loss.item() - loss_r.item()) / abs(loss_r.item()):.1e}, worst gradient error under the HIP decisions "
          f"{max(errs.values()):.2e} ({max(errs, key=errs.get)})")


@pytest.mark.parametrize("shape", [(3, 2, 32, 128), (2, 1, 80, 256), (3, 1, 400, 512)])
def test_unet_bn_groups_equal_sequential_passes(shape):
    """[new] UNet.forward(x, bn_groups=N) (the patch flow's --docs_per_step): N groups of images through ONE pass with
    BatchNorm statistics per group == N sequential train-mode passes of the reference (one document per call,
    train_nn_patch.py:318-321): outputs, the gradients of a sum of per-group losses, running statistics after the N ordered
    updates, num_batches_tracked.  (3, 1, 400, 512) = three POS-sized documents: levels 4-5 are 50x64 and 25x32 pixels, which run
    on the generic tile whose statistics blocks straddle images — why the grouped path takes its statistics in a separate pass."""
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    N, per, Hh, Ww = shape
    su = mo.default_init_state(mo.unet_state_shapes(), 17)
    x = torch.rand(N * per, 1, Hh, Ww, generator=torch.Generator().manual_seed(18)).cuda()
    wgt = torch.rand(N * per, 1, Hh, Ww, generator=torch.Generator().manual_seed(19)).cuda()

    def make():
        net = UNet()
        net.load_state_dict(su)
        net = net.cuda().train()
        net.zero_grad()
        return net
    seq = make()
    outs = []
    for g in range(N):
        sl = slice(g * per, (g + 1) * per)
        o = seq(x[sl])
        (F.mse_loss(o, torch.ones_like(o)) + (o * wgt[sl]).mean()).backward()
        outs.append(o.detach())
    fused = make()
    o_all = fused(x, bn_groups=N)
    total = 0
    for g in range(N):
        sl = slice(g * per, (g + 1) * per)
        total = total + F.mse_loss(o_all[sl], torch.ones_like(o_all[sl])) + (o_all[sl] * wgt[sl]).mean()
    total.backward()
    torch.cuda.synchronize()
    # the grouped pass sums its statistics in another order (separate fp64 pass vs the conv epilogue's partials): equal to fp64
    # rounding, i.e. the fp32 coefficients agree except at rounding boundaries
    assert (o_all.detach() - torch.cat(outs)).abs().max().item() <= 2e-6
    for (n, a), (_, b) in zip(fused.named_parameters(), seq.named_parameters()):
        assert _rel(a.grad, b.grad) < 2e-5, n
    for (n, a), (_, b) in zip(fused.named_buffers(), seq.named_buffers()):
        if a.is_floating_point():
            assert (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item()), n
        else:
            assert int(a) == int(b) == N, n


@pytest.mark.parametrize("mode", ["split_f16", "split_bf16"])
def test_replica_groups_equal_sequential_passes(mode):
    """R jitter replicas fused into the batch dimension with per-replica-group BatchNorm == R sequential
    CRNN passes of the reference loop: same log-probs, same gradients (sum of the replica losses), same
    running statistics.  Bit for bit in the three-way bf16 split (every output element has the same summation
    order in both forms); in the two-way fp16 split the operands' power-of-two scale comes from the abs-max of the WHOLE
    tensor a launch consumes — R*k strips there, k strips here — so the forms agree to fp32 rounding instead."""
    from models.model_crnn import CRNN
    from oracle import model_oracle as mo
    from qea import ops
    from qea.loss import CTCLoss
    prev_mode = ops.set_mfma_mode(mode)
    R, k = 3, 4
    sc = mo.seeded_state(mo.crnn_state_shapes(), 7)
    x = torch.stack([H.synth_images(k, 50 + r) for r in range(R)]).reshape(R * k, 1, 32, 128).cuda()
    labels = [H.synth_labels(k, 60 + r, 1, 8) for r in range(R)]
    ins = torch.full((k,), 31, dtype=torch.int)

    def make():
        net = CRNN(95, False)
        net.load_state_dict(sc)
        net = net.cuda().train()
        net.register_backward_hook(net.backward_hook)
        net.zero_grad()
        return net
    seq = make()
    lps = []
    for r in range(R):
        lp = seq(x[r * k:(r + 1) * k])
        y, ysz = H.encode(labels[r])
        CTCLoss()(lp, y, ins, ysz).backward()
        lps.append(lp.detach())
    fused = make()
    lp_all = fused(x, replica_groups=R)
    total = 0
    for r in range(R):
        y, ysz = H.encode(labels[r])
        total = total + CTCLoss()(lp_all[:, r * k:(r + 1) * k, :], y, ins, ysz)
    total.backward()
    ops.set_mfma_mode(prev_mode)
    if mode == "split_bf16":
        assert torch.equal(lp_all.detach(), torch.cat(lps, dim=1))
    else:
        assert (lp_all.detach() - torch.cat(lps, dim=1)).abs().max().item() < 5e-6
    for (n, a), (_, b) in zip(fused.named_parameters(), seq.named_parameters()):
        if mode != "split_bf16" and n in ("convo.conv5.bias", "convo.conv6.bias"):
            continue                                    # exactly zero in exact arithmetic (a bias in front of a batch-statistics BN): rounding noise
        assert _rel(a.grad, b.grad) < 2e-5, n
    for (n, a), (_, b) in zip(fused.named_buffers(), seq.named_buffers()):
        if mode == "split_bf16" or not a.is_floating_point():
            assert torch.equal(a, b), n
        else:
            assert (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item()), n


@pytest.mark.parametrize("mode", ["split_f16", "split_bf16"])
def test_ragged_bn_groups_equal_sequential_passes(mode):
    """CRNN.forward(group_sizes=[...]) (round 4: the jittered strips of SEVERAL documents in one pass, one BatchNorm group per (document,
    replica), groups of different sizes) == one CRNN pass per group in that order (train_nn_patch.py:288-303 document after document):
    same log-probs, gradients of the summed losses, running statistics.  Tolerances as in the equal-groups test."""
    from models.model_crnn import CRNN
    from oracle import model_oracle as mo
    from qea import ops
    from qea.loss import CTCLoss
    prev_mode = ops.set_mfma_mode(mode)
    sizes = [3, 3, 5, 5, 2]                                            # e.g. two replicas of a 3-strip and of a 5-strip document, then 2 strips
    sc = mo.seeded_state(mo.crnn_state_shapes(), 9)
    xs = [H.synth_images(n, 70 + i).cuda() for i, n in enumerate(sizes)]
    labels = [H.synth_labels(n, 80 + i, 1, 8) for i, n in enumerate(sizes)]

    def make():
        net = CRNN(95, False)
        net.load_state_dict(sc)
        net = net.cuda().train()
        net.register_backward_hook(net.backward_hook)
        net.zero_grad()
        return net
    seq = make()
    lps = []
    for x, lab in zip(xs, labels):
        lp = seq(x)
        y, ysz = H.encode(lab)
        CTCLoss()(lp, y, torch.full((x.shape[0],), 31, dtype=torch.int), ysz).backward()
        lps.append(lp.detach())
    fused = make()
    lp_all = fused(torch.cat(xs), group_sizes=sizes)
    total, a = 0, 0
    for n, lab in zip(sizes, labels):
        y, ysz = H.encode(lab)
        total = total + CTCLoss()(lp_all[:, a:a + n, :], y, torch.full((n,), 31, dtype=torch.int), ysz)
        a += n
    total.backward()
    ops.set_mfma_mode(prev_mode)
    if mode == "split_bf16":
        assert torch.equal(lp_all.detach(), torch.cat(lps, dim=1))
    else:
        assert (lp_all.detach() - torch.cat(lps, dim=1)).abs().max().item() < 5e-6
    for (n, a_), (_, b) in zip(fused.named_parameters(), seq.named_parameters()):
        if mode != "split_bf16" and n in ("convo.conv5.bias", "convo.conv6.bias"):
            continue
        assert _rel(a_.grad, b.grad) < 2e-5, n
    for (n, a_), (_, b) in zip(fused.named_buffers(), seq.named_buffers()):
        if mode == "split_bf16" or not a_.is_floating_point():
            assert torch.equal(a_, b), n
        else:
            assert (a_ - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item()), n
    with pytest.raises(ValueError):
        fused(torch.cat(xs), group_sizes=[3, 3])


@pytest.mark.parametrize("g", [2, 0])
def test_backward_group_equals_sequential_last_replica(g):
    """train_nn_area.py:269-271 back-propagates the LAST replica only.  CRNN.forward(backward_group=g) returns the same
    log-probs (the other groups detached) and its backward visits group g's samples only: the gradients must equal those of
    the reference-style loop (R sequential passes, backward of pass g alone), the input gradient is zero outside the group,
    and a loss that touches another group cannot leak a gradient."""
    from models.model_crnn import CRNN
    from oracle import model_oracle as mo
    from qea.loss import CTCLoss
    R, k = 3, 5
    sc = mo.seeded_state(mo.crnn_state_shapes(), 11)
    x = torch.stack([H.synth_images(k, 70 + r) for r in range(R)]).reshape(R * k, 1, 32, 128).cuda()
    labels = [H.synth_labels(k, 80 + r, 1, 8) for r in range(R)]
    ins = torch.full((k,), 31, dtype=torch.int)

    def make():
        net = CRNN(95, False)
        net.load_state_dict(sc)
        net = net.cuda().train()
        net.register_backward_hook(net.backward_hook)
        net.zero_grad()
        return net
    seq = make()
    xs = x.clone().requires_grad_()
    lps = []
    for r in range(R):
        lp = seq(xs[r * k:(r + 1) * k])
        lps.append(lp.detach())
        if r == g:
            y, ysz = H.encode(labels[r])
            CTCLoss()(lp, y, ins, ysz).backward()
    fused = make()
    xf = x.clone().requires_grad_()
    lp_all = fused(xf, replica_groups=R, backward_group=g)
    assert torch.equal(lp_all.detach(), torch.cat(lps, dim=1))
    y, ysz = H.encode(labels[g])
    other = (g + 1) % R
    yo, yosz = H.encode(labels[other])
    # the second term touches a detached group: it must contribute nothing
    loss = CTCLoss()(lp_all[:, g * k:(g + 1) * k, :], y, ins, ysz) + CTCLoss()(lp_all[:, other * k:(other + 1) * k, :], yo, ins, yosz)
    loss.backward()
    for (n, a), (_, b) in zip(fused.named_parameters(), seq.named_parameters()):
        assert _rel(a.grad, b.grad) < 2e-5, n
    for (n, a), (_, b) in zip(fused.named_buffers(), seq.named_buffers()):
        assert torch.equal(a, b), n
    assert _rel(xf.grad[g * k:(g + 1) * k], xs.grad[g * k:(g + 1) * k]) < 2e-5
    outside = torch.cat([xf.grad[:g * k], xf.grad[(g + 1) * k:]])
    assert outside.abs().max().item() == 0.0


def test_fused_adam_resume_matches_uninterrupted(tmp_path):
    """optim_*_latest checkpoints (train_nn_patch.py:153-156,446-454): save after 2 steps, load into a fresh FusedAdam,
    continue -> identical to 3 uninterrupted steps; the state_dict has torch.optim.Adam's layout and loads into it."""
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    from qea.optim import FusedAdam
    su = mo.seeded_state(mo.unet_state_shapes(), 13)
    def make():
        net = UNet()
        net.load_state_dict(su)
        return net.cuda()

    def set_grads(net, k):
        from qea.params import ensure_flat
        ensure_flat(net).attach_grads()
        for i, p in enumerate(net.parameters()):
            p.grad.copy_(torch.randn(p.shape, generator=torch.Generator().manual_seed(k * 1000 + i)).cuda())
    a = make()
    oa = FusedAdam(a.parameters(), lr=1e-3, weight_decay=5e-4)
    for k in range(3):
        set_grads(a, k)
        oa.step()
    b = make()
    ob = FusedAdam(b.parameters(), lr=1e-3, weight_decay=5e-4)
    for k in range(2):
        set_grads(b, k)
        ob.step()
    torch.save(ob.state_dict(), tmp_path / "optim_prep_latest")
    torch.save(b, tmp_path / "Prep_model")
    c = torch.load(tmp_path / "Prep_model", weights_only=False).cuda()
    oc = FusedAdam(c.parameters(), lr=1e-3, weight_decay=5e-4)
    oc.load_state_dict(torch.load(tmp_path / "optim_prep_latest", weights_only=False))
    set_grads(c, 2)
    oc.step()
    for (n, p), (_, q) in zip(a.named_parameters(), c.named_parameters()):
        assert torch.equal(p, q), n
    # the same file loads into torch's own Adam (layout compatibility in the other direction)
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in b.parameters()], lr=1e-3, weight_decay=5e-4)
    ref.load_state_dict(torch.load(tmp_path / "optim_prep_latest", weights_only=False))
    assert int(ref.state_dict()["state"][0]["step"]) == 2


def test_unet_inference_fused_bn_epilogue_is_bit_identical():
    """Eval-mode no-grad forward folds BatchNorm+ReLU into the conv epilogue (halo and generic kernels): same bits as conv -> bn_apply."""
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    from qea import unet_engine
    net = _load(UNet(), mo.unet_state_shapes, 1).eval()
    x = H.synth_images(6, 77).cuda()
    outs = []
    for fuse in (True, False):
        unet_engine.FUSE_EVAL_BN = fuse
        try:
            with torch.no_grad():
                outs.append(net(x).clone())
        finally:
            unet_engine.FUSE_EVAL_BN = True
    assert torch.equal(outs[0], outs[1])


def test_bn_apply_and_pool_in_one_pass_is_bit_identical():
    """qea_bn_apply_pool (the encoder's BatchNorm apply + ReLU and the 2x2 max-pool behind it in one pass, models/model_unet.py:52-59)
    against the two separate launches: kernel level (both outputs and both abs-max slots, NaN and tie cases, (2,1) windows, strided
    buffers) and through the UNet in train mode (outputs, every gradient, running statistics; one and two BatchNorm groups)."""
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    from qea import ops, unet_engine
    g = torch.Generator().manual_seed(3)
    for (B, Hh, Ww, Cc, kh, kw) in ((3, 8, 32, 64, 2, 2), (2, 6, 10, 32, 2, 1), (1, 2, 2, 4, 2, 2)):
        y = torch.randn(B, Hh, Ww, Cc + 8, generator=g).cuda()
        y[0, 0, 0, 0] = float("nan")
        y[0, 1, 1, 1] = y[0, 0, 1, 1] = y[0, 1, 0, 1] = y[0, 0, 0, 1] = 0.5                 # a four-way tie
        sc, sh = (torch.rand(Cc, generator=g) + 0.5).cuda(), torch.randn(Cc, generator=g).cuda()
        a0 = torch.full((B, Hh, Ww, 2 * Cc), -7.0, device="cuda")
        a1 = a0.clone()
        p0 = torch.empty(B, Hh // kh, Ww // kw, Cc, device="cuda")
        p1 = torch.empty_like(p0)
        am = torch.zeros(4, device="cuda")
        ops.bn_apply(y, Cc + 8, a0[..., Cc:], 2 * Cc, B * Hh * Ww, Cc, sc, sh, relu=True, amax=am[0:1])
        ops.maxpool_fwd(a0[..., Cc:], 2 * Cc, p0, Cc, B, Hh, Ww, Cc, kh, kw, amax=am[1:2])
        ops.bn_apply_pool(y, Cc + 8, a1[..., Cc:], 2 * Cc, p1, Cc, B, Hh, Ww, Cc, sc, sh, kh, kw, relu=True, amax=am[2:3], pooled_amax=am[3:4])
        torch.cuda.synchronize()
        assert torch.equal(a0.view(torch.int32), a1.view(torch.int32)) and torch.equal(p0.view(torch.int32), p1.view(torch.int32))
        assert am[0].item() == am[2].item() and am[1].item() == am[3].item()
    res = []
    for fuse in (True, False):                                   # inference pass: the pool leaves with the conv epilogue (pool_y)
        unet_engine.FUSE_BN_POOL = fuse
        try:
            with torch.no_grad():
                res.append(_load(UNet(), mo.unet_state_shapes, 1).eval()(H.synth_images(6, 22).cuda()).clone())
        finally:
            unet_engine.FUSE_BN_POOL = True
    assert torch.equal(res[0], res[1])
    for groups in (1, 2):
        res = []
        for fuse in (True, False):
            unet_engine.FUSE_BN_POOL = fuse
            try:
                net = _load(UNet(), mo.unet_state_shapes, 1).train()
                x = H.synth_images(4, 21).cuda()
                out = net(x, bn_groups=groups)
                (out * torch.linspace(0.5, 1.5, out.numel(), device="cuda").view_as(out)).sum().backward()
                res.append([out.detach().clone()] + [p.grad.clone() for p in net.parameters()] + [b.clone() for b in net.buffers()])
            finally:
                unet_engine.FUSE_BN_POOL = True
        assert all(torch.equal(u, v) for u, v in zip(*res))


def test_crnn_fused_pools_are_bit_identical():
    """qea_conv_c1_fwd_pool (conv1 + ReLU + max_pool2d(2,2), models/model_crnn.py:48) and qea_bn_apply_pool behind BatchNorm2
    (models/model_crnn.py:53-54) against the separate launches: kernel level (both outputs, the pooled abs-max, NaN in the input) and
    through the CRNN (log-probs, input gradient, every parameter gradient, running statistics; train-mode BN, one and two groups)."""
    from models.model_crnn import CRNN
    from oracle import model_oracle as mo
    from qea import crnn_engine, ops
    g = torch.Generator().manual_seed(11)
    for (B, Hh, Ww, Co) in ((3, 32, 128, 64), (2, 6, 12, 32), (1, 2, 4, 128)):
        x = torch.rand(B, Hh, Ww, generator=g).cuda()
        x[0, 1, 2] = float("nan")
        w, b = torch.randn(Co, 9, generator=g).cuda(), torch.randn(Co, generator=g).cuda()
        y0 = torch.empty(B, Hh, Ww, Co, device="cuda")
        y1 = torch.empty_like(y0)
        p0 = torch.empty(B, Hh // 2, Ww // 2, Co, device="cuda")
        p1 = torch.empty_like(p0)
        am = torch.zeros(2, device="cuda")
        ops.conv_c1_fwd(x, w, b, y0, Co, B, Hh, Ww, Co, relu=True)
        ops.maxpool_fwd(y0, Co, p0, Co, B, Hh, Ww, Co, 2, 2, amax=am[0:1])
        assert ops.conv_c1_fwd_pool(x, w, b, y1, Co, p1, Co, B, Hh, Ww, Co, relu=True, pooled_amax=am[1:2])
        torch.cuda.synchronize()
        assert torch.equal(y0.view(torch.int32), y1.view(torch.int32)) and torch.equal(p0.view(torch.int32), p1.view(torch.int32))
        assert am[0].item() == am[1].item()
    assert not ops.conv_c1_fwd_pool(x, w, b, y1, Co, p1, Co, 1, 3, 4, Co)               # odd height: the caller runs the two launches
    for groups in (1, 2):
        res = []
        for fuse in (True, False):
            crnn_engine.FUSE_POOL = fuse
            try:
                net = _load(CRNN(95, False), mo.crnn_state_shapes, 2).train()
                x = H.synth_images(4, 5).cuda().requires_grad_()
                lp = net(x, replica_groups=groups)
                (lp * torch.linspace(0.5, 1.5, lp.numel(), device="cuda").view_as(lp)).sum().backward()
                res.append([lp.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in net.parameters()] + [bf.clone() for bf in net.buffers()])
            finally:
                crnn_engine.FUSE_POOL = True
        assert all(torch.equal(u, v) for u, v in zip(*res))


def test_conv1_relu_pool_backward_from_the_pooled_gradient():
    """qea_conv_c1_pool_bwd (ABI v8): conv1 -> ReLU -> max_pool2d(2, 2) backward (models/model_crnn.py:37-38,47-48 under autograd) from the
    pooled tensor's gradient and the 1-channel input alone — the activation rebuilt with the forward's own multiply-add chain — against
    qea_conv_c1_fwd + qea_maxpool_bwd(relu_mask) + qea_conv_c1_wgrad + qea_conv_c1_dgrad: (i) kernel level incl. windows of equal
    activations (first in scan order wins), all-negative windows and image borders: the masked pooled gradient equals the window sums of
    the routed gradient BIT FOR BIT (same winners, same mask), dW / db / dx to fp32 summation order; (ii) through the CRNN with the
    engine's switch, with and without the input gradient, and with the activation kept or not."""
    from models.model_crnn import CRNN
    from oracle import model_oracle as mo
    from qea import crnn_engine, ops
    g = torch.Generator().manual_seed(11)
    B, Hh, Ww, Co = 5, 32, 128, 64
    x = torch.rand(B, 1, Hh, Ww, generator=g)
    x[0, 0, :8, :16] = 0.5                                             # constant patch: four equal activations per window
    x[1, 0, 8:16, 32:64] = 0.0
    x = x.cuda()
    w = (torch.randn(Co, 1, 3, 3, generator=g) * 0.4).cuda()
    b = (torch.randn(Co, generator=g) * 0.2).cuda()
    b[:8] = -5.0                                                       # channels whose activations are all zero
    dpool = torch.randn(B * (Hh // 2) * (Ww // 2), Co, generator=g).cuda()
    M = B * Hh * Ww
    a1 = torch.empty(M, Co, device="cuda")
    ops.conv_c1_fwd(x, w, b, a1, Co, B, Hh, Ww, Co, relu=True)
    dy1 = torch.empty(M, Co, device="cuda")
    ops.maxpool_bwd(a1, Co, dpool, Co, dy1, Co, B, Hh, Ww, Co, 2, 2, relu_mask=True)
    dw0, db0 = torch.zeros(Co, 1, 3, 3, device="cuda"), torch.zeros(Co, device="cuda")
    ops.conv_c1_wgrad(x, dy1, Co, dw0, db0, B, Hh, Ww, Co)
    dx0 = torch.empty(B, 1, Hh, Ww, device="cuda")
    ops.conv_c1_dgrad(dy1, Co, w, dx0, B, Hh, Ww, Co)
    for need_dx in (True, False):
        gm = dpool.clone()
        dw1, db1 = torch.full((Co, 1, 3, 3), 0.5, device="cuda"), torch.full((Co,), 0.5, device="cuda")
        dx1 = torch.empty(B, 1, Hh, Ww, device="cuda") if need_dx else None
        ops.conv_c1_pool_bwd(x, w, b, gm, Co, dw1, db1, dx1, B, Hh, Ww, Co, accumulate=True)
        torch.cuda.synchronize()
        wsum = dy1.view(B, Hh // 2, 2, Ww // 2, 2, Co).sum(dim=(2, 4)).reshape(-1, Co)      # one non-zero per window: an exact sum
        assert torch.equal(gm, wsum)
        assert ((dw1 - 0.5) - dw0).abs().max().item() <= 2e-5 * dw0.abs().max().item()
        assert ((db1 - 0.5) - db0).abs().max().item() <= 2e-5 * db0.abs().max().item()
        if need_dx:
            assert (dx1 - dx0).abs().max().item() <= 2e-5 * dx0.abs().max().item()
    res = []
    try:
        for fuse, keep in ((True, False), (True, True), (False, False)):
            crnn_engine.FUSE_C1_BWD, crnn_engine.KEEP_A1 = fuse, keep
            crnn = _load(CRNN(95, False), mo.crnn_state_shapes, 3).train()
            xi = H.synth_images(5, 41).cuda().requires_grad_(True)
            lp = crnn(xi)
            assert (lp.grad_fn.saved["acts"]["a1"] is None) == (fuse and not keep)
            (lp * torch.linspace(0.5, 1.5, lp.numel(), device="cuda").view_as(lp)).sum().backward()
            res.append([p.grad.clone() for p in crnn.parameters()] + [xi.grad.clone()])
    finally:
        crnn_engine.FUSE_C1_BWD, crnn_engine.KEEP_A1 = True, False
    for other in res[1:]:
        for u, v in zip(res[0], other):
            den = v.double().norm().item()
            if den > 0:
                assert (u.double() - v.double()).norm().item() / den <= 2e-6


def test_pool_backward_inside_the_bn_backward():
    """qea_bn_bwd_pool (ABI v8): the max-pool backward + the skip path's sum rebuilt per window inside the two passes of the BatchNorm
    backward (models/model_unet.py:52-59, models/model_crnn.py:53-54 under autograd) against qea_maxpool_bwd(accumulate) followed by
    qea_bn_bwd: (i) kernel level with ties, an all-negative window and a NaN — dy bit for bit where the per-channel constants agree bit
    for bit, the parameter gradients to fp64-order noise; (ii) through the whole UNet and CRNN backward with the engines' switch."""
    from models.model_crnn import CRNN
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    from qea import crnn_engine, ops, unet_engine
    g = torch.Generator().manual_seed(5)
    for kw in (2, 1):
        B, Hh, Ww, Cc = 3, 8, 16, 64
        y = torch.randn(B, Hh, Ww, Cc, generator=g)
        y[0, 0, 0:2, :8] = 0.25                                            # a tie inside a window: first in scan order wins
        y[1, 2:4, 4:6, 8:16] = -3.0                                        # a window whose activations are all zero
        y[2, 4, 6, 3] = float("nan")
        y = y.cuda()
        gamma = (torch.rand(Cc, generator=g) + 0.5).cuda()
        beta = torch.randn(Cc, generator=g).mul(0.3).cuda()
        M = B * Hh * Ww
        coef = torch.empty(4, Cc, device="cuda")
        st = torch.empty(2, Cc, device="cuda", dtype=torch.float64)
        rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
        yfin = torch.nan_to_num(y, nan=0.0)
        ops.bn_train_stats(yfin.view(M, Cc), Cc, M, Cc, gamma, beta, 1e-5, 0.1, rm, rv, coef[0], coef[1], coef[2], coef[3], st)
        a = torch.empty(M, Cc, device="cuda")
        ops.bn_apply(y.view(M, Cc), Cc, a, Cc, M, Cc, coef[2], coef[3], relu=True)
        dskip = torch.randn(M, Cc, generator=g).cuda()
        dpool = torch.randn(B * (Hh // 2) * (Ww // kw), Cc, generator=g).cuda()
        outs = []
        for fused in (False, True):
            dg, db = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
            dy = torch.empty(M, Cc, device="cuda")
            if fused:
                ops.bn_bwd_pool(dskip, Cc, dpool, Cc, kw, y.view(M, Cc), Cc, B, Hh, Ww, Cc, gamma, coef[0], coef[1], True, dg, db, dy, Cc,
                                accumulate=True, stat64=st, relu_scale=coef[2], relu_shift=coef[3])
            else:
                da = dskip.clone()
                ops.maxpool_bwd(a, Cc, dpool, Cc, da, Cc, B, Hh, Ww, Cc, 2, kw, relu_mask=False, accumulate=True)
                ops.bn_bwd(da, Cc, None, 0, y.view(M, Cc), Cc, M, Cc, gamma, coef[0], coef[1], True, dg, db, dy, Cc, accumulate=True, stat64=st,
                           relu_scale=coef[2], relu_shift=coef[3])
            outs.append((dy.cpu(), dg.cpu(), db.cpu()))
        (dy0, dg0, db0), (dy1, dg1, db1) = outs
        ok = torch.isfinite(dg0)                                           # (the NaN's channel is NaN in both)
        assert torch.equal(torch.isfinite(dg0), torch.isfinite(dg1)) and torch.equal(torch.isnan(dy0), torch.isnan(dy1))
        assert (dg0[ok] - dg1[ok]).abs().max().item() <= 1e-6 * dg0[ok].abs().max().item()
        assert (db0[ok] - db1[ok]).abs().max().item() <= 1e-6 * db0[ok].abs().max().item()
        fin = torch.isfinite(dy0)
        assert (dy0[fin] - dy1[fin]).abs().max().item() <= 1e-6 * dy0[fin].abs().max().item()
    res = []
    try:
        for fuse in (True, False):
            unet_engine.FUSE_POOL_BWD = crnn_engine.FUSE_POOL_BWD = fuse
            net = _load(UNet(), mo.unet_state_shapes, 1).train()
            crnn = _load(CRNN(95, False), mo.crnn_state_shapes, 2).train()
            x = H.synth_images(6, 31).cuda()
            out = net(x)
            lp = crnn(out)
            (lp * torch.linspace(0.5, 1.5, lp.numel(), device="cuda").view_as(lp)).sum().backward()
            res.append([p.grad.clone() for p in list(net.parameters()) + list(crnn.parameters())])
    finally:
        unet_engine.FUSE_POOL_BWD = crnn_engine.FUSE_POOL_BWD = True
    worst = 0.0
    for u, v in zip(*res):
        den = v.double().norm().item()
        if den > 0:
            worst = max(worst, (u.double() - v.double()).norm().item() / den)
    assert worst <= 2e-6, worst


def test_bn_backward_sums_from_the_dgrad_epilogue():
    """qea_conv_desc.bst_y (ABI v7): the two per-channel reductions of BatchNorm1's backward (sum dz, sum dz * xhat with the ReLU mask
    recomputed from the pre-BN tensor; models/model_unet.py:78-109 under autograd) written as fp64 partials by the epilogue of the 3x3
    dgrad that produces da, consumed by qea_bn_bwd_from_partials — against the separate pass (colreduce) through the whole UNet
    backward: every gradient to fp64-summation-order noise, and the fused launch really ran (profiling tag of the BST instances is the
    plain one, so count launches of the reduction kernel instead: fewer colreduce passes is checked through the result only)."""
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    from qea import ops, unet_engine
    res, fused_calls = [], []
    real_bn_bwd = ops.bn_bwd

    def counting_bn_bwd(*a, **k):
        fused_calls[-1] += k.get("partials") is not None
        return real_bn_bwd(*a, **k)
    ops.bn_bwd = counting_bn_bwd
    try:
        for fuse in (True, False):
            unet_engine.FUSE_BN_BWD_SUMS = fuse
            fused_calls.append(0)
            net = _load(UNet(), mo.unet_state_shapes, 1).train()
            x = H.synth_images(6, 31).cuda()
            out = net(x)
            (out * torch.linspace(0.5, 1.5, out.numel(), device="cuda").view_as(out)).sum().backward()
            res.append([p.grad.clone() for p in net.parameters()])
    finally:
        ops.bn_bwd = real_bn_bwd
        unet_engine.FUSE_BN_BWD_SUMS = True
    # BatchNorm1 of the nine blocks took its sums from the dgrad epilogue (an epilogue of the fp16-split LDS-halo kernel only)
    assert fused_calls == ([9, 0] if ops.mfma_mode() == "split_f16" else [0, 0]), fused_calls
    worst = 0.0
    for u, v in zip(*res):
        den = v.double().norm().item()
        if den > 0:
            worst = max(worst, (u.double() - v.double()).norm().item() / den)
    assert worst <= 2e-6, worst                        # (fp64 partials in another grouping: equal or a few last-bit flips)
    # kernel level: the partials of one launch against fp64 sums of the definition
    g = torch.Generator().manual_seed(8)
    B, Hh, Ww, Cc = 3, 8, 32, 128
    dy = torch.randn(B, Hh, Ww, Cc, generator=g).cuda()
    w = (torch.randn(Cc, 3, 3, Cc, generator=g) / (9 * Cc) ** 0.5).cuda()
    yref = torch.randn(B, Hh, Ww, Cc, generator=g).cuda()
    sc, sh = (torch.rand(Cc, generator=g) + 0.5).cuda(), torch.randn(Cc, generator=g).cuda()
    st = torch.stack([torch.randn(Cc, generator=g).double(), (torch.rand(Cc, generator=g) + 0.5).double()]).cuda()
    da = torch.empty(B, Hh, Ww, Cc, device="cuda")
    prev = ops.set_mfma_mode("split_f16")
    try:
        got = ops.conv_igemm(dy, w, da, B=B, H=Hh, W=Ww, Cin=Cc, OH=Hh, OW=Ww, N=Cc, KH=3, KW=3, pad=(1, 1), ldx=Cc, ldy=Cc,
                             bwd_stats=(yref, Cc, st, sc, sh))
    finally:
        ops.set_mfma_mode(prev)
    assert got is not None
    part = got[0][:got[1]].sum(0).cpu()                                   # [C][2]
    dz = (da.double() * (torch.addcmul(sh, yref, sc) > 0)).reshape(-1, Cc)
    xh = ((yref.double() - st[0]) * st[1]).reshape(-1, Cc)
    assert (part[:, 0] - dz.sum(0).cpu()).abs().max().item() <= 1e-9 * dz.abs().sum(0).max().item()
    assert (part[:, 1] - (dz * xh).sum(0).cpu()).abs().max().item() <= 1e-9 * (dz * xh).abs().sum(0).max().item()


def test_weights_loaded_after_a_forward_get_a_fresh_filter_scale():
    """the fp16 planes of every filter are scaled from ONE abs-max of the model's flat parameter buffer; the
    parameters are views re-homed with `p.data = view`, so load_state_dict / p.copy_ move the PARAMETER's version counter and not the
    buffer's.  Forward, load weights 8x larger (the old bound would put them at inf in the h plane), forward again: the second result
    must be the oracle's for the new weights, and finite."""
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    from qea import ops
    if ops.mfma_mode() != "split_f16":
        pytest.skip("the scale only exists in the fp16 split")
    torch.manual_seed(5)
    net = _load(UNet(), mo.unet_state_shapes, 3).eval()
    x = torch.rand(2, 1, 32, 128).cuda()
    with torch.no_grad():
        y0 = net(x)
        big = {k: (v * 8 if (v.dtype == torch.float32 and k.endswith("weight") and v.dim() == 4) else v) for k, v in mo.seeded_state(mo.unet_state_shapes(), 3).items()}
        net.load_state_dict(big)
        y1 = net(x)
        P, Bf = mo.split_state({k: (v.double() if v.is_floating_point() else v) for k, v in big.items()}, requires_grad=False)
        want = mo.unet_forward(P, Bf, x.cpu().double(), False)
    assert torch.isfinite(y1).all()
    assert not torch.equal(y0, y1)
    assert _rel(y1, want.float()) < 1e-4
    # ... and a write to ONE parameter through torch (copy_) is seen too
    with torch.no_grad():
        p = net.encoder1.enc1conv2.weight if hasattr(net, "encoder1") and hasattr(net.encoder1, "enc1conv2") else next(q for q in net.parameters() if q.dim() == 4 and q.shape[1] > 1)
        p.copy_(p * 16)
        y2 = net(x)
    assert torch.isfinite(y2).all()


def test_batched_weight_forms_give_the_same_step():
    """ops.BATCH_FORMS (round 4): all stale derived forms of a model (flipped filters, fragment planes) in one launch per group on the
    first miss after a weight update.  Two optimiser steps of the UNet -> CRNN -> CTC chain with and without it: identical losses
    and gradients bit for bit, a handful of multi-form launches instead of one launch per form, and a weight written through torch
    between the steps is seen."""
    from models.model_crnn import CRNN
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    from qea import ops
    from qea.loss import CTCLoss
    labels = ["abc", "hello", "MI355X", "q"]
    y, ysz = H.encode(labels)
    x = torch.rand(4, 1, 32, 128, generator=torch.Generator().manual_seed(3)).cuda()

    def run(on):
        old = ops.BATCH_FORMS["on"]
        ops.BATCH_FORMS["on"] = on
        calls = []
        orig = ops._batched_forms

        def counting(key, w, also, job, token):
            out = orig(key, w, also, job, token)
            calls.append(key[0])
            return out
        ops._batched_forms = counting
        try:
            unet = _load(UNet(), mo.unet_state_shapes, 1).train()
            crnn = _load(CRNN(95, False), mo.crnn_state_shapes, 2).train()
            opt = torch.optim.SGD(list(unet.parameters()) + list(crnn.parameters()), lr=1e-3)
            res = []
            for it in range(2):
                calls.append("step")
                opt.zero_grad(set_to_none=True)
                lp = crnn(unet(x))
                T = lp.shape[0]
                loss = CTCLoss()(lp, y, torch.full((4,), T, dtype=torch.int), ysz)
                loss.backward()
                res.append([loss.detach().clone()] + [p.grad.detach().clone() for p in list(unet.parameters()) + list(crnn.parameters())])
                opt.step()
                if it == 0:
                    with torch.no_grad():
                        next(q for q in unet.parameters() if q.dim() == 4 and q.shape[1] > 1).mul_(1.5)      # a write through torch
            torch.cuda.synchronize()
            return res, calls
        finally:
            ops._batched_forms = orig
            ops.BATCH_FORMS["on"] = old

    a, calls_on = run(True)
    b, calls_off = run(False)
    assert [c for c in calls_off if c != "step"] == []
    for ra, rb in zip(a, b):
        for ta, tb in zip(ra, rb):
            assert torch.equal(ta, tb)
    if ops.mfma_mode() == "split_f16":
        # the first step registers the forms one by one; from the second step on, per model: one pack launch at the first conv (which
        # asks for the flipped filters of its input-gradient forms first: one flip launch) instead of ~70 single launches
        second = calls_on[len(calls_on) - calls_on[::-1].index("step"):]
        assert len(calls_on) - len(second) - 2 >= 40 and 2 <= len(second) <= 8, calls_on


def test_fused_adam_of_one_model_leaves_the_other_models_forms_alone():
    """qea.optim.FusedAdam writes the weights through raw pointers and declares it with ops.bump_weight_epoch(its parameters): only
    the forms (planes, flipped filters, the flat abs-max) of THAT model go stale.  After a CRNN-only step the UNet's cached forms
    are the same objects, the CRNN's are new, and the next pass equals a pass with the cache switched off bit for bit."""
    import weakref
    from models.model_crnn import CRNN
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    from qea import ops
    from qea.loss import CTCLoss
    from qea.optim import FusedAdam
    from qea.params import flat_state_of
    labels = ["abc", "hello", "MI355X", "q"]
    y, ysz = H.encode(labels)
    x = torch.rand(4, 1, 32, 128, generator=torch.Generator().manual_seed(4)).cuda()
    unet = _load(UNet(), mo.unet_state_shapes, 1).train()
    crnn = _load(CRNN(95, False), mo.crnn_state_shapes, 2).train()
    opt = FusedAdam(crnn.parameters(), lr=1e-3)

    def pass_():
        for m in (unet, crnn):
            m.zero_grad(set_to_none=True)
        lp = crnn(unet(x))
        loss = CTCLoss()(lp, y, torch.full((4,), lp.shape[0], dtype=torch.int), ysz)
        loss.backward()
        return [loss.detach().clone()] + [p.grad.detach().clone() for p in list(unet.parameters()) + list(crnn.parameters())]

    pass_()
    pass_()                                                  # (the CRNN's flat buffer is made in the first pass, after the UNet's forward: that bumps everything once)
    fs_u = flat_state_of(next(unet.parameters()))
    import gc
    gc.collect()                                             # (entries of dead models leave through weakref callbacks: not while we walk the dict)
    before = {k: (v[1], v[2]()) for k, v in list(ops._wcache.items())}
    opt.step()
    got = pass_()
    kept = rebuilt = 0
    for k, (val, w) in before.items():
        if w is None or k not in ops._wcache or not (torch.is_tensor(w) and w.is_cuda):
            continue
        fs = flat_state_of(w, check=False)
        if fs is None:
            continue
        same = ops._wcache[k][1] is val
        if fs is fs_u:
            assert same, k
            kept += 1
        elif not same:
            rebuilt += 1
    assert kept >= 20 and rebuilt >= 10, (kept, rebuilt)
    old = ops.WEIGHT_CACHE["on"]
    ops.WEIGHT_CACHE["on"] = False
    try:
        want = pass_()
    finally:
        ops.WEIGHT_CACHE["on"] = old
    for a, b in zip(got, want):
        assert torch.equal(a, b)
