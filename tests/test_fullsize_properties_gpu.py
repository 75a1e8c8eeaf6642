"""Parity at the BENCH batch (B = 512), where the CPU oracle would take minutes: size-independent properties.

  * batch-split invariance: per-sample outputs of the CRNN (BN eval) and of the UNet (BN eval) do not depend on what
    else is in the batch (the kernels tile over the batch dimension, so this exercises tile seams);
  * a random 8-sample subset of the B = 512 result equals the CPU oracle run on just those 8 samples;
  * gradient linearity: scaling the loss by 2 scales every gradient by exactly 2; accumulation over two backward
    calls equals the gradient of the sum;
  * an infeasible CTC target contributes an inf loss and exactly zero gradient rows;
  * the wgrad split-K reduction is bit-reproducible from run to run.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import helpers as H

pytestmark = pytest.mark.gpu
B = 512


def _models(seed_u=31, seed_c=32):
    from models.model_crnn import CRNN
    from models.model_unet import UNet
    from oracle import model_oracle as mo
    su, sc = mo.seeded_state(mo.unet_state_shapes(), seed_u), mo.seeded_state(mo.crnn_state_shapes(), seed_c)
    prep = UNet()
    prep.load_state_dict(su)
    crnn = CRNN(95, False)
    crnn.load_state_dict(sc)
    crnn.register_backward_hook(crnn.backward_hook)
    return prep.cuda(), crnn.cuda(), su, sc


def _bn_eval(m):
    for x in m.modules():
        if isinstance(x, torch.nn.modules.batchnorm._BatchNorm):
            x.eval()


def test_batch_split_invariance_and_oracle_subset():
    from oracle import model_oracle as mo
    prep, crnn, su, sc = _models()
    prep.eval()
    crnn.eval()
    x = H.synth_images(B, 123).cuda()
    with torch.no_grad():
        img = prep(x)
        lp = crnn(img)
        img_h = torch.cat([prep(x[:200]), prep(x[200:])])
        lp_h = torch.cat([crnn(img[:, :][:72]), crnn(img[72:])], dim=1)
    # a different batch picks different conv tiles (16- or 32-deep K slices = another fp32 summation order):
    # the outputs agree to accumulated rounding, not bit for bit
    assert (img - img_h).abs().max().item() < 2e-5
    assert (lp - lp_h).abs().max().item() < 1e-4
    idx = torch.tensor([0, 17, 127, 128, 255, 256, 300, 511])
    Pu, Bu = mo.split_state(su, requires_grad=False)
    Pc, Bc = mo.split_state(sc, requires_grad=False)
    with torch.no_grad():
        img_r = mo.unet_forward(Pu, Bu, x[idx.cuda()].cpu(), training=False)
        lp_r = mo.crnn_forward(Pc, Bc, img_r, bn_training=False)
    assert (img[idx.cuda()].cpu() - img_r).abs().max().item() < 2e-5
    assert (lp[:, idx.cuda()].cpu() - lp_r).abs().max().item() < 5e-4


def test_gradient_linearity_accumulation_and_infeasible_rows():
    from qea.loss import CTCLoss
    prep, crnn, _, _ = _models()
    x = H.synth_images(B, 321).cuda()
    labels = H.synth_labels(B, 5, 1, 12)
    labels[7] = "zz" * 10                                     # 20 chars, 19 repeats: needs 39 > 31 frames
    y, ysz = H.encode(labels)
    ins = torch.full((B,), 31, dtype=torch.int)

    def grads(scale, twice=False):
        prep.train()
        crnn.train()
        _bn_eval(crnn)
        prep.zero_grad()
        crnn.zero_grad()
        xi = x.clone().requires_grad_()
        for _ in range(2 if twice else 1):
            lp = crnn(xi)
            per = CTCLoss(reduction="none")(lp, y, ins, ysz)
            fin = torch.isfinite(per)
            (per[fin].sum() * scale).backward()
        return torch.cat([p.grad.flatten().clone() for p in crnn.parameters()]), xi.grad.clone(), per.detach()

    g1, dx1, per = grads(1.0)
    g2, dx2, _ = grads(2.0)
    g3, dx3, _ = grads(1.0, twice=True)
    assert torch.isinf(per[7]) and torch.isfinite(per[torch.arange(B) != 7]).all()
    assert torch.equal(g2, 2 * g1) and torch.equal(dx2, 2 * dx1)          # power-of-two scaling is exact in fp32
    assert (g3 - 2 * g1).abs().max().item() <= 1e-5 * g1.abs().max().item()
    # mean-reduced loss with the infeasible member: inf loss, finite gradients, zero rows for that sample
    crnn.zero_grad()
    xi = x.clone().requires_grad_()
    loss = CTCLoss()(crnn(xi), y, ins, ysz)
    assert torch.isinf(loss)
    loss.backward()
    assert torch.isfinite(xi.grad).all() and xi.grad[7].abs().max().item() == 0.0 and xi.grad[8].abs().max().item() > 0
    assert all(torch.isfinite(p.grad).all() for p in crnn.parameters())


def test_wgrad_bit_reproducible_and_conv_linear():
    from qea import ops
    g = torch.Generator().manual_seed(3)
    Bn, Hh, Ww, Ci, Co = B, 8, 32, 128, 128
    x = torch.randn(Bn, Hh, Ww, Ci, generator=g).cuda()
    dy = torch.randn(Bn, Hh, Ww, Co, generator=g).cuda()
    outs = []
    for _ in range(3):
        dw = torch.empty(Co, 3, 3, Ci, device="cuda")
        ops.conv_wgrad(dy, x, dw, B=Bn, PH=Hh, PW=Ww, QH=Hh, QW=Ww, R=Co, Cc=Ci, KH=3, KW=3, pad=(1, 1), ldp=Co, ldq=Ci)
        outs.append(dw.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])     # no float atomics anywhere
    w = torch.randn(Co, 3, 3, Ci, generator=g).cuda()
    x2 = torch.randn(Bn, Hh, Ww, Ci, generator=g).cuda()

    def conv(inp):
        out = torch.empty(Bn, Hh, Ww, Co, device="cuda")
        ops.conv_igemm(inp, w, out, B=Bn, H=Hh, W=Ww, Cin=Ci, OH=Hh, OW=Ww, N=Co, KH=3, KW=3, pad=(1, 1), ldx=Ci, ldy=Co)
        return out
    a, b, ab = conv(x), conv(x2), conv(x + x2)
    assert (ab - (a + b)).abs().max().item() <= 2e-5 * ab.abs().max().item()
    assert torch.equal(conv(2 * x), 2 * a)


def test_bilstm_layer_at_full_size_one_launch_vs_steps():
    """BASELINE configs[2] size (T = 31, B = 2048, 128-row workgroups, 256 of them, both exchange parities in use for 30 steps):
    the one-launch layer kernels (csrc/lstm_seq.hip) against the per-step split-bf16 kernels on the same inputs (forward output and
    gate gradients within 1e-6 relative: two fp32-class arithmetics), bit-reproducible run to run, the carried abs-max values exact,
    and a batch-split property: rows 0..1279 of the full batch equal a B = 1280 launch of those rows bit for bit (a row block's result
    depends on no other row block; both sizes run the 128-row workgroups — the 32-row shape of smaller batches sums the eight K chunks in
    another association and agrees to rounding only)."""
    from qea import ops
    if ops.mfma_mode() != "split_f16":
        pytest.skip("the one-launch layer kernels are the fp16-split mode's")
    dev, T, B = "cuda", 31, 2048
    g = torch.Generator().manual_seed(77)
    wf, wr = (torch.randn(1024, 256, generator=g) / 16).to(dev), (torch.randn(1024, 256, generator=g) / 16).to(dev)
    gx = (torch.randn(T, B, 2048, generator=g) * 0.5).to(dev)
    dy = torch.randn(T, B, 512, generator=g).to(dev)

    def layer(mode_on, gx_, dy_, Bn):
        old = ops.LSTM_SEQ["on"]
        ops.LSTM_SEQ["on"] = mode_on
        try:
            pf, pb, mode = ops.lstm_packs(wf, wr)
        finally:
            ops.LSTM_SEQ["on"] = old
        gates, c, y = gx_.clone(), torch.empty(T, Bn, 512, device=dev), torch.empty(T, Bn, 512, device=dev)
        ya, ga = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        ops.lstm_layer_fwd_any(gates, c, y, pf, mode, T, Bn, y_amax=ya)
        dc = None if mode == "seq" else torch.empty(Bn, 512, device=dev)
        ops.lstm_layer_bwd_any(gates, c, dy_, pb, mode, dc, T, Bn, g_amax=ga)
        torch.cuda.synchronize()
        return y, gates, ya, ga

    y1, g1, ya, ga = layer(True, gx, dy, B)
    y2, g2, _, _ = layer(True, gx, dy, B)
    assert torch.equal(y1, y2) and torch.equal(g1, g2)                      # run to run
    assert torch.isfinite(y1).all() and torch.isfinite(g1).all()
    assert ya.item() == y1.abs().max().item() and ga.item() == g1.abs().max().item()
    ys, gs, _, _ = layer(False, gx, dy, B)                                  # the per-step kernels
    assert ((y1 - ys).norm() / ys.norm()).item() < 1e-6
    assert ((g1 - gs).norm() / gs.norm()).item() < 1e-6
    yh, gh, _, _ = layer(True, gx[:, :1280].contiguous(), dy[:, :1280].contiguous(), 1280)
    assert torch.equal(yh, y1[:, :1280]) and torch.equal(gh, g1[:, :1280])
    yq, gq, _, _ = layer(True, gx[:, :512].contiguous(), dy[:, :512].contiguous(), 512)          # 32-row workgroups
    assert ((yq - y1[:, :512]).norm() / y1[:, :512].norm()).item() < 1e-6   # (forward: the four gates meet through LDS, same sums)
    assert ((gq - g1[:, :512]).norm() / g1[:, :512].norm()).item() < 1e-6
