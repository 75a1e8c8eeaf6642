"""Kernel-level parity of the HIP library (through the C ABI) against the CPU oracle / torch-CPU
autograd, on seeded inputs.  Integer / index results bit-exact, fp32 within stated tolerances."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import helpers as H

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def rel_err(got, ref):
    return (got.double() - ref.double()).abs().max().item() / max(ref.double().abs().max().item(), 1e-12)


# ----------------------------------------------------------------------------- BN
@pytest.mark.parametrize("C,M", [(32, 2 * 32 * 128), (64, 999), (512, 48), (96, 130)])
def test_bn_train_fwd_bwd(C, M):
    from qea import ops
    g = torch.Generator().manual_seed(C + M)
    y = (torch.randn(M, C, generator=g) * 2 + 0.7)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    da = torch.randn(M, C, generator=g)
    # oracle: torch CPU double
    yr = y.double().t().reshape(1, C, M, 1).requires_grad_()
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    rm_r, rv_r = rm.double().clone(), rv.double().clone()
    a_r = F.relu(F.batch_norm(yr, rm_r, rv_r, gr, br, True, 0.1, 1e-5))
    a_r.backward(da.double().t().reshape(1, C, M, 1))
    dev = "cuda"
    yd, dad = y.to(dev), da.to(dev)
    gd, bd, rmd, rvd = gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev)
    mean, invstd, scale, shift = (torch.empty(C, device=dev) for _ in range(4))
    stat64 = torch.empty(2, C, device=dev, dtype=torch.float64)
    ops.bn_train_stats(yd, C, M, C, gd, bd, 1e-5, 0.1, rmd, rvd, mean, invstd, scale, shift, stat64)
    a = torch.empty(M, C, device=dev)
    ops.bn_apply(yd, C, a, C, M, C, scale, shift, relu=True)
    dgamma, dbeta = torch.empty(C, device=dev), torch.empty(C, device=dev)
    dy = torch.empty(M, C, device=dev)
    ops.bn_bwd(dad, C, a, C, yd, C, M, C, gd, mean, invstd, True, dgamma, dbeta, dy, C, stat64=stat64)
    torch.cuda.synchronize()
    assert rel_err(a.cpu(), a_r.detach().reshape(C, M).t()) < 2e-6
    assert rel_err(rmd.cpu(), rm_r) < 1e-6 and rel_err(rvd.cpu(), rv_r) < 1e-6
    assert rel_err(dy.cpu(), yr.grad.reshape(C, M).t()) < 1e-6
    # the per-channel sum of dy is exactly 0 in exact arithmetic: no correlated rounding may survive
    assert (dy.double().sum(0).abs() / dy.double().abs().sum(0).clamp_min(1e-30)).max().item() < 1e-7
    assert rel_err(dgamma.cpu(), gr.grad) < 1e-5 and rel_err(dbeta.cpu(), br.grad) < 1e-5


@pytest.mark.parametrize("C,M", [(32, 2 * 32 * 128 + 5), (512, 777), (64, 4096)])
def test_bn_bwd_mask_recomputed_from_y_is_bit_identical(C, M):
    """qea_bn_bwd(relu_scale, relu_shift) must give exactly what qea_bn_bwd(a) gives (same mask, one tensor read less)."""
    from qea import ops
    g = torch.Generator().manual_seed(11)
    dev = "cuda"
    y = (torch.randn(M, C, generator=g) * 2 + 0.3).to(dev)
    da = torch.randn(M, C, generator=g).to(dev)
    gamma, beta = torch.randn(C, generator=g).to(dev), (0.01 * torch.randn(C, generator=g)).to(dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean, invstd, scale, shift = (torch.empty(C, device=dev) for _ in range(4))
    stat64 = torch.empty(2, C, device=dev, dtype=torch.float64)
    ops.bn_train_stats(y, C, M, C, gamma, beta, 1e-5, 0.1, rm, rv, mean, invstd, scale, shift, stat64)
    a = torch.empty(M, C, device=dev)
    ops.bn_apply(y, C, a, C, M, C, scale, shift, relu=True)
    out = []
    for kw in (dict(a=a, lda=C), dict(a=None, lda=0, relu_scale=scale, relu_shift=shift)):
        dgamma, dbeta, dy = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(M, C, device=dev)
        ops.bn_bwd(da, C, kw.pop("a"), kw.pop("lda"), y, C, M, C, gamma, mean, invstd, True, dgamma, dbeta, dy, C, stat64=stat64, **kw)
        out.append((dgamma, dbeta, dy))
    torch.cuda.synchronize()
    for u, v in zip(*out):
        assert torch.equal(u, v)


def test_bn_eval_fwd_bwd():
    from qea import ops
    C, M = 512, 3 * 4 * 32
    g = torch.Generator().manual_seed(3)
    y = torch.randn(M, C, generator=g)
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    da = torch.randn(M, C, generator=g)
    yr = y.double().t().reshape(1, C, M, 1).requires_grad_()
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    a_r = F.relu(F.batch_norm(yr, rm.double(), rv.double(), gr, br, False, 0.1, 1e-5))
    a_r.backward(da.double().t().reshape(1, C, M, 1))
    dev = "cuda"
    mean, invstd, scale, shift = (torch.empty(C, device=dev) for _ in range(4))
    ops.bn_eval_coeff(C, gamma.to(dev), beta.to(dev), rm.to(dev), rv.to(dev), 1e-5, None, mean, invstd, scale, shift)
    yd = y.to(dev)
    a = torch.empty(M, C, device=dev)
    ops.bn_apply(yd, C, a, C, M, C, scale, shift, relu=True)
    dgamma, dbeta, dy = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(M, C, device=dev)
    ops.bn_bwd(da.to(dev), C, a, C, yd, C, M, C, gamma.to(dev), mean, invstd, False, dgamma, dbeta, dy, C)
    torch.cuda.synchronize()
    assert rel_err(a.cpu(), a_r.detach().reshape(C, M).t()) < 2e-6
    assert rel_err(dy.cpu(), yr.grad.reshape(C, M).t()) < 1e-5
    assert rel_err(dgamma.cpu(), gr.grad) < 1e-5 and rel_err(dbeta.cpu(), br.grad) < 1e-5


# ----------------------------------------------------------------------------- pool / layout
@pytest.mark.parametrize("kh,kw", [(2, 2), (2, 1)])
def test_maxpool_fwd_bwd_with_ties(kh, kw):
    from qea import ops
    B, Cc, Hh, Ww = 3, 64, 8, 32
    g = torch.Generator().manual_seed(kh * 10 + kw)
    x = torch.randn(B, Cc, Hh, Ww, generator=g).relu()          # many exact-zero ties, like a ReLU output
    x[:, :, ::2, ::2] = x[:, :, 1::2, 1::2] if kw == 2 else x[:, :, ::2, ::2]   # positive ties too
    xr = x.double().requires_grad_()
    yr = F.max_pool2d(xr, (kh, kw))
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    mask = (x > 0).double()
    xd = nhwc(x).cuda()
    y = torch.empty(B, Hh // kh, Ww // kw, Cc, device="cuda")
    ops.maxpool_fwd(xd, Cc, y, Cc, B, Hh, Ww, Cc, kh, kw)
    dx = torch.full((B, Hh, Ww, Cc), 0.25, device="cuda")
    ops.maxpool_bwd(xd, Cc, nhwc(dy).cuda(), Cc, dx, Cc, B, Hh, Ww, Cc, kh, kw, relu_mask=True, accumulate=True)
    torch.cuda.synchronize()
    assert torch.equal(y.cpu().permute(0, 3, 1, 2), yr.detach().float())
    ref = (xr.grad * mask).float()
    assert torch.equal((dx.cpu() - 0.25).permute(0, 3, 1, 2), (ref + 0.25) - 0.25)


def test_colsum_transpose_flip():
    from qea import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1234, 96, generator=g)
    out = torch.ones(96, device="cuda")
    ops.colsum(x.cuda(), 96, 1234, 96, out, accumulate=True)
    assert rel_err(out.cpu(), x.double().sum(0) + 1) < 1e-6
    a = torch.randn(100, 70, generator=g)
    t = torch.empty(70, 100, device="cuda")
    ops.transpose2d(a.cuda(), t, 100, 70)
    assert torch.equal(t.cpu(), a.t().contiguous())
    w = torch.randn(12, 3, 3, 20, generator=g)            # [Co][kh][kw][Ci]
    wt = torch.empty(20, 3, 3, 12, device="cuda")
    ops.filter_flip_transpose(w.cuda(), wt, 12, 20, 3, 3)
    assert torch.equal(wt.cpu(), w.flip(1, 2).permute(3, 1, 2, 0).contiguous())


def test_conv_dgrad_via_flipped_filter():
    """input gradient of a 3x3 conv = the same implicit GEMM on the flipped/transposed filter."""
    from qea import ops
    B, Ci, Co, Hh, Ww = 2, 64, 96, 8, 16
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, Ci, Hh, Ww, generator=g).double().requires_grad_()
    w = torch.randn(Co, Ci, 3, 3, generator=g).double()
    dy = torch.randn(B, Co, Hh, Ww, generator=g)
    F.conv2d(x, w, padding=1).backward(dy.double())
    wd = w.float().permute(0, 2, 3, 1).contiguous().cuda()
    wt = torch.empty(Ci, 3, 3, Co, device="cuda")
    ops.filter_flip_transpose(wd, wt, Co, Ci, 3, 3)
    dx = torch.empty(B, Hh, Ww, Ci, device="cuda")
    ops.conv_igemm(nhwc(dy).cuda(), wt, dx, B=B, H=Hh, W=Ww, Cin=Co, OH=Hh, OW=Ww, N=Ci, KH=3, KW=3, pad=(1, 1), ldx=Co, ldy=Ci)
    assert rel_err(dx.cpu().permute(0, 3, 1, 2), x.grad) < 2e-5


def test_convtranspose_fwd_and_dgrad():
    from qea import ops
    B, Ci, Co, Hh, Ww = 2, 64, 32, 4, 16
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Ci, Hh, Ww, generator=g).double().requires_grad_()
    w = torch.randn(Ci, Co, 2, 2, generator=g).double()
    b = torch.randn(Co, generator=g).double()
    y = F.conv_transpose2d(x, w, b, stride=2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    wp = w.float().permute(0, 2, 3, 1).contiguous().cuda()          # [Ci][a][b][Co]  (channels_last of IOHW)
    wT = torch.empty(4 * Co, Ci, device="cuda")
    ops.transpose2d(wp, wT, Ci, 4 * Co)
    yd = torch.zeros(B, 2 * Hh, 2 * Ww, 2 * Co, device="cuda")        # written into the first half of a concat buffer
    ops.conv_igemm(nhwc(x.detach().float()).cuda(), wT, yd, B=B, H=Hh, W=Ww, Cin=Ci, OH=Hh, OW=Ww, N=4 * Co, KH=1, KW=1,
                   ldx=Ci, ldy=2 * Co, bias=b.float().cuda(), out_mode=ops.OUT_CONVT)
    assert rel_err(yd.cpu()[..., :Co].permute(0, 3, 1, 2), y.detach()) < 2e-5
    assert yd[..., Co:].abs().max().item() == 0
    dx = torch.empty(B, Hh, Ww, Ci, device="cuda")
    ops.conv_igemm(nhwc(dy).cuda(), wp, dx, B=B, H=2 * Hh, W=2 * Ww, Cin=Co, OH=Hh, OW=Ww, N=Ci, KH=2, KW=2, stride=(2, 2),
                   ldx=Co, ldy=Ci)
    assert rel_err(dx.cpu().permute(0, 3, 1, 2), x.grad) < 2e-5


# ----------------------------------------------------------------------------- C_in = 1 convs, head
@pytest.mark.parametrize("Co,relu,bias,Hh,Ww", [(32, False, False, 32, 128), (64, True, True, 32, 128), (64, True, True, 12, 48), (32, False, True, 9, 30),
                                                (128, True, False, 16, 200)])
def test_conv_c1(Co, relu, bias, Hh, Ww):
    """C_in = 1 convolutions (UNet enc1conv1, CRNN conv1; /root/reference/models/model_unet.py:85, model_crnn.py:37): forward (4-pixel
    groups when W % 4 == 0, else the one-pixel kernel), weight / bias gradient, input gradient (8 x 64 tiles; partial tiles at 12 x 48,
    9 x 30 and 16 x 200) against fp64 autograd"""
    from qea import ops
    B = 3
    g = torch.Generator().manual_seed(Co)
    x = torch.rand(B, 1, Hh, Ww, generator=g).double().requires_grad_()
    w = torch.randn(Co, 1, 3, 3, generator=g).double().requires_grad_()
    b = torch.randn(Co, generator=g).double().requires_grad_() if bias else None
    pre = F.conv2d(x, w, b, padding=1)
    dy = torch.randn(pre.shape, generator=g)
    pre.backward(dy.double())
    ref = pre.relu() if relu else pre
    xd = x.detach().float().cuda().reshape(B, Hh, Ww)
    wd = w.detach().float().reshape(Co, 9).cuda()
    y = torch.empty(B, Hh, Ww, Co, device="cuda")
    ops.conv_c1_fwd(xd, wd, b.detach().float().cuda() if bias else None, y, Co, B, Hh, Ww, Co, relu=relu)
    assert rel_err(y.cpu().permute(0, 3, 1, 2), ref.detach()) < 2e-6
    dyd = nhwc(dy).cuda()
    dw, db = torch.empty(Co, 9, device="cuda"), torch.empty(Co, device="cuda")
    ops.conv_c1_wgrad(xd, dyd, Co, dw, db if bias else None, B, Hh, Ww, Co)
    assert rel_err(dw.cpu(), w.grad.reshape(Co, 9)) < 1e-5
    if bias:
        assert rel_err(db.cpu(), b.grad) < 1e-5
    dx = torch.empty(B, Hh, Ww, device="cuda")
    ops.conv_c1_dgrad(dyd, Co, wd, dx, B, Hh, Ww, Co)
    assert rel_err(dx.cpu(), x.grad.reshape(B, Hh, Ww)) < 1e-5
    ops.conv_c1_dgrad(dyd, Co, wd, dx, B, Hh, Ww, Co, accumulate=True)
    assert rel_err(dx.cpu(), 2 * x.grad.reshape(B, Hh, Ww)) < 1e-5


def test_head_fwd_bwd():
    from qea import ops
    B, Cc, Hh, Ww = 2, 32, 32, 128
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, Cc, Hh, Ww, generator=g).double().requires_grad_()
    w = torch.randn(1, Cc, 1, 1, generator=g).double().requires_grad_()
    b = torch.randn(1, generator=g).double().requires_grad_()
    y = torch.sigmoid(F.conv2d(x, w, b))
    dyy = torch.randn(y.shape, generator=g)
    y.backward(dyy.double())
    M = B * Hh * Ww
    xd = nhwc(x.detach().float()).cuda()
    wd, bd = w.detach().float().reshape(Cc).cuda(), b.detach().float().cuda()
    yd = torch.empty(M, device="cuda")
    ops.head_fwd(xd, Cc, wd, bd, yd, M, Cc)
    assert rel_err(yd.cpu(), y.detach().reshape(M)) < 2e-6
    dx = torch.empty(M, Cc, device="cuda")
    dw, db = torch.empty(Cc, device="cuda"), torch.empty(1, device="cuda")
    ops.head_bwd(xd, Cc, yd, dyy.reshape(M).cuda(), wd, dx, Cc, dw, db, M, Cc)
    assert rel_err(dx.cpu().reshape(B, Hh, Ww, Cc).permute(0, 3, 1, 2), x.grad) < 1e-5
    assert rel_err(dw.cpu(), w.grad.reshape(Cc)) < 1e-5 and rel_err(db.cpu(), b.grad) < 1e-5


# ----------------------------------------------------------------------------- derived weight forms, several per launch
def test_weight_forms_multi_equals_the_single_launches():
    """qea_weight_forms_multi (csrc/weight_forms.hip): flip-transposed filters, 3x3 fragment planes (both chunk orders) and 1x1
    fragment planes of several layers in ONE launch each group — the same bytes as the single calls."""
    import ctypes as C
    from qea import _lib, ops
    L = _lib.lib()
    dev = "cuda"
    g = torch.Generator().manual_seed(11)
    st = torch.cuda.current_stream().cuda_stream
    shapes3 = [(64, 32), (32, 32), (128, 64), (256, 128), (512, 512), (64, 64)]          # (Co, Ci) of 3x3 layers
    ws = [(torch.randn(co, 3, 3, ci, generator=g) * (0.1 + i)).to(dev) for i, (co, ci) in enumerate(shapes3)]
    # flips
    single = []
    for w, (co, ci) in zip(ws, shapes3):
        wt = torch.empty(ci, 3, 3, co, device=dev)
        ops.filter_flip_transpose(w, wt, co, ci, 3, 3)
        single.append(wt)
    jobs = (_lib.WformJob * len(ws))()
    multi = [torch.empty_like(t) for t in single]
    for i, (w, (co, ci)) in enumerate(zip(ws, shapes3)):
        jobs[i].src, jobs[i].dst, jobs[i].amax, jobs[i].kind = w.data_ptr(), multi[i].data_ptr(), None, 0
        jobs[i].a, jobs[i].b, jobs[i].c, jobs[i].d = co, ci, 3, 3
    _lib.check(L.qea_weight_forms_multi(jobs, len(ws), st), "multi")
    for a, b in zip(single, multi):
        assert torch.equal(a, b)
    # 3x3 planes of the forward filters (N = Co, Cin = Ci) and of the flipped ones (N = Ci, Cin = Co)
    srcs = [(w, co, ci) for w, (co, ci) in zip(ws, shapes3)] + [(wt, ci, co) for wt, (co, ci) in zip(single, shapes3) if co >= 32 and ci >= 32]
    srcs = [(w, n, c) for (w, n, c) in srcs if n in (32, 64) or n % 128 == 0]
    amax = [ops.absmax(w, w.numel(), 1, w.numel()) for w, _, _ in srcs]
    single = []
    for (w, n, c), am in zip(srcs, amax):
        out = torch.zeros(L.qea_pack_frag_planes_f16_bytes(n, c), dtype=torch.uint8, device=dev)
        _lib.check(L.qea_pack_frag_planes_f16(w.data_ptr(), n, c, am.data_ptr(), out.data_ptr(), st), "single")
        single.append(out)
    jobs = (_lib.WformJob * len(srcs))()
    multi = [torch.zeros_like(t) for t in single]
    for i, ((w, n, c), am) in enumerate(zip(srcs, amax)):
        jobs[i].src, jobs[i].dst, jobs[i].amax, jobs[i].kind = w.data_ptr(), multi[i].data_ptr(), am.data_ptr(), 1
        jobs[i].a, jobs[i].b, jobs[i].c, jobs[i].d = n, c, 0, 0
    _lib.check(L.qea_weight_forms_multi(jobs, len(srcs), st), "multi")
    for a, b in zip(single, multi):
        assert torch.equal(a, b)
    # 1x1 planes
    shapes1 = [(1024, 512), (128, 64), (512, 2048)]
    w1 = [(torch.randn(n, k, generator=g) / k ** 0.5).to(dev) for n, k in shapes1]
    amax = [ops.absmax(w, w.numel(), 1, w.numel()) for w in w1]
    jobs = (_lib.WformJob * len(w1))()
    single, multi = [], []
    for i, (w, (n, k), am) in enumerate(zip(w1, shapes1, amax)):
        out = torch.zeros(L.qea_pack_frag_planes_f16_1x1_bytes(n, k), dtype=torch.uint8, device=dev)
        _lib.check(L.qea_pack_frag_planes_f16_1x1(w.data_ptr(), n, k, am.data_ptr(), out.data_ptr(), st), "single")
        single.append(out)
        multi.append(torch.zeros_like(out))
        jobs[i].src, jobs[i].dst, jobs[i].amax, jobs[i].kind = w.data_ptr(), multi[i].data_ptr(), am.data_ptr(), 2
        jobs[i].a, jobs[i].b, jobs[i].c, jobs[i].d = n, k, 0, 0
    _lib.check(L.qea_weight_forms_multi(jobs, len(w1), st), "multi")
    torch.cuda.synchronize()
    for a, b in zip(single, multi):
        assert torch.equal(a, b)
    assert L.qea_weight_forms_multi(jobs, 65, st) < 0          # more than 64 jobs: refused, not truncated


# ----------------------------------------------------------------------------- LSTM
@pytest.mark.parametrize("B", [70, 600, 1300])
def test_lstm_seq_is_bit_identical_under_concurrent_load(B):
    """The in-launch hand-off of csrc/lstm_seq.hip (write-through fragment stores, one arrival counter per (row block, direction),
    sc1 loads) under UNEVEN load: a side stream keeps the chip busy with large GEMMs while the layer runs, so workgroups of a group
    start at different times and some wait for a CU.  The kernels are deterministic: every run must equal the idle run bit for bit
    (B = 70: groups spread over the XCDs; 600: 128-row forward, 32-row backward; 1300: 128-row both, ragged last row block)."""
    from qea import ops
    if ops.mfma_mode() != "split_f16":
        pytest.skip("the one-launch layer kernels are the fp16-split mode's")
    dev, T = "cuda", 31
    g = torch.Generator().manual_seed(B)
    wf, wr = (torch.randn(1024, 256, generator=g) / 16).to(dev), (torch.randn(1024, 256, generator=g) / 16).to(dev)
    old = ops.LSTM_SEQ["on"]
    ops.LSTM_SEQ["on"] = True
    try:
        pf, pb, mode = ops.lstm_packs(wf, wr)
    finally:
        ops.LSTM_SEQ["on"] = old
    assert mode == "seq"
    gx = (torch.randn(T, B, 2048, generator=g) * 0.5).to(dev)
    dy = torch.randn(T, B, 512, generator=g).to(dev)

    def layer():
        gates, c, y = gx.clone(), torch.empty(T, B, 512, device=dev), torch.empty(T, B, 512, device=dev)
        ops.lstm_layer_fwd_any(gates, c, y, pf, mode, T, B)
        acts = gates.clone()
        ops.lstm_layer_bwd_any(gates, c, dy, pb, mode, None, T, B)
        return y, acts, c, gates

    ref = layer()
    torch.cuda.synchronize()
    assert all(torch.isfinite(t).all() for t in ref)
    a = torch.randn(4096, 4096, device=dev)
    side = torch.cuda.Stream()
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(6 + 3 * rep):
                a @ a
        out = layer()
        torch.cuda.synchronize()
        for r, o in zip(ref, out):
            assert torch.equal(r, o), rep


@pytest.mark.parametrize("T,B,step", [(31, 5, "f32"), (7, 70, "f32"), (31, 5, "split"), (7, 70, "split"), (3, 1571, "split"), (4, 2048, "split"),
                                      (3, 1571, "f32"), (31, 5, "seq"), (7, 70, "seq"), (3, 1571, "seq"), (4, 2048, "seq"), (31, 512, "seq"),
                                      (31, 600, "seq")])
def test_lstm_layer_fwd_bwd(T, B, step):
    """step = "f32": lstm_step_kernel (v_mfma_f32_32x32x2_f32); "split": lstm_step_bf3_kernel (split-bf16 recurrent GEMMs; 32-row
    workgroups below 1 536 rows, 128-row workgroups with LDS-DMA weight stages above, ragged last row block included)."""
    from oracle import model_oracle as mo
    from qea import ops
    g = torch.Generator().manual_seed(T * 100 + B)
    In = 512
    x = torch.randn(T, B, In, generator=g) * 0.5
    P = {}
    for suf in ("", "_reverse"):
        P["w_ih" + suf] = (torch.randn(1024, In, generator=g) / In ** 0.5).requires_grad_()
        P["w_hh" + suf] = (torch.randn(1024, 256, generator=g) / 16).requires_grad_()
        P["b_ih" + suf] = (torch.randn(1024, generator=g) * 0.1).requires_grad_()
        P["b_hh" + suf] = (torch.randn(1024, generator=g) * 0.1).requires_grad_()
    xr = x.clone().requires_grad_()
    out = torch.cat([mo.lstm_layer_dir(xr, P["w_ih" + s], P["w_hh" + s], P["b_ih" + s], P["b_hh" + s], rev)
                     for s, rev in (("", False), ("_reverse", True))], dim=2)
    dy = torch.randn(out.shape, generator=g)
    out.backward(dy)
    dev = "cuda"
    gates = torch.empty(T, B, 2048, device=dev)
    xd = x.to(dev)
    for d, s in enumerate(("", "_reverse")):
        bias = (P["b_ih" + s] + P["b_hh" + s]).detach().to(dev)
        ops.conv_igemm(xd, P["w_ih" + s].detach().to(dev), gates[:, :, d * 1024:], B=1, H=1, W=T * B, Cin=In, OH=1, OW=T * B,
                       N=1024, KH=1, KW=1, ldx=In, ldy=2048, bias=bias)
    split = step == "split"
    if step == "seq" and ops.mfma_mode() != "split_f16":
        pytest.skip("the one-launch layer kernels are the fp16-split mode's")
    if step == "seq":
        # one launch per pass: W_hh in LDS as fp16 planes, h / gate gradients exchanged inside the launch (csrc/lstm_seq.hip);
        # 32-row workgroups up to 512 rows, 128-row ones above
        from qea import _lib
        assert _lib.lib().qea_lstm_seq_pack_bytes() == 1024 * 256 * 2 * 2
        old = ops.LSTM_SEQ["on"]
        ops.LSTM_SEQ["on"] = True
        try:
            pf, pb, split = ops.lstm_packs(P["w_hh"].detach().to(dev), P["w_hh_reverse"].detach().to(dev))
        finally:
            ops.LSTM_SEQ["on"] = old
        assert split == "seq"
    elif split:
        from qea import _lib
        nb = _lib.lib().qea_lstm_pack_whh_split_bytes()
        assert nb == 1024 * 256 * 3 * 2
        pf, pb = torch.empty(2, nb, dtype=torch.uint8, device=dev), torch.empty(2, nb, dtype=torch.uint8, device=dev)
    else:
        pf, pb = torch.empty(2, 1024 * 256, device=dev), torch.empty(2, 1024 * 256, device=dev)
    for d, s in enumerate(("", "_reverse")):
        if step != "seq":
            (ops.lstm_pack_whh_split if split else ops.lstm_pack_whh)(P["w_hh" + s].detach().to(dev), pf[d], pb[d])
    c, y = torch.empty(T, B, 512, device=dev), torch.empty(T, B, 512, device=dev)
    ya, ga = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    carried = ops.lstm_layer_fwd_any(gates, c, y, pf, split, T, B, y_amax=ya)
    torch.cuda.synchronize()
    assert rel_err(y.cpu(), out.detach()) < 2e-5
    assert bool(carried) == (step == "seq")
    if carried:                                           # the one-launch kernel leaves the layer output's abs-max (exactly)
        assert ya.item() == y.abs().max().item()
    dc = torch.empty(B, 512, device=dev)
    carried = ops.lstm_layer_bwd_any(gates, c, dy.to(dev), pb, split, dc, T, B, g_amax=ga)
    torch.cuda.synchronize()
    if carried:                                           # ... and the gate gradients'
        assert ga.item() == gates.abs().max().item()
    # dgates -> dW_ih, db, dX via the generic kernels
    for d, s in enumerate(("", "_reverse")):
        dg = gates[:, :, d * 1024:]
        dw = torch.empty(1024, In, device=dev)
        ops.conv_wgrad(dg, xd, dw, B=1, PH=1, PW=T * B, QH=1, QW=T * B, R=1024, Cc=In, KH=1, KW=1, ldp=2048, ldq=In)
        assert rel_err(dw.cpu(), P["w_ih" + s].grad) < 1e-4, s
        db = torch.empty(1024, device=dev)
        ops.colsum(dg, 2048, T * B, 1024, db)
        assert rel_err(db.cpu(), P["b_ih" + s].grad) < 1e-4
        # dW_hh = sum_t dgates[t]^T h[t_prev]
        dwh = torch.empty(1024, 256, device=dev)
        if d == 0:
            ops.conv_wgrad(gates[1:, :, :1024], y[:T - 1, :, :256], dwh, B=1, PH=1, PW=(T - 1) * B, QH=1, QW=(T - 1) * B,
                           R=1024, Cc=256, KH=1, KW=1, ldp=2048, ldq=512)
        else:
            ops.conv_wgrad(gates[:T - 1, :, 1024:], y[1:, :, 256:], dwh, B=1, PH=1, PW=(T - 1) * B, QH=1, QW=(T - 1) * B,
                           R=1024, Cc=256, KH=1, KW=1, ldp=2048, ldq=512)
        assert rel_err(dwh.cpu(), P["w_hh" + s].grad) < 1e-4, s
    wcat = torch.cat([P["w_ih"].detach(), P["w_ih_reverse"].detach()], 0).to(dev)      # [2048][In]
    wT = torch.empty(In, 2048, device=dev)
    ops.transpose2d(wcat, wT, 2048, In)
    dx = torch.empty(T, B, In, device=dev)
    ops.conv_igemm(gates, wT, dx, B=1, H=1, W=T * B, Cin=2048, OH=1, OW=T * B, N=In, KH=1, KW=1, ldx=2048, ldy=In)
    assert rel_err(dx.cpu(), xr.grad) < 1e-4


# ----------------------------------------------------------------------------- log_softmax / CTC
def _run_ctc(lp, targets, tls, reduction="mean"):
    from qea import ops
    T, N, Cc = lp.shape
    dev = "cuda"
    lpd = lp.to(dev).contiguous()
    tg = torch.as_tensor(np.asarray(targets), dtype=torch.int32, device=dev)
    tl = torch.as_tensor(np.asarray(tls), dtype=torch.int32, device=dev)
    off = torch.zeros(N, dtype=torch.int64)
    off[1:] = torch.cumsum(torch.as_tensor(np.asarray(tls), dtype=torch.int64), 0)[:-1]
    il = torch.full((N,), T, dtype=torch.int32, device=dev)
    nll, loss = torch.empty(N, device=dev), torch.empty(1, device=dev)
    grad = torch.empty(T, N, Cc, device=dev)
    S_max = 2 * max(int(max(tls)), 1) + 1
    ops.ctc_loss(lpd, N * Cc, Cc, tg, off.to(dev), il, tl, T, N, Cc, 0, S_max, 1 if reduction == "mean" else 0, 1.0, nll, loss,
                 grad, N * Cc, Cc)
    torch.cuda.synchronize()
    return nll.cpu(), loss.cpu().item(), grad.cpu()


def test_ctc_golden_cases():
    fx = H.golden("ctc_cases.npz")
    nll, loss, grad = _run_ctc(torch.from_numpy(fx["lp"]), fx["targets"], fx["target_lengths"])
    assert np.allclose(nll.numpy(), fx["nll"], rtol=1e-5)
    assert abs(loss - float(fx["loss_mean"])) <= 1e-5 * abs(loss)
    assert np.abs(grad.numpy() - fx["grad_mean"]).max() <= 1e-4 * np.abs(fx["grad_mean"]).max()
    nll, loss, grad = _run_ctc(torch.from_numpy(fx["inf_lp"]), fx["inf_targets"], fx["inf_target_lengths"])
    assert (np.isinf(nll.numpy()) == np.isinf(fx["inf_nll"])).all() and np.isinf(loss)
    fin = np.isfinite(fx["inf_nll"])
    assert np.allclose(nll.numpy()[fin], fx["inf_nll"][fin], rtol=1e-5)
    ref = fx["inf_grad_mean"]
    assert (np.isnan(grad.numpy()) == np.isnan(ref)).all()
    ok = ~np.isnan(ref)
    assert np.abs(grad.numpy()[ok] - ref[ok]).max() <= 1e-4 * np.abs(ref[ok]).max()


def test_ctc_vs_aten_batch_and_logsoftmax():
    from qea import ops
    T, N, Cc = 31, 64, 95
    g = torch.Generator().manual_seed(12)
    logits = torch.randn(T, N, Cc, generator=g) * 3
    rng = np.random.RandomState(5)
    tls = rng.randint(0, 16, N)
    tls[3] = 31
    targets = np.concatenate([rng.randint(1, Cc, l) for l in tls]).astype(np.int32)
    lr = logits.clone().requires_grad_()
    lp_r = F.log_softmax(lr, 2)
    per = F.ctc_loss(lp_r, torch.from_numpy(targets), torch.full((N,), T, dtype=torch.int32), torch.from_numpy(tls.astype(np.int32)),
                     reduction="none")
    lp_r.retain_grad()
    loss_r = (per / torch.from_numpy(np.maximum(tls, 1)).float()).mean()
    loss_r.backward()
    dev = "cuda"
    lgd = torch.zeros(T * N, 96, device=dev)
    lgd[:, :95] = logits.reshape(T * N, Cc).to(dev)
    lp = torch.empty(T * N, 96, device=dev)
    ops.log_softmax_fwd(lgd, 96, lp, 96, T * N, Cc)
    torch.cuda.synchronize()
    assert rel_err(lp[:, :95].cpu(), lp_r.detach().reshape(T * N, Cc)) < 1e-6
    nll, loss, grad = _run_ctc(lp_r.detach(), targets, tls)
    fin = torch.isfinite(per.detach())
    assert (torch.isfinite(nll) == fin).all()
    assert torch.allclose(nll[fin], per.detach()[fin], rtol=1e-5)
    gref = lp_r.grad
    ok = ~torch.isnan(gref)
    assert (torch.isnan(grad) == torch.isnan(gref)).all()
    assert (grad[ok] - gref[ok]).abs().max().item() <= 1e-4 * gref[ok].abs().max().item()
    # fused log_softmax backward + NaN scrub == autograd through log_softmax with the scrub hook
    dl = torch.empty(T * N, 96, device=dev)
    ops.log_softmax_bwd(grad.to(dev).reshape(T * N, Cc), Cc, lp, 96, dl, 96, T * N, Cc, 96, True)
    torch.cuda.synchronize()
    ref = torch.nan_to_num(lr.grad, nan=0.0).reshape(T * N, Cc)
    assert torch.isfinite(dl).all() and dl[:, 95].abs().max().item() == 0
    assert (dl[:, :95].cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


# ----------------------------------------------------------------------------- Adam / jitter / top-k / crops / decode
def test_adam_matches_torch():
    from qea import ops
    g = torch.Generator().manual_seed(0)
    n = 100003
    for wd in (0.0, 5e-4):
        p0 = torch.randn(n, generator=g)
        q = p0.clone().requires_grad_()
        opt = torch.optim.Adam([q], lr=1e-4, weight_decay=wd)
        pad = (n + 3) // 4 * 4
        p, m, v = torch.zeros(pad, device="cuda"), torch.zeros(pad, device="cuda"), torch.zeros(pad, device="cuda")
        p[:n] = p0.cuda()
        for step in range(1, 4):
            gr = torch.randn(n, generator=g)
            q.grad = gr.clone()
            opt.step()
            gd = torch.zeros(pad, device="cuda")
            gd[:n] = gr.cuda()
            ops.adam_step(p, gd, m, v, n, 1e-4, 0.9, 0.999, 1e-8, wd, step)
            assert (p[:n].cpu() - q.detach()).abs().max().item() < 2e-7


def test_jitter_apply_bit_exact_and_moments():
    from oracle import path_oracle as po
    from qea import ops
    fx = H.golden("helpers.npz")
    img = torch.from_numpy(fx["jit|img"]).reshape(1, -1)
    for tag, coef in (("jit0", 1.0), ("jit1", 1.0), ("jitc", 0.5)):
        out = torch.empty(1, img.shape[1], device="cuda")
        ops.jitter_apply(img.cuda(), torch.from_numpy(fx[f"{tag}|noise"]).reshape(1, -1).cuda(), out, 1, 1, img.shape[1], coef)
        assert np.array_equal(out.cpu().numpy().reshape(fx[f"{tag}|out"].shape), fx[f"{tag}|out"])
    # generated noise: replicas fused in the batch dim, per-image sigma, N(0, sigma^2) moments
    K, R, HW = 16, 4, 4096
    base = torch.full((K, HW), 0.5, device="cuda")
    sigma = torch.tensor([0.01 * (1 + (i % 5)) for i in range(R * K)], device="cuda")
    out, noise = torch.empty(R * K, HW, device="cuda"), torch.empty(R * K, HW, device="cuda")
    ops.jitter(base, sigma, out, noise, K, R, HW, 1.0, 1234, 0)
    z = (noise / sigma[:, None]).cpu().double()
    assert abs(z.mean().item()) < 0.01 and abs(z.std().item() - 1) < 0.01
    assert abs((z ** 3).mean().item()) < 0.03 and abs((z ** 4).mean().item() - 3) < 0.1
    assert np.array_equal(out.cpu().numpy(), po.jitter(base.cpu().numpy().repeat(R, 0).reshape(R, K, HW).reshape(R * K, HW),
                                                        noise.cpu().numpy()))
    out2 = torch.empty_like(out)
    ops.jitter(base, sigma, out2, None, K, R, HW, 1.0, 1234, 1)      # different offset -> different stream
    assert not torch.equal(out, out2)
    ops.jitter(base, sigma, out2, None, K, R, HW, 1.0, 1234, 0)
    assert torch.equal(out, out2)


def test_topk_bit_exact(golden_dir):
    from oracle import path_oracle as po
    from qea import ops
    cases = json.load(open(os.path.join(golden_dir, "topk_cases.json")))["cases"]
    rng = np.random.RandomState(0)
    extra = [(rng.choice([0.0, 0.5, 1.0, 1 / 3, 0.25], 8192).astype(np.float32), 410), (rng.rand(1000).astype(np.float32), 1000),
             (np.zeros(17, np.float32), 5)]
    runs = [(np.float32([c["cers"][n] for n in c["names"] if n in c["cers"]]), c["k"], c) for c in cases] + [(v, k, None) for v, k in extra]
    for vals, k, c in runs:
        idx = torch.empty(k, dtype=torch.int64, device="cuda")
        ops.topk_desc_stable(torch.from_numpy(vals).cuda(), len(vals), k, idx)
        got = idx.cpu().tolist()
        assert got == po.topk_desc_stable(vals, k).tolist()
        if c is not None and "build-specific" not in c["note"] and "real slice" not in c["note"]:
            assert got == c["idx"], c["note"]


@pytest.mark.parametrize("fixture,pk,bk,sk", [("helpers.npz", "crop|page", "crop|boxes", "crop|stack"),
                                               ("crop_oversize.npz", "page", "boxes", "stack")])
def test_crop_pad_gather_scatter(fixture, pk, bk, sk):
    """crop_oversize.npz: boxes larger than 32x128 by odd / even amounts (negative ConstantPad2d padding crops; Python
    floor division decides which side loses the extra pixel — utils.py:118-126)."""
    from qea import ops
    fx = H.golden(fixture)
    page = torch.from_numpy(fx[pk])[0]
    Hh, Ww = page.shape
    boxes = torch.from_numpy(fx[bk].astype(np.int32))
    N = boxes.shape[0]
    out = torch.empty(N, 32, 128, device="cuda")
    ops.crop_pad_gather(page.cuda(), Hh, Ww, boxes.cuda(), N, 32, 128, out)
    assert np.array_equal(out.cpu().numpy().reshape(fx[sk].shape), fx[sk])
    g = torch.Generator().manual_seed(1)
    dout = torch.randn(N, 32, 128, generator=g)
    pr = page.clone().requires_grad_()
    stack = []
    for b in boxes.tolist():
        crop = pr[b[1]:b[3], b[0]:b[2]]
        ch, cw = crop.shape
        left, top = (128 - cw) // 2, (32 - ch) // 2
        stack.append(F.pad(crop, (left, 128 - left - cw, top, 32 - top - ch), value=1.0))
    torch.stack(stack).backward(dout)
    dimg = torch.zeros(Hh, Ww, device="cuda")
    ops.crop_pad_scatter(dout.cuda(), boxes.cuda(), N, 32, 128, dimg, Hh, Ww)
    assert (dimg.cpu() - pr.grad).abs().max().item() < 1e-6


def test_greedy_decode_matches_reference_strings():
    from qea import ops
    fx = H.golden("helpers.npz")
    scores = torch.from_numpy(fx["dec|scores"]).cuda()
    T, N, Cc = scores.shape
    tokens = torch.zeros(N, T, dtype=torch.int32, device="cuda")
    lengths = torch.zeros(N, dtype=torch.int32, device="cuda")
    ops.greedy_decode(scores, N * Cc, Cc, T, N, Cc, 0, tokens, lengths)
    tk, ln = tokens.cpu().tolist(), lengths.cpu().tolist()
    got = ["".join(H.I2C[i] for i in tk[n][:ln[n]]) for n in range(N)]
    assert got == [str(s) for s in fx["dec|strings"]]


def test_device_cer_matches_host_reference_loop():
    """decode + edit distance on the device == pred_to_string + compare_labels per sample (bit-exact ints)."""
    import utils
    from oracle import path_oracle as po
    g = torch.Generator().manual_seed(23)
    T, B, Cc = 31, 200, 95
    scores = torch.randn(T, B, Cc, generator=g)
    scores[:, :, 0] += 1.5
    labels = H.synth_labels(B, 9, 0, 20)
    labels[3] = ""
    labels[7] = "x" * 100
    cers = utils.batch_cers(scores.cuda(), labels, H.C2I)
    preds = po.greedy_decode(scores.numpy(), H.I2C)
    ref = [po.compare_labels([p], [l])[1] for p, l in zip(preds, labels)]
    assert cers == ref
