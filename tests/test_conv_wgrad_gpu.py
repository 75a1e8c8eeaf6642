"""HIP weight-gradient kernel (qea_conv_wgrad) vs torch-CPU autograd, through the C ABI."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _conv_case(B, H, W, Cin, Cout, K=3, pad=1, tile=0, splits=0, seed=0, accumulate=False):
    from qea import ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g).double()
    w = torch.zeros(Cout, Cin, K, K, dtype=torch.double, requires_grad=True)
    y = F.conv2d(x, w, padding=pad)
    dy = torch.randn(y.shape, generator=g).double()
    y.backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).contiguous()          # [Cout][kh][kw][Cin]
    OH, OW = y.shape[2], y.shape[3]
    xd = x.float().permute(0, 2, 3, 1).contiguous().cuda()
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
    dw = torch.full(ref.shape, 0.5 if accumulate else float("nan"), device="cuda")
    ops.conv_wgrad(dyd, xd, dw, B=B, PH=OH, PW=OW, QH=H, QW=W, R=Cout, Cc=Cin, KH=K, KW=K, pad=(pad, pad),
                   ldp=Cout, ldq=Cin, accumulate=accumulate, tile=tile, splits=splits)
    torch.cuda.synchronize()
    got = dw.cpu().double() - (0.5 if accumulate else 0.0)
    err = (got - ref).abs().max().item()
    assert err <= 3e-5 * ref.abs().max().item() + 1e-6, (err, ref.abs().max().item())


@pytest.mark.parametrize("H,W,Cin,Cout", [(32, 128, 32, 32), (32, 128, 64, 32), (16, 64, 32, 64), (16, 64, 64, 64),
                                          (8, 32, 128, 128), (8, 32, 256, 128), (4, 16, 128, 256), (2, 8, 512, 512),
                                          (4, 32, 256, 512)])
def test_wgrad_conv3x3(H, W, Cin, Cout):
    _conv_case(2, H, W, Cin, Cout)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 20, 21, 22])
def test_wgrad_tiles_ragged_and_splits(tile):
    _conv_case(3, 5, 7, 36, 44, tile=tile, splits=1)
    _conv_case(3, 5, 7, 36, 44, tile=tile, splits=3, accumulate=True)


def test_wgrad_conv7_2x2():
    _conv_case(3, 2, 32, 512, 512, K=2, pad=0)


def test_wgrad_convtranspose():
    from qea import ops
    g = torch.Generator().manual_seed(1)
    B, H, W, Ci, Co = 2, 4, 16, 64, 32
    x = torch.randn(B, Ci, H, W, generator=g).double()
    w = torch.zeros(Ci, Co, 2, 2, dtype=torch.double, requires_grad=True)
    y = F.conv_transpose2d(x, w, stride=2)
    dy = torch.randn(y.shape, generator=g).double()
    y.backward(dy)
    ref = w.grad.permute(0, 2, 3, 1).contiguous()          # [Cin][a][b][Cout]
    xd = x.float().permute(0, 2, 3, 1).contiguous().cuda()
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
    dw = torch.empty(ref.shape, device="cuda")
    ops.conv_wgrad(xd, dyd, dw, B=B, PH=H, PW=W, QH=2 * H, QW=2 * W, R=Ci, Cc=Co, KH=2, KW=2, stride=(2, 2), ldp=Ci, ldq=Co)
    torch.cuda.synchronize()
    assert (dw.cpu().double() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()


def test_wgrad_linear():
    from qea import ops
    g = torch.Generator().manual_seed(2)
    M, K, N = 31 * 7, 512, 96
    x = torch.randn(M, K, generator=g)
    dy = torch.randn(M, N, generator=g)
    ref = dy.double().t() @ x.double()
    dw = torch.empty(N, K, device="cuda")
    ops.conv_wgrad(dy.cuda(), x.cuda(), dw, B=1, PH=1, PW=M, QH=1, QW=M, R=N, Cc=K, KH=1, KW=1, ldp=N, ldq=K)
    torch.cuda.synchronize()
    assert (dw.cpu().double() - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 8, 32, 128, 128), (3, 8, 32, 256, 128), (5, 4, 16, 128, 256), (2, 4, 32, 256, 512),
                                            (2, 16, 64, 64, 128), (1, 4, 32, 512, 512), (9, 4, 16, 256, 256),
                                            (3, 32, 128, 32, 32), (2, 32, 128, 64, 32), (2, 16, 64, 32, 64), (3, 8, 16, 96, 32), (2, 8, 32, 32, 96)])
@pytest.mark.parametrize("splits", [0, 1, 3])
@pytest.mark.parametrize("tile", [23, 29])
def test_wgrad_nine_tap_split_bf16(B, H, W, Cin, Cout, splits, tile):
    """tile 23 with 64-channel blocks in the fp16 form = wgrad_halo9_spec_kernel (round 4: four MFMA waves + four staging waves per
    workgroup, two LDS buffers); tile 29 (and the 32-wide blocks / the bf16 form under tile 23) = wgrad_halo9_bf3_kernel: a workgroup accumulates all nine taps of a 64x64 (or 32-wide) channel block from ONE staged
    dY tile + X halo (split once) instead of gathering and splitting both operands per tap; 32- and 16-pixel-wide tiles,
    image borders, pixel splits (order-fixed slab reduction), accumulate; against torch-CPU autograd in fp64."""
    _conv_case(B, H, W, Cin, Cout, tile=tile, splits=splits, seed=B * 100 + H, accumulate=(splits == 3))


def test_wgrad_nine_tap_bit_reproducible_and_default():
    from qea import ops
    g = torch.Generator().manual_seed(3)
    Bn, Hh, Ww, Ci, Co = 64, 8, 32, 128, 256
    x = torch.randn(Bn, Hh, Ww, Ci, generator=g).cuda()
    dy = torch.randn(Bn, Hh, Ww, Co, generator=g).cuda()
    outs = []
    for tile in (23, 23, 0):                                   # tile 0 (auto) must take the nine-tap kernel for this shape
        dw = torch.empty(Co, 3, 3, Ci, device="cuda")
        ops.conv_wgrad(dy, x, dw, B=Bn, PH=Hh, PW=Ww, QH=Hh, QW=Ww, R=Co, Cc=Ci, KH=3, KW=3, pad=(1, 1), ldp=Co, ldq=Ci, tile=tile)
        outs.append(dw.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # the round-3 form of the same kernel (every wave stages): other slabs, same products — equal to fp32 rounding, and reproducible
    for tile in (29, 29):
        dw = torch.empty(Co, 3, 3, Ci, device="cuda")
        ops.conv_wgrad(dy, x, dw, B=Bn, PH=Hh, PW=Ww, QH=Hh, QW=Ww, R=Co, Cc=Ci, KH=3, KW=3, pad=(1, 1), ldp=Co, ldq=Ci, tile=tile)
        outs.append(dw.clone())
    assert torch.equal(outs[3], outs[4])
    assert (outs[3] - outs[0]).abs().max().item() <= 2e-6 * outs[0].abs().max().item()


@pytest.mark.parametrize("B,H,W,Cin,Cout,splits,acc", [(6, 8, 32, 128, 256, 0, False), (3, 4, 16, 256, 128, 3, True), (5, 16, 64, 64, 64, 1, True),
                                                       (2, 8, 32, 64, 32, 0, True)])
def test_wgrad_bias_gradient_rides_with_the_staging_waves(B, H, W, Cin, Cout, splits, acc):
    """qea_wgrad_desc.dbias (ABI v8): the bias gradient = column sums of dY (nn.Conv2d(bias=True) under autograd,
    models/model_crnn.py:38-45) from the producer waves of the producer / consumer nine-tap kernel — fp64 sums of what they stage,
    one partial row per (split, wave), fixed-order reduction — against fp64 sums on the host; accumulate; a shape the kernel does not
    take (32-wide blocks) gets the separate colsum pass through the same call; dW unchanged by the option."""
    from qea import ops
    g = torch.Generator().manual_seed(B * 7 + H)
    x = torch.randn(B, H, W, Cin, generator=g).cuda()
    dy = torch.randn(B, H, W, Cout, generator=g).cuda()
    base = 0.25 if acc else float("nan")
    dw0 = torch.full((Cout, 3, 3, Cin), base, device="cuda")
    dw1 = dw0.clone()
    db = torch.full((Cout,), base, device="cuda")
    kw = dict(B=B, PH=H, PW=W, QH=H, QW=W, R=Cout, Cc=Cin, KH=3, KW=3, pad=(1, 1), ldp=Cout, ldq=Cin, accumulate=acc, splits=splits)
    ops.conv_wgrad(dy, x, dw0, **kw)
    ops.conv_wgrad(dy, x, dw1, dbias=db, **kw)
    torch.cuda.synchronize()
    assert torch.equal(dw0, dw1)
    ref = dy.double().reshape(-1, Cout).sum(0).cpu() + (base if acc else 0.0)
    err = (db.cpu().double() - ref).abs().max().item()
    assert err <= 2e-6 * ref.abs().max().item() + 1e-6, err
    # bit-reproducible
    db2 = torch.full((Cout,), base, device="cuda")
    ops.conv_wgrad(dy, x, dw1.fill_(base), dbias=db2, **kw)
    assert torch.equal(db, db2)
